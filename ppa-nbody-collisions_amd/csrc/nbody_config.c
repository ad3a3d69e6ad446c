/* csrc/nbody_config.c -- nbodyConfig.txt reader, C restatement of parseConfigFile
 * (include/nbodyConfig.h:22-227 of the reference): same grammar, same number syntax (std::stoi / std::stof
 * = strtol / strtof with "no conversion" and ERANGE as errors), same echo text, same messages.  The only
 * behavioural differences are the documented ones in include/nbody.h: errors are returned, not exit(1)'d,
 * and keys that are absent from the file are zero and flagged in `present` instead of uninitialised. */
#include "nbody.h"
#include "nbody_error.h"
#include <errno.h>
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void echo(int fd, const char* fmt, ...) {
    if (fd < 0) return;
    char buf[NBODY_IMAGE_PATH_MAX + 128];
    va_list ap;
    va_start(ap, fmt);
    int len = vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (len < 0) return;
    if ((size_t)len >= sizeof(buf)) len = (int)sizeof(buf) - 1;
    const char* p = buf;
    while (len > 0) {
        ssize_t w = write(fd, p, (size_t)len);
        if (w <= 0) return;
        p += w;
        len -= (int)w;
    }
}

/* std::stoi: strtol base 10; nothing converted -> invalid_argument, ERANGE or outside int -> out_of_range.
 * Both exceptions' what() is "stoi". Trailing characters are ignored. */
static int parse_int(const char* s, int* out) {
    char* end;
    errno = 0;
    long v = strtol(s, &end, 10);
    if (end == s) return -1;
    if (errno == ERANGE || v < INT_MIN || v > INT_MAX) return -1;
    *out = (int)v;
    return 0;
}

/* std::stof: strtof; nothing converted -> invalid_argument, ERANGE -> out_of_range; what() is "stof".
 * So "0.2f", "1e4f", "50.f" parse (the suffix is trailing text). */
static int parse_float(const char* s, float* out) {
    char* end;
    errno = 0;
    float v = strtof(s, &end);
    if (end == s) return -1;
    if (errno == ERANGE) return -1;
    *out = v;
    return 0;
}

typedef enum { K_INT, K_FLOAT, K_STRING } key_kind;

typedef struct key_desc {
    const char* file_key;     /* left-hand side in the file                                  */
    const char* echo_key;     /* spelling of the reference's echo line                       */
    const char* error_key;    /* spelling of the reference's "<x> invalid value: " line      */
    key_kind kind;
    size_t offset;
    int bit;
} key_desc;

#define OFF(f) offsetof(nbody_config, f)
/* Order of the if/else chain, include/nbodyConfig.h:36-221 */
static const key_desc k_keys[] = {
    {"particleCount", "particleCount", "particleCount", K_INT, OFF(particleCount), NBODY_KEY_particleCount},
    {"totalIterations", "totalIterations", "totalIterations", K_INT, OFF(totalIterations),
     NBODY_KEY_totalIterations},
    {"save_Image_Every_Xth_Iteration", "save_Image_Every_Xth_Iteration", "save_Image_Every_Xth_Iteration",
     K_INT, OFF(save_Image_Every_Xth_Iteration), NBODY_KEY_save_Image_Every_Xth_Iteration},
    {"timestep", "timestep", "timestep", K_FLOAT, OFF(timestep), NBODY_KEY_timestep},
    /* :101 echoes "minRandBodymass=" (lower-case m) */
    {"minRandBodyMass", "minRandBodymass", "minRandBodyMass", K_FLOAT, OFF(minRandBodyMass),
     NBODY_KEY_minRandBodyMass},
    {"maxRandBodyMass", "maxRandBodyMass", "maxRandBodyMass", K_FLOAT, OFF(maxRandBodyMass),
     NBODY_KEY_maxRandBodyMass},
    {"minRadius", "minRadius", "minRadius", K_FLOAT, OFF(minRadius), NBODY_KEY_minRadius},
    {"maxRadius", "maxRadius", "maxRadius", K_FLOAT, OFF(maxRadius), NBODY_KEY_maxRadius},
    {"imgWidth", "imgWidth", "imgWidth", K_INT, OFF(imgWidth), NBODY_KEY_imgWidth},
    {"imgHeight", "imgHeight", "imgHeight", K_INT, OFF(imgHeight), NBODY_KEY_imgHeight},
    {"fieldWidth", "fieldWidth", "fieldWidth", K_INT, OFF(fieldWidth), NBODY_KEY_fieldWidth},
    {"fieldHeight", "fieldHeight", "fieldHeight", K_INT, OFF(fieldHeight), NBODY_KEY_fieldHeight},
    {"imagePath", "imagePath", "imagePath", K_STRING, OFF(imagePath), NBODY_KEY_imagePath},
    /* :208-220 file key radiusGrowthRate, echo and error text say growthRate */
    {"radiusGrowthRate", "growthRate", "growthRate", K_FLOAT, OFF(growthRate), NBODY_KEY_radiusGrowthRate},
};
#undef OFF

static int handle_line(const char* line, size_t len, nbody_config* cfg, int fd) {
    /* :34-35  delimPos = line.find("="); variableName = line.substr(0, delimPos) */
    const char* eq = memchr(line, '=', len);
    size_t name_len = eq ? (size_t)(eq - line) : len;
    /* value = line.substr(delimPos + 1); with no '=' npos + 1 wraps to 0: the whole line */
    const char* val = eq ? eq + 1 : line;
    size_t val_len = eq ? len - name_len - 1 : len;

    for (size_t k = 0; k < sizeof(k_keys) / sizeof(k_keys[0]); ++k) {
        const key_desc* kd = &k_keys[k];
        if (strlen(kd->file_key) != name_len || memcmp(kd->file_key, line, name_len) != 0) continue;
        char* field = (char*)cfg + kd->offset;
        /* NUL-terminated copy of the value for strtol/strtof */
        char* tmp = (char*)malloc(val_len + 1);
        if (!tmp) return nbody_fail(NBODY_ERR_NOMEM, "config: out of memory");
        memcpy(tmp, val, val_len);
        tmp[val_len] = '\0';
        int rc = NBODY_OK;
        if (kd->kind == K_INT) {
            int v;
            if (parse_int(tmp, &v) != 0) {
                echo(fd, "%s invalid value: stoi\n", kd->error_key);
                rc = nbody_fail(NBODY_ERR_PARSE, "%s invalid value: stoi", kd->error_key);
            } else {
                echo(fd, "%s=%d\n", kd->echo_key, v);
                memcpy(field, &v, sizeof(v));
            }
        } else if (kd->kind == K_FLOAT) {
            float v;
            if (parse_float(tmp, &v) != 0) {
                echo(fd, "%s invalid value: stof\n", kd->error_key);
                rc = nbody_fail(NBODY_ERR_PARSE, "%s invalid value: stof", kd->error_key);
            } else {
                echo(fd, "%s=%g\n", kd->echo_key, (double)v);   /* operator<<(float): %g, precision 6 */
                memcpy(field, &v, sizeof(v));
            }
        } else {
            if (val_len >= NBODY_IMAGE_PATH_MAX) {
                free(tmp);
                return nbody_fail(NBODY_ERR_CAPACITY, "imagePath longer than %d bytes", NBODY_IMAGE_PATH_MAX - 1);
            }
            echo(fd, "%s=%s\n", kd->echo_key, tmp);
            memcpy(field, tmp, val_len + 1);
        }
        free(tmp);
        if (rc == NBODY_OK) cfg->present |= 1u << kd->bit;
        return rc;
    }
    /* :222-224 */
    echo(fd, "Invalid variable: %.*s\n", (int)name_len, line);
    return NBODY_OK;
}

int nbody_config_parse_fd(const char* path, nbody_config* out, int echo_fd) {
    if (!path || !out) return nbody_fail(NBODY_ERR_INVALID, "nbody_config_parse: NULL argument");
    memset(out, 0, sizeof(*out));
    FILE* f = fopen(path, "rb");
    if (!f) {
        echo(echo_fd, "Error opening config file! Exiting...\n");    /* :26 */
        return nbody_fail(NBODY_ERR_IO, "Error opening config file: %s", path);
    }
    size_t cap = 4096, len = 0;
    char* buf = (char*)malloc(cap);
    if (!buf) { fclose(f); return nbody_fail(NBODY_ERR_NOMEM, "config: out of memory"); }
    for (;;) {
        if (len == cap) {
            char* nb = (char*)realloc(buf, cap * 2);
            if (!nb) { free(buf); fclose(f); return nbody_fail(NBODY_ERR_NOMEM, "config: out of memory"); }
            buf = nb;
            cap *= 2;
        }
        size_t got = fread(buf + len, 1, cap - len, f);
        len += got;
        if (got == 0) break;
    }
    fclose(f);
    /* std::getline: lines end at '\n'; a final unterminated line counts; a final '\n' adds no empty line */
    int rc = NBODY_OK;
    size_t pos = 0;
    while (pos < len) {
        const char* nl = memchr(buf + pos, '\n', len - pos);
        size_t line_len = nl ? (size_t)(nl - (buf + pos)) : len - pos;
        rc = handle_line(buf + pos, line_len, out, echo_fd);
        if (rc != NBODY_OK) break;
        pos += line_len + 1;
    }
    free(buf);
    return rc;
}

int nbody_config_parse(const char* path, nbody_config* out) {
    fflush(stdout);
    return nbody_config_parse_fd(path, out, STDOUT_FILENO);
}

void nbody_config_stock(nbody_config* c) {
    /* nbodyConfig.txt:1-14 */
    memset(c, 0, sizeof(*c));
    c->particleCount = 16384;
    c->totalIterations = 2000;
    c->save_Image_Every_Xth_Iteration = 10;
    c->timestep = 0.2f;
    c->growthRate = 0.1f;
    c->minRandBodyMass = 1e4f;
    c->maxRandBodyMass = 1e17f;
    c->minRadius = 50.f;
    c->maxRadius = 200.f;
    c->imgWidth = 1024;
    c->imgHeight = 1024;
    c->fieldWidth = 100000;
    c->fieldHeight = 100000;
    strcpy(c->imagePath, "iter_img");
    c->present = (1u << NBODY_KEY_COUNT) - 1u;
}
