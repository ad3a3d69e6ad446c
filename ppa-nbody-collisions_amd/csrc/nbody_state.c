/* csrc/nbody_state.c -- state dump / restore over the C ABI (the reference keeps its state
 * only in RAM and has no checkpoint of any kind).  The payload is the reference's own body block
 * layout (src/nbody.cu:66-77) for the current survivors, so a dump is also a teacher-forcing fixture. */
#include "nbody.h"
#include "nbody_error.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct state_header {
    char magic[8];          /* "NBODYST1" */
    int32_t precision;
    int32_t n;
    int64_t steps;
    double timestep;
    double growthRate;
    int32_t fieldWidth;
    int32_t fieldHeight;
    int32_t semantics;
    int32_t reserved[3];
} state_header;

_Static_assert(sizeof(state_header) == 64, "state header is 64 bytes");

int nbody_state_save(nbody_ctx* ctx, const char* path) {
    if (!ctx || !path) return nbody_fail(NBODY_ERR_INVALID, "nbody_state_save: NULL argument");
    nbody_ctx_desc d;
    int64_t steps = 0;
    int rc = nbody_ctx_info(ctx, &d, &steps);
    if (rc != NBODY_OK) return rc;
    void* block = nbody_block_alloc(d.capacity, d.precision);
    if (!block) return NBODY_ERR_NOMEM;
    int n = 0;
    rc = nbody_download(ctx, block, &n);
    if (rc != NBODY_OK) { nbody_block_free(block); return rc; }
    state_header h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "NBODYST1", 8);
    h.precision = d.precision; h.n = n; h.steps = steps;
    h.timestep = d.timestep; h.growthRate = d.growthRate;
    h.fieldWidth = d.fieldWidth; h.fieldHeight = d.fieldHeight; h.semantics = d.semantics;
    FILE* f = fopen(path, "wb");
    if (!f) { nbody_block_free(block); return nbody_fail(NBODY_ERR_IO, "cannot open %s for writing", path); }
    const size_t bytes = nbody_block_bytes(n, d.precision);
    const int okw = fwrite(&h, sizeof(h), 1, f) == 1 && (bytes == 0 || fwrite(block, bytes, 1, f) == 1);
    const int okc = fclose(f) == 0;
    nbody_block_free(block);
    if (!okw || !okc) return nbody_fail(NBODY_ERR_IO, "short write to %s", path);
    return NBODY_OK;
}

static int read_header(FILE* f, const char* path, state_header* h) {
    if (fread(h, sizeof(*h), 1, f) != 1 || memcmp(h->magic, "NBODYST1", 8) != 0)
        return nbody_fail(NBODY_ERR_PARSE, "%s is not an nbody state file", path);
    if ((h->precision != NBODY_F32 && h->precision != NBODY_F64) || h->n < 0 || h->steps < 0)
        return nbody_fail(NBODY_ERR_PARSE, "%s: corrupt state header", path);
    return NBODY_OK;
}

int nbody_state_peek(const char* path, int* precision, int* n, int64_t* steps) {
    if (!path) return nbody_fail(NBODY_ERR_INVALID, "nbody_state_peek: NULL path");
    FILE* f = fopen(path, "rb");
    if (!f) return nbody_fail(NBODY_ERR_IO, "cannot open %s", path);
    state_header h;
    int rc = read_header(f, path, &h);
    fclose(f);
    if (rc != NBODY_OK) return rc;
    if (precision) *precision = h.precision;
    if (n) *n = h.n;
    if (steps) *steps = h.steps;
    return NBODY_OK;
}

int nbody_state_load(nbody_ctx* ctx, const char* path) {
    if (!ctx || !path) return nbody_fail(NBODY_ERR_INVALID, "nbody_state_load: NULL argument");
    nbody_ctx_desc d;
    int rc = nbody_ctx_info(ctx, &d, NULL);
    if (rc != NBODY_OK) return rc;
    FILE* f = fopen(path, "rb");
    if (!f) return nbody_fail(NBODY_ERR_IO, "cannot open %s", path);
    state_header h;
    rc = read_header(f, path, &h);
    if (rc != NBODY_OK) { fclose(f); return rc; }
    if (h.precision != d.precision) { fclose(f); return nbody_fail(NBODY_ERR_INVALID, "%s holds precision %d, context is %d", path, h.precision, d.precision); }
    if (h.n > d.capacity) { fclose(f); return nbody_fail(NBODY_ERR_CAPACITY, "%s holds %d bodies, context capacity is %d", path, h.n, d.capacity); }
    void* block = nbody_block_alloc(h.n, h.precision);
    if (!block) { fclose(f); return NBODY_ERR_NOMEM; }
    const size_t bytes = nbody_block_bytes(h.n, h.precision);
    const int okr = bytes == 0 || fread(block, bytes, 1, f) == 1;
    fclose(f);
    if (!okr) { nbody_block_free(block); return nbody_fail(NBODY_ERR_PARSE, "%s: truncated body block", path); }
    rc = nbody_upload(ctx, block, h.n);
    nbody_block_free(block);
    if (rc != NBODY_OK) return rc;
    return nbody_ctx_set_steps(ctx, h.steps);
}

/* saveImageToDisk, src/nbody.cu:350-371 */
int nbody_write_pgm(const char* path, const unsigned char* img, int width, int height) {
    if (!path || !img || width <= 0 || height <= 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_write_pgm: bad argument");
    printf("Saving (%dx%d) to disk\n", width, height);                         /* :356 */
    FILE* f = fopen(path, "wb");
    if (!f) {
        fprintf(stderr, "Error writing image to file:%s\nEnsure the the folder exists\n", path);   /* :367-368 */
        return nbody_fail(NBODY_ERR_IO, "Error writing image to file:%s", path);
    }
    fprintf(f, "P5\n%d %d\n255\n", width, height);                             /* :359 */
    const size_t bytes = (size_t)width * height;
    const int ok = fwrite(img, 1, bytes, f) == bytes;
    if (fclose(f) != 0 || !ok) return nbody_fail(NBODY_ERR_IO, "short write to %s", path);
    return NBODY_OK;
}
