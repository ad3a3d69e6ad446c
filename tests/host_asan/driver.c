/* tests/host_asan/driver.c -- the C host code (config parser, body block helpers, RNG / initial condition, state
 * file peek, PGM writer) under AddressSanitizer + UBSan.  Built and run by tests/test_host_cpu.py; no GPU code is
 * linked (the sanitizers are for host code only).
 *   driver parse <file>      parse a config, print status and every field
 *   driver blocks            exercise alloc / carve / init / compact / rng on several sizes
 *   driver peek <file>       nbody_state_peek on an arbitrary file
 *   driver state <file> <p>  save / peek / load round trip and refusals through a host-only stand-in context
 *   driver pgm <file>        nbody_write_pgm of a small image
 *   driver partition         nbody_partition over many (n, world): covering, contiguous, block-aligned, level */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "nbody.h"

/* A host-only stand-in for the device context, so that the state file code (csrc/nbody_state.c) can run under
 * the sanitizers: it keeps one body block in RAM behind the four context entry points that code calls. */
struct nbody_ctx { nbody_ctx_desc d; void* block; int n; int64_t steps; };
int nbody_ctx_info(nbody_ctx* c, nbody_ctx_desc* desc_out, int64_t* steps) {
    if (desc_out) *desc_out = c->d;
    if (steps) *steps = c->steps;
    return NBODY_OK;
}
int nbody_download(nbody_ctx* c, void* block, int* n_out) {
    memcpy(block, c->block, nbody_block_bytes(c->n, c->d.precision));
    *n_out = c->n;
    return NBODY_OK;
}
int nbody_upload(nbody_ctx* c, const void* block, int n) {
    if (n > c->d.capacity) return NBODY_ERR_CAPACITY;
    memcpy(c->block, block, nbody_block_bytes(n, c->d.precision));
    c->n = n;
    return NBODY_OK;
}
int nbody_ctx_set_steps(nbody_ctx* c, int64_t steps) { c->steps = steps; return NBODY_OK; }

static int do_state(const char* path, int prec) {
    nbody_config cfg;
    nbody_config_stock(&cfg);
    cfg.particleCount = 777;
    nbody_ctx a, b;
    memset(&a, 0, sizeof(a));
    nbody_ctx_desc_from_config(&a.d, &cfg, prec);
    a.d.capacity = 1000;
    b = a;
    a.block = nbody_block_alloc(1000, prec);
    b.block = nbody_block_alloc(1000, prec);
    if (!a.block || !b.block || nbody_init_bodies(&cfg, a.block, prec) != NBODY_OK) return 1;
    a.n = 777; a.steps = 41;
    if (nbody_state_save(&a, path) != NBODY_OK) { printf("save: %s\n", nbody_last_error_string()); return 1; }
    int p2 = -1, n2 = -1; int64_t s2 = -1;
    if (nbody_state_peek(path, &p2, &n2, &s2) != NBODY_OK || p2 != prec || n2 != 777 || s2 != 41) return 1;
    if (nbody_state_load(&b, path) != NBODY_OK) { printf("load: %s\n", nbody_last_error_string()); return 1; }
    const int same = b.n == 777 && b.steps == 41 && !memcmp(a.block, b.block, nbody_block_bytes(777, prec));
    b.d.capacity = 100;                                    /* too small a context: must be refused, not overrun */
    const int rc_small = nbody_state_load(&b, path);
    b.d.capacity = 1000;
    b.d.precision = 1 - prec;                              /* wrong precision: refused */
    const int rc_prec = nbody_state_load(&b, path);
    b.d.precision = prec;
    FILE* f = fopen(path, "r+b");                          /* truncate the body block: refused */
    if (!f) return 1;
    fclose(f);
    if (truncate(path, 64 + 100) != 0) return 1;
    const int rc_trunc = nbody_state_load(&b, path);
    printf("state same=%d small=%d prec=%d trunc=%d\n", same, rc_small, rc_prec, rc_trunc);
    nbody_block_free(a.block);
    nbody_block_free(b.block);
    return 0;
}

static int do_parse(const char* path) {
    nbody_config c;
    memset(&c, 0x5a, sizeof(c));
    int rc = nbody_config_parse_fd(path, &c, -1);
    printf("rc=%d present=%u\n", rc, (unsigned)c.present);
    if (rc == NBODY_OK)
        printf("%d %d %d %.9g %.9g %.9g %.9g %.9g %.9g %d %d %d %d [%s]\n", c.particleCount, c.totalIterations,
               c.save_Image_Every_Xth_Iteration, c.timestep, c.minRandBodyMass, c.maxRandBodyMass, c.minRadius,
               c.maxRadius, c.growthRate, c.imgWidth, c.imgHeight, c.fieldWidth, c.fieldHeight, c.imagePath);
    else
        printf("error: %s\n", nbody_last_error_string());
    return 0;
}

static int do_blocks(void) {
    static const int sizes[] = {0, 1, 2, 127, 128, 129, 1000, 4097};
    for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); ++k) {
        for (int prec = 0; prec < 2; ++prec) {
            const int n = sizes[k];
            nbody_config cfg;
            nbody_config_stock(&cfg);
            cfg.particleCount = n;
            void* b = nbody_block_alloc(n, prec);
            if (!b) { printf("alloc failed n=%d\n", n); return 1; }
            if (nbody_init_bodies(&cfg, b, prec) != NBODY_OK) { printf("init failed: %s\n", nbody_last_error_string()); return 1; }
            int alive = n;
            if (prec == NBODY_F32) {
                nbody_vec2f *P, *V; float *M, *R;
                if (nbody_block_carve_f32(b, n, &P, &V, &M, &R) != NBODY_OK) return 1;
                for (int i = 0; i < n; i += 3) { M[i] = 0.f; --alive; }
                if (n) { P[n - 1].X += 1.f; V[n - 1].Y = R[n - 1]; }
            } else {
                nbody_vec2 *P, *V; double *M, *R;
                if (nbody_block_carve_f64(b, n, &P, &V, &M, &R) != NBODY_OK) return 1;
                for (int i = 0; i < n; i += 3) { M[i] = 0.0; --alive; }
                if (n) { P[n - 1].X += 1.0; V[n - 1].Y = R[n - 1]; }
            }
            const int got = nbody_block_compact(b, n, prec);
            if (got != alive) { printf("compact: %d != %d (n=%d)\n", got, alive, n); return 1; }
            nbody_block_free(b);
        }
    }
    nbody_rng g;
    nbody_rng_seed(&g, 1024);
    double acc = 0;
    for (int i = 0; i < 100000; ++i) acc += nbody_rng_fval_range(&g, -3.5, 1e17) * 1e-17 + (double)(nbody_rng_ival64(&g) & 1);
    printf("blocks ok %.6f num_blocks %d %d %d\n", acc, nbody_num_blocks(1), nbody_num_blocks(128), nbody_num_blocks(1000));
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "parse")) return do_parse(argv[2]);
    if (argc >= 2 && !strcmp(argv[1], "blocks")) return do_blocks();
    if (argc >= 3 && !strcmp(argv[1], "peek")) {
        int prec = -1, n = -1; int64_t steps = -1;
        int rc = nbody_state_peek(argv[2], &prec, &n, &steps);
        printf("rc=%d prec=%d n=%d steps=%lld\n", rc, prec, n, (long long)steps);
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "state")) return do_state(argv[2], atoi(argv[3]));
    if (argc >= 3 && !strcmp(argv[1], "pgm")) {
        unsigned char img[7 * 5];
        for (int i = 0; i < 35; ++i) img[i] = (unsigned char)(i * 7);
        printf("rc=%d\n", nbody_write_pgm(argv[2], img, 7, 5));
        return 0;
    }
    if (argc >= 2 && !strcmp(argv[1], "partition")) {
        long checked = 0;
        for (int n = 0; n <= 70000; n += (n < 600 ? 1 : 997))
            for (int world = 1; world <= 9; ++world) {
                int next = 0, lo = -1, cnt = -1, cmin = 1 << 30, cmax = 0;
                for (int g = 0; g < world; ++g) {
                    if (nbody_partition(n, g, world, &lo, &cnt) != NBODY_OK || lo != next || cnt < 0) return 1;
                    if (cnt > 0 && lo % 128 != 0) return 1;
                    next = lo + cnt;
                    if (cnt < cmin) cmin = cnt;
                    if (cnt > cmax) cmax = cnt;
                    ++checked;
                }
                if (next != n || cmax - cmin > 255) return 1;
            }
        int lo, cnt;
        if (nbody_partition(-1, 0, 1, &lo, &cnt) != NBODY_ERR_INVALID || nbody_partition(5, 2, 2, &lo, &cnt) != NBODY_ERR_INVALID ||
            nbody_partition(5, 0, 1, NULL, &cnt) != NBODY_ERR_INVALID)
            return 1;
        printf("partition ok (%ld ranges)\n", checked);
        return 0;
    }
    fprintf(stderr, "usage: driver parse|blocks|peek|pgm|partition ...\n");
    return 2;
}
