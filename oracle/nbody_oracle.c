/* oracle/nbody_oracle.c -- TEST INFRASTRUCTURE ONLY.  See nbody_oracle.h for scope, citations and pinning. */
#include "nbody_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Bodies that have a thread: blocks = N<128 ? 1 : N/128 of 128 threads, guarded by i < N
 * (/root/reference/src/nbody.cu:473,142-143).  Clean semantics updates every body. */
static int oracle_active_count(int n, int semantics) {
    if (semantics != ORACLE_LITERAL) return n;
    const int nb = n < 128 ? 1 : n / 128;
    const int64_t t = (int64_t)nb * 128;
    return t < n ? (int)t : n;
}

#define REAL float
#define FN(x) x##_f32
#define RSQRT_ sqrtf
#include "nbody_oracle_step.inc"
#undef REAL
#undef FN
#undef RSQRT_

#define REAL double
#define FN(x) x##_f64
#define RSQRT_ sqrt
#include "nbody_oracle_step.inc"
#undef REAL
#undef FN
#undef RSQRT_

void oracle_render_f32(const void* block, int n, int blocks, unsigned char* img, int w, int h, int fieldW,
                       int fieldH) {
    const float* P = (const float*)block;
    const float* R = P + 5 * (size_t)n;
    memset(img, 254, (size_t)w * h);                                   /* src/nbody.cu:534 */
    const int img_width = w, img_height = h;
    const int doubleFieldWidth = fieldW << 1, doubleFieldHeight = fieldH << 1;   /* :314-315 */
    const long long lim = (long long)blocks * 128;
    for (int i = 0; i < n && i < lim; ++i) {
        const float px = P[2 * i], py = P[2 * i + 1];
        const float r = (R[i] * w) / fieldW;                            /* :310 */
        const int xc = (int)(((px + fieldW) / doubleFieldWidth) * img_width);    /* :318 */
        const int yc = (int)(((py + fieldH) / doubleFieldHeight) * img_height);  /* :319 */
        const int y_min = yc - r < 0 ? 0 : yc - r;                      /* :323 */
        const int y_max = yc + r >= img_height ? img_height : yc + r;   /* :324 */
        const int x_min = xc - r < 0 ? 0 : xc - r;                      /* :325 */
        const int x_max = xc + r > img_width ? img_width : xc + r;      /* :326 */
        for (int y = y_min; y < y_max; ++y) {                           /* :328 (the clamps at :330-331,334-335 */
            for (int x = x_min; x < x_max; ++x) {                       /*  cannot fire inside these bounds)    */
                const int x_sq = (x - xc) * (x - xc), y_sq = (y - yc) * (y - yc);   /* :336-337 */
                if (x_sq + y_sq <= (int)(r * r)) img[(size_t)img_width * y + x] = 0; /* :338-344 */
            }
        }
    }
}

int oracle_jlist(int n, int i, int semantics, int32_t* out) {
    if (n <= 0 || i < 0 || i >= n || !out) return -1;
    if (i >= oracle_active_count(n, semantics)) return -1;
    int len = 0;
    if (semantics == ORACLE_LITERAL) {
        ORACLE_VISIT_LITERAL(n, i, { out[len++] = j; });
    } else {
        ORACLE_VISIT_CLEAN(n, i, { out[len++] = j; });
    }
    return len;
}

int64_t oracle_pairs_per_step(int n, int semantics) {
    if (n <= 0) return 0;
    if (semantics != ORACLE_LITERAL) return (int64_t)n * (n - 1);
    /* per active body: 128*(nb-1) + L visited entries, minus the one skipped at k=0,off=0 when L_0 > 0 */
    const int nb = n < 128 ? 1 : n / 128;
    const int L = n % 129;
    const int64_t per = (nb == 1) ? (L > 0 ? L - 1 : 0) : (int64_t)128 * (nb - 1) + L - 1;
    return per * oracle_active_count(n, semantics);
}

void oracle_set_threads(int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
}

int oracle_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
