/* csrc/nbody_bodies.c -- host side of the body container and the initial-condition generator.
 *   block layout / carving   src/nbody.cu:63-79 (BodiesData::alloc)
 *   stable compaction        src/nbody.cu:488-510
 *   random generator         include/jbutil.h:514-562 (jbutil::randgen, a Numerical-Recipes "Ranq"-family
 *                            64-bit generator) -- re-derived from its recurrences, KATs in tests/golden
 *   initial conditions       src/nbody.cu:401-416
 */
#include "nbody.h"
#include "nbody_error.h"
#include "nbody_partition.h"
#include <stdlib.h>
#include <string.h>

size_t nbody_block_bytes(int n, int precision) {
    if (n < 0) return 0;
    return (size_t)n * (precision == NBODY_F64 ? 48u : 24u);   /* :66 */
}

void* nbody_block_alloc(int n, int precision) {
    if (n < 0) { nbody_fail(NBODY_ERR_INVALID, "nbody_block_alloc: negative count"); return NULL; }
    size_t bytes = nbody_block_bytes(n, precision);
    void* p = malloc(bytes ? bytes : 1);
    if (!p) nbody_fail(NBODY_ERR_NOMEM, "Failed to allocate body data");   /* :70 */
    return p;
}

void nbody_block_free(void* block) { free(block); }

int nbody_block_carve_f32(void* block, int n, nbody_vec2f** P, nbody_vec2f** V, float** M, float** R) {
    if (!block || n < 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_block_carve_f32: bad argument");
    nbody_vec2f* p = (nbody_vec2f*)block;    /* :74 */
    nbody_vec2f* v = p + n;                  /* :75 */
    float* m = (float*)(v + n);              /* :76 */
    float* r = m + n;                        /* :77 */
    if (P) *P = p;
    if (V) *V = v;
    if (M) *M = m;
    if (R) *R = r;
    return NBODY_OK;
}

int nbody_block_carve_f64(void* block, int n, nbody_vec2** P, nbody_vec2** V, double** M, double** R) {
    if (!block || n < 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_block_carve_f64: bad argument");
    nbody_vec2* p = (nbody_vec2*)block;
    nbody_vec2* v = p + n;
    double* m = (double*)(v + n);
    double* r = m + n;
    if (P) *P = p;
    if (V) *V = v;
    if (M) *M = m;
    if (R) *R = r;
    return NBODY_OK;
}

int nbody_block_compact(void* block, int n, int precision) {
    if (!block || n < 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_block_compact: bad argument");
    /* Every destination lies at or below its source, arrays are moved in layout order, so the forward
     * element-wise copy is safe in place. */
    int k = 0;
    if (precision == NBODY_F64) {
        nbody_vec2 *P, *V; double *M, *R;
        nbody_block_carve_f64(block, n, &P, &V, &M, &R);
        int newN = 0;
        for (int i = 0; i < n; ++i) if (M[i] != 0.0) ++newN;      /* :488-494 */
        unsigned char* keep = (unsigned char*)malloc((size_t)n + 1);
        if (!keep) return nbody_fail(NBODY_ERR_NOMEM, "nbody_block_compact: out of memory");
        for (int i = 0; i < n; ++i) keep[i] = M[i] != 0.0;
        nbody_vec2 *nP, *nV; double *nM, *nR;
        nbody_block_carve_f64(block, newN, &nP, &nV, &nM, &nR);
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nP[k++] = P[i];
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nV[k++] = V[i];
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nM[k++] = M[i];
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nR[k++] = R[i];
        free(keep);
        return newN;
    } else {
        nbody_vec2f *P, *V; float *M, *R;
        nbody_block_carve_f32(block, n, &P, &V, &M, &R);
        int newN = 0;
        for (int i = 0; i < n; ++i) if (M[i] != 0.f) ++newN;
        unsigned char* keep = (unsigned char*)malloc((size_t)n + 1);
        if (!keep) return nbody_fail(NBODY_ERR_NOMEM, "nbody_block_compact: out of memory");
        for (int i = 0; i < n; ++i) keep[i] = M[i] != 0.f;
        nbody_vec2f *nP, *nV; float *nM, *nR;
        nbody_block_carve_f32(block, newN, &nP, &nV, &nM, &nR);
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nP[k++] = P[i];
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nV[k++] = V[i];
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nM[k++] = M[i];
        k = 0; for (int i = 0; i < n; ++i) if (keep[i]) nR[k++] = R[i];
        free(keep);
        return newN;
    }
}

/* ------------------------------------------------------------------------------------------------------
 * Generator: three 64-bit state words; an LCG (u), a 64-bit xorshift (v) and a multiply-with-carry (w),
 * combined per draw (include/jbutil.h:535-552).
 * ---------------------------------------------------------------------------------------------------- */
static inline void rng_advance(nbody_rng* g) {
    g->u = g->u * 2862933555777941757ULL + 7046029254386353087ULL;
    g->v ^= g->v >> 17;
    g->v ^= g->v << 31;
    g->v ^= g->v >> 8;
    g->w = 4294957665ULL * (g->w & 0xffffffffULL) + (g->w >> 32);
}

uint64_t nbody_rng_ival64(nbody_rng* g) {
    rng_advance(g);
    uint64_t x = g->u ^ (g->u << 21);
    x ^= x >> 35;
    x ^= x << 4;
    return (x + g->v) ^ g->w;
}

void nbody_rng_seed(nbody_rng* g, uint64_t s) {
    g->v = 4101842887655102017ULL;
    g->w = 1;
    g->u = s ^ g->v;
    nbody_rng_ival64(g);
    g->v = g->u;
    nbody_rng_ival64(g);
    g->w = g->v;
    nbody_rng_ival64(g);
}

double nbody_rng_fval(nbody_rng* g) { return 5.42101086242752217E-20 * (double)nbody_rng_ival64(g); }

double nbody_rng_fval_range(nbody_rng* g, double a, double b) { return nbody_rng_fval(g) * (b - a) + a; }

int nbody_init_bodies(const nbody_config* cfg, void* block, int precision) {
    if (!cfg || !block || cfg->particleCount < 0)
        return nbody_fail(NBODY_ERR_INVALID, "nbody_init_bodies: bad argument");
    const int n = cfg->particleCount;
    const int fieldWidth = cfg->fieldWidth, fieldHeight = cfg->fieldHeight;
    const int doubleFieldWidth = fieldWidth << 1;       /* src/nbody.cu:388 */
    const int doubleFieldHeight = fieldHeight << 1;     /* :390 */
    nbody_rng gen;
    nbody_rng_seed(&gen, 1024);                         /* :403 */
    if (precision == NBODY_F64) {
        nbody_vec2 *P, *V; double *M, *R;
        nbody_block_carve_f64(block, n, &P, &V, &M, &R);
        for (int i = 0; i < n; ++i) {
            /* the draws are double already (jbutil.h:553-560); fp64 keeps them unrounded */
            double x = nbody_rng_fval_range(&gen, 0, doubleFieldWidth) - fieldWidth;
            double y = nbody_rng_fval_range(&gen, 0, doubleFieldHeight) - fieldHeight;
            double m = nbody_rng_fval_range(&gen, cfg->minRandBodyMass, cfg->maxRandBodyMass);
            double r = nbody_rng_fval_range(&gen, cfg->minRadius, cfg->maxRadius);
            P[i].X = x; P[i].Y = y; V[i].X = 0.0; V[i].Y = 0.0; M[i] = m; R[i] = r;
        }
    } else {
        nbody_vec2f *P, *V; float *M, *R;
        nbody_block_carve_f32(block, n, &P, &V, &M, &R);
        for (int i = 0; i < n; ++i) {
            float x = (float)(nbody_rng_fval_range(&gen, 0, doubleFieldWidth) - fieldWidth);     /* :408 */
            float y = (float)(nbody_rng_fval_range(&gen, 0, doubleFieldHeight) - fieldHeight);   /* :409 */
            float m = (float)nbody_rng_fval_range(&gen, cfg->minRandBodyMass, cfg->maxRandBodyMass); /* :410 */
            float r = (float)nbody_rng_fval_range(&gen, cfg->minRadius, cfg->maxRadius);         /* :411 */
            P[i].X = x; P[i].Y = y;        /* :412 */
            V[i].X = 0.f; V[i].Y = 0.f;    /* :413 */
            M[i] = m;                      /* :414 */
            R[i] = r;                      /* :415 */
        }
    }
    return NBODY_OK;
}

void nbody_ctx_desc_from_config(nbody_ctx_desc* d, const nbody_config* cfg, int precision) {
    memset(d, 0, sizeof(*d));
    d->precision = precision;
    d->semantics = NBODY_LITERAL;
    d->capacity = cfg->particleCount;
    d->device = 0;
    d->rank = 0;
    d->world = 1;
    d->timestep = cfg->timestep;
    d->growthRate = cfg->growthRate;
    d->fieldWidth = cfg->fieldWidth;
    d->fieldHeight = cfg->fieldHeight;
}

int nbody_num_blocks(int numBodies) {
    return numBodies < 128 ? 1 : numBodies / 128;     /* src/nbody.cu:473 */
}

/* The partition rule as a C entry point (csrc/nbody_partition.h). */
int nbody_partition(int n, int rank, int world, int* lo, int* cnt) {
    if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !cnt)
        return nbody_fail(NBODY_ERR_INVALID, "nbody_partition: bad argument");
    nbody_own_range_of(n, rank, world, lo, cnt);
    return NBODY_OK;
}
