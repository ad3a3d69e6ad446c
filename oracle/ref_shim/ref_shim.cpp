// oracle/ref_shim/ref_shim.cpp -- TEST INFRASTRUCTURE, not product code.
//
// "Literal oracle": runs the reference's OWN device code (predicate + ComputeForces + MoveBodies,
// generateImage, /root/reference/src/nbody.cu:126-348) on the CPU, unmodified.  The kernel text is sliced out of the
// reference by line range AT BUILD TIME into a temporary file (see oracle/Makefile; the slice never enters
// this repository and is deleted after the build) and #included below as REF_SLICE.  The reference's own
// headers (vec2f.h, jbutil.h, nbodyConfig.h) are included from /root/reference/include by -I.
//
// CUDA execution model shim: one ucontext fiber per CUDA thread, one block at a time per OS thread;
// __syncthreads() yields to the block scheduler, which round-robins the fibers that have not returned, so
// every fiber reaches barrier k before any fiber leaves it.  Blocks are independent in ComputeForces (a
// thread only writes its own body's velocity / scratch slots and only reads other bodies' P, M, R, which
// ComputeForces never writes), so blocks are distributed over OpenMP threads.
//
// The host loop (scratch arrays, launch geometry, stable compaction) follows src/nbody.cu:463-510 and the
// initial-condition loop follows src/nbody.cu:401-416, using the reference's own jbutil::randgen.
//
// Build: oracle/Makefile target `ref` -> oracle/_ref/libnbody_ref.so (git-ignored, travels with gpurun).
#include <ucontext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <assert.h>
#include <unistd.h>
#include <vector>
#include <string>

#define __global__
#define __device__
#define __shared__ thread_local
#define THREADS_PER_BLOCK 128
#define GRAV_CONSTANT 6.67408e-11f

#include "vec2f.h"
#include "jbutil.h"
#include "nbodyConfig.h"

struct shim_dim3 { unsigned x, y, z; };
static thread_local shim_dim3 threadIdx, blockIdx, blockDim;
// dynamic shared memory of one block: the reference asks for 6144 B (src/nbody.cu:451); give it 16 KiB
thread_local Vec2f sharedMem[2048];

static void __syncthreads();

#include REF_SLICE   // /root/reference/src/nbody.cu lines 126-348, verbatim, from a temp file

// ------------------------------------------------------------------------------------------------------
// fiber scheduler
// ------------------------------------------------------------------------------------------------------
namespace {

struct Launch {
    void (*body)(void*);
    void* arg;
};

struct Fiber {
    ucontext_t ctx;
    bool done;
};

constexpr size_t kStack = 64 * 1024;
thread_local Fiber* t_fibers = nullptr;
thread_local char* t_stacks = nullptr;
thread_local ucontext_t t_main;
thread_local int t_cur = 0;
thread_local Launch t_launch;

void fiber_entry() {
    t_launch.body(t_launch.arg);
    t_fibers[t_cur].done = true;
    // uc_link returns to t_main
}

void run_block(unsigned block, unsigned nthreads, void (*body)(void*), void* arg) {
    if (!t_fibers) {
        t_fibers = new Fiber[1024];
        t_stacks = (char*)malloc(kStack * 1024);
    }
    t_launch.body = body;
    t_launch.arg = arg;
    blockIdx.x = block; blockIdx.y = blockIdx.z = 0;
    blockDim.x = nthreads; blockDim.y = blockDim.z = 1;
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = t_fibers[t];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = t_stacks + kStack * t;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = &t_main;
        f.done = false;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    bool any = true;
    while (any) {
        any = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = t_fibers[t];
            if (f.done) continue;
            t_cur = (int)t;
            threadIdx.x = t; threadIdx.y = threadIdx.z = 0;
            swapcontext(&t_main, &f.ctx);
            if (!f.done) any = true;
        }
    }
}

struct ForcesArgs {
    void* bodyData; float* updM; float* updR; int n; float dt; int fw; int fh; int nb; float growth;
};
void forces_body(void* p) {
    ForcesArgs* a = (ForcesArgs*)p;
    // updatedVelocities is a never-allocated pointer in the reference (src/nbody.cu:441,482) and unused
    ComputeForces(a->bodyData, a->updM, (Vec2f*)nullptr, a->updR, a->n, a->dt, a->fw, a->fh, a->nb, a->growth);
}
struct ImageArgs { void* bodyData; int n; char* img; int w, h, fw, fh; };
void image_body(void* p) {
    ImageArgs* a = (ImageArgs*)p;
    // the reference kernel has no `i < numBodies` guard (src/nbody.cu:302-310) and is launched with a stale
    // block count (:535): threads past the body count read out of bounds there.  The shim only runs threads
    // that have a body, which is the guarded behaviour SURVEY.md 8 f3 asks for.
    if ((int)(blockIdx.x * blockDim.x + threadIdx.x) >= a->n) return;
    generateImage(a->bodyData, a->n, a->img, a->w, a->h, a->fw, a->fh);
}
struct MoveArgs { void* bodyData; float* updM; float* updR; int n; float dt; };
void move_body(void* p) {
    MoveArgs* a = (MoveArgs*)p;
    MoveBodies(a->bodyData, a->updM, (Vec2f*)nullptr, a->updR, a->n, a->dt);
}

}  // namespace

static void __syncthreads() {
    swapcontext(&t_fibers[t_cur].ctx, &t_main);
}

// ------------------------------------------------------------------------------------------------------
// C entry points (ctypes / dlopen)
// ------------------------------------------------------------------------------------------------------
extern "C" {

// One literal step on a host block in the reference layout [P Vec2f[N] | V Vec2f[N] | M f32[N] | R f32[N]]
// (src/nbody.cu:66-77).  *n is updated to the survivor count and the block is re-carved for it, as the
// reference's newData copy does (src/nbody.cu:488-510).  If pre_compaction != NULL it receives a copy of
// the 24*N-byte block as it was after MoveBodies and before compaction (step-t index space).
int ref_step(void* block, int* n, float dt, int fieldW, int fieldH, float growth, void* pre_compaction) {
    int N = *n;
    if (N <= 0) return 0;
    Vec2f* P = (Vec2f*)block;
    Vec2f* V = P + N;
    float* M = (float*)(V + N);
    float* R = M + N;
    // src/nbody.cu:463-470: scratch pre-filled with current M, R
    std::vector<float> updM(M, M + N), updR(R, R + N);
    // src/nbody.cu:473
    int blocks = N < THREADS_PER_BLOCK ? 1 : N / THREADS_PER_BLOCK;
    ForcesArgs fa{block, updM.data(), updR.data(), N, dt, fieldW, fieldH, blocks, growth};
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < blocks; ++b) run_block((unsigned)b, THREADS_PER_BLOCK, forces_body, &fa);
    MoveArgs ma{block, updM.data(), updR.data(), N, dt};
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < blocks; ++b) run_block((unsigned)b, THREADS_PER_BLOCK, move_body, &ma);
    if (pre_compaction) memcpy(pre_compaction, block, (size_t)N * 24);
    // src/nbody.cu:488-510 stable compaction on `mass != 0.f`
    int newN = 0;
    for (int i = 0; i < N; ++i) if (M[i] != 0.f) ++newN;
    std::vector<char> tmp((size_t)newN * 24 + 8);
    Vec2f* nP = (Vec2f*)tmp.data();
    Vec2f* nV = nP + newN;
    float* nM = (float*)(nV + newN);
    float* nR = nM + newN;
    int k = 0;
    for (int i = 0; i < N; ++i) {
        if (M[i] != 0.f) { nP[k] = P[i]; nV[k] = V[i]; nM[k] = M[i]; nR[k] = R[i]; ++k; }
    }
    memcpy(block, tmp.data(), (size_t)newN * 24);
    *n = newN;
    return 0;
}

// generateImage (src/nbody.cu:294-348) on a host block of n bodies, launched with `blocks` blocks of 128 threads
// as at :535 (the caller passes the STALE block count of the step that produced the block); the image is
// pre-set to 254 as cudaMemsetAsync does at :534.
void ref_render(void* block, int n, int blocks, char* img, int w, int h, int fieldW, int fieldH) {
    memset(img, 254, (size_t)w * h);
    ImageArgs ia{block, n, img, w, h, fieldW, fieldH};
    for (int b = 0; b < blocks; ++b) run_block((unsigned)b, THREADS_PER_BLOCK, image_body, &ia);
}

// Initial conditions exactly as src/nbody.cu:401-416 (seed 1024, draws x,y,m,r per body, v = 0).
void ref_init_bodies(void* block, int n, int fieldW, int fieldH, float minMass, float maxMass,
                     float minRadius, float maxRadius) {
    Vec2f* P = (Vec2f*)block;
    Vec2f* V = P + n;
    float* M = (float*)(V + n);
    float* R = M + n;
    int doubleFieldWidth = fieldW << 1, doubleFieldHeight = fieldH << 1;
    jbutil::randgen gen;
    gen.seed(1024);
    float x, y, m, r;
    for (int i = 0; i < n; ++i) {
        x = gen.fval(0, doubleFieldWidth) - fieldW;
        y = gen.fval(0, doubleFieldHeight) - fieldH;
        m = gen.fval(minMass, maxMass);
        r = gen.fval(minRadius, maxRadius);
        P[i] = Vec2f(x, y);
        V[i] = Vec2f(0.f, 0.f);
        M[i] = m;
        R[i] = r;
    }
}

// Raw generator outputs for known-answer tests (include/jbutil.h:525-561).
void ref_rng_ival64(uint64_t seed, int count, uint64_t* out) {
    jbutil::randgen gen;
    gen.seed(seed);
    for (int i = 0; i < count; ++i) out[i] = gen.ival64();
}
void ref_rng_fval(uint64_t seed, int count, double a, double b, double* out) {
    jbutil::randgen gen;
    gen.seed(seed);
    for (int i = 0; i < count; ++i) out[i] = gen.fval(a, b);
}

struct ref_config {
    int particleCount, totalIterations, saveEvery;
    float timestep, minMass, maxMass, minRadius, maxRadius, growthRate;
    int imgWidth, imgHeight, fieldWidth, fieldHeight;
    char imagePath[256];
};

// Runs the reference's own parseConfigFile (include/nbodyConfig.h:22-227).  It echoes to stdout and calls
// exit(1) on errors, so callers that want to capture either run this in a child process.  Keys the file
// does not set come back as whatever the reference left in its uninitialised struct: golden generators only
// record the keys their test file sets.
void ref_parse_config(const char* path, ref_config* out) {
    ConfigData r = parseConfigFile(path);
    fflush(stdout);
    std::cout.flush();
    out->particleCount = r.particleCount; out->totalIterations = r.totalIterations;
    out->saveEvery = r.save_Image_Every_Xth_Iteration;
    out->timestep = r.timestep; out->minMass = r.minRandBodyMass; out->maxMass = r.maxRandBodyMass;
    out->minRadius = r.minRadius; out->maxRadius = r.maxRadius; out->growthRate = r.growthRate;
    out->imgWidth = r.imgWidth; out->imgHeight = r.imgHeight;
    out->fieldWidth = r.fieldWidth; out->fieldHeight = r.fieldHeight;
    memset(out->imagePath, 0, sizeof(out->imagePath));
    strncpy(out->imagePath, r.imagePath.c_str(), sizeof(out->imagePath) - 1);
}

int ref_sizeof_vec2f(void) { return (int)sizeof(Vec2f); }
int ref_alignof_vec2f(void) { return (int)alignof(Vec2f); }

}  // extern "C"
