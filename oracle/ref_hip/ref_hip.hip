// oracle/ref_hip/ref_hip.hip -- TEST INFRASTRUCTURE, not product code.
//
// "Reference on the GPU": the reference's OWN device code - areParticlesColliding, ComputeForces, MoveBodies
// (/root/reference/src/nbody.cu:126-292) with its three #defines (:35-37) and its own Vec2f header - compiled UNMODIFIED by
// this image's GPU compiler (hipcc, gfx950) and launched with the reference's own launch geometry and shared-memory
// size (:451,473,481-483).  Nothing of CUDA is stood in for: __global__, __shared__, threadIdx, __syncthreads and <<<>>>
// are HIP's own language features, executed by the MI355X's real thread blocks and barriers.  The kernel text is sliced
// out of the reference by line range AT BUILD TIME into a temporary file (oracle/Makefile; it never enters this
// repository) and #included below as REF_SLICE; vec2f.h comes from /root/reference/include by -I, with -D__CUDACC__ on the
// command line so that its CUDA_CALLABLE_MEMBER (include/vec2f.h:7-11) marks the members __host__ __device__.
//
// The host loop below is OURS: it restates src/nbody.cu:463-510 (scratch arrays, block count, two launches, D2H,
// stable host compaction) with HIP runtime calls.  Two builds (oracle/Makefile): -ffp-contract=off (one rounding per
// written operation: the oracle of record's FP model) and hipcc's default contraction (what `nvcc -O3`, cudaCmd.txt:1,
// also does by default: the FMA reading).
//
// A third build, REF_FLOAT_AS_DOUBLE (libnbody_ref_hip_f64.so): the fp64 READING of the same text.  The reference has no
// fp64 code at all (its templated include/vec2.h is never included, src/nbody.cu:15), yet BASELINE.json's fifth
// configuration is an fp64 stepper.  The closest thing to "the reference in fp64" that can exist without writing one is
// its own kernel text and its own Vec2f header with the token `float` read as `double` - ONE macro around the two
// #includes below, nothing else touched: every variable, array element, kernel parameter and Vec2f member becomes a
// double, sqrt() resolves to the double overload, and the float LITERALS of the text (GRAV_CONSTANT 6.67408e-11f,
// `1.0f / val` in vec2f.h:52) stay float literals and are widened where they are used - exactly the convention the product's
// fp64 path and the CPU oracle's fp64 instantiation follow.  System headers are included before the macro
// is defined.  This is an fp64 pin that does not pass through any line of the product or of oracle/nbody_oracle.c.
//
// Used by tests/test_gpu_reference_kernels.py (parity of the product against the reference's kernels at full size, on
// the GPU) and by bench.py's optional reference-timing leg.  Only tests/ and bench.py may load this library.
#include <hip/hip_runtime.h>
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <vector>

#ifdef REF_FLOAT_AS_DOUBLE
#define float double
#endif
#include "vec2f.h"

#include REF_SLICE   // /root/reference/src/nbody.cu lines 35-37 and 126-292, verbatim, from a temp file
#ifdef REF_FLOAT_AS_DOUBLE
#undef float
typedef double ref_real;
#else
typedef float ref_real;
#endif
static_assert(sizeof(Vec2f) == 2 * sizeof(ref_real), "Vec2f holds two reals of the build's precision");

namespace {
thread_local char g_err[256];
#define RH_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            snprintf(g_err, sizeof(g_err), "%s: %s", #expr, hipGetErrorString(e__));                        \
            return -1;                                                                                      \
        }                                                                                                   \
    } while (0)
}  // namespace

extern "C" const char* refhip_last_error(void) { return g_err; }

// `steps` iterations of the reference's loop body (src/nbody.cu:463-510) on a host block [P|V|M|R] of *n bodies.
// The block is updated in place and re-carved for the survivors after every step; *n becomes the survivor count.
// pre_compaction (optional, 24 * n_at_last_step bytes): the block after the last step's kernels, before its compaction.
// kernel_ms (optional): sum over the steps of the time of ComputeForces + MoveBodies measured with HIP events.
// (fp64 build: the block holds doubles, 48 bytes per body; timestep and growthRate are doubles.)
extern "C" int refhip_real_bytes(void) { return (int)sizeof(ref_real); }
extern "C" int refhip_run(void* host_block, int* n_io, int steps, ref_real timestep, int fieldWidth, int fieldHeight,
                          ref_real growthRate, void* pre_compaction, double* kernel_ms) {
    if (!host_block || !n_io || *n_io < 0 || steps < 0) return -2;
    int numBodies = *n_io;
    ref_real* block = (ref_real*)host_block;
    const int threadsPerBlock = THREADS_PER_BLOCK;                                              // :445
    const size_t sharedMemSize =
        threadsPerBlock * ((2 * (sizeof(Vec2f) + sizeof(ref_real) + sizeof(ref_real))) + 2 * sizeof(Vec2f));   // :451
    hipEvent_t e0, e1;
    RH_TRY(hipEventCreate(&e0));
    RH_TRY(hipEventCreate(&e1));
    double total_ms = 0.0;
    for (int iteration = 0; iteration < steps && numBodies > 0; ++iteration) {
        const size_t bytes = (size_t)numBodies * 6 * sizeof(ref_real);                          // :66
        void* d_block = nullptr;
        ref_real *d_updatedMasses = nullptr, *d_updatedRadii = nullptr;
        RH_TRY(hipMalloc(&d_block, bytes));                                                     // :93
        RH_TRY(hipMalloc((void**)&d_updatedMasses, numBodies * sizeof(ref_real)));              // :463
        RH_TRY(hipMalloc((void**)&d_updatedRadii, numBodies * sizeof(ref_real)));               // :464
        const int blocks = numBodies < threadsPerBlock ? 1 : numBodies / threadsPerBlock;       // :473
        RH_TRY(hipMemcpy(d_block, block, bytes, hipMemcpyHostToDevice));                        // :94
        // :467-470,477-478: the scratch arrays start as copies of the masses and radii
        RH_TRY(hipMemcpy(d_updatedMasses, block + 4 * (size_t)numBodies, numBodies * sizeof(ref_real), hipMemcpyHostToDevice));
        RH_TRY(hipMemcpy(d_updatedRadii, block + 5 * (size_t)numBodies, numBodies * sizeof(ref_real), hipMemcpyHostToDevice));
        RH_TRY(hipEventRecord(e0, 0));
        ComputeForces<<<blocks, threadsPerBlock, sharedMemSize, 0>>>(d_block, d_updatedMasses, (Vec2f*)nullptr, d_updatedRadii,
                                                                      numBodies, timestep, fieldWidth, fieldHeight, blocks,
                                                                      growthRate);                // :481-482
        MoveBodies<<<blocks, threadsPerBlock, 0, 0>>>(d_block, d_updatedMasses, (Vec2f*)nullptr, d_updatedRadii, numBodies,
                                                       timestep);                                 // :483
        RH_TRY(hipEventRecord(e1, 0));
        RH_TRY(hipGetLastError());
        RH_TRY(hipMemcpy(block, d_block, bytes, hipMemcpyDeviceToHost));                        // :486
        float ms = 0.f;
        RH_TRY(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        hipFree(d_block); hipFree(d_updatedMasses); hipFree(d_updatedRadii);                    // :84,541-542
        if (pre_compaction && iteration == steps - 1) memcpy(pre_compaction, block, bytes);
        // :488-510 stable compaction on mass != 0, block re-carved for the new count
        const ref_real* M = block + 4 * (size_t)numBodies;
        int newN = 0;
        for (int i = 0; i < numBodies; ++i) newN += M[i] != 0.f;
        if (newN != numBodies) {
            std::vector<ref_real> nb_((size_t)newN * 6);
            ref_real *nP = nb_.data(), *nV = nP + 2 * (size_t)newN, *nM = nV + 2 * (size_t)newN, *nR = nM + newN;
            const ref_real *P = block, *V = P + 2 * (size_t)numBodies, *R = M + numBodies;
            int k = 0;
            for (int i = 0; i < numBodies; ++i)
                if (M[i] != 0.f) {
                    nP[2 * k] = P[2 * i]; nP[2 * k + 1] = P[2 * i + 1];
                    nV[2 * k] = V[2 * i]; nV[2 * k + 1] = V[2 * i + 1];
                    nM[k] = M[i]; nR[k] = R[i];
                    ++k;
                }
            memcpy(block, nb_.data(), (size_t)newN * 6 * sizeof(ref_real));
            numBodies = newN;
        }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *n_io = numBodies;
    if (kernel_ms) *kernel_ms = total_ms;
    return 0;
}
