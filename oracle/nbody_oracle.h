/* oracle/nbody_oracle.h -- TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline), never the product path.
 *
 * CPU restatement ("port") of the reference's step semantics for the hot path:
 *   areParticlesColliding  /root/reference/src/nbody.cu:126-134
 *   ComputeForces          /root/reference/src/nbody.cu:139-271
 *   MoveBodies             /root/reference/src/nbody.cu:277-292
 *   launch geometry        /root/reference/src/nbody.cu:473,481-483
 *   host compaction        /root/reference/src/nbody.cu:488-510
 *   Vec2f rounding order   /root/reference/include/vec2f.h:45-93
 *
 * Pinning: the reference holds no tests, fixtures or golden vectors and this image has no nvcc.  What
 * pins this restatement is the reference's OWN device code: oracle/_ref/libnbody_ref_hip.so compiles it unmodified with
 * hipcc and runs it on the MI355X (oracle/ref_hip; tests/test_gpu_reference_kernels.py: bit-identical to this file and
 * to the product up to N = 262144), and oracle/_ref/libnbody_ref.so runs the same text on the CPU behind a shim for the
 * CUDA execution model (oracle/ref_shim; generated tests/golden/; bit-identical too).  The FP model (no contraction) is
 * our choice; the other plausible one (default FMA contraction) is bounded: <= 4.3e-7 per step, identical collision
 * outcomes.  The fp64 variant: the reference has no fp64 code; the pin is the reference's own kernel text with `float` read
 * as `double` (oracle/_ref/libnbody_ref_hip_f64.so, one macro in oracle/ref_hip), run on the GPU: bit-identical to the fp64
 * instantiation of this file and to the product up to N = 1048576 (tests/test_gpu_reference_kernels.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef NBODY_ORACLE_H
#define NBODY_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_LITERAL = 0, ORACLE_CLEAN = 1 };

typedef struct oracle_stats {
    int64_t pairs;      /* ordered (i,j) pairs evaluated by the stepper in this call        */
    int32_t n_absorb;   /* |E_t| : (i,j) with hit and Mi >= Mj               */
    int32_t n_deleted;  /* |D_t| : i with some hit and Mi <  Mj                              */
    int32_t n_active;   /* bodies that were updated (literal: i < (N/128)*128, src :473)     */
    int32_t n_after;    /* survivor count after compaction (full-step entry points only)     */
} oracle_stats;

/* One full step (forces+collisions, drift+commit, stable compaction) on a host block laid out as the
 * reference does (src/nbody.cu:66-77): [P vec2f[N] | V vec2f[N] | M f32[N] | R f32[N]], 24*N bytes.
 * *n becomes the survivor count and the block is re-carved for it.
 *   absorb_pairs : optional int32[2*absorb_cap], receives E_t as (i,j) in i-major, visit-order-minor order
 *   deleted      : optional int32[deleted_cap], receives D_t ascending
 *   pre_compaction : optional 24*N bytes, the block after the drift and before compaction
 * Returns 0, or -1 on bad arguments. */
int oracle_step_f32(void* block, int* n, float dt, int fieldW, int fieldH, float growth, int semantics,
                    int32_t* absorb_pairs, int absorb_cap, int32_t* deleted, int deleted_cap,
                    oracle_stats* stats, void* pre_compaction);

/* fp64 twin on [P vec2[N] | V vec2[N] | M f64[N] | R f64[N]], 48*N bytes (include/vec2.h layout). */
int oracle_step_f64(void* block, int* n, double dt, int fieldW, int fieldH, double growth, int semantics,
                    int32_t* absorb_pairs, int absorb_cap, int32_t* deleted, int deleted_cap,
                    oracle_stats* stats, void* pre_compaction);

/* Range form used for sharded tests and for the bounded cpu_baseline sample: computes the post-drift state
 * of bodies [lo,hi) from the step-start block WITHOUT modifying it and without compaction.
 * outP/outV are vec2 arrays of (hi-lo), outM/outR scalars of (hi-lo); del_flags (optional) uint8 of (hi-lo). */
int oracle_range_f32(const void* block, int n, int lo, int hi, float dt, int fieldW, int fieldH,
                     float growth, int semantics, float* outP, float* outV, float* outM, float* outR,
                     uint8_t* del_flags, oracle_stats* stats);
int oracle_range_f64(const void* block, int n, int lo, int hi, double dt, int fieldW, int fieldH,
                     double growth, int semantics, double* outP, double* outV, double* outM, double* outR,
                     uint8_t* del_flags, oracle_stats* stats);

/* generateImage (src/nbody.cu:294-348) with the missing `i < numBodies` guard: bodies i < min(n, blocks*128) are
 * rasterised as filled discs (value 0) into img[w*h], pre-set to 254 (:534).  `blocks` is the block count of the
 * launch (:535 passes the stale count of the step that produced the block).  fp32 arithmetic as written. */
void oracle_render_f32(const void* block, int n, int blocks, unsigned char* img, int w, int h, int fieldW,
                       int fieldH);

/* Number of ordered pairs the stepper evaluates in one step at body count n. */
int64_t oracle_pairs_per_step(int n, int semantics);

/* J(i) enumeration for tests of the quirk mask: writes the visit-ordered j list of body i, returns its
 * length (or -1 if i is inactive). out must hold n entries. */
int oracle_jlist(int n, int i, int semantics, int32_t* out);

void oracle_set_threads(int nthreads);   /* 0 = OpenMP default */
int oracle_get_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
