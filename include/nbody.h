/* include/nbody.h -- C ABI of the MI355X-native direct N-body gravity + collision stepper.
 *
 * Drop-in boundary for the hot path of Aidan900/ppa-nbody-collisions.  The reference has no plugin / FFI
 * layer: main() (src/nbody.cu:373-551) calls its kernels directly.  Each entry point below names the
 * reference code it replaces (paths are relative to the reference tree).  Plain C types only; no
 * exceptions, no exit(): every function returns an nbody_status (0 = OK, negative = error) and
 * nbody_last_error_string() describes the last failure on the calling thread.
 *
 * Library: ppa-nbody-collisions_amd/libnbody_mi355x.so (built by __graft_entry__.build() / csrc/Makefile).
 * All compute entry points need a gfx950 device and FAIL (NBODY_ERR_NO_DEVICE / NBODY_ERR_HIP) without one:
 * there is no CPU fallback anywhere in the library.
 */
#ifndef NBODY_MI355X_H
#define NBODY_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBODY_ABI_VERSION 2

/* ---------------------------------------------------------------------------------------------------
 * Status codes
 * ------------------------------------------------------------------------------------------------- */
typedef enum nbody_status {
    NBODY_OK = 0,
    NBODY_ERR_INVALID = -1,      /* bad argument                                                       */
    NBODY_ERR_IO = -2,           /* config file cannot be opened  (include/nbodyConfig.h:25-28)        */
    NBODY_ERR_PARSE = -3,        /* "<key> invalid value"          (include/nbodyConfig.h:41-45 etc.)  */
    NBODY_ERR_NOMEM = -4,        /* host or device allocation failed (src/nbody.cu:68-72)              */
    NBODY_ERR_NO_DEVICE = -5,    /* no gfx950 device visible                                           */
    NBODY_ERR_HIP = -6,          /* a HIP runtime call failed (replaces CUDA_SYNC_CHECK, :20-33)       */
    NBODY_ERR_CAPACITY = -7,     /* more bodies / events than the context was created for             */
    NBODY_ERR_COMM = -8,         /* RCCL failure or collective library unavailable                     */
    NBODY_ERR_STATE = -9         /* call sequence error (e.g. step before upload)                      */
} nbody_status;

const char* nbody_last_error_string(void);
const char* nbody_status_string(int status);
int nbody_abi_version(void);

/* ---------------------------------------------------------------------------------------------------
 * Body layout -- include/vec2f.h:13-20 (Vec2f: 8 bytes, align 4) and include/vec2.h:6-17 (Vec2<double>).
 * A host "block" is ONE allocation [Positions vec2[N] | Velocities vec2[N] | Masses real[N] | Radii real[N]]
 * exactly as BodiesData::alloc carves it (src/nbody.cu:63-79): 24*N bytes (fp32) or 48*N bytes (fp64).
 * ------------------------------------------------------------------------------------------------- */
typedef struct nbody_vec2f { float X, Y; } nbody_vec2f;
typedef struct nbody_vec2 { double X, Y; } nbody_vec2;

typedef enum nbody_precision { NBODY_F32 = 0, NBODY_F64 = 1 } nbody_precision;

size_t nbody_block_bytes(int n, int precision);
/* BodiesData::alloc (src/nbody.cu:63-79) / freeData (:81-86), host part. */
void* nbody_block_alloc(int n, int precision);
void nbody_block_free(void* block);
/* Pointer carving of src/nbody.cu:74-77 (and :147-150, :283-286). Any output pointer may be NULL. */
int nbody_block_carve_f32(void* block, int n, nbody_vec2f** P, nbody_vec2f** V, float** M, float** R);
int nbody_block_carve_f64(void* block, int n, nbody_vec2** P, nbody_vec2** V, double** M, double** R);
/* Stable compaction on `mass != 0` with re-carving for the new count (src/nbody.cu:488-510). Host side
 * utility for callers that step through the reference-shaped launches below. Returns the new count or <0. */
int nbody_block_compact(void* block, int n, int precision);

/* ---------------------------------------------------------------------------------------------------
 * nbodyConfig.txt -- include/nbodyConfig.h:4-19 (struct ConfigData) and :22-227 (parseConfigFile)
 * ------------------------------------------------------------------------------------------------- */
#define NBODY_IMAGE_PATH_MAX 1024

enum { /* bit k of nbody_config.present is set when key k was accepted */
    NBODY_KEY_particleCount = 0, NBODY_KEY_totalIterations, NBODY_KEY_save_Image_Every_Xth_Iteration,
    NBODY_KEY_timestep, NBODY_KEY_minRandBodyMass, NBODY_KEY_maxRandBodyMass, NBODY_KEY_minRadius,
    NBODY_KEY_maxRadius, NBODY_KEY_radiusGrowthRate, NBODY_KEY_imgWidth, NBODY_KEY_imgHeight,
    NBODY_KEY_fieldWidth, NBODY_KEY_fieldHeight, NBODY_KEY_imagePath, NBODY_KEY_COUNT
};

typedef struct nbody_config {
    int particleCount;
    int totalIterations;
    int save_Image_Every_Xth_Iteration;
    float timestep;
    float minRandBodyMass;
    float maxRandBodyMass;
    float minRadius;
    float maxRadius;
    float growthRate;               /* file key "radiusGrowthRate" (include/nbodyConfig.h:13,208-220) */
    int imgWidth;
    int imgHeight;
    int fieldWidth;
    int fieldHeight;
    char imagePath[NBODY_IMAGE_PATH_MAX];
    uint32_t present;               /* deviation: the reference leaves missing keys uninitialised;  */
                                    /* we zero them and record which keys were seen                 */
} nbody_config;

/* parseConfigFile (include/nbodyConfig.h:22-227): same grammar (`key=value` per line, std::stoi/std::stof
 * number syntax so `0.2f`, `1e4f`, `50.f` parse), same echo text on stdout including the reference's
 * `minRandBodymass=` / `growthRate=` spellings and `Invalid variable: <name>` for unknown lines.  Instead of
 * exit(1) it returns NBODY_ERR_IO / NBODY_ERR_PARSE after printing the reference's message. */
int nbody_config_parse(const char* path, nbody_config* out);
/* Same, echo written to file descriptor echo_fd (-1: no echo). */
int nbody_config_parse_fd(const char* path, nbody_config* out, int echo_fd);
/* The stock nbodyConfig.txt values (nbodyConfig.txt:1-14). */
void nbody_config_stock(nbody_config* out);

/* ---------------------------------------------------------------------------------------------------
 * Initial conditions -- src/nbody.cu:401-416 with jbutil::randgen (include/jbutil.h:514-562)
 * ------------------------------------------------------------------------------------------------- */
typedef struct nbody_rng { uint64_t u, v, w; } nbody_rng;
void nbody_rng_seed(nbody_rng* g, uint64_t s);            /* randgen::seed   jbutil.h:525-534 */
uint64_t nbody_rng_ival64(nbody_rng* g);                  /* randgen::ival64 jbutil.h:545-552 */
double nbody_rng_fval(nbody_rng* g);                      /* randgen::fval() jbutil.h:553-556 */
double nbody_rng_fval_range(nbody_rng* g, double a, double b); /* fval(a,b)  jbutil.h:557-560 */

/* Fills a block of cfg->particleCount bodies: seed 1024, draws x,y,m,r per body in that order, v = 0.
 * fp32 rounds each draw to float exactly where the reference does; fp64 keeps the double draws. */
int nbody_init_bodies(const nbody_config* cfg, void* block, int precision);

/* ---------------------------------------------------------------------------------------------------
 * Stepper context -- replaces the per-iteration host loop src/nbody.cu:460-545 (cudaMalloc scratch,
 * H2D, ComputeForces, MoveBodies, D2H, host compaction) with device-resident state.
 * ------------------------------------------------------------------------------------------------- */
typedef enum nbody_semantics {
    NBODY_LITERAL = 0,  /* exactly what src/nbody.cu computes, index quirks included */
    NBODY_CLEAN = 1     /* every body active, true all-pairs, j ascending                                */
} nbody_semantics;

enum { /* nbody_ctx_desc.flags */
    NBODY_FLAG_RECORD_EVENTS = 1u << 0,   /* keep the collision event log (E_t, D_t of SURVEY.md A.2)   */
    NBODY_FLAG_GROUP_EXCHANGE = 1u << 1,  /* world>1, every rank is a context of this process: the       */
                                          /* exchange is done by nbody_group_step with peer copies        */
    NBODY_FLAG_FORCE_COMM = 1u << 2       /* create the RCCL communicator and run the slot all-gather    */
                                          /* even when world == 1 (exercises the multi-rank path on one  */
                                          /* GPU)                                                         */
};

typedef struct nbody_ctx nbody_ctx;

typedef struct nbody_ctx_desc {
    int precision;          /* nbody_precision                                                          */
    int semantics;          /* nbody_semantics                                                          */
    int capacity;           /* max bodies (>= first upload's n)                                         */
    int device;             /* HIP device ordinal                                                       */
    int rank, world;        /* range partition of bodies over ranks; world = 1 for a single GPU         */
    uint32_t flags;
    int event_capacity;     /* max logged events (0: default)                                           */
    double timestep;        /* cfg.timestep  (kernel arg `timestep`,  src/nbody.cu:482)                 */
    double growthRate;      /* cfg.growthRate (kernel arg `growthRate`, :482)                           */
    int fieldWidth;         /* kernel args :482                                                         */
    int fieldHeight;
    const void* comm_id;    /* world>1 with RCCL: 128-byte id from nbody_comm_unique_id on rank 0       */
    int kernel_variant;     /* 0 = automatic (by own-range size); tuning / A-B testing only: 1 general kernel,  */
                            /* 11/12/14/18 one-lane..eight-lanes-per-body kernel, 31/32 its 256-thread form,    */
                            /* 50/52/54 ring-of-waves kernel with 2x8 / 4x4 / 1x8 (rings x waves) workgroups    */
                            /* (53, 55, 56, 58: its tuning forms).  fp64: 1 selects the general kernel, anything else */
                            /* the fp64 production kernel                                                       */
} nbody_ctx_desc;

void nbody_ctx_desc_from_config(nbody_ctx_desc* d, const nbody_config* cfg, int precision);

int nbody_ctx_create(nbody_ctx** out, const nbody_ctx_desc* desc);
int nbody_ctx_destroy(nbody_ctx* ctx);

/* BodiesData::uploadToDevice (src/nbody.cu:88-96): copies a host block of n bodies (the FULL set on every
 * rank) to the device; the context keeps it resident between steps. */
int nbody_upload(nbody_ctx* ctx, const void* block, int n);
/* nsteps iterations of the loop body src/nbody.cu:463-510 (forces+collisions, drift+commit, stable
 * compaction), asynchronous: returns after enqueueing. */
int nbody_step(nbody_ctx* ctx, int nsteps);
/* cudaMemcpyAsync D2H of the state (src/nbody.cu:486) + the survivors' re-carved block (:496-510):
 * writes 24*n (48*n) bytes laid out for the CURRENT count n and stores n. block must hold `capacity`.
 * On an RCCL context (world > 1) this is a COLLECTIVE: velocities live only on their owner, every rank must
 * call it (the same holds for nbody_state_save, which downloads). */
int nbody_download(nbody_ctx* ctx, void* block, int* n);
int nbody_body_count(nbody_ctx* ctx, int* n);   /* synchronises */
int nbody_sync(nbody_ctx* ctx);                 /* CUDA_SYNC_CHECK (src/nbody.cu:20-33,546)           */
/* Context introspection used by the state files. */
int nbody_ctx_info(nbody_ctx* ctx, nbody_ctx_desc* desc_out, int64_t* steps);
int nbody_ctx_set_steps(nbody_ctx* ctx, int64_t steps);

typedef struct nbody_event {  /* one collision event, in the index space of the step it happened in    */
    int32_t step;             /* step counter since upload (0-based)                                   */
    int32_t i;                /* the body whose thread saw the collision                               */
    int32_t j;                /* the other body; j < 0 is never produced                               */
    int32_t kind;             /* 0: i absorbs j (E_t)   1: i deleted because of j (D_t witness)        */
} nbody_event;
/* Copies up to cap logged events (unordered within a step) and the total number logged since the last
 * clear; total > cap means the caller's buffer was too small, total > event_capacity means the log
 * overflowed (extra events were counted, not stored). */
int nbody_get_events(nbody_ctx* ctx, nbody_event* out, int cap, int64_t* total);
int nbody_clear_events(nbody_ctx* ctx);

typedef struct nbody_stats {
    int64_t steps;            /* steps enqueued since upload                                           */
    int64_t pairs;            /* ordered (i,j) pairs evaluated by THIS rank since upload (device count) */
    double force_kernel_ms;   /* sum of force-kernel durations measured with HIP events (0 if off)      */
    int64_t force_kernel_launches;
    int n_bodies;             /* current global body count                                             */
    int n_own;                /* bodies owned by this rank                                             */
    double exchange_ms;       /* sum of the per-step slot all-gather durations, HIP events (0 if off or no exchange) */
    int64_t exchange_launches;
    int64_t exchange_bytes;   /* bytes this rank received through all-gathers since upload              */
    int64_t slot_bytes_now;   /* bytes ONE rank contributes to the next step's all-gather (0: no exchange): follows */
                              /* the live body count, not the capacity                                   */
} nbody_stats;
int nbody_get_stats(nbody_ctx* ctx, nbody_stats* out);  /* synchronises */
/* Bracket every force-kernel launch with HIP events on the context's stream (bench / profiling). */
int nbody_set_kernel_timing(nbody_ctx* ctx, int enable);
/* Which force kernel the next step of this context launches (static string; reporting only). */
const char* nbody_force_kernel_name(nbody_ctx* ctx);

/* Image output.  nbody_render_image = cudaMemsetAsync(254) + generateImage + D2H
 * (src/nbody.cu:531-537, kernel :294-348): bodies drawn as filled discs of value 0 into img[width*height]; the
 * kernel's missing `i < numBodies` guard is present.  On a multi-rank context every rank can render (the
 * replica holds all positions and radii).  nbody_write_pgm = saveImageToDisk (:350-371): prints
 * "Saving (WxH) to disk", writes "P5\nW H\n255\n" + bytes; where the reference prints its error and exit(1)s
 * it prints the same text to stderr and returns NBODY_ERR_IO. */
int nbody_render_image(nbody_ctx* ctx, unsigned char* img, int width, int height);
int nbody_write_pgm(const char* path, const unsigned char* img, int width, int height);

/* State dump / restore (the reference has none): a 64-byte header {magic "NBODYST1", precision,
 * body count, steps since upload, timestep, growthRate, field} followed by the [P|V|M|R] block of the current
 * survivors, i.e. exactly what nbody_download returns.  nbody_state_load uploads the block into ctx (which must
 * have the same precision and enough capacity) and restores the step counter; parameters in the file are
 * informational, the context keeps its own. */
int nbody_state_save(nbody_ctx* ctx, const char* path);
int nbody_state_load(nbody_ctx* ctx, const char* path);
/* Reads only the header of a state file. Any output pointer may be NULL. */
int nbody_state_peek(const char* path, int* precision, int* n, int64_t* steps);

/* Multi-rank plumbing (world > 1).  One process per GPU; the host language moves the 128-byte id. */
#define NBODY_COMM_ID_BYTES 128
int nbody_comm_unique_id(void* out128);   /* ncclGetUniqueId via dlopen("librccl.so.1") */

/* Single-process form of the same partition: ctxs[g] is rank g of `world`, all created in this process
 * with NBODY_FLAG_GROUP_EXCHANGE (one per device, or several on one device).  The per-step all-gather is
 * done with stream-ordered device-to-device copies; no RCCL, no host synchronisation inside the loop. */
int nbody_group_step(nbody_ctx** ctxs, int world, int nsteps);
int nbody_group_download(nbody_ctx** ctxs, int world, void* block, int* n);
/* Global index range [lo, lo+cnt) currently owned by this rank (synchronises). */
int nbody_own_range(nbody_ctx* ctx, int* lo, int* cnt);
/* The partition rule itself (pure host function, no device needed): whole reference blocks of 128 bodies
 * (THREADS_PER_BLOCK, src/nbody.cu:36), as evenly as the block count allows, in rank order.  nbody_upload draws it
 * for the uploaded count and the device re-draws it from the survivor count after every step, so ranks stay level
 * as bodies are deleted (the reference's compaction is global, src/nbody.cu:488-510). */
int nbody_partition(int n, int rank, int world, int* lo, int* cnt);
void* nbody_ctx_stream(nbody_ctx* ctx);   /* hipStream_t of the context */

/* ---------------------------------------------------------------------------------------------------
 * Reference-shaped launches on caller-owned DEVICE memory: one-to-one replacements of the two <<<>>> sites
 * src/nbody.cu:481-483.  d_bodyData is a device block in the reference layout for numBodies bodies;
 * velocities are updated in place, updatedMasses/updatedRadii are the scratch arrays of :463-464.  The
 * never-allocated `updatedVelocities` argument of the reference (:441) is dropped.  `stream` is a
 * hipStream_t (NULL = default stream).  numBlocks follows :473; pass nbody_num_blocks(numBodies): with that
 * block count the production kernel of nbody_step runs (same speed), through a process-wide workspace on the current
 * device that holds what a context keeps resident (the {x,y,m,r} replica and the staged output; grown to the largest
 * numBodies seen, released by nbody_launch_workspace_release; not re-entrant, as the reference's loop is one host
 * thread); any other count is honoured by a general kernel directly on the block (it changes which bodies are
 * active and how many tiles are walked).
 * ------------------------------------------------------------------------------------------------- */
int nbody_num_blocks(int numBodies);      /* src/nbody.cu:473 */
int nbody_launch_compute_forces_f32(void* d_bodyData, float* d_updatedMasses, float* d_updatedRadii,
                                    int numBodies, float timestep, int fieldWidth, int fieldHeight,
                                    int numBlocks, float growthRate, void* stream);
int nbody_launch_move_bodies_f32(void* d_bodyData, const float* d_updatedMasses, const float* d_updatedRadii,
                                 int numBodies, float timestep, int numBlocks, void* stream);
int nbody_launch_workspace_release(void);   /* frees the workspace of nbody_launch_compute_forces_f32 (if any) */

/* Device self-test used by the GPU test-suite, exhaustive over all 2^32 fp32 inputs: the kernels' general
 * sqrt and reciprocal against fp64-then-round, and the fast evaluation chain of the fp32 force kernel
 * (rsq/rcp + fma corrections) against the general code on its whole guarded domain.
 * mismatches = {sqrt, reciprocal, fast chain} mismatch counts; all must be 0. */
int nbody_selftest_ieee_f32(int device, uint64_t mismatches[3]);
/* The fp64 counterpart cannot be exhaustive: the fast chain of the fp64 force kernel against the compiler's IEEE
 * sqrt and 1/x on inputs_per_mode inputs of each of three families of its guarded domain [2^-500, 2^500] (random;
 * mantissas next to powers of two; perfect squares +- 4 ulps).  mismatches = {sqrt, 1/d^3}; both must be 0. */
int nbody_selftest_chain_f64(int device, uint64_t inputs_per_mode, uint64_t mismatches[2]);
/* The one known exception of Newton-type fp64 reciprocals (a significand of all ones: the last fma sees an exact tie and
 * rounds one ulp low), on c = (2 - 2^-52) 2^k for every k in [-750, 750] against the closed form 2^-(k+1) (1 + 2^-52):
 * result = {mismatches of the general code's reciprocal (must be 0), of the compiler's bare 1.0 / c (informational), of
 * the fast chain's refinement (informational: the reason the fp64 kernel screens such c and redoes them with the general
 * code), inputs the kernel's screen would miss (must be 0), inputs checked (1501)}. */
int nbody_selftest_rcp_ones_f64(int device, uint64_t result[5]);

/* The ring kernel's hand-off relies on a lane's 16-byte LDS record being written (ds_write_b128) and read
 * (ds_read_b128) in one LDS-array cycle, i.e. never seen half old, half new.  512 workgroups: one wave rewrites its 64
 * records `iters` times while seven waves poll them.  result = {torn records seen, sequence numbers going
 * backwards, records read}; the first two must be 0. */
int nbody_selftest_lds_record(int device, int iters, uint64_t result[3]);

/* Profiling aid: launches ONLY the force kernel of the context's current state `reps` times, back to back; the
 * results land in the staging buffers and are never committed, so the state does not change (the pair and event
 * counters do count).  Lets one rank's kernel of a G-rank partition be timed / profiled in steady state on one GPU. */
int nbody_debug_force_only(nbody_ctx* ctx, int reps);

/* Tuning aid (kernel_variant 58 only): cycle totals of the ring kernel's phases since upload, summed over waves:
 * {evaluate, wait, chain+publish, window check, polls, turns, shader clocks of one wave's life, the same in 100 MHz
 * ticks}. */
int nbody_debug_ring_probe(nbody_ctx* ctx, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_MI355X_H */
