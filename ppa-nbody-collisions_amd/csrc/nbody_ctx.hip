// csrc/nbody_ctx.hip -- host side of the stepper context behind the C ABI (include/nbody.h).
//
// Replaces the reference's per-iteration host loop (src/nbody.cu:460-545: cudaMalloc scratch, H2D of the
// whole state, two launches, D2H of the whole state, O(N) host compaction, cudaFree) with device-resident
// state: per step three stream-ordered phases and no host synchronisation,
//     compute : forces+collisions+drift of the own range  -> staged records (step-t index space)
//               stable compaction of the own range        -> this rank's send slot {count, survivors}
//     exchange: all-gather of the slots over RCCL/xGMI (world > 1 only)
//     commit  : gathered slots -> replica of step t+1, new {N, lo, cnt} published on the device
// Global stable order is rank order, so the reference's index-dependent semantics survive
// sharding unchanged.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "nbody.h"
#include "nbody_error.h"
#include "nbody_kernels.hpp"

using namespace nbk;

#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess)                                                                            \
            return nbody_fail(NBODY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),      \
                              __FILE__, __LINE__);                                                        \
    } while (0)

// ---------------------------------------------------------------------------------------------------------
// RCCL through dlopen: no link-time dependency, and inside a process that already loaded a librccl.so.1
// (PyTorch ships one) the same copy is reused.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct Id128 { char b[NBODY_COMM_ID_BYTES]; };   // ncclUniqueId, passed by value

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return NBODY_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    // a copy that is already in the process (PyTorch's) wins: never two RCCLs in one process
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (lib) break;
    }
    for (const char* n : names) {
        if (lib) break;
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!lib) return nbody_fail(NBODY_ERR_COMM, "cannot dlopen librccl.so.1: %s", dlerror());
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = (int (*)(void*))dlsym(lib, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(lib, "ncclCommInitRank");
    r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
    r.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString)
        return nbody_fail(NBODY_ERR_COMM, "librccl is missing a required symbol");
    g_rccl = r;
    return NBODY_OK;
}

#define RCCL_TRY(expr)                                                                                    \
    do {                                                                                                  \
        int r__ = (expr);                                                                                 \
        if (r__ != 0)                                                                                     \
            return nbody_fail(NBODY_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(r__));        \
    } while (0)

constexpr int kNcclInt8 = 0;   // ncclInt8 / ncclChar

}  // namespace

// ---------------------------------------------------------------------------------------------------------
constexpr size_t kCountersOffset = 64;                                    // Meta is 32 bytes
constexpr size_t kMetaBlockBytes = kCountersOffset + sizeof(Counters);
static_assert(sizeof(Meta) <= kCountersOffset, "Meta fits before the counters");

struct nbody_ctx {
    nbody_ctx_desc desc{};
    size_t real_bytes = 4;      // sizeof(T)
    size_t rec_bytes = 16;      // sizeof(Rec<T>)
    int cap = 0;                // global capacity
    int cap_own = 0;            // own-range capacity
    hipStream_t stream = nullptr;
    // device memory
    void* J = nullptr;          // Rec<T>[cap]
    void* Vown = nullptr;       // Vec2<T>[cap_own]
    void* S_J = nullptr;        // Rec<T>[cap_own]
    void* S_V = nullptr;        // Vec2<T>[cap_own]
    unsigned char* slot = nullptr;    // SlotHeader + Rec<T>[cap_own]          (send slot)
    unsigned char* gather = nullptr;  // world * slot_bytes                    (== slot when world == 1)
    size_t slot_bytes = 0;
    int* blk_counts = nullptr;
    unsigned* tile_rmax = nullptr;  // per aligned 128-body tile of J: bits of max |radius| (NaN skipped); see unpack_slots
    float* Jt = nullptr;            // fp32 contexts: the replica once more, tile by tile component-major (see store_tiled)
    int n_tiles = 0;                // cap / 128 + 2
    Meta* meta = nullptr;
    Meta* meta_all = nullptr;       // RCCL contexts: every rank's Meta, all-gathered by nbody_download
    Counters* counters = nullptr;
    Event* events = nullptr;
    int ev_cap = 0;
    unsigned char* d_img = nullptr;   // raster target, grown on demand
    size_t d_img_bytes = 0;
    // host mirrors / staging
    void* h_stage = nullptr;    // pinned, max(cap*rec, cap*2*real ...)
    size_t h_stage_bytes = 0;
    Meta* h_meta = nullptr;     // pinned
    Meta* h_meta_async = nullptr;   // pinned; refreshed by an un-waited D2H copy after every step
    Counters* h_counters = nullptr;
    Counters* h_counters_async = nullptr;   // pinned; refreshed like h_meta_async (nbody_step looks at .errors)
    int spin_limit = 1 << 24;       // ring kernel: polls before a hand-off wait is declared failed
    int num_cus = 256;              // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    bool device_failed = false;     // sticky until the next nbody_upload: a kernel reported a failed hand-off wait
    int n_upper = 0;            // host-side upper bound of the global count (exact after a sync)
    int own_upper = 0;          // upper bound of the own count
    // The slot layout of a step - records | velocities of at most `U` bodies per rank - must be the SAME on every rank,
    // so it may only depend on values every rank sees identically: the count at upload and the count after the step
    // kLag steps back, whose Meta copy the host WAITS for (long landed by then) instead of looking at whatever has
    // arrived.  A count only shrinks, so the older one bounds the current one.
    static constexpr int kLag = 4;
    struct Landed {
        hipEvent_t ev = nullptr;            // recorded behind the copy
        unsigned char* block = nullptr;     // pinned: Meta | Counters as of the end of step `step`
        int64_t step = -1;
    };
    Landed lag[kLag];
    int64_t enq = 0;            // steps enqueued since upload (index of the next step)
    int xchg_n = 0;             // the deterministic bound of the body count (see above)
    nbody_ctx** peers = nullptr;    // group contexts: every rank of the partition, set for the duration of a group call
    int64_t xchg_bytes = 0;     // bytes this rank received through the exchange since upload
    bool uploaded = false;
    int64_t steps = 0;
    // RCCL
    void* comm = nullptr;
    // kernel timing
    bool timing = false;
    struct Timed { hipEvent_t e0, e1; int what; };        // what: 0 force kernel, 1 exchange
    std::vector<Timed> ev_pending;     // recorded, not yet resolved
    std::vector<hipEvent_t> ev_free;   // created by nbody_set_kernel_timing, outside any timed region
    double force_ms = 0.0;
    int64_t force_launches = 0;
    double xchg_ms = 0.0;
    int64_t xchg_launches = 0;
};

namespace {

// Polls of a hand-off record before the ring kernel declares the wait failed (a poll is ~100 cycles: the default is
// about a second).  NBODY_RING_SPIN_LIMIT shrinks it so that the failure path can be exercised (tests).
int ring_spin_limit() {                                    // read when a context is created
    const char* e = getenv("NBODY_RING_SPIN_LIMIT");
    const long v = e ? strtol(e, nullptr, 10) : 0;
    return v > 0 && v < (1l << 30) ? (int)v : (1 << 24);
}

inline bool not_plus_zero_host(float x) { uint32_t u; memcpy(&u, &x, 4); return u != 0; }
inline bool not_plus_zero_host(double x) { uint64_t u; memcpy(&u, &x, 8); return u != 0; }

template <typename T>
StepParams<T> make_params(const nbody_ctx_desc& d, int spin_limit = 1 << 24) {
    StepParams<T> p;
    p.dt = (T)d.timestep;            // cfg values are floats; double holds them exactly
    p.growth = (T)d.growthRate;
    p.G = (T)6.67408e-11f;           // src/nbody.cu:37, float literal (widened for fp64: SURVEY.md H6)
    p.wall_hi_x = (T)d.fieldWidth;
    p.wall_lo_x = (T)(-d.fieldWidth);
    p.wall_hi_y = (T)d.fieldHeight;
    p.wall_lo_y = (T)(-d.fieldHeight);
    p.literal = d.semantics == NBODY_LITERAL;
    p.spin_limit = spin_limit;
    p.rotate_priority = 0;
    return p;
}

// Timing events come from a pool that nbody_set_kernel_timing fills: nothing is created inside a timed region.  A pool
// that runs dry resolves what is pending (a wait for the oldest launches) and reuses those events.
constexpr int kTimingPool = 1024;
int resolve_timing(nbody_ctx* c) {
    for (const nbody_ctx::Timed& t : c->ev_pending) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(t.e1));
        HIP_TRY(hipEventElapsedTime(&ms, t.e0, t.e1));
        if (t.what == 0) { c->force_ms += ms; c->force_launches += 1; }
        else { c->xchg_ms += ms; c->xchg_launches += 1; }
        c->ev_free.push_back(t.e0);
        c->ev_free.push_back(t.e1);
    }
    c->ev_pending.clear();
    return NBODY_OK;
}
int timing_begin(nbody_ctx* c, int what, hipStream_t stream, nbody_ctx::Timed* t) {
    if (c->ev_free.size() < 2) {
        int rc = resolve_timing(c);
        if (rc != NBODY_OK) return rc;
    }
    if (c->ev_free.size() < 2) return nbody_fail(NBODY_ERR_STATE, "timing is on but its event pool is empty");
    t->e1 = c->ev_free.back(); c->ev_free.pop_back();
    t->e0 = c->ev_free.back(); c->ev_free.pop_back();
    t->what = what;
    HIP_TRY(hipEventRecord(t->e0, stream));
    return NBODY_OK;
}
int timing_end(nbody_ctx* c, hipStream_t stream, const nbody_ctx::Timed& t) {
    HIP_TRY(hipEventRecord(t.e1, stream));
    c->ev_pending.push_back(t);
    return NBODY_OK;
}

// The convention of CUDA_SYNC_CHECK (src/nbody.cu:20-33): a failure on the device surfaces at the next point where
// the host looks.  The only in-kernel failure is a ring hand-off wait that gave up (the kernel then poisons its
// output with NaNs); once seen it is sticky until the next nbody_upload.
// Upper bound of any rank's own count when the body count is at most n: the partition is re-drawn every step, a
// rank's range can grow by a block while the total shrinks, but never beyond ceil(blocks / world) blocks.
int own_upper_of(const nbody_ctx* c, int n) { return nbody_own_upper_of(n, c->desc.world); }

int device_failure(nbody_ctx* c, unsigned long long errors) {
    if (errors != 0) c->device_failed = true;
    if (!c->device_failed) return NBODY_OK;
    return nbody_fail(NBODY_ERR_HIP, "device reported %llu in-kernel hand-off time-out(s) (the state is poisoned: NaN) and "
                                     "%llu failed index check(s) (the access was skipped): upload again",
                      errors % kIndexError, errors / kIndexError);
}

// Asynchronous look: errors of the step records that have landed so far.
int landed_failure(nbody_ctx* c) {
    unsigned long long errors = 0;
    for (const nbody_ctx::Landed& L : c->lag)
        if (L.step >= 0 && hipEventQuery(L.ev) == hipSuccess)
            errors |= reinterpret_cast<const Counters*>(L.block + kCountersOffset)->errors;
    (void)hipGetLastError();                               // hipErrorNotReady of a pending record is not an error
    return device_failure(c, errors);
}

// Synchronises the stream and refreshes the host copies of Meta and Counters; fails if the device reported a failure.
int read_meta(nbody_ctx* c) {
    HIP_TRY(hipMemcpyAsync(c->h_meta, c->meta, sizeof(Meta), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_counters, c->counters, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_upper = c->h_meta->n;
    c->own_upper = own_upper_of(c, c->n_upper);
    return device_failure(c, c->h_counters->errors);
}

// kernel_variant: 0 automatic | 1 v1 (one body per lane, compiler IEEE sqrt/div) |
//                 11,12,14,18 v3 with K = 1,2,4,8 lanes per body, 128-thread workgroups |
//                 31,32 v3 K = 1 with 256-thread workgroups, registers sized for 4 / 2 waves per SIMD |
//                 50,52,54 ring of waves with 2x8, 4x4, 1x8 (rings x waves) per workgroup; 53,55,56,58 its tuning forms
// Where a force launch goes: the context's own stream and velocities, or - the reference-shaped launch through the
// workspace context - the caller's stream and the velocities where they lie in the caller's block.
struct LaunchTarget {
    hipStream_t stream;
    const void* vown;
};
LaunchTarget own_target(const nbody_ctx* c) { return LaunchTarget{c->stream, c->Vown}; }

template <typename T>
void launch_forces(nbody_ctx* c, const StepParams<T>& p, int nblocks, bool log, LaunchTarget to);

#define NB_FORCES_ARGS(T)                                                                                 \
    (const Rec<T>*)c->J, (const Vec2<T>*)to.vown, (Rec<T>*)c->S_J, (Vec2<T>*)c->S_V, (const Meta*)c->meta, p, \
        c->events, c->ev_cap, c->counters

template <>
void launch_forces<double>(nbody_ctx* c, const StepParams<double>& p, int nblocks, bool log, LaunchTarget to) {
    if (c->desc.kernel_variant == 1) {                     // general kernel: compiler IEEE sqrt / divide per pair
        if (log) hipLaunchKernelGGL((forces_v1<double, true>), dim3(nblocks), dim3(kTile), 0, to.stream, NB_FORCES_ARGS(double));
        else hipLaunchKernelGGL((forces_v1<double, false>), dim3(nblocks), dim3(kTile), 0, to.stream, NB_FORCES_ARGS(double));
        return;
    }
    const int grid = (nblocks + 1) / 2;                    // two 128-lane groups per workgroup
    // priority rotation (nbody_forces_v3.inc) only when every workgroup of the launch is resident at once: 168 VGPRs
    // = 3 waves per SIMD = 3 of these 4-wave workgroups per CU
    StepParams<double> pr = p;
    pr.rotate_priority = grid <= 3 * c->num_cus;
#define NB_FORCES_ARGS_PR                                                                                  \
    (const Rec<double>*)c->J, (const Vec2<double>*)to.vown, (Rec<double>*)c->S_J, (Vec2<double>*)c->S_V,    \
        (const Meta*)c->meta, pr, c->events, c->ev_cap, c->counters
    if (log) hipLaunchKernelGGL((forces_v3w_f64<true>), dim3(grid), dim3(2 * kTile), 0, to.stream, NB_FORCES_ARGS_PR);
    else hipLaunchKernelGGL((forces_v3w_f64<false>), dim3(grid), dim3(2 * kTile), 0, to.stream, NB_FORCES_ARGS_PR);
#undef NB_FORCES_ARGS_PR
}

template <int K>
void launch_v3(nbody_ctx* c, const StepParams<float>& p, int nblocks, bool log, LaunchTarget to) {
    const int grid = nblocks * K;
    if (log) hipLaunchKernelGGL((forces_v3_f32<K, true>), dim3(grid), dim3(kTile), 0, to.stream, NB_FORCES_ARGS(float));
    else hipLaunchKernelGGL((forces_v3_f32<K, false>), dim3(grid), dim3(kTile), 0, to.stream, NB_FORCES_ARGS(float));
}

template <int K, int kOcc>
void launch_v3w(nbody_ctx* c, const StepParams<float>& p, int nblocks, bool log, LaunchTarget to) {
    const int grid = (nblocks * K + 1) / 2;                // two 128-lane groups per workgroup
    if (log) hipLaunchKernelGGL((forces_v3w_f32<K, true, kOcc>), dim3(grid), dim3(2 * kTile), 0, to.stream, NB_FORCES_ARGS(float));
    else hipLaunchKernelGGL((forces_v3w_f32<K, false, kOcc>), dim3(grid), dim3(2 * kTile), 0, to.stream, NB_FORCES_ARGS(float));
}

template <int kW, int kT, int kSleep, bool kProbe, int kRings>
void launch_ring(nbody_ctx* c, const StepParams<float>& p, int nblocks, bool log, LaunchTarget to) {
    const int grid = (nblocks * 2 + kRings - 1) / kRings;  // a workgroup serves kRings rings of 64 bodies, two per reference block
    RingArgs a;
    a.Vown = (const Vec2<float>*)to.vown; a.S_J = (Rec<float>*)c->S_J; a.S_V = (Vec2<float>*)c->S_V;
    a.meta = c->meta; a.p = p; a.ev = c->events; a.ev_cap = c->ev_cap; a.ctr = c->counters;
    a.tile_rmax = (const float*)c->tile_rmax; a.Jt = c->Jt; a.cap_own = c->cap_own; a.n_tiles = c->n_tiles;
    // testing aid: make the index checks fire (the kernel is told the tiled copy is one tile long / the staging arrays hold
    // one body), to see them reported instead of trusted
    if (const char* e = getenv("NBODY_TEST_INDEX_CHECKS")) {
        if (e[0] == 't') a.n_tiles = 1;
        if (e[0] == 'q') a.cap_own = 1;
    }
    if (log) hipLaunchKernelGGL((forces_ring_f32<true, kW, kT, kSleep, kProbe, kRings>), dim3(grid), dim3(kRings * kW * kWave), 0, to.stream, a);
    else hipLaunchKernelGGL((forces_ring_f32<false, kW, kT, kSleep, kProbe, kRings>), dim3(grid), dim3(kRings * kW * kWave), 0, to.stream, a);
}
template <>
void launch_forces<float>(nbody_ctx* c, const StepParams<float>& p, int nblocks, bool log, LaunchTarget to) {
    switch (c->desc.kernel_variant) {
        case 1:
            if (log) hipLaunchKernelGGL((forces_v1<float, true>), dim3(nblocks), dim3(kTile), 0, to.stream, NB_FORCES_ARGS(float));
            else hipLaunchKernelGGL((forces_v1<float, false>), dim3(nblocks), dim3(kTile), 0, to.stream, NB_FORCES_ARGS(float));
            return;
        case 11: launch_v3<1>(c, p, nblocks, log, to); return;
        case 12: launch_v3<2>(c, p, nblocks, log, to); return;
        case 14: launch_v3<4>(c, p, nblocks, log, to); return;
        case 18: launch_v3<8>(c, p, nblocks, log, to); return;
        case 31: launch_v3w<1, 4>(c, p, nblocks, log, to); return;
        case 32: launch_v3w<1, 2>(c, p, nblocks, log, to); return;
        case 50: launch_ring<8, 32, 8, false, 2>(c, p, nblocks, log, to); return;   // 2 rings of 8 waves per workgroup
        case 52: launch_ring<4, 32, 8, false, 4>(c, p, nblocks, log, to); return;   // 4 rings of 4 waves per workgroup
        case 54: launch_ring<8, 32, 2, false, 1>(c, p, nblocks, log, to); return;   // one ring of 8 waves per workgroup
        case 53: launch_ring<8, 16, 8, false, 2>(c, p, nblocks, log, to); return;   // tuning: turns of 16 positions
        case 55: launch_ring<8, 32, 2, false, 2>(c, p, nblocks, log, to); return;   // tuning: 2 x 8 with a 128-cycle poll interval
        case 56: launch_ring<8, 32, 4, false, 2>(c, p, nblocks, log, to); return;   // tuning: 2 x 8 with a 256-cycle poll interval
        case 58: launch_ring<8, 32, 8, true, 2>(c, p, nblocks, log, to); return;    // tuning: in-kernel phase stamps
        default: break;
    }
    // default: chosen by how many bodies this rank owns, i.e. how many ordered chains there are to fill the chip with
    // (measured on MI355X with csrc/tune/ring_probe.py, profiles/r02_ring_*):
    //   >= 48k bodies : ring kernel, workgroups of 4 rings x 4 waves (256 bodies): 32.2 ms at 262144 own bodies
    //                   (one lane per body: 33.6), 16.6 / 8.4 ms at 131072 / 65536 (18.5 / 10.2)
    //   >= 24k bodies : ring kernel, workgroups of 2 rings x 8 waves (128 bodies): fills the chip with half the bodies
    //                   (4.35 ms at 32768 own bodies of 262144; 4 x 4: 5.4)
    //   below         : ring kernel, one ring of 8 waves per workgroup: twice the workgroups to spread over the CUs
    //                   (0.21 ms at N = 16384; 2 x 8: 0.29; the producer/consumer kernel of round 1: 0.31)
    // Poll interval of the hand-off wait: s_sleep 8 (512 cycles) where the launch is bound by evaluation - a poll takes
    // issue slots from the waves that evaluate: -0.5 ... -1 % against s_sleep 2 -, s_sleep 2 for the small launches, which
    // are bound by the chain (N = 16384: 0.185 ms; with s_sleep 8: 0.218).
    if (c->own_upper >= 49152) launch_ring<4, 32, 8, false, 4>(c, p, nblocks, log, to);
    else if (c->own_upper >= 24576) launch_ring<8, 32, 8, false, 2>(c, p, nblocks, log, to);
    else launch_ring<8, 32, 2, false, 1>(c, p, nblocks, log, to);
}

template <typename T>
int launch_force_kernel(nbody_ctx* c, const StepParams<T>& p, int nblocks) {
    nbody_ctx::Timed t{};
    if (c->timing) {
        int rc = timing_begin(c, 0, c->stream, &t);
        if (rc != NBODY_OK) return rc;
    }
    const bool log = (c->desc.flags & NBODY_FLAG_RECORD_EVENTS) != 0;
    launch_forces<T>(c, p, nblocks, log, own_target(c));
    HIP_TRY(hipGetLastError());
    if (c->timing) return timing_end(c, c->stream, t);
    return NBODY_OK;
}

// Bodies per rank the slots of the NEXT step are laid out for, and the bytes one slot then takes in the exchange.
int slot_bodies(const nbody_ctx* c) { return own_upper_of(c, c->xchg_n); }
size_t slot_stride(const nbody_ctx* c) {
    const size_t b = sizeof(SlotHeader) + (size_t)slot_bodies(c) * (c->rec_bytes + 2 * c->real_bytes);
    return (b + 255) & ~(size_t)255;
}

// Before step `enq` is enqueued: the Meta copy of step enq - kLag is waited for and becomes the bound of this step.
int refresh_bound(nbody_ctx* c) {
    if (c->enq < nbody_ctx::kLag) return NBODY_OK;        // still the uploaded count
    nbody_ctx::Landed& L = c->lag[c->enq % nbody_ctx::kLag];
    if (L.step != c->enq - nbody_ctx::kLag) return nbody_fail(NBODY_ERR_STATE, "step record ring out of sequence");
    HIP_TRY(hipEventSynchronize(L.ev));
    const Meta* m = reinterpret_cast<const Meta*>(L.block);
    const Counters* k = reinterpret_cast<const Counters*>(L.block + kCountersOffset);
    if (m->n >= 0 && m->n < c->xchg_n) c->xchg_n = m->n;
    if (c->xchg_n < c->n_upper) { c->n_upper = c->xchg_n; c->own_upper = own_upper_of(c, c->n_upper); }
    return device_failure(c, k->errors);
}

template <typename T>
int launch_compute(nbody_ctx* c) {
    const StepParams<T> p = make_params<T>(c->desc, c->spin_limit);
    {
        int rc = refresh_bound(c);
        if (rc != NBODY_OK) return rc;
    }
    // Workgroups cover every reference block of the own range, which starts on a block boundary (nbody_own_range_of).  The
    // body count only shrinks between syncs, so the host-side upper bound is safe; the kernel takes the exact range
    // from the device-side Meta and workgroups past it exit at once.
    const int nblocks = c->own_upper / kTile > 0 ? c->own_upper / kTile : 1;
    int rc = launch_force_kernel<T>(c, p, nblocks);
    if (rc != NBODY_OK) return rc;
    const int nblk = (c->own_upper + kCompactBlock - 1) / kCompactBlock > 0
                         ? (c->own_upper + kCompactBlock - 1) / kCompactBlock : 1;
    hipLaunchKernelGGL((compact_count<T>), dim3(nblk), dim3(kCompactBlock), 0, c->stream,
                       (const Rec<T>*)c->S_J, (const Meta*)c->meta, c->blk_counts, c->tile_rmax, c->n_tiles);
    hipLaunchKernelGGL((compact_scatter<T>), dim3(nblk), dim3(kCompactBlock), 0, c->stream,
                       (const Rec<T>*)c->S_J, (const Vec2<T>*)c->S_V, (const Meta*)c->meta,
                       (const int*)c->blk_counts, nblk, (SlotHeader*)c->slot,
                       (Rec<T>*)(c->slot + sizeof(SlotHeader)),
                       (Vec2<T>*)(c->slot + sizeof(SlotHeader) + (size_t)slot_bodies(c) * sizeof(Rec<T>)));
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

template <typename T>
int launch_commit(nbody_ctx* c) {
    // every rank's slot holds up to slot_bodies() survivors: the grid covers the largest count any rank can have
    const int gx_all = (slot_bodies(c) + 255) / 256 > 0 ? (slot_bodies(c) + 255) / 256 : 1;
    hipLaunchKernelGGL((unpack_slots<T>), dim3(gx_all, c->desc.world), dim3(256), 0, c->stream,
                       (const unsigned char*)c->gather, slot_stride(c), slot_bodies(c), c->desc.world, c->desc.rank,
                       (Rec<T>*)c->J, (Vec2<T>*)c->Vown, c->meta, c->tile_rmax, c->Jt, c->counters);
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

// ---------------------------------------------------------------------------------------------------------
// ONE all-gather interface, two transports.  Enqueued on c->stream: for every rank g, `bytes` bytes of rank g's `what`
// land at dst + g * bytes of THIS rank.  RCCL contexts: one ncclAllGather (every rank calls it).  Group contexts (all
// ranks are contexts of this process, c->peers): device-to-device copies out of the peers' buffers; the caller orders
// them behind the peers' work (nbody_group_step's events, nbody_group_download's synchronisation).  Everything around
// the transport - the slot layout, the unpack, the padded velocity gather and the Meta gather of a download - is the
// same code for both, so the single-GPU group tests execute what a multi-GPU RCCL run executes, the ncclAllGather call
// itself excepted.
// ---------------------------------------------------------------------------------------------------------
enum class Part { Slot, Velocities, MetaBlock };
const unsigned char* part_of(const nbody_ctx* c, Part what) {
    switch (what) {
        case Part::Slot: return c->slot;
        case Part::Velocities: return (const unsigned char*)c->Vown;
        default: return (const unsigned char*)c->meta;
    }
}
int all_gather(nbody_ctx* c, Part what, unsigned char* dst, size_t bytes) {
    if (c->comm) {
        RCCL_TRY(g_rccl.AllGather(part_of(c, what), dst, bytes, kNcclInt8, c->comm, c->stream));
    } else if (c->desc.world == 1) {
        if (dst != part_of(c, what))
            HIP_TRY(hipMemcpyAsync(dst, part_of(c, what), bytes, hipMemcpyDeviceToDevice, c->stream));
    } else if (c->peers) {
        for (int g = 0; g < c->desc.world; ++g)
            HIP_TRY(hipMemcpyAsync(dst + (size_t)g * bytes, part_of(c->peers[g], what), bytes, hipMemcpyDeviceToDevice,
                                   c->stream));
    } else {
        return nbody_fail(NBODY_ERR_STATE, "context of a %d-rank partition has neither a communicator nor its peers",
                          c->desc.world);
    }
    c->xchg_bytes += (int64_t)bytes * c->desc.world;
    return NBODY_OK;
}

// The per-step exchange: every rank's slot {count | records | velocities}, laid out for slot_bodies() bodies - the
// bound of the LIVE count, not the capacity: the gather shrinks with the body count.
int do_exchange(nbody_ctx* c) {
    if (c->gather == c->slot) return NBODY_OK;   // single rank without a communicator: the gather buffer IS the slot
    nbody_ctx::Timed t{};
    if (c->timing) {
        int rc = timing_begin(c, 1, c->stream, &t);
        if (rc != NBODY_OK) return rc;
    }
    int rc = all_gather(c, Part::Slot, c->gather, slot_stride(c));
    if (rc != NBODY_OK) return rc;
    if (c->timing) return timing_end(c, c->stream, t);
    return NBODY_OK;
}

int compute_phase(nbody_ctx* c) {
    return c->desc.precision == NBODY_F64 ? launch_compute<double>(c) : launch_compute<float>(c);
}
int commit_phase(nbody_ctx* c) {
    int rc = c->desc.precision == NBODY_F64 ? launch_commit<double>(c) : launch_commit<float>(c);
    if (rc != NBODY_OK) return rc;
    c->steps += 1;
    // Meta and Counters of this step go to a pinned record; the step kLag steps on waits for it (refresh_bound)
    nbody_ctx::Landed& L = c->lag[c->enq % nbody_ctx::kLag];
    HIP_TRY(hipMemcpyAsync(L.block, c->meta, kMetaBlockBytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(L.ev, c->stream));
    L.step = c->enq;
    c->enq += 1;
    return NBODY_OK;
}

void free_all(nbody_ctx* c) {
    if (!c) return;
    hipSetDevice(c->desc.device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (const nbody_ctx::Timed& t : c->ev_pending) { hipEventDestroy(t.e0); hipEventDestroy(t.e1); }
    for (hipEvent_t e : c->ev_free) hipEventDestroy(e);
    for (nbody_ctx::Landed& L : c->lag) {
        if (L.ev) hipEventDestroy(L.ev);
        if (L.block) hipHostFree(L.block);
    }
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    hipFree(c->J); hipFree(c->Vown); hipFree(c->S_J); hipFree(c->S_V);
    if (c->gather && c->gather != c->slot) hipFree(c->gather);
    hipFree(c->slot);
    hipFree(c->blk_counts); hipFree(c->tile_rmax); hipFree(c->Jt); hipFree(c->meta); hipFree(c->meta_all); hipFree(c->events); hipFree(c->d_img);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_meta) hipHostFree(c->h_meta);
    if (c->h_meta_async) hipHostFree(c->h_meta_async);
    if (c->h_counters) hipHostFree(c->h_counters);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" {

int nbody_comm_unique_id(void* out128) {
    if (!out128) return nbody_fail(NBODY_ERR_INVALID, "nbody_comm_unique_id: NULL");
    int rc = rccl_load();
    if (rc != NBODY_OK) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(out128));
    return NBODY_OK;
}

int nbody_ctx_create(nbody_ctx** out, const nbody_ctx_desc* d) {
    if (!out || !d) return nbody_fail(NBODY_ERR_INVALID, "nbody_ctx_create: NULL argument");
    *out = nullptr;
    if (d->capacity <= 0 || d->world < 1 || d->rank < 0 || d->rank >= d->world)
        return nbody_fail(NBODY_ERR_INVALID, "nbody_ctx_create: bad capacity/rank/world");
    if (d->precision != NBODY_F32 && d->precision != NBODY_F64)
        return nbody_fail(NBODY_ERR_INVALID, "nbody_ctx_create: bad precision");
    if (d->semantics != NBODY_LITERAL && d->semantics != NBODY_CLEAN)
        return nbody_fail(NBODY_ERR_INVALID, "nbody_ctx_create: bad semantics");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return nbody_fail(NBODY_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU path",
                          e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (d->device < 0 || d->device >= ndev)
        return nbody_fail(NBODY_ERR_INVALID, "device ordinal %d out of range (0..%d)", d->device, ndev - 1);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, d->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return nbody_fail(NBODY_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 only", d->device,
                          prop.gcnArchName);
    HIP_TRY(hipSetDevice(d->device));

    nbody_ctx* c = new (std::nothrow) nbody_ctx();
    if (!c) return nbody_fail(NBODY_ERR_NOMEM, "nbody_ctx_create: out of host memory");
    c->desc = *d;
    c->desc.comm_id = nullptr;
    c->spin_limit = ring_spin_limit();
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->real_bytes = d->precision == NBODY_F64 ? 8 : 4;
    c->rec_bytes = 4 * c->real_bytes;
    c->cap = d->capacity;
    c->cap_own = nbody_own_upper_of(d->capacity, d->world);   // largest own range of the block-aligned partition
    c->ev_cap = d->event_capacity > 0 ? d->event_capacity : (1 << 20);
    c->slot_bytes = sizeof(SlotHeader) + (size_t)c->cap_own * (c->rec_bytes + 2 * c->real_bytes);   // records | velocities
    c->slot_bytes = (c->slot_bytes + 255) & ~(size_t)255;

#define CTX_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) {                                                                          \
            int rc__ = nbody_fail(e__ == hipErrorOutOfMemory ? NBODY_ERR_NOMEM : NBODY_ERR_HIP,           \
                                  "%s failed: %s", #expr, hipGetErrorString(e__));                        \
            free_all(c);                                                                                  \
            return rc__;                                                                                  \
        }                                                                                                 \
    } while (0)

    CTX_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CTX_TRY(hipMalloc(&c->J, (size_t)c->cap * c->rec_bytes));
    CTX_TRY(hipMalloc(&c->Vown, (size_t)c->cap_own * 2 * c->real_bytes));
    CTX_TRY(hipMalloc(&c->S_J, (size_t)c->cap_own * c->rec_bytes));
    CTX_TRY(hipMalloc(&c->S_V, (size_t)c->cap_own * 2 * c->real_bytes));
    CTX_TRY(hipMalloc((void**)&c->slot, c->slot_bytes));
    const bool use_comm = (d->world > 1 && !(d->flags & NBODY_FLAG_GROUP_EXCHANGE)) || (d->flags & NBODY_FLAG_FORCE_COMM);
    if (d->world > 1 || use_comm) CTX_TRY(hipMalloc((void**)&c->gather, c->slot_bytes * d->world));
    else c->gather = c->slot;
    CTX_TRY(hipMalloc((void**)&c->blk_counts, sizeof(int) * (size_t)(c->cap_own / kCompactBlock + 2)));
    c->n_tiles = c->cap / kTile + 2;
    CTX_TRY(hipMalloc((void**)&c->tile_rmax, sizeof(unsigned) * (size_t)c->n_tiles));
    CTX_TRY(hipMemset(c->tile_rmax, 0, sizeof(unsigned) * (size_t)c->n_tiles));
    if (d->precision != NBODY_F64) {
        CTX_TRY(hipMalloc((void**)&c->Jt, sizeof(float) * 4 * kTile * (size_t)c->n_tiles));
        CTX_TRY(hipMemset(c->Jt, 0, sizeof(float) * 4 * kTile * (size_t)c->n_tiles));
    }
    // Meta and Counters share one device block (and one pinned host block): the per-step look at them is ONE copy
    CTX_TRY(hipMalloc((void**)&c->meta, kMetaBlockBytes));
    c->counters = reinterpret_cast<Counters*>(reinterpret_cast<unsigned char*>(c->meta) + kCountersOffset);
    if (use_comm || d->world > 1) CTX_TRY(hipMalloc((void**)&c->meta_all, sizeof(Meta) * (size_t)d->world));
    for (nbody_ctx::Landed& L : c->lag) {
        CTX_TRY(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
        CTX_TRY(hipHostMalloc((void**)&L.block, kMetaBlockBytes, hipHostMallocDefault));
        memset(L.block, 0, kMetaBlockBytes);
    }
    CTX_TRY(hipMalloc((void**)&c->events, sizeof(Event) * (size_t)c->ev_cap));
    // on the context's own stream and waited for: hipMemset() on device memory runs on the NULL stream and may
    // return before the fill has executed; the context's stream is non-blocking, so a late fill could land
    // after nbody_upload's copy of Meta (seen once as a step that found n = 0)
    CTX_TRY(hipMemsetAsync(c->counters, 0, sizeof(Counters), c->stream));
    CTX_TRY(hipMemsetAsync(c->meta, 0, sizeof(Meta), c->stream));
    CTX_TRY(hipStreamSynchronize(c->stream));
    c->h_stage_bytes = (size_t)c->cap * c->rec_bytes;
    CTX_TRY(hipHostMalloc(&c->h_stage, c->h_stage_bytes, hipHostMallocDefault));
    CTX_TRY(hipHostMalloc((void**)&c->h_meta, sizeof(Meta), hipHostMallocDefault));
    CTX_TRY(hipHostMalloc((void**)&c->h_meta_async, kMetaBlockBytes, hipHostMallocDefault));
    c->h_counters_async = reinterpret_cast<Counters*>(reinterpret_cast<unsigned char*>(c->h_meta_async) + kCountersOffset);
    CTX_TRY(hipHostMalloc((void**)&c->h_counters, sizeof(Counters), hipHostMallocDefault));
    memset(c->h_counters, 0, sizeof(Counters));
    memset(c->h_counters_async, 0, sizeof(Counters));
#undef CTX_TRY

    if (use_comm) {
        int rc = rccl_load();
        if (rc == NBODY_OK && !d->comm_id)
            rc = nbody_fail(NBODY_ERR_INVALID, "an RCCL context needs comm_id (nbody_comm_unique_id on rank 0)");
        if (rc == NBODY_OK) {
            Id128 id;
            memcpy(id.b, d->comm_id, sizeof(id.b));
            int r = g_rccl.CommInitRank(&c->comm, d->world, id, d->rank);
            if (r != 0) rc = nbody_fail(NBODY_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        }
        if (rc != NBODY_OK) { free_all(c); return rc; }
    }
    *out = c;
    return NBODY_OK;
}

int nbody_ctx_destroy(nbody_ctx* ctx) {
    free_all(ctx);
    return NBODY_OK;
}

void* nbody_ctx_stream(nbody_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int nbody_upload(nbody_ctx* c, const void* block, int n) {
    if (!c || !block || n < 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_upload: bad argument");
    if (n > c->cap) return nbody_fail(NBODY_ERR_CAPACITY, "nbody_upload: %d bodies > capacity %d", n, c->cap);
    HIP_TRY(hipSetDevice(c->desc.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int lo = 0, cnt = 0;
    nbody_own_range_of(n, c->desc.rank, c->desc.world, &lo, &cnt);
    if (cnt > c->cap_own) return nbody_fail(NBODY_ERR_CAPACITY, "own range %d > own capacity %d", cnt, c->cap_own);
    // pack [P|V|M|R] (src/nbody.cu:66-77) into {x,y,m,r} records; Meta::summary as unpack_slots computes it per step
    int summary = 0;
    if (c->desc.precision == NBODY_F64) {
        const double* P = (const double*)block;
        const double* V = P + 2 * (size_t)n;
        const double* M = V + 2 * (size_t)n;
        const double* R = M + (size_t)n;
        Rec<double>* st = (Rec<double>*)c->h_stage;
        for (int i = 0; i < n; ++i) {
            st[i] = Rec<double>{P[2 * i], P[2 * i + 1], M[i], R[i]};
            const bool bounded = fabs(st[i].x) < FastDomain<double>::coord && fabs(st[i].y) < FastDomain<double>::coord;
            summary |= (bounded ? 0 : kSummaryUnbounded) | (not_plus_zero_host(R[i]) ? kSummaryRadius : 0);
        }
        HIP_TRY(hipMemcpyAsync(c->J, st, (size_t)n * sizeof(Rec<double>), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->Vown, V + 2 * (size_t)lo, (size_t)cnt * 16, hipMemcpyHostToDevice, c->stream));
    } else {
        const float* P = (const float*)block;
        const float* V = P + 2 * (size_t)n;
        const float* M = V + 2 * (size_t)n;
        const float* R = M + (size_t)n;
        Rec<float>* st = (Rec<float>*)c->h_stage;
        for (int i = 0; i < n; ++i) {
            st[i] = Rec<float>{P[2 * i], P[2 * i + 1], M[i], R[i]};
            const bool bounded = fabsf(st[i].x) < FastDomain<float>::coord && fabsf(st[i].y) < FastDomain<float>::coord;
            const bool small = !(fabsf(st[i].x) >= kCoordFloor && fabsf(st[i].y) >= kCoordFloor);
            summary |= (bounded ? 0 : kSummaryUnbounded) | (not_plus_zero_host(R[i]) ? kSummaryRadius : 0) |
                       (small ? kSummarySmall : 0) | (fabsf(st[i].m) < kMassBound ? 0 : kSummaryMass);
        }
        HIP_TRY(hipMemcpyAsync(c->J, st, (size_t)n * sizeof(Rec<float>), hipMemcpyHostToDevice, c->stream));
        if (n > 0) hipLaunchKernelGGL(records_to_tiles_f32, dim3((n + 255) / 256), dim3(256), 0, c->stream,
                                      (const Rec<float>*)c->J, n, c->Jt);
        HIP_TRY(hipMemcpyAsync(c->Vown, V + 2 * (size_t)lo, (size_t)cnt * 8, hipMemcpyHostToDevice, c->stream));
    }
    {   // per-tile radius bounds, as unpack_slots maintains them from then on
        std::vector<unsigned> tr((size_t)c->n_tiles, 0u);
        for (int i = 0; i < n; ++i) {
            const float ar = c->desc.precision == NBODY_F64 ? (float)fabs(((const double*)block)[5 * (size_t)n + i])
                                                            : fabsf(((const float*)block)[5 * (size_t)n + i]);
            unsigned bits;
            memcpy(&bits, &ar, 4);
            if (ar == ar && bits > tr[i / kTile]) tr[i / kTile] = bits;
        }
        HIP_TRY(hipMemcpyAsync(c->tile_rmax, tr.data(), sizeof(unsigned) * tr.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));   // tr goes out of scope
    }
    c->h_meta->n = n; c->h_meta->lo = lo; c->h_meta->cnt = cnt; c->h_meta->step = 0; c->h_meta->n_prev = n;
    c->h_meta->summary = summary; c->h_meta->pad[0] = c->h_meta->pad[1] = 0;
    *c->h_meta_async = *c->h_meta;
    HIP_TRY(hipMemcpyAsync(c->meta, c->h_meta, sizeof(Meta), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->counters, 0, sizeof(Counters), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_upper = n; c->own_upper = own_upper_of(c, n);
    c->xchg_n = n; c->enq = 0; c->xchg_bytes = 0;
    for (nbody_ctx::Landed& L : c->lag) L.step = -1;
    memset(c->h_counters, 0, sizeof(Counters));
    memset(c->h_counters_async, 0, sizeof(Counters));
    c->device_failed = false;
    c->uploaded = true;
    c->steps = 0;
    int rt = resolve_timing(c);
    if (rt != NBODY_OK) return rt;
    c->force_ms = 0; c->force_launches = 0; c->xchg_ms = 0; c->xchg_launches = 0;
    return NBODY_OK;
}

// Single-process multi-context stepping: every rank of the partition is a context of THIS process (one per
// device, or several on one device for tests).  The per-step exchange is done with stream-ordered
// device-to-device copies and events, no host synchronisation and no RCCL.
int nbody_group_step(nbody_ctx** ctxs, int world, int nsteps) {
    if (!ctxs || world < 1 || nsteps < 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_group_step: bad argument");
    for (int g = 0; g < world; ++g) {
        nbody_ctx* c = ctxs[g];
        if (!c || c->desc.world != world || c->desc.rank != g || !c->uploaded)
            return nbody_fail(NBODY_ERR_STATE, "nbody_group_step: context %d is not rank %d of %d (or not uploaded)", g, g, world);
        if (world > 1 && !(c->desc.flags & NBODY_FLAG_GROUP_EXCHANGE))
            return nbody_fail(NBODY_ERR_STATE, "nbody_group_step: context %d lacks NBODY_FLAG_GROUP_EXCHANGE", g);
        // the ranks lay their slots out from the same history (refresh_bound): they must have been uploaded and stepped together
        if (c->enq != ctxs[0]->enq || c->xchg_n != ctxs[0]->xchg_n)
            return nbody_fail(NBODY_ERR_STATE, "nbody_group_step: context %d is at step %lld with bound %d, context 0 at step %lld "
                                               "with bound %d (upload and step the group together)",
                              g, (long long)c->enq, c->xchg_n, (long long)ctxs[0]->enq, ctxs[0]->xchg_n);
    }
    // direct xGMI copies between the ranks' devices (ignored where already enabled / same device)
    for (int g = 0; g < world; ++g)
        for (int h = 0; h < world; ++h)
            if (ctxs[g]->desc.device != ctxs[h]->desc.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, ctxs[g]->desc.device, ctxs[h]->desc.device) == hipSuccess && can) {
                    HIP_TRY(hipSetDevice(ctxs[g]->desc.device));
                    hipError_t pe = hipDeviceEnablePeerAccess(ctxs[h]->desc.device, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                        return nbody_fail(NBODY_ERR_HIP, "hipDeviceEnablePeerAccess failed: %s", hipGetErrorString(pe));
                    (void)hipGetLastError();
                }
            }
    for (int g = 0; g < world; ++g) {
        int failed = landed_failure(ctxs[g]);
        if (failed != NBODY_OK) return failed;
    }
    std::vector<hipEvent_t> ready(world), done(world);
    for (int g = 0; g < world; ++g) {
        HIP_TRY(hipSetDevice(ctxs[g]->desc.device));
        HIP_TRY(hipEventCreateWithFlags(&ready[g], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&done[g], hipEventDisableTiming));
    }
    // NBODY_GROUP_SERIALIZE=1 (tuning aid): ranks that share one device run their compute phases one after the
    // other, so each rank's kernel time is what a dedicated GPU would see
    const bool serialize = getenv("NBODY_GROUP_SERIALIZE") != nullptr;
    int rc = NBODY_OK;
    for (int s = 0; s < nsteps && rc == NBODY_OK; ++s) {
        for (int g = 0; g < world && rc == NBODY_OK; ++g) {
            nbody_ctx* c = ctxs[g];
            HIP_TRY(hipSetDevice(c->desc.device));
            if (serialize && g > 0) HIP_TRY(hipStreamWaitEvent(c->stream, ready[g - 1], 0));
            if (serialize && g == 0 && s > 0) HIP_TRY(hipStreamWaitEvent(c->stream, done[world - 1], 0));
            // the slot of rank g may be rewritten only after every rank has copied it out (previous step)
            if (s > 0) for (int h = 0; h < world; ++h) if (h != g) HIP_TRY(hipStreamWaitEvent(c->stream, done[h], 0));
            rc = compute_phase(c);
            if (rc == NBODY_OK) HIP_TRY(hipEventRecord(ready[g], c->stream));
        }
        for (int h = 0; h < world && rc == NBODY_OK; ++h) {
            nbody_ctx* c = ctxs[h];
            HIP_TRY(hipSetDevice(c->desc.device));
            if (world > 1) {
                for (int g = 0; g < world; ++g)
                    if (g != h) HIP_TRY(hipStreamWaitEvent(c->stream, ready[g], 0));
                c->peers = ctxs;
                rc = do_exchange(c);                       // the same call an RCCL context makes, peer-copy transport
                c->peers = nullptr;
                if (rc != NBODY_OK) break;
                HIP_TRY(hipEventRecord(done[h], c->stream));
            }
            rc = commit_phase(c);
        }
    }
    for (int g = 0; g < world; ++g) {
        hipSetDevice(ctxs[g]->desc.device);
        hipStreamSynchronize(ctxs[g]->stream);
        hipEventDestroy(ready[g]);
        hipEventDestroy(done[g]);
    }
    return rc;
}

// Full state of a single-process group: rank 0 downloads exactly as a rank of an RCCL run does - replica, velocity
// gather, Meta gather (nbody_download) -, with the peer-copy transport.  Every rank is synchronised first: the copies
// read the peers' buffers.
int nbody_group_download(nbody_ctx** ctxs, int world, void* block, int* n_out) {
    if (!ctxs || world < 1 || !block || !n_out) return nbody_fail(NBODY_ERR_INVALID, "nbody_group_download: bad argument");
    int n0 = -1;
    for (int g = 0; g < world; ++g) {
        nbody_ctx* c = ctxs[g];
        if (!c || c->desc.world != world || c->desc.rank != g || !c->uploaded)
            return nbody_fail(NBODY_ERR_STATE, "nbody_group_download: context %d is not rank %d of %d (or not uploaded)", g, g, world);
        HIP_TRY(hipSetDevice(c->desc.device));
        int rc = read_meta(c);
        if (rc != NBODY_OK) return rc;
        if (g == 0) n0 = c->h_meta->n;
        if (c->h_meta->n != n0) return nbody_fail(NBODY_ERR_STATE, "group ranks disagree on the body count");
    }
    ctxs[0]->peers = ctxs;
    int rc = nbody_download(ctxs[0], block, n_out);
    ctxs[0]->peers = nullptr;
    return rc;
}

int nbody_step(nbody_ctx* c, int nsteps) {
    if (!c || nsteps < 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_step: bad argument");
    if (!c->uploaded) return nbody_fail(NBODY_ERR_STATE, "nbody_step before nbody_upload");
    if (c->desc.world > 1 && (c->desc.flags & NBODY_FLAG_GROUP_EXCHANGE))
        return nbody_fail(NBODY_ERR_STATE, "group context: step it with nbody_group_step");
    HIP_TRY(hipSetDevice(c->desc.device));
    // asynchronous: a device failure of an earlier step is reported as soon as its counters have landed
    int failed = landed_failure(c);
    if (failed != NBODY_OK) return failed;
    for (int s = 0; s < nsteps; ++s) {
        int rc = compute_phase(c);
        if (rc != NBODY_OK) return rc;
        rc = do_exchange(c);
        if (rc != NBODY_OK) return rc;
        rc = commit_phase(c);
        if (rc != NBODY_OK) return rc;
    }
    return NBODY_OK;
}

int nbody_sync(nbody_ctx* c) {
    if (!c) return nbody_fail(NBODY_ERR_INVALID, "NULL context");
    HIP_TRY(hipSetDevice(c->desc.device));
    HIP_TRY(hipGetLastError());
    return c->uploaded ? read_meta(c) : (hipStreamSynchronize(c->stream) == hipSuccess ? NBODY_OK
                                          : nbody_fail(NBODY_ERR_HIP, "hipStreamSynchronize failed"));
}

int nbody_ctx_info(nbody_ctx* c, nbody_ctx_desc* desc_out, int64_t* steps) {
    if (!c) return nbody_fail(NBODY_ERR_INVALID, "NULL context");
    if (desc_out) *desc_out = c->desc;
    if (steps) *steps = c->steps;
    return NBODY_OK;
}

int nbody_ctx_set_steps(nbody_ctx* c, int64_t steps) {
    if (!c || steps < 0 || steps > 0x7fffffff) return nbody_fail(NBODY_ERR_INVALID, "nbody_ctx_set_steps: bad argument");
    if (!c->uploaded) return nbody_fail(NBODY_ERR_STATE, "nbody_ctx_set_steps before nbody_upload");
    HIP_TRY(hipSetDevice(c->desc.device));
    int rc = read_meta(c);
    if (rc != NBODY_OK) return rc;
    c->h_meta->step = (int)steps;
    HIP_TRY(hipMemcpyAsync(c->meta, c->h_meta, sizeof(Meta), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->steps = steps;
    return NBODY_OK;
}

int nbody_body_count(nbody_ctx* c, int* n) {
    if (!c || !n) return nbody_fail(NBODY_ERR_INVALID, "nbody_body_count: NULL");
    HIP_TRY(hipSetDevice(c->desc.device));
    int rc = read_meta(c);
    if (rc != NBODY_OK) return rc;
    *n = c->h_meta->n;
    return NBODY_OK;
}

int nbody_own_range(nbody_ctx* c, int* lo, int* cnt) {
    if (!c || !lo || !cnt) return nbody_fail(NBODY_ERR_INVALID, "nbody_own_range: NULL");
    HIP_TRY(hipSetDevice(c->desc.device));
    int rc = read_meta(c);
    if (rc != NBODY_OK) return rc;
    *lo = c->h_meta->lo;
    *cnt = c->h_meta->cnt;
    return NBODY_OK;
}

int nbody_download(nbody_ctx* c, void* block, int* n_out) {
    if (!c || !block || !n_out) return nbody_fail(NBODY_ERR_INVALID, "nbody_download: NULL argument");
    if (!c->uploaded) return nbody_fail(NBODY_ERR_STATE, "nbody_download before nbody_upload");
    HIP_TRY(hipSetDevice(c->desc.device));
    int rc = read_meta(c);
    if (rc != NBODY_OK) return rc;
    const int n = c->h_meta->n, lo = c->h_meta->lo, cnt = c->h_meta->cnt;
    const size_t rb = c->real_bytes;
    HIP_TRY(hipMemcpyAsync(c->h_stage, c->J, (size_t)n * c->rec_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned char* P = (unsigned char*)block;
    unsigned char* V = P + 2 * rb * (size_t)n;
    unsigned char* M = V + 2 * rb * (size_t)n;
    unsigned char* R = M + rb * (size_t)n;
    const unsigned char* st = (const unsigned char*)c->h_stage;
    for (int i = 0; i < n; ++i) {
        memcpy(P + 2 * rb * i, st + c->rec_bytes * i, 2 * rb);
        memcpy(M + rb * i, st + c->rec_bytes * i + 2 * rb, rb);
        memcpy(R + rb * i, st + c->rec_bytes * i + 3 * rb, rb);
    }
    // velocities: own range from this rank; other ranks' through the same all-gather interface as the step's exchange
    if (c->desc.world == 1 && !c->comm) {
        HIP_TRY(hipMemcpyAsync(V, c->Vown, (size_t)cnt * 2 * rb, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else if (c->comm || c->peers) {
        // padded all-gather of the own velocities into the slot receive area: own_upper_of(n) vec2 entries per rank (n is
        // exact and the same on every rank here: all have synchronised); world of them always fit there (a slot holds
        // records of 4 reals + velocities of 2 reals for at least that many bodies)
        const size_t vbytes = (size_t)own_upper_of(c, n) * 2 * rb;
        if (vbytes * c->desc.world > c->slot_bytes * c->desc.world)
            return nbody_fail(NBODY_ERR_CAPACITY, "velocity gather does not fit the slot area");
        rc = all_gather(c, Part::Velocities, c->gather, vbytes);
        if (rc != NBODY_OK) return rc;
        // every rank's {lo, cnt}: a second, tiny all-gather of the device-resident Meta (the slot headers hold
        // counts only after a step has run)
        rc = all_gather(c, Part::MetaBlock, (unsigned char*)c->meta_all, sizeof(Meta));
        if (rc != NBODY_OK) return rc;
        std::vector<Meta> h_all(c->desc.world);
        HIP_TRY(hipMemcpyAsync(h_all.data(), c->meta_all, sizeof(Meta) * c->desc.world, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int g = 0; g < c->desc.world; ++g) {
            if (h_all[g].n != n || h_all[g].lo < 0 || h_all[g].cnt < 0 || (size_t)h_all[g].cnt * 2 * rb > vbytes ||
                (long long)h_all[g].lo + h_all[g].cnt > n)
                return nbody_fail(NBODY_ERR_STATE, "rank %d reports range [%d, +%d) of %d bodies, this rank has %d bodies",
                                  g, h_all[g].lo, h_all[g].cnt, h_all[g].n, n);
            HIP_TRY(hipMemcpy(V + 2 * rb * (size_t)h_all[g].lo, c->gather + (size_t)g * vbytes,
                              (size_t)h_all[g].cnt * 2 * rb, hipMemcpyDeviceToHost));
        }
    } else {
        // a group context on its own: only the own range lives here; nbody_group_download assembles the whole state
        memset(V, 0, 2 * rb * (size_t)n);
        HIP_TRY(hipMemcpyAsync(V + 2 * rb * (size_t)lo, c->Vown, (size_t)cnt * 2 * rb, hipMemcpyDeviceToHost,
                               c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    *n_out = n;
    return NBODY_OK;
}

int nbody_render_image(nbody_ctx* c, unsigned char* img, int width, int height) {
    if (!c || !img || width <= 0 || height <= 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_render_image: bad argument");
    if (!c->uploaded) return nbody_fail(NBODY_ERR_STATE, "nbody_render_image before nbody_upload");
    HIP_TRY(hipSetDevice(c->desc.device));
    int rc0 = read_meta(c);
    if (rc0 != NBODY_OK) return rc0;
    const size_t bytes = (size_t)width * height;
    if (bytes > c->d_img_bytes) {
        hipFree(c->d_img);
        c->d_img = nullptr; c->d_img_bytes = 0;
        HIP_TRY(hipMalloc((void**)&c->d_img, bytes));
        c->d_img_bytes = bytes;
    }
    HIP_TRY(hipMemsetAsync(c->d_img, 254, bytes, c->stream));                 // src/nbody.cu:534
    const int literal = c->desc.semantics == NBODY_LITERAL;
    const int grid = (c->n_upper + 255) / 256 > 0 ? (c->n_upper + 255) / 256 : 1;
    if (c->desc.precision == NBODY_F64)
        hipLaunchKernelGGL(render_discs<double>, dim3(grid), dim3(256), 0, c->stream, (const Rec<double>*)c->J,
                           (const Meta*)c->meta, literal, c->d_img, width, height, c->desc.fieldWidth, c->desc.fieldHeight);
    else
        hipLaunchKernelGGL(render_discs<float>, dim3(grid), dim3(256), 0, c->stream, (const Rec<float>*)c->J,
                           (const Meta*)c->meta, literal, c->d_img, width, height, c->desc.fieldWidth, c->desc.fieldHeight);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(img, c->d_img, bytes, hipMemcpyDeviceToHost, c->stream));   // :537
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NBODY_OK;
}

int nbody_get_events(nbody_ctx* c, nbody_event* out, int cap, int64_t* total) {
    if (!c || !total || cap < 0 || (cap > 0 && !out)) return nbody_fail(NBODY_ERR_INVALID, "nbody_get_events: bad argument");
    HIP_TRY(hipSetDevice(c->desc.device));
    HIP_TRY(hipMemcpyAsync(c->h_counters, c->counters, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const unsigned long long tot = c->h_counters->events;
    *total = (int64_t)tot;
    unsigned long long ncopy = tot;
    if (ncopy > (unsigned long long)c->ev_cap) ncopy = c->ev_cap;
    if (ncopy > (unsigned long long)cap) ncopy = cap;
    static_assert(sizeof(nbody_event) == sizeof(Event), "event layouts must match");
    if (ncopy) HIP_TRY(hipMemcpy(out, c->events, ncopy * sizeof(Event), hipMemcpyDeviceToHost));
    return NBODY_OK;
}

int nbody_clear_events(nbody_ctx* c) {
    if (!c) return nbody_fail(NBODY_ERR_INVALID, "NULL context");
    HIP_TRY(hipSetDevice(c->desc.device));
    HIP_TRY(hipMemsetAsync(&c->counters->events, 0, sizeof(unsigned long long), c->stream));
    return NBODY_OK;
}

int nbody_set_kernel_timing(nbody_ctx* c, int enable) {
    if (!c) return nbody_fail(NBODY_ERR_INVALID, "NULL context");
    HIP_TRY(hipSetDevice(c->desc.device));
    if (enable)                                            // the events of the timed launches are created HERE
        while (c->ev_free.size() + 2 * c->ev_pending.size() < (size_t)kTimingPool) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            c->ev_free.push_back(e);
        }
    c->timing = enable != 0;
    return NBODY_OK;
}

int nbody_get_stats(nbody_ctx* c, nbody_stats* out) {
    if (!c || !out) return nbody_fail(NBODY_ERR_INVALID, "nbody_get_stats: NULL");
    HIP_TRY(hipSetDevice(c->desc.device));
    int rc = read_meta(c);
    if (rc != NBODY_OK) return rc;
    rc = resolve_timing(c);
    if (rc != NBODY_OK) return rc;
    out->steps = c->steps;
    out->pairs = (int64_t)c->h_counters->pairs;
    out->force_kernel_ms = c->force_ms;
    out->force_kernel_launches = c->force_launches;
    out->n_bodies = c->h_meta->n;
    out->n_own = c->h_meta->cnt;
    out->exchange_ms = c->xchg_ms;
    out->exchange_launches = c->xchg_launches;
    out->exchange_bytes = c->xchg_bytes;
    out->slot_bytes_now = c->gather == c->slot ? 0 : (int64_t)slot_stride(c);
    return NBODY_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Reference-shaped launches (src/nbody.cu:481-483)
// ---------------------------------------------------------------------------------------------------------
// Workspace of the reference-shaped launch: a context on the CURRENT device, grown to the largest body count seen.
// Process-wide and not re-entrant, like the reference's own loop (one host thread, src/nbody.cu:373).
namespace {
nbody_ctx* g_ref_ws = nullptr;
int g_ref_ws_device = -1;
int ref_launch_workspace(int n, nbody_ctx** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (g_ref_ws && (g_ref_ws_device != dev || g_ref_ws->cap < n)) {
        nbody_ctx_destroy(g_ref_ws);
        g_ref_ws = nullptr;
    }
    if (!g_ref_ws) {
        nbody_ctx_desc d{};
        d.precision = NBODY_F32; d.semantics = NBODY_LITERAL; d.capacity = n; d.device = dev; d.rank = 0; d.world = 1;
        d.event_capacity = 16;
        int rc = nbody_ctx_create(&g_ref_ws, &d);
        if (rc != NBODY_OK) return rc;
        g_ref_ws_device = dev;
    }
    *out = g_ref_ws;
    return NBODY_OK;
}
}  // namespace

int nbody_launch_workspace_release(void) {
    int rc = NBODY_OK;
    if (g_ref_ws) {
        hipSetDevice(g_ref_ws->desc.device);
        hipDeviceSynchronize();                            // launches through the workspace ran on the callers' streams
        rc = device_failure(g_ref_ws, *(volatile unsigned long long*)&g_ref_ws->h_counters_async->errors);
        nbody_ctx_destroy(g_ref_ws);
        g_ref_ws = nullptr;
        g_ref_ws_device = -1;
    }
    return rc;
}

int nbody_launch_compute_forces_f32(void* d_bodyData, float* d_updM, float* d_updR, int numBodies,
                                    float timestep, int fieldWidth, int fieldHeight, int numBlocks,
                                    float growthRate, void* stream) {
    if (!d_bodyData || !d_updM || !d_updR || numBodies <= 0 || numBlocks <= 0)
        return nbody_fail(NBODY_ERR_INVALID, "nbody_launch_compute_forces_f32: bad argument");
    nbody_ctx_desc d{};
    d.timestep = timestep; d.growthRate = growthRate; d.fieldWidth = fieldWidth; d.fieldHeight = fieldHeight;
    d.semantics = NBODY_LITERAL;
    const StepParams<float> p = make_params<float>(d);
    const char* force_general = getenv("NBODY_REF_LAUNCH_GENERAL");        // testing aid
    const char* one_lane = getenv("NBODY_REF_LAUNCH_ONE_LANE");            // testing aid
    if (numBlocks == nbody_num_blocks(numBodies) && !(force_general && force_general[0] == '1') &&
        !(one_lane && one_lane[0] == '1')) {
        // the reference's own launch geometry (src/nbody.cu:473): the production (ring) kernel, through a process-wide
        // workspace that holds the {x,y,m,r} replica and the staged output - what a context keeps resident.  Bodies
        // past the last full block get no thread in the reference and are left untouched here too.
        nbody_ctx* c = nullptr;
        int rc = ref_launch_workspace(numBodies, &c);
        if (rc != NBODY_OK) return rc;
        hipStream_t s = (hipStream_t)stream;
        // CUDA_SYNC_CHECK's convention (src/nbody.cu:20-33): a device-side failure of an EARLIER launch through the
        // workspace (a hand-off time-out poisons the caller's block with NaN) is reported by the next call that looks
        rc = device_failure(c, *(volatile unsigned long long*)&c->h_counters_async->errors);
        if (rc != NBODY_OK) return rc;
        HIP_TRY(hipMemsetAsync(c->meta, 0, sizeof(Meta), s));
        HIP_TRY(hipMemsetAsync(c->tile_rmax, 0, sizeof(unsigned) * (size_t)c->n_tiles, s));
        hipLaunchKernelGGL(ref_layout_pack_f32, dim3((numBodies + 255) / 256), dim3(256), 0, s, (const void*)d_bodyData,
                           numBodies, (Rec<float>*)c->J, c->meta, c->tile_rmax, c->Jt);
        // the caller's stream, and the velocities where they lie in the caller's block
        c->own_upper = c->n_upper = numBodies;             // (kernel choice by size, as in a context of this many bodies)
        StepParams<float> pr = p;
        pr.spin_limit = c->spin_limit;
        launch_forces<float>(c, pr, numBodies / kTile > 0 ? numBodies / kTile : 1, false,
                             LaunchTarget{s, (const float*)d_bodyData + 2 * (size_t)numBodies});
        const int n_active = numBodies < kTile ? numBodies : (numBodies / kTile) * kTile;
        hipLaunchKernelGGL(ref_layout_finish_f32, dim3((n_active + 255) / 256), dim3(256), 0, s, d_bodyData, d_updM, d_updR,
                           numBodies, n_active, (const Rec<float>*)c->S_J, (const Vec2<float>*)c->S_V);
        HIP_TRY(hipMemcpyAsync(c->h_counters_async, c->counters, sizeof(Counters), hipMemcpyDeviceToHost, s));
    } else if (numBlocks == nbody_num_blocks(numBodies) && !(force_general && force_general[0] == '1')) {
        // NBODY_REF_LAUNCH_ONE_LANE=1: the one-lane-per-body kernel directly on the block layout, no workspace (round 1's
        // form of this launch; kept as a second implementation the tests compare)
        hipLaunchKernelGGL(ref_layout_forces_v3_f32, dim3((numBodies / kTile + 2) / 2), dim3(2 * kTile), 0,
                           (hipStream_t)stream, d_bodyData, d_updM, d_updR, numBodies, p);
    } else {
        // any other block count changes which bodies are active and how many tiles are walked: general kernel
        hipLaunchKernelGGL(ref_layout_forces_f32, dim3(numBlocks), dim3(kTile), 0, (hipStream_t)stream,
                           d_bodyData, d_updM, d_updR, numBodies, numBlocks, p);
    }
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

int nbody_launch_move_bodies_f32(void* d_bodyData, const float* d_updM, const float* d_updR, int numBodies,
                                 float timestep, int numBlocks, void* stream) {
    if (!d_bodyData || !d_updM || !d_updR || numBodies <= 0 || numBlocks <= 0)
        return nbody_fail(NBODY_ERR_INVALID, "nbody_launch_move_bodies_f32: bad argument");
    hipLaunchKernelGGL(ref_layout_move_f32, dim3(numBlocks), dim3(kTile), 0, (hipStream_t)stream, d_bodyData,
                       d_updM, d_updR, numBodies, timestep);
    HIP_TRY(hipGetLastError());
    return NBODY_OK;
}

const char* nbody_force_kernel_name(nbody_ctx* c) {
    if (!c) return "";
    if (c->desc.precision == NBODY_F64)
        return c->desc.kernel_variant == 1 ? "forces_v1<double>" : "forces_v3w_f64";
    switch (c->desc.kernel_variant) {
        case 1: return "forces_v1<float>";
        case 11: case 12: case 14: case 18: return "forces_v3_f32";
        case 31: case 32: return "forces_v3w_f32";
        case 50: case 55: case 56: case 58: return "forces_ring_f32 (2 rings x 8 waves per workgroup)";
        case 52: return "forces_ring_f32 (4 rings x 4 waves per workgroup)";
        case 53: return "forces_ring_f32 (2 rings x 8 waves, 16-position turns)";
        case 54: return "forces_ring_f32 (1 ring x 8 waves per workgroup)";
        default: break;
    }
    if (c->own_upper >= 49152) return "forces_ring_f32 (4 rings x 4 waves per workgroup)";
    if (c->own_upper >= 24576) return "forces_ring_f32 (2 rings x 8 waves per workgroup)";
    return "forces_ring_f32 (1 ring x 8 waves per workgroup)";
}

int nbody_debug_force_only(nbody_ctx* c, int reps) {
    if (!c || reps < 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_debug_force_only: bad argument");
    if (!c->uploaded) return nbody_fail(NBODY_ERR_STATE, "nbody_debug_force_only before nbody_upload");
    HIP_TRY(hipSetDevice(c->desc.device));
    const int nblocks = c->own_upper / kTile > 0 ? c->own_upper / kTile : 1;
    for (int r = 0; r < reps; ++r) {
        int rc = c->desc.precision == NBODY_F64
                     ? launch_force_kernel<double>(c, make_params<double>(c->desc, c->spin_limit), nblocks)
                     : launch_force_kernel<float>(c, make_params<float>(c->desc, c->spin_limit), nblocks);
        if (rc != NBODY_OK) return rc;
    }
    return NBODY_OK;
}

int nbody_debug_ring_probe(nbody_ctx* c, uint64_t out[8]) {
    if (!c || !out) return nbody_fail(NBODY_ERR_INVALID, "nbody_debug_ring_probe: NULL");
    HIP_TRY(hipSetDevice(c->desc.device));
    int rc = read_meta(c);
    if (rc != NBODY_OK) return rc;
    for (int k = 0; k < 8; ++k) out[k] = c->h_counters->probe[k];
    return NBODY_OK;
}

int nbody_selftest_lds_record(int device, int iters, uint64_t result[3]) {
    if (!result || iters <= 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_selftest_lds_record: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nbody_fail(NBODY_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 3 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d, 0, 3 * sizeof(unsigned long long), 0));
    hipLaunchKernelGGL(selftest_lds_record, dim3(512), dim3(8 * kWave), 0, 0, d, iters);
    HIP_TRY(hipGetLastError());
    unsigned long long h[3];
    HIP_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(d);
    for (int k = 0; k < 3; ++k) result[k] = h[k];
    return NBODY_OK;
}

int nbody_selftest_ieee_f32(int device, uint64_t mismatches[3]) {
    if (!mismatches) return nbody_fail(NBODY_ERR_INVALID, "nbody_selftest_ieee_f32: NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nbody_fail(NBODY_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 3 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(d, 0, 3 * sizeof(unsigned long long)));
    hipLaunchKernelGGL(selftest_ieee_f32, dim3(256 * 16), dim3(256), 0, 0, d);
    HIP_TRY(hipGetLastError());
    unsigned long long h[3];
    HIP_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(d);
    mismatches[0] = h[0];
    mismatches[1] = h[1];
    mismatches[2] = h[2];
    return NBODY_OK;
}

int nbody_selftest_chain_f64(int device, uint64_t inputs_per_mode, uint64_t mismatches[2]) {
    if (!mismatches || inputs_per_mode == 0) return nbody_fail(NBODY_ERR_INVALID, "nbody_selftest_chain_f64: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nbody_fail(NBODY_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 2 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d, 0, 2 * sizeof(unsigned long long), 0));
    const int blocks = 4096;
    const uint64_t per_thread = (inputs_per_mode + (uint64_t)blocks * 256 - 1) / ((uint64_t)blocks * 256);
    const int iters = per_thread > 0x7fffffff ? 0x7fffffff : (int)per_thread;
    for (int mode = 0; mode < 3; ++mode)
        hipLaunchKernelGGL(selftest_chain_f64, dim3(blocks), dim3(256), 0, 0, d, mode, iters,
                           0xabcdefull + (unsigned long long)mode);
    HIP_TRY(hipGetLastError());
    unsigned long long h[2];
    HIP_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(d);
    mismatches[0] = h[0];
    mismatches[1] = h[1];
    return NBODY_OK;
}

int nbody_selftest_rcp_ones_f64(int device, uint64_t result[5]) {
    if (!result) return nbody_fail(NBODY_ERR_INVALID, "nbody_selftest_rcp_ones_f64: NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return nbody_fail(NBODY_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 5 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d, 0, 5 * sizeof(unsigned long long), 0));
    hipLaunchKernelGGL(selftest_rcp_ones_f64, dim3(6), dim3(256), 0, 0, d);
    HIP_TRY(hipGetLastError());
    unsigned long long h[5];
    HIP_TRY(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(d);
    for (int k = 0; k < 5; ++k) result[k] = h[k];
    return NBODY_OK;
}

}  // extern "C"
