// csrc/nbody_kernels.hpp -- gfx950 (CDNA4, MI355X) device code of the stepper.
//
// What is computed is fixed by the reference (src/nbody.cu:126-292 of the reference);
// how it is computed is not: nothing here follows the reference's kernel structure.
//
//   * Bodies live on the device as 16-byte (fp32) / 32-byte (fp64) records {x, y, m, r}: one vector load
//     per body when a tile is staged, one ds_read_b128 per pair in the hot loop.
//   * The force/collision kernel keeps the reference's per-body ACCUMULATION ORDER (tile k of block b holds
//     the cyclic bodies 128(b+k)..+127 mod N; lane t walks a tile as (t+off) mod L) because fp32 sums are
//     order-sensitive beyond the 1e-5 budget: with IEEE sqrt/divide and no contraction the
//     results are bit-identical to the CPU oracle.  The rotated walk is a conflict-free LDS pattern on
//     CDNA4: the 64 lanes of a wave read 64 consecutive 16-byte records.
//   * The drift (MoveBodies) is fused into the force kernel's epilogue; results go to a staging array in
//     step-t index space, a stable on-device compaction then builds step t+1 (no host round trip).
//
// All floating-point statements are written one rounding per operation and the file is compiled with
// -ffp-contract=off; `#pragma clang fp contract(off)` below makes that independent of the command line.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "nbody_partition.h"

#pragma clang fp contract(off)

namespace nbk {

constexpr int kTile = 128;          // bodies per reference block / tile (THREADS_PER_BLOCK, src/nbody.cu:36)
constexpr int kWave = 64;

template <typename T> struct Rec;   // {x, y, m, r}
template <> struct alignas(16) Rec<float> { float x, y, m, r; };
template <> struct alignas(32) Rec<double> { double x, y, m, r; };
template <typename T> struct alignas(2 * sizeof(T)) Vec2 { T x, y; };

// An fp64 record as it is kept in an LDS tile: 48 bytes apart instead of 32.  64 lanes reading consecutive 32-byte records
// as two 16-byte halves are 2-way bank conflicted (a 16-lane group spans 512 bytes = two bank rows: 66 % of the LDS-array
// cycles of the fp64 kernel were conflicts, profiles/r02_pmc_f64.txt); at a 48-byte stride the 16 lanes of a group fall
// on 16 different 4-bank slots: conflict-free for both halves.
struct alignas(16) RecPadded {
    double x, y, m, r, pad0, pad1;
    __device__ __forceinline__ RecPadded& operator=(const Rec<double>& o) { x = o.x; y = o.y; m = o.m; r = o.r; return *this; }
    __device__ __forceinline__ operator Rec<double>() const { return Rec<double>{x, y, m, r}; }
};
static_assert(sizeof(RecPadded) == 48, "48-byte LDS stride");

// Step-resident scalars, device memory.  Written by the unpack kernel, read by everything else.
struct Meta {
    int n;        // global body count N_t
    int lo;       // first global index owned by this rank
    int cnt;      // number of bodies owned by this rank
    int step;     // step counter since upload
    int n_prev;   // body count before the last step's compaction (the reference renders with its block count)
    int summary;  // of the replica: bit 0 some coordinate is not below kCoordBound in magnitude (or NaN), bit 1 some
                  // radius is not +0, bit 2 some coordinate is below kCoordFloor in magnitude (zero included), bit 3 some
                  // mass is not below kMassBound in magnitude (or NaN).  OR-ed together by unpack_slots (cleared by
                  // compact_count), set by nbody_upload.
    int pad[2];
};
constexpr int kSummaryUnbounded = 1, kSummaryRadius = 2, kSummarySmall = 4, kSummaryMass = 8;

struct Event { int32_t step, i, j, kind; };

struct Counters {
    unsigned long long pairs;     // ordered pairs evaluated (this rank)
    unsigned long long events;    // events logged (may exceed capacity: overflow is counted, not stored)
    unsigned long long errors;    // device-side failures: in-kernel hand-off waits that timed out (low half) and index
                                  // checks that failed (kIndexError each: high half); either makes the host fail loudly
    unsigned long long probe[8];  // tuning builds of the ring kernel: cycle totals per phase (kProbe)
};

// Send slot of one rank for the per-step exchange: header + compacted survivors of the own range.
struct SlotHeader {
    int count;
    int layout;                  // bodies the slot is laid out for (records at +32, velocities behind `layout` records)
    int pad[6];
};
static_assert(sizeof(SlotHeader) == 32, "slot header is 32 bytes so fp64 records stay 32-byte aligned");

template <typename T>
struct StepParams {
    T dt;
    T growth;
    T G;
    T wall_hi_x, wall_lo_x;   // (T)fieldWidth, (T)(-fieldWidth): the int->real conversions of :256-257
    T wall_hi_y, wall_lo_y;
    int literal;              // 1: reference index semantics, 0: clean all-pairs
    int spin_limit;           // ring kernel: polls of a hand-off record before the wait is declared failed
    int rotate_priority;      // one-lane kernels built with NB_V3_ROTATE_PRIORITY: the launch is a single round
};

template <typename T> __device__ __forceinline__ T ieee_sqrt(T x);
template <> __device__ __forceinline__ float ieee_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ __forceinline__ double ieee_sqrt<double>(double x) { return __builtin_sqrt(x); }

// 1 / c, correctly rounded (vec2f.h:52 / vec2.h:48: `1.0f / val`).  fp32: the compiler's expansion, proved on all 2^32 inputs
// (nbody_selftest_ieee_f32).  fp64 cannot be enumerated, and Newton-type refinements - the compiler's expansion and
// the fast chain below alike - have ONE known exception (Markstein): a significand of all ones, c = (2 - 2^-52) 2^k.
// There 1 / c = 2^-(k+1) (1 + 2^-53 + 2^-106 + ...) lies 2^-107 relative ABOVE the midpoint of two neighbours and the
// last fma of the refinement sees an exact tie, which it rounds to even - one ulp low.  The correctly rounded value is
// known in closed form, 2^-(k+1) (1 + 2^-52), and is written down for that significand (nbody_selftest_rcp_ones_f64).
template <typename T> __device__ __forceinline__ T ieee_rcp(T c);
template <> __device__ __forceinline__ float ieee_rcp<float>(float c) { return 1.0f / c; }
constexpr unsigned long long kF64Frac = 0x000fffffffffffffull;
__device__ __forceinline__ bool significand_all_ones(double c) {
    return ((unsigned long long)__double_as_longlong(c) & kF64Frac) == kF64Frac;
}
template <> __device__ __forceinline__ double ieee_rcp<double>(double c) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(c);
    const unsigned long long e = (b >> 52) & 0x7ffull;    // biased exponent; the result's is 2045 - e
    if (significand_all_ones(c) && e >= 1 && e <= 2044)
        return __longlong_as_double((long long)((b & 0x8000000000000000ull) | ((2045ull - e) << 52) | 1ull));
    return 1.0 / c;
}

// ---------------------------------------------------------------------------------------------------------
// One (i, j) interaction, SURVEY.md A.1 step 2 (src/nbody.cu:210-239, include/vec2f.h:45-93).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct BodyAcc {
    T xi, yi, mi, ri;      // start-of-step snapshot (:169-171)
    T fx, fy;              // force accumulator (:153)
    T mnew, rnew;          // :174-175
    int deleted;           // :180
};

template <typename T, bool kLog>
__device__ __forceinline__ void interact(BodyAcc<T>& a, const Rec<T>& bj, T growth, int i, int j,
                                         Event* ev, int ev_cap, Counters* ctr, int step) {
    // :129-133 areParticlesColliding
    const T dx = bj.x - a.xi;
    const T dy = bj.y - a.yi;
    const T d2 = (dx * dx) + (dy * dy);
    const T rs = a.ri + bj.r;
    const bool hit = d2 <= rs * rs;
    const bool ge = a.mi >= bj.m;
    const bool lt = a.mi < bj.m;
    if (__builtin_expect(hit && (ge || lt), 0)) {
        if (ge) {                       // :215-221
            a.mnew = a.mnew + bj.m;
            a.rnew = a.rnew + bj.r * growth;
        } else {                        // :222-226
            a.deleted = 1;
        }
        if (kLog) {
            const unsigned long long slot = atomicAdd(&ctr->events, 1ull);
            if (slot < (unsigned long long)ev_cap) ev[slot] = Event{step, i, j, ge ? 0 : 1};
        }
        return;
    }
    // :230-239 (a NaN mass with hit falls through to here, as in the reference's if / else-if)
    const T d = ieee_sqrt<T>(d2);       // same operands, same rounding as the recomputation at :232
    const T c = (d * d) * d;
    const T inv = ieee_rcp<T>(c);       // vec2f.h:52
    const T mx = bj.m * dx;             // vec2f.h:45-47
    const T my = bj.m * dy;
    a.fx = a.fx + inv * mx;             // vec2f.h:52, :83-85
    a.fy = a.fy + inv * my;
}

// Epilogue of one active body: SURVEY.md A.1 steps 3-7 (src/nbody.cu:245-264 and MoveBodies :288-290).
template <typename T>
__device__ __forceinline__ void finish_body(const BodyAcc<T>& a, Vec2<T> v, const StepParams<T>& p,
                                            Rec<T>& out, Vec2<T>& vout) {
    const T um = a.deleted ? (T)0 : a.mnew;                      // :245
    const T ur = a.rnew;                                         // :246
    const T ax = p.G * a.fx, ay = p.G * a.fy;                    // :250
    const T dvx = p.dt * ax, dvy = p.dt * ay;                    // :252
    const T tx = a.xi + (ax * p.dt), ty = a.yi + (ay * p.dt);    // :256-260
    if (tx > p.wall_hi_x - a.ri || tx < p.wall_lo_x + a.ri) v.x = v.x * (T)(-1);
    if (ty > p.wall_hi_y - a.ri || ty < p.wall_lo_y + a.ri) v.y = v.y * (T)(-1);
    v.x = v.x + dvx;                                             // :264
    v.y = v.y + dvy;
    vout = v;
    out.x = a.xi + p.dt * v.x;                                   // :288
    out.y = a.yi + p.dt * v.y;
    out.m = um;                                                  // :289
    out.r = ur;                                                  // :290
}

// ---------------------------------------------------------------------------------------------------------
// Force + collision + drift kernel, variant "v1": 128-lane workgroup = one reference block, one body per
// lane, tiles double-buffered in LDS (one barrier per tile).
//
//   J        replica of all bodies, global index space of step t
//   Vown     velocities of the own range, local index q = i - lo
//   S_J,S_V  staged post-step state of the own range (still step-t index space, before compaction)
// ---------------------------------------------------------------------------------------------------------
template <typename T, bool kLog>
__global__ __launch_bounds__(kTile) void forces_v1(const Rec<T>* __restrict__ J,
                                                   const Vec2<T>* __restrict__ Vown,
                                                   Rec<T>* __restrict__ S_J, Vec2<T>* __restrict__ S_V,
                                                   const Meta* __restrict__ meta, StepParams<T> p,
                                                   Event* ev, int ev_cap, Counters* ctr) {
    __shared__ Rec<T> tile[2][kTile];
    const int N = meta->n, lo = meta->lo, cnt = meta->cnt, step = meta->step;
    const int t = threadIdx.x;
    const int b = lo / kTile + blockIdx.x;                 // reference block index
    const long long blk0 = (long long)b * kTile;
    if (blk0 >= (long long)lo + cnt) return;               // grid is sized for the capacity; whole-WG exit
    const long long i64 = blk0 + t;
    const int i = (int)(i64 < 0x7fffffff ? i64 : 0x7fffffff);
    const int nb = N < kTile ? 1 : N / kTile;              // src/nbody.cu:473
    const bool lit = p.literal != 0;                       // else NBODY_CLEAN: see forces_v3_f32
    const int ntiles = lit ? nb : (N + kTile - 1) / kTile;
    const bool mine = i64 >= lo && i64 < (long long)lo + cnt;
    // literal: only bodies with a thread are updated (quirk Q2); N < 128: the single block is guarded by i < N
    const bool active = mine && i64 < N && (!lit || i64 < (long long)nb * kTile);

    BodyAcc<T> a;
    Vec2<T> v{0, 0};
    if (mine) {
        const Rec<T> me = J[i];
        a.xi = me.x; a.yi = me.y; a.mi = me.m; a.ri = me.r;
        v = Vown[i - lo];
    } else {
        a.xi = a.yi = a.mi = a.ri = 0;
    }
    a.fx = 0; a.fy = 0; a.mnew = a.mi; a.rnew = a.ri; a.deleted = 0;
    unsigned long long pairs = 0;

    long long start = lit ? blk0 % N : 0;
    auto entry_index = [&](long long st) -> int {          // this lane's entry of the tile starting at st, or -1
        long long src = st + t;
        if (!lit) return src < N ? (int)src : -1;
        if (N < kTile && t >= N) return -1;                // lanes >= N load nothing (:143)
        if (src >= N) src -= N;
        if (src >= N) src %= N;
        return (int)src;                                   // :186
    };
    {
        const int e = entry_index(start);
        if (e >= 0) tile[0][t] = J[e];
    }
    __syncthreads();
    for (int k = 0; k < ntiles; ++k) {
        const int cur = k & 1;
        // prefetch tile k+1 into registers while tile k is consumed
        Rec<T> nxt{};
        const bool have_next = (k + 1 < ntiles);
        long long next_start = start + kTile;
        if (lit) while (next_start >= N) next_start -= N;
        const int e_next = have_next ? entry_index(next_start) : -1;
        if (e_next >= 0) nxt = J[e_next];
        int L;
        if (lit) L = (k == nb - 1) ? N % (kTile + 1) : kTile;                 // :194 (quirk Q1)
        else L = (N - start) < kTile ? (int)(N - start) : kTile;
        if (active) {
            if (lit) {
                for (int off = (k == 0 ? 1 : 0); off < L; ++off) {           // :200-204 skip (k=0, off=0)
                    const int s = (L == kTile) ? ((t + off) & (kTile - 1)) : ((t + off) % L);   // :207
                    long long j = start + s;
                    if (j >= N) j %= N;
                    interact<T, kLog>(a, tile[cur][s], p.growth, i, (int)j, ev, ev_cap, ctr, step);
                }
                pairs += (k == 0) ? (L > 0 ? L - 1 : 0) : L;
            } else {
                for (int off = 0; off < L; ++off) {
                    if (start + off == i64) continue;
                    interact<T, kLog>(a, tile[cur][off], p.growth, i, (int)(start + off), ev, ev_cap, ctr, step);
                }
                pairs += L - ((i64 >= start && i64 < start + L) ? 1 : 0);
            }
        }
        if (e_next >= 0) tile[cur ^ 1][t] = nxt;
        __syncthreads();
        start = next_start;
    }

    if (mine) {
        const int q = i - lo;
        if (active) {
            Rec<T> out; Vec2<T> vout;
            finish_body<T>(a, v, p, out, vout);
            S_J[q] = out;
            S_V[q] = vout;
        } else {   // frozen body: no thread exists for it in the reference, state carried over unchanged
            S_J[q] = Rec<T>{a.xi, a.yi, a.mi, a.ri};
            S_V[q] = v;
        }
    }
    // one atomic per wave for the pair counter
    for (int sh = kWave / 2; sh > 0; sh >>= 1) pairs += __shfl_down(pairs, sh, kWave);
    if ((t & (kWave - 1)) == 0 && pairs) atomicAdd(&ctr->pairs, pairs);
}

// ---------------------------------------------------------------------------------------------------------
// Fast exact fp32 pair evaluation.
//
// The reference's per-pair arithmetic needs d = RN(sqrt(d2)) and inv = RN(1 / RN(RN(d*d)*d)).  hipcc's
// correctly-rounded sqrt/divide expansions cost ~30 VALU instructions per pair (range scaling, div_scale /
// div_fmas / div_fixup).  For d2 in [2^-80, 2^80] the short chains below give the SAME BITS for every input
// (proved by enumeration over all fp32 values on the device: csrc/tune/chain_probe.hip during development,
// nbody_selftest_ieee_f32 in the test-suite):
//     y = v_rsq_f32(d2);  g = d2*y;  h = 0.5*y;  d = fma(fma(-g, g, d2), h, g)          (sqrt, 1 trans + 4)
//     r = v_rcp_f32(c);   inv = fma(fma(-c, r, 1), r, r)                                 (rcp,  1 trans + 2)
// Pairs outside that domain, colliding pairs and non-finite inputs never take this path: a conservative
// per-pair test (d2 <= fma(rs, rs, 2^-80), which is a superset of `hit` and of d2 <= 2^-80) is OR-ed into a
// wave mask, coordinates are bounded per tile, and a flagged (wave, tile) is redone from a snapshot with the
// general code (`interact`).
// ---------------------------------------------------------------------------------------------------------
constexpr float kFastLo = 0x1p-80f;        // lower edge of the proved domain (and the flag margin)
constexpr float kFastHi = 0x1p80f;         // upper edge of the proved domain
constexpr float kCoordBound = 0x1p38f;     // |x|,|y| < 2^38 for both bodies  =>  d2 <= 2^79 < kFastHi
// |x|,|y| >= 2^-16 for both bodies  =>  a coordinate difference is 0 or at least 2^-39 (the spacing of fp32 at 2^-16), so
// d2 is exactly 0 or at least 2^-78: with all radii +0 the only pairs outside the fast chain's domain - and the only
// collisions - are COINCIDENT bodies, and those turn their term into NaN (v_rsq_f32(0) = inf, 0 * inf = NaN): the ring
// kernel then needs no screen per pair at all, it looks at the sum after the turn's adds (see its turn loop).
constexpr float kCoordFloor = 0x1p-16f;
// That look at the sum also fires for a sum that is NaN for any other reason, and a lane whose sum is NaN takes the exact
// path in every turn from then on: right, but slow.  A NaN or infinite mass would do that to every body at once, so the
// NaN-sum screen is only used while every mass of the replica is finite and below 2^90 (a term then only overflows for
// bodies closer than the coordinates' spacing allows at ordinary magnitudes).
constexpr float kMassBound = 0x1p90f;
__device__ __forceinline__ int coord_summary(float x, float y, float m) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    return ((ax < kCoordBound && ay < kCoordBound) ? 0 : kSummaryUnbounded) |
           ((ax >= kCoordFloor && ay >= kCoordFloor) ? 0 : kSummarySmall) |
           (__builtin_fabsf(m) < kMassBound ? 0 : kSummaryMass);
}

struct FastChain { float d, inv; };
__device__ __forceinline__ FastChain fast_chain(float d2) {
    const float y = __builtin_amdgcn_rsqf(d2);
    const float g = d2 * y;
    const float h = 0.5f * y;
    const float e = __builtin_fmaf(-g, g, d2);
    const float d = __builtin_fmaf(e, h, g);
    const float c = (d * d) * d;
    const float r = __builtin_amdgcn_rcpf(c);
    const float e2 = __builtin_fmaf(-c, r, 1.0f);
    return FastChain{d, __builtin_fmaf(e2, r, r)};
}

// fp64 counterpart.  The steps are the ones hipcc's own correctly-rounded fp64 sqrt and divide expansions perform
// between their range scaling (v_ldexp / v_div_scale) and their special-case fix-ups (v_div_fixup, class tests):
// for operands whose exponents keep every intermediate normal - d2 in [2^-500, 2^500], hence d^3 in
// [2^-750, 2^750] - the scaling is by 2^0 and the fix-ups select the computed value, so the results are the same
// bits.  Checked on the device against the compiler's sqrt and 1/x (csrc/tune/chain_probe_f64.hip: 1.3e10 random
// and structured inputs; nbody_selftest_chain_f64 in the test-suite): SAMPLED, not proved.  36 instead of 71 instructions
// per pair.  Known exception of the reciprocal (see ieee_rcp): c with a significand of all ones comes out one ulp low.
// The kernels therefore never let such a pair through: the low word of every c goes into a running v_max3_u32 (half an
// instruction per pair) and a chunk that has seen 0xffffffff there is redone by the general code (2^-32 of the pairs).
struct FastChainD { double d, inv, c; };
__device__ __forceinline__ double fast_rcp_cube(double c, double h1);
__device__ __forceinline__ FastChainD fast_chain(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double g0 = x * y;
    const double h0 = 0.5 * y;
    const double r0 = __builtin_fma(-h0, g0, 0.5);
    const double g1 = __builtin_fma(g0, r0, g0);
    const double h1 = __builtin_fma(h0, r0, h0);
    const double d0 = __builtin_fma(-g1, g1, x);
    const double g2 = __builtin_fma(d0, h1, g1);
    const double d1 = __builtin_fma(-g2, g2, x);
    const double d = __builtin_fma(d1, h1, g2);
    const double c = (d * d) * d;
    return FastChainD{d, fast_rcp_cube(c, h1), c};
}
// 1 / c for c = d^3, seeded from h1 = 1 / (2 d) to ~2^-50 (the square root's refinement leaves it)
__device__ __forceinline__ double fast_rcp_cube(double c, double h1) {
    // 1 / c without a second transcendental (v_rcp_f64 costs about five fp64 multiplies on this chip): h1 is 1 / (2 sqrt x)
    // to ~2^-50, so 8 h1^3 is 1 / c to ~2^-47 - a far better seed than the instruction's ~2^-26 - and ONE Newton step plus
    // the final residual correction of the compiler's own expansion (the step that makes its result correctly rounded:
    // after it the error is the square of a half-ulp residual) replace the instruction and two of its three steps.
    const double q0 = ((h1 * h1) * h1) * 8.0;
    const double e0 = __builtin_fma(-c, q0, 1.0);
    const double q1 = __builtin_fma(q0, e0, q0);
    const double e1 = __builtin_fma(-c, q1, 1.0);
    return __builtin_fma(e1, q1, q1);
}

// Two chains at once on 2-vectors, element-wise (the same operations per element as fast_chain, hence the same
// bits): on gfx950 the fp32 multiplies and fmas become v_pk_mul_f32 / v_pk_fma_f32, one instruction for both
// pairs.  Same-box, bare sequence (csrc/tune/pair_probe.hip): +14 % pairs/s at 4 waves per SIMD, +19 % at 1.
template <typename T> struct Pair;
template <> struct Pair<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Pair<double> { typedef double type __attribute__((ext_vector_type(2))); };
__device__ __forceinline__ Pair<float>::type fast_inv_cube2(Pair<float>::type d2) {
    typedef Pair<float>::type V;
    V y;
    y.x = __builtin_amdgcn_rsqf(d2.x);
    y.y = __builtin_amdgcn_rsqf(d2.y);
    const V g = d2 * y;
    const V h = y * 0.5f;
    const V e = __builtin_elementwise_fma(-g, g, d2);
    const V d = __builtin_elementwise_fma(e, h, g);
    const V c = (d * d) * d;
    V r;
    r.x = __builtin_amdgcn_rcpf(c.x);
    r.y = __builtin_amdgcn_rcpf(c.y);
    const V one = {1.0f, 1.0f};
    const V e2 = __builtin_elementwise_fma(-c, r, one);
    return __builtin_elementwise_fma(e2, r, r);
}
// Two packed chains written side by side: a packed instruction is then followed by an independent one and not by its
// dependent, which would need a wait state (hipcc keeps the order it is given here).
__device__ __forceinline__ void fast_inv_cube2x2(Pair<float>::type d2a, Pair<float>::type d2b, Pair<float>::type& inva,
                                                 Pair<float>::type& invb) {
    typedef Pair<float>::type V;
    V ya, yb;
    ya.x = __builtin_amdgcn_rsqf(d2a.x);
    ya.y = __builtin_amdgcn_rsqf(d2a.y);
    yb.x = __builtin_amdgcn_rsqf(d2b.x);
    yb.y = __builtin_amdgcn_rsqf(d2b.y);
    const V ga = d2a * ya, gb = d2b * yb;
    const V ha = ya * 0.5f, hb = yb * 0.5f;
    const V ea = __builtin_elementwise_fma(-ga, ga, d2a), eb = __builtin_elementwise_fma(-gb, gb, d2b);
    const V da = __builtin_elementwise_fma(ea, ha, ga), db = __builtin_elementwise_fma(eb, hb, gb);
    const V sa = da * da, sb = db * db;
    const V ca = sa * da, cb = sb * db;
    V ra, rb;
    ra.x = __builtin_amdgcn_rcpf(ca.x);
    ra.y = __builtin_amdgcn_rcpf(ca.y);
    rb.x = __builtin_amdgcn_rcpf(cb.x);
    rb.y = __builtin_amdgcn_rcpf(cb.y);
    const V one = {1.0f, 1.0f};
    const V fa = __builtin_elementwise_fma(-ca, ra, one), fb = __builtin_elementwise_fma(-cb, rb, one);
    inva = __builtin_elementwise_fma(fa, ra, ra);
    invb = __builtin_elementwise_fma(fb, rb, rb);
}
// `ones`: running maximum of the low words of every c = d^3 the chain has inverted (see FastChainD)
__device__ __forceinline__ Pair<double>::type fast_inv_cube2(Pair<double>::type d2, unsigned& ones) {
    Pair<double>::type inv;                                // no packed fp64 instructions: two scalar chains
    const FastChainD a = fast_chain(d2.x), b = fast_chain(d2.y);
    inv.x = a.inv;
    inv.y = b.inv;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(ones) : "v"(ones), "v"((unsigned)__double_as_longlong(a.c)),
        "v"((unsigned)__double_as_longlong(b.c)));
    return inv;
}
__device__ __forceinline__ Pair<float>::type fast_inv_cube2(Pair<float>::type d2, unsigned&) { return fast_inv_cube2(d2); }
// wave mask of the lanes whose chunk must be redone because of it (fp32: none - its chain is proved on every input)
__device__ __forceinline__ unsigned long long ones_mask(float, unsigned) { return 0ull; }
__device__ __forceinline__ unsigned long long ones_mask(double, unsigned ones) { return __ballot(ones == 0xffffffffu); }

// a + b as an instruction the SLP vectoriser cannot see: left alone it merges the two pairs' (dx^2 + dy^2) adds into
// one v_pk_add_f32 and pays for it with three v_mov_b32 that gather the operands
__device__ __forceinline__ float add_unmerged(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double add_unmerged(double a, double b) { return a + b; }

// Orders this wave's LDS accesses (and keeps the compiler from moving memory accesses across) WITHOUT waiting for
// outstanding global loads: a workgroup-scope fence or __syncthreads() emits s_waitcnt vmcnt(0), which puts the
// latency of a prefetch from the replica (a microsecond or two) on the critical path of whoever waits.
__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the guarded domain of the fast chains per precision: pairs with d2 <= fma(rs, rs, lo) or a coordinate at or beyond
// `coord` (so d2 could exceed the upper edge) are left to the general code
template <typename T> struct FastDomain;
// `floor`: with both bodies' coordinates at or above it in magnitude a coordinate difference is 0 or at least one spacing of
// the format there, so d2 is exactly 0 or above `lo` (fp32: 2^-39 squared = 2^-78; fp64: 2^-242 squared = 2^-484): with
// all radii +0 the only pairs outside the fast domain are coincident bodies (see kCoordFloor)
template <> struct FastDomain<float> { static constexpr float lo = kFastLo, coord = kCoordBound, floor = kCoordFloor; };
template <> struct FastDomain<double> { static constexpr double lo = 0x1p-500, coord = 0x1p249, floor = 0x1p-190; };

__device__ __forceinline__ float abs_(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double abs_(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ unsigned long long le_mask(float a, float b) {     // wave mask of a <= b (ordered)
    return __builtin_amdgcn_fcmpf(a, b, 5 /* llvm::CmpInst::FCMP_OLE */);
}
__device__ __forceinline__ unsigned long long le_mask(double a, double b) {
    return __builtin_amdgcn_fcmp(a, b, 5 /* llvm::CmpInst::FCMP_OLE */);
}
__device__ __forceinline__ unsigned long long unordered_mask(float a, float b) {   // wave mask of "a or b is NaN"
    return __builtin_amdgcn_fcmpf(a, b, 8 /* llvm::CmpInst::FCMP_UNO */);
}
__device__ __forceinline__ unsigned long long unordered_mask(double a, double b) {
    return __builtin_amdgcn_fcmp(a, b, 8 /* llvm::CmpInst::FCMP_UNO */);
}
__device__ __forceinline__ bool not_plus_zero(float x) { return __float_as_uint(x) != 0u; }
__device__ __forceinline__ bool not_plus_zero(double x) { return __double_as_longlong(x) != 0ll; }

// ---------------------------------------------------------------------------------------------------------
// Force + collision + drift kernel, variant "v3": the production kernel.  Its text lives in nbody_forces_v3.inc
// and is instantiated below for: fp32 with 128-thread workgroups and K lanes per body (forces_v3_f32), fp32 and
// fp64 with 256-thread workgroups (forces_v3w_f32 / forces_v3w_f64, the defaults), and the reference's device
// block layout (ref_layout_forces_v3_f32).
//
//   * 128-lane group = 128/K bodies of one reference block, K consecutive lanes per body.  Lane h of a
//     group evaluates the tile entries off = h, h+K, h+2K, ... of the body's walk; the running force sum
//     lives in lane h = 0 and takes the K terms of a round strictly in walk order (own term, then the
//     neighbours' through DPP row_shl), so the fp32 accumulation order - and therefore every bit of the
//     result - is independent of K.  K > 1 buys parallelism when a rank owns fewer bodies than the chip has
//     lanes (strong scaling); K = 1 is one body per lane.
//   * The tile is stored TWICE back to back in LDS, so entry (t+off) mod 128 is at index t+off without a wrap:
//     the address of every pair's ds_read_b128 is one per-lane base plus an immediate offset.
//   * Fast path (fast_chain) in chunks of 32 walk positions with a wave-wide flag per chunk; a flagged chunk
//     is restored from a 2-register snapshot and redone by the general code in the chain lanes.
// ---------------------------------------------------------------------------------------------------------
constexpr int kChunk = 32;

template <int U>
__device__ __forceinline__ float dpp_row_shl(float v) {
    // bound_ctrl:1 (out-of-row lanes read 0) lets the DPP combiner fold the move into the consuming v_add_f32
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x100 + U, 0xf, 0xf, true));
}
template <int U>
__device__ __forceinline__ double dpp_row_shl(double v) {   // K > 1 is fp32 only; present so the text compiles
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, 0x100 + U, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x100 + U, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

#define NB_V3_REAL float
#define NB_V3_SIGNATURE                                                                                      \
    template <int K, bool kLog>                                                                              \
    __global__ __launch_bounds__(kTile, 4) void forces_v3_f32(                                               \
        const Rec<float>* __restrict__ J, const Vec2<float>* __restrict__ Vown, Rec<float>* __restrict__ S_J, \
        Vec2<float>* __restrict__ S_V, const Meta* __restrict__ meta, StepParams<float> p, Event* ev,        \
        int ev_cap, Counters* ctr)
#define NB_V3_LDS                                                                                            \
    __shared__ Rec<T> tile[2][2 * kTile];                  /* each tile stored twice: no wrap in the walk */ \
    __shared__ int tile_bad[2][kTile / kWave];                                                               \
    __shared__ int tile_rnz[2][kTile / kWave];             /* some radius in the staged tile is not +0 */
#define NB_V3_LANE const int lane = threadIdx.x;
#define NB_V3_WG const int wg = blockIdx.x;
#define NB_V3_BATCH 8
#define NB_V3_CONSTANTS
#define NB_V3_RANGE const int N = meta->n, lo = meta->lo, cnt = meta->cnt, step = meta->step;
#define NB_V3_REC(j) J[j]
#define NB_V3_VEL(i) Vown[i - lo]
#define NB_V3_PUT(q, i, out, vout) S_J[q] = out; S_V[q] = vout;
#define NB_V3_KEEP(q, i, a, v) S_J[q] = Rec<T>{a.xi, a.yi, a.mi, a.ri}; S_V[q] = v;
#define NB_V3_COUNT(pairs)                                                                                   \
    for (int sh = kWave / 2; sh > 0; sh >>= 1) pairs += __shfl_down(pairs, sh, kWave);                      \
    if ((lane & (kWave - 1)) == 0 && pairs) atomicAdd(&ctr->pairs, pairs);
#include "nbody_forces_v3.inc"
#undef NB_V3_REAL
#undef NB_V3_SIGNATURE
#undef NB_V3_LDS
#undef NB_V3_LANE
#undef NB_V3_WG
#undef NB_V3_BATCH
#undef NB_V3_CONSTANTS
#undef NB_V3_RANGE
#undef NB_V3_REC
#undef NB_V3_VEL
#undef NB_V3_PUT
#undef NB_V3_KEEP
#undef NB_V3_COUNT

// The same kernel with 256-thread workgroups: two independent 128-lane groups (each one reference block or a
// K-th of it, each with its own tiles) share a workgroup and its barriers, so that a workgroup has four waves and
// the dispatcher puts one on every SIMD (two 2-wave workgroups per CU land 2-1-1-0,
// profiles/r01_wave_placement.txt).  kOcc = waves per SIMD the register budget is sized for: 4 (8 tile reads per
// batch, 96 VGPRs) when the own range fills the chip, 2 (16 reads per batch) below.  This is the one-lane-per-body
// production kernel: 33.5 ms per step at N=262144 on one GPU.
#define NB_V3_REAL float
#define NB_V3_SIGNATURE                                                                                      \
    template <int K, bool kLog, int kOcc>                                                                    \
    __global__ __launch_bounds__(2 * kTile, kOcc) void forces_v3w_f32(                                       \
        const Rec<float>* __restrict__ J, const Vec2<float>* __restrict__ Vown, Rec<float>* __restrict__ S_J, \
        Vec2<float>* __restrict__ S_V, const Meta* __restrict__ meta, StepParams<float> p, Event* ev,        \
        int ev_cap, Counters* ctr)
#define NB_V3_LDS                                                                                            \
    __shared__ Rec<T> tile_all[2][2][2 * kTile];                                                             \
    __shared__ int tile_bad_all[2][2][kTile / kWave];                                                        \
    __shared__ int tile_rnz_all[2][2][kTile / kWave];                                                        \
    Rec<T>(&tile)[2][2 * kTile] = tile_all[threadIdx.x / kTile];                                             \
    int(&tile_bad)[2][kTile / kWave] = tile_bad_all[threadIdx.x / kTile];                                    \
    int(&tile_rnz)[2][kTile / kWave] = tile_rnz_all[threadIdx.x / kTile];
#define NB_V3_LANE const int lane = threadIdx.x % kTile;
#define NB_V3_WG const int wg = blockIdx.x * 2 + threadIdx.x / kTile;
#define NB_V3_BATCH (kOcc >= 4 ? 8 : 16)
#define NB_V3_CONSTANTS
#define NB_V3_RANGE const int N = meta->n, lo = meta->lo, cnt = meta->cnt, step = meta->step;
#define NB_V3_REC(j) J[j]
#define NB_V3_VEL(i) Vown[i - lo]
#define NB_V3_PUT(q, i, out, vout) S_J[q] = out; S_V[q] = vout;
#define NB_V3_KEEP(q, i, a, v) S_J[q] = Rec<T>{a.xi, a.yi, a.mi, a.ri}; S_V[q] = v;
#define NB_V3_COUNT(pairs)                                                                                   \
    for (int sh = kWave / 2; sh > 0; sh >>= 1) pairs += __shfl_down(pairs, sh, kWave);                      \
    if ((lane & (kWave - 1)) == 0 && pairs) atomicAdd(&ctr->pairs, pairs);
#include "nbody_forces_v3.inc"
#undef NB_V3_REAL
#undef NB_V3_SIGNATURE
#undef NB_V3_LDS
#undef NB_V3_LANE
#undef NB_V3_WG
#undef NB_V3_BATCH
#undef NB_V3_CONSTANTS
#undef NB_V3_RANGE
#undef NB_V3_REC
#undef NB_V3_VEL
#undef NB_V3_PUT
#undef NB_V3_KEEP
#undef NB_V3_COUNT

// The same kernel in fp64 (256-thread form, one lane per body): the fp64 production kernel.  A record is 32 bytes,
// 8 reads per batch (same-box A/B: 4 reads with the 4-waves register budget -0.4 %, 16 reads -4 %).
#define NB_V3_ROTATE_PRIORITY
#define NB_V3_TILE_REC RecPadded
#define NB_V3_REAL double
#define NB_V3_SIGNATURE                                                                                      \
    template <bool kLog>                                                                                     \
    __global__ __launch_bounds__(2 * kTile, 2) void forces_v3w_f64(                                          \
        const Rec<double>* __restrict__ J, const Vec2<double>* __restrict__ Vown,                            \
        Rec<double>* __restrict__ S_J, Vec2<double>* __restrict__ S_V, const Meta* __restrict__ meta,        \
        StepParams<double> p, Event* ev, int ev_cap, Counters* ctr)
#define NB_V3_LDS                                                                                            \
    __shared__ RecPadded tile_all[2][2][2 * kTile];                                                          \
    __shared__ int tile_bad_all[2][2][kTile / kWave];                                                        \
    __shared__ int tile_rnz_all[2][2][kTile / kWave];                                                        \
    RecPadded(&tile)[2][2 * kTile] = tile_all[threadIdx.x / kTile];                                          \
    int(&tile_bad)[2][kTile / kWave] = tile_bad_all[threadIdx.x / kTile];                                    \
    int(&tile_rnz)[2][kTile / kWave] = tile_rnz_all[threadIdx.x / kTile];
#define NB_V3_LANE const int lane = threadIdx.x % kTile;
#define NB_V3_WG const int wg = blockIdx.x * 2 + threadIdx.x / kTile;
#define NB_V3_BATCH 8
#define NB_V3_CONSTANTS constexpr int K = 1;
#define NB_V3_RANGE const int N = meta->n, lo = meta->lo, cnt = meta->cnt, step = meta->step;
#define NB_V3_REC(j) J[j]
#define NB_V3_VEL(i) Vown[i - lo]
#define NB_V3_PUT(q, i, out, vout) S_J[q] = out; S_V[q] = vout;
#define NB_V3_KEEP(q, i, a, v) S_J[q] = Rec<T>{a.xi, a.yi, a.mi, a.ri}; S_V[q] = v;
#define NB_V3_COUNT(pairs)                                                                                   \
    for (int sh = kWave / 2; sh > 0; sh >>= 1) pairs += __shfl_down(pairs, sh, kWave);                      \
    if ((lane & (kWave - 1)) == 0 && pairs) atomicAdd(&ctr->pairs, pairs);
#include "nbody_forces_v3.inc"
#undef NB_V3_ROTATE_PRIORITY
#undef NB_V3_TILE_REC
#undef NB_V3_REAL
#undef NB_V3_SIGNATURE
#undef NB_V3_LDS
#undef NB_V3_LANE
#undef NB_V3_WG
#undef NB_V3_BATCH
#undef NB_V3_CONSTANTS
#undef NB_V3_RANGE
#undef NB_V3_REC
#undef NB_V3_VEL
#undef NB_V3_PUT
#undef NB_V3_KEEP
#undef NB_V3_COUNT

// The same kernel (256-thread form) on the reference's device block [P|V|M|R] and its two scratch arrays: drop-in
// for the ComputeForces<<<>>> site when it is launched with the reference's own block count
// (src/nbody.cu:473,481-482).  Velocities are updated in place (:264), updatedMasses / updatedRadii written
// (:245-246), positions are left to MoveBodies, bodies without a thread in the reference are not touched.
#define NB_V3_REAL float
#define NB_V3_SIGNATURE                                                                                      \
    __global__ __launch_bounds__(2 * kTile, 4) void ref_layout_forces_v3_f32(                                \
        void* bodyData, float* __restrict__ updM, float* __restrict__ updR, const int N, StepParams<float> p)
#define NB_V3_LDS                                                                                            \
    __shared__ Rec<T> tile_all[2][2][2 * kTile];                                                             \
    __shared__ int tile_bad_all[2][2][kTile / kWave];                                                        \
    __shared__ int tile_rnz_all[2][2][kTile / kWave];                                                        \
    Rec<T>(&tile)[2][2 * kTile] = tile_all[threadIdx.x / kTile];                                             \
    int(&tile_bad)[2][kTile / kWave] = tile_bad_all[threadIdx.x / kTile];                                    \
    int(&tile_rnz)[2][kTile / kWave] = tile_rnz_all[threadIdx.x / kTile];
#define NB_V3_LANE const int lane = threadIdx.x % kTile;
#define NB_V3_WG const int wg = blockIdx.x * 2 + threadIdx.x / kTile;
#define NB_V3_BATCH 8
#define NB_V3_CONSTANTS                                                                                      \
    constexpr int K = 1;                                                                                     \
    constexpr bool kLog = false;
#define NB_V3_RANGE                                                                                          \
    constexpr int lo = 0, step = 0, ev_cap = 0;                                                              \
    const int cnt = N;                                                                                       \
    Event* const ev = nullptr;                                                                               \
    Counters* const ctr = nullptr;                                                                           \
    const Vec2<float>* __restrict__ P = reinterpret_cast<const Vec2<float>*>(bodyData); /* :147-150 */       \
    Vec2<float>* __restrict__ V = reinterpret_cast<Vec2<float>*>(bodyData) + N;                              \
    const float* __restrict__ M = reinterpret_cast<const float*>(V + N);                                     \
    const float* __restrict__ R = M + N;
#define NB_V3_REC(j) Rec<float>{P[j].x, P[j].y, M[j], R[j]}
#define NB_V3_VEL(i) V[i]
#define NB_V3_PUT(q, i, out, vout) (void)q; updM[i] = out.m; updR[i] = out.r; V[i] = vout;
#define NB_V3_KEEP(q, i, a, v) (void)q;
#define NB_V3_COUNT(pairs) (void)pairs;
#include "nbody_forces_v3.inc"
#undef NB_V3_REAL
#undef NB_V3_SIGNATURE
#undef NB_V3_LDS
#undef NB_V3_LANE
#undef NB_V3_WG
#undef NB_V3_BATCH
#undef NB_V3_CONSTANTS
#undef NB_V3_RANGE
#undef NB_V3_REC
#undef NB_V3_VEL
#undef NB_V3_PUT
#undef NB_V3_KEEP
#undef NB_V3_COUNT

// ---------------------------------------------------------------------------------------------------------
// Force + collision + drift kernel, variant "ring" (fp32): the fp32 force kernel at every size.
//
// The per-body sum is a strictly ordered chain of N adds; only the terms are independent.  A RING of kW waves serves 64
// bodies; every wave of the ring holds the same 64 bodies (one per lane).  The walk is cut into turns of kT positions and
// the turns go round the waves: wave w takes turns w, w + kW, w + 2 kW, ...  For its turn a wave (1) has the tile
// entries its lanes need loaded straight from the replica into a private LDS window (direct-to-LDS loads issued one own
// turn ahead; the window is component-major, so that two consecutive walk positions of a lane are one ds_read2_b32
// into the register pair of a packed instruction), (2) evaluates the kT terms of every lane into registers - all but
// the two adds per pair, dependent on nothing, two walk positions per packed instruction throughout -, (3) polls
// until the wave before it has published the running sum, (4) adds its kT terms in walk order
// (or, for a flagged lane / a special tile, runs the general code on the kT positions, from the same window), and (5)
// publishes the state for the next wave.  No wave is a dedicated chain wave, no term goes through LDS, there is no
// workgroup barrier in the loop.  A workgroup is kRings rings (1, 2 or 4: 64, 128 or 256 bodies) that fill a CU together.
//
// Hand-off = ONE 16-byte LDS record per lane {fx, fy, seq, flags}: written with one ds_write_b128, polled with one
// ds_read_b128.  The LDS services a lane's whole 16 bytes in one array cycle for both instructions (MI355X_MICROARCH.md,
// LDS table: lane groups), so a reader sees a record entirely old or entirely new (nbody_selftest_lds_record checks
// exactly this); seq == tau means "the state after turn tau - 1"; flags = deleted.  Absorbed mass / radius change rarely:
// they live, with the start-of-step mass, in a second record {mnew, rnew, mi} that is rewritten only when they change and
// read only by the general code and the epilogue.  Between turns a lane keeps NOTHING of its running state in registers
// (position and radius only): the last turn publishes like every other and the epilogue reads the records back.
// Lessons built in (profiles/r02_ring_*): (a) a generic pointer to the sequence number compiles to flat_load / flat_store
// + s_waitcnt vmcnt(0), and hipcc waits vmcnt(0) for the builtin form of the direct-to-LDS load before ANY later LDS
// read: either way the window prefetch was never in flight.  Typed LDS accesses and inline assembly now; no
// compiler-tracked vector-memory operation lives in the turn loop, hence no compiler-placed vmcnt wait in it.
// (b) Rings that share a CU must be kept level, or the arbiter's older-wave-first rule lets one finish early and the CU
// runs half empty: see the priority rule at the top of the loop.  (c) Every turn reads from the window: tile 0 is a
// fast tile whose self position is masked, the truncated last tile is loaded whole across both window buffers.
// (d) Meta::summary says whether ANY coordinate of the replica is unbounded / any radius non-zero: windows are scanned
// only then.  (e) The collision / tiny-distance screen is the smallest d2 of a lane's kT pairs against ONE threshold from the
// largest |radius| of the tiles the window touches (tile_rmax, kept by unpack_slots); with all radii +0, all masses finite
// and all coordinates in [2^-16, 2^38) there is no screen per pair at all: a coincident pair - the only collision left - makes
// its term NaN and the wave looks at the sum after the turn's adds (kCoordFloor).  Flagged lanes get the exact status of
// their pairs in parallel across the wave (see the turn loop).  (f) The LDS reads of a batch of four positions are issued one
// batch ahead; the kernel's arguments travel as one struct and the cold ones are loaded where they are used (RingArgs).
// A wait that exceeds p.spin_limit polls is reported (Counters::errors, sticky on the host) and POISONS the chain:
// the state becomes NaN and a dead mark travels with the sequence number, so every later turn passes at once and
// the step's output cannot be mistaken for a result.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* LdsPtr;
typedef int Int4 __attribute__((ext_vector_type(4)));
typedef float Float2 __attribute__((ext_vector_type(2)));
typedef float Float4 __attribute__((ext_vector_type(4)));
typedef volatile __attribute__((address_space(3))) Int4* LdsInt4Ptr;
typedef volatile __attribute__((address_space(3))) Float4* LdsFloat4Ptr;

constexpr int kRingDeadSeq = 0x40000000;                   // sequence number of a poisoned chain
// Index checks: the role of CUDA_SYNC_CHECK (src/nbody.cu:20-33) for the addresses the ring kernel forms itself.  The
// staging store of the epilogue is checked in every build; the event-logging builds (kLog - the builds of round 2's
// unexplained fault) also check the source range of every window gather and the radius-bound lookups.  A
// failed check SKIPS the access and adds kIndexError to Counters::errors: the host then fails like after a time-out.
constexpr unsigned long long kIndexError = 1ull << 32;

// Global memory straight into LDS (gfx950 LDS-DMA).  Inline assembly on purpose: hipcc tracks the builtin form as a writer
// of all LDS and waits vmcnt(0) before the next LDS read.  M0 is written by nothing else in these kernels (gfx9 DS
// instructions do not use it).
// 4 bytes per active lane: lane l's dword lands at lds_base + 4 * l.  The SOURCE address is per lane, so a strided gather
// from the {x, y, m, r} records turns one component of 64 bodies into 64 consecutive LDS words.
// (The instruction's immediate offset applies to the global AND the LDS address; offsets of 4 ... 12 bytes cost nothing,
// 512 ... 1536 made these loads slow - +28 % kernel time -, so none is used.)
// M0 is declared clobbered although hipcc treats it as reserved (it never keeps a value of its own there: the ISA of
// every build of these kernels is identical with and without the clobber) and says so with -Winline-asm.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void load_to_lds_b32(const void* base, unsigned byte_offset, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                 ::"s"(lds_dst), "v"(byte_offset), "s"(base) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(unsigned long long)(LdsPtr)p; }

// The kernel's arguments travel as ONE struct.  What the turn loop needs (meta, Jt, tile_rmax, two flags of p) is read
// from it the usual way; everything else - the staging arrays and step parameters of the epilogue, the event log of the
// rare path - is loaded from the kernarg segment WHERE IT IS USED (ring_late_arg): hipcc loads every kernel argument
// it can see at the kernel's entry and keeps it in scalar registers to the end, and this kernel has none to spare
// (spilled scalars cost v_readlanes per turn, a spilled vector register a scratch access inside the loop).
struct RingArgs {
    const Vec2<float>* Vown;
    Rec<float>* S_J;
    Vec2<float>* S_V;
    const Meta* meta;
    StepParams<float> p;
    Event* ev;
    int ev_cap;
    Counters* ctr;
    const float* tile_rmax;
    const float* Jt;
    int cap_own;                 // entries of Vown / S_J / S_V
    int n_tiles;                 // entries of tile_rmax, 2 KiB tiles of Jt
};
template <typename A>
__device__ __forceinline__ A ring_late_arg(unsigned byte_offset) {
#if defined(__HIP_DEVICE_COMPILE__)                        // (the host pass of hipcc only needs the declaration)
    typedef const __attribute__((address_space(4))) char* KernargBytes;
    const KernargBytes ka = (KernargBytes)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(byte_offset));                  // an offset hipcc cannot see through: the load stays here
    return *(const __attribute__((address_space(4))) A*)(ka + byte_offset);   // a scalar load (s_load_*)
#else
    (void)byte_offset;
    return A{};
#endif
}
#define NB_RING_LATE(field) \
    ring_late_arg<decltype(((const RingArgs*)nullptr)->field)>((unsigned)__builtin_offsetof(RingArgs, field))

template <bool kLog, int kW, int kT, int kSleep, bool kProbe, int kRings>
__global__ __launch_bounds__(kRings * kW * kWave) __attribute__((amdgpu_waves_per_eu(4, 4)))
void forces_ring_f32(const RingArgs args) {
    typedef float T;
    const Meta* __restrict__ const meta = args.meta;
    const float* __restrict__ const tile_rmax = args.tile_rmax;
    const float* __restrict__ const Jt = args.Jt;
    typedef Pair<float>::type V2;
    static_assert(kTile % kT == 0 && kT % 8 == 0 && kT <= kWave, "turn length");
    constexpr int kTurnsPerTile = kTile / kT;
    static_assert(kW % kTurnsPerTile == 0, "a wave's successive turns are whole tiles apart");
    constexpr int kTilesPerRound = kW / kTurnsPerTile;
    constexpr int kWin = kWave + kT;                       // window entries a turn can touch (kWin - 1 used)
    static_assert(kRings == 1 || kRings == 2 || kRings == 4, "rings (64 bodies each) per workgroup");
    // Per wave, double buffered, COMPONENT-MAJOR: x[kWin] y[kWin] m[kWin] r[kWin].  A lane's walk positions r, r + 1 are
    // then two consecutive words of each component: one ds_read2_b32 fills the register pair a packed fp32 instruction
    // takes, so two walk positions share EVERY instruction of the term (with records in LDS the differences and squares
    // were packed per position: one more instruction per pair, plus moves that paired the mass with its operand).
    __shared__ float win_all[kRings][kW][2][4][kWin];
    __shared__ Int4 hand_all[kRings][kWave];               // {fx, fy, seq, flags = deleted} per lane
    __shared__ Float4 hand_m_all[kRings][kWave];           // {mnew, rnew, mi, -}: rewritten only when mnew / rnew change
    const int N = meta->n, lo = meta->lo, cnt = meta->cnt;
    const bool all_bounded = (meta->summary & kSummaryUnbounded) == 0, any_radius = (meta->summary & kSummaryRadius) != 0;
    // every coordinate of the replica in [2^-16, 2^38), every radius +0, every mass finite: coincident bodies are the only
    // pairs a fast turn must not add up, and they show as a NaN sum (kCoordFloor, kMassBound)
    const bool nan_screen = meta->summary == 0;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int ring = wv / kW;                              // which 64 bodies of the block
    const int w = wv % kW;                                 // place in the ring
    const int l = tid % kWave;
    const int wg = blockIdx.x;
    const int ring_g = wg * kRings + ring;                 // ring of the launch: two per reference block
    const int half = ring_g % 2;
    const int b = lo / kTile + ring_g / 2;                 // reference block
    const int t0 = half * kWave;
    const int t = t0 + l;                                  // threadIdx.x of this lane's body in the reference
    const long long blk0 = (long long)b * kTile;
    float(&win)[kW][2][4][kWin] = win_all[ring];
    Int4(&hand)[kWave] = hand_all[ring];
    Float4(&hand_m)[kWave] = hand_m_all[ring];
    const int nb = N < kTile ? 1 : N / kTile;              // src/nbody.cu:473
    const bool lit = args.p.literal != 0;
    const int ntiles = lit ? nb : (N + kTile - 1) / kTile;
    const int nturns = ntiles * kTurnsPerTile;

    const bool mine = blk0 + t >= lo && blk0 + t < (long long)lo + cnt;
    const bool active = mine && blk0 + t < N && (!lit || blk0 + t < (long long)nb * kTile);
    // The lane's body index, formed where it is needed (rare path, epilogue) from a lane number hipcc cannot see through:
    // kept across the turn loop it is two vector registers the evaluation cannot spare.
    auto body_index64 = [&]() -> long long {
        int lane = l;
        asm volatile("" : "+v"(lane));
        return blk0 + t0 + lane;
    };
    // What a lane keeps in REGISTERS across the turns is its position and radius only: every fast turn needs them.  The
    // rest of its state lives in LDS between turns - the running sum and `deleted` in the hand-off record, {mnew, rnew, mi}
    // in the rare record - and is picked up where it is needed: by the general code and by the epilogue.
    T xi = 0, yi = 0, ri = 0;
    {
        T mi = 0;
        if (mine) {                                        // own start-of-step state, from the tile-planar copy like the windows
            const long long i64 = blk0 + t;
            const int i = (int)(i64 < 0x7fffffff ? i64 : 0x7fffffff);
            const float* me = Jt + (size_t)(i / kTile) * (4 * kTile) + (i % kTile);
            xi = me[0]; yi = me[kTile]; mi = me[2 * kTile]; ri = me[3 * kTile];
        }
        // the loads are waited for HERE: left to hipcc, the wait for `ri` lands at its first use - inside the turn loop,
        // behind the next window's prefetch, which it would then wait for in every turn
        asm volatile("" : "+v"(xi), "+v"(yi), "+v"(mi), "+v"(ri));
        if (w == 0) {
            *(LdsInt4Ptr)&hand[l] = Int4{0, 0, 0, 0};
            *(LdsFloat4Ptr)&hand_m[l] = Float4{mi, ri, mi, 0.0f};            // mnew = mi, rnew = ri (:174-175)
        }
    }
    const int spin_limit = __builtin_amdgcn_readfirstlane(args.p.spin_limit);   // (kept out of the poll loop's reach)
    bool dead = false;
    int timeouts = 0;
    const bool lane_ok = !active || ((abs_(xi) < kCoordBound) && (abs_(yi) < kCoordBound));
    const bool wave_ok = __ballot(!lane_ok) == 0ull;
    int pairs_rare = 0;                                    // pairs of this lane's turns that left the common path
    const LdsInt4Ptr hand_l = (LdsInt4Ptr)&hand[l];
    const LdsFloat4Ptr hand_m_l = (LdsFloat4Ptr)&hand_m[l];
    __syncthreads();                                       // the only workgroup barrier: seq = 0 everywhere
    if (blk0 + t0 >= (long long)lo + cnt) return;          // a ring without own bodies (after the barrier)
    // Two rings share a CU's SIMDs (two per workgroup here, or two workgroups per CU), and the instruction arbiter
    // serves the OLDER wave first at equal priority: left alone, the older ring runs at full speed, the younger one
    // on what is left, and the CU then spends a third of the kernel with one ring only (measured: workgroups ended at
    // 3.2 and 4.8 ms of a 4.8 ms launch, profiles/r02_ring_fairness.txt).  With both rings in one workgroup each
    // wave compares the two chains' progress at the start of a turn and evaluates at priority 1 when its own ring is
    // behind, 0 otherwise: the rings stay level and finish together.
    typedef const volatile __attribute__((address_space(3))) int* LdsSeqPtr;
    const LdsSeqPtr seq_mine = (LdsSeqPtr)&hand_all[ring][0] + 2;
    const LdsSeqPtr seq_o1 = (LdsSeqPtr)&hand_all[(ring + 1) % kRings][0] + 2;    // the other rings of the workgroup
    const LdsSeqPtr seq_o2 = (LdsSeqPtr)&hand_all[(ring + 2) % kRings][0] + 2;
    const LdsSeqPtr seq_o3 = (LdsSeqPtr)&hand_all[(ring + 3) % kRings][0] + 2;
    unsigned long long pr_eval = 0, pr_wait = 0, pr_chain = 0, pr_check = 0, pr_polls = 0, pr_t0 = 0, pr_r0 = 0;
    if (kProbe) { pr_t0 = __builtin_readcyclecounter(); pr_r0 = wall_clock64(); }

    // First body of the tile of this wave's current turn (literal: cyclic tile b + kk), kept incrementally: a wave
    // moves on by kW turns = kTilesPerRound tiles at a time, and a division here would cost as much as the turn's
    // arithmetic.
    auto tile_start_slow = [&](int kk) -> long long {
        if (!lit) return (long long)kk * kTile;
        return (blk0 % N + (long long)kk * kTile) % N;
    };
    auto round_on = [&](long long st) -> long long {
        st += kTilesPerRound * kTile;
        if (lit) while (st >= N) st -= N;                  // a next turn exists only when N > kTilesPerRound tiles: once
        return st;
    };
    auto tile_len = [&](int kk, long long st) -> int {
        if (lit) return (kk == nb - 1) ? N % (kTile + 1) : kTile;             // :194 (quirk Q1)
        return (N - st) < kTile ? (int)(N - st) : kTile;
    };
    // Every turn reads its tile entries from the wave's LDS window, never from the replica.  Kind of a turn's window:
    // 0 none (past the walk), 1 the standard window of a full tile, 2 a truncated tile (literal: the last one, of
    // N mod 129 entries; clean: the partial last one), held whole.
    auto turn_kind = [&](int tau, long long st) -> int {
        if (tau >= nturns) return 0;
        return tile_len(tau / kTurnsPerTile, st) == kTile ? 1 : 2;
    };
    // may a full tile take the fast path?  literal: yes (the self position, tile 0 / walk position 0, is masked in the
    // first turn); clean: not the tile that holds the workgroup's own bodies (the self position differs per lane)
    auto fast_tile = [&](int kk) -> bool { return lit || kk != b; };
    // The standard window: entry j of the window is tile entry (wbase0 + off0 + j) mod 128, and off0 is the same
    // for every turn of a wave (kW is a multiple of the turns per tile): the tile entries this lane fetches are two
    // per-lane constants, the body index is that plus the tile's first body, wrapped at most once (st < N, e < 128 <= N).
    const int wbase0 = lit ? t0 : 0;                       // literal: lane l reads window[l + r]; clean: window[r]
    const int nwin = lit ? (kWave + kT - 1) : kT;          // entries of the window that are used
    // (recomputed per window from the lane number - two instructions - instead of living in two registers)
    auto first_entry = [&]() -> unsigned {
        int lane = l;
        asm volatile("" : "+v"(lane));
        return (unsigned)(wbase0 + (w % kTurnsPerTile) * kT + lane) & (kTile - 1);
    };
    auto window_offset = [&](long long st, unsigned e) -> unsigned {   // byte offset of the body's x in the tiled copy Jt
        const unsigned src = (unsigned)st + e;
        const unsigned wrapped = src - (unsigned)N;        // huge when src < N
        const unsigned idx = src < wrapped ? src : wrapped;
        return ((idx / kTile) * (4u * kTile) + (idx % kTile)) * (unsigned)sizeof(T);
    };
    // Loaded from the replica STRAIGHT INTO LDS, one component of 64 bodies per instruction (lane l's word lands at
    // base + 4 l; the per-lane source address does the transposition): the prefetch holds no registers and stays in
    // flight for a whole turn.  The radii are only fetched when some radius of the replica is not +0 (Meta::summary).
    auto issue_entries = [&](unsigned tile_byte, unsigned byte_offset, unsigned base, unsigned comp_bytes) {
        // Source: Jt, the replica once more, tile by tile component-major (x[128] y[128] m[128] r[128] per aligned
        // 128-body tile, written next to J by unpack_slots): a wave's 64 entries of one component are 256 contiguous
        // bytes - two or three cache lines per instruction where the 16-byte records took eight or nine -, and the
        // radius lines are never touched while every radius is +0: 3 of the 4 MiB, which an XCD's 4 MiB L2 keeps from one
        // round of workgroups to the next (fabric-side reads halved, profiles/r02_traffic_pmc.json).
        // The planes of a tile are 512 bytes apart.  Their base pointers are formed HERE, from an offset hipcc cannot see
        // through: hoisted out of the turn loop they are three more scalar register pairs the kernel does not have
        // (spilled pairs cost v_readlanes per turn).
        unsigned long long plane = kTile * sizeof(T);
        asm volatile("" : "+s"(plane));
        if (kLog) {                                        // the four words this lane fetches lie inside Jt
            const unsigned long long last = (unsigned long long)tile_byte + byte_offset + 3 * kTile * sizeof(T) + sizeof(T);
            if (last > (unsigned long long)NB_RING_LATE(n_tiles) * (4 * kTile * sizeof(T))) {
                atomicAdd(&NB_RING_LATE(ctr)->errors, kIndexError);
                return;
            }
        }
        const char* const src = (const char*)Jt + tile_byte;
        load_to_lds_b32(src, byte_offset, base);
        load_to_lds_b32(src + plane, byte_offset, base + comp_bytes);
        load_to_lds_b32(src + 2 * plane, byte_offset, base + 2 * comp_bytes);
        if (any_radius) load_to_lds_b32(src + 3 * plane, byte_offset, base + 3 * comp_bytes);
    };
    auto issue_window = [&](long long st, int buf) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds_offset_of(&win[w][buf][0][0]));
        const unsigned e0 = first_entry();
        const unsigned e1 = e0 ^ kWave;                    // the entry 64 further on
        if (((unsigned)st & (kTile - 1)) == 0u && st + kTile <= N) {
            // the common case - the window's tile is an aligned tile of Jt -: the tile goes into the scalar base address,
            // the lanes' offsets are the two per-lane constants, no vector arithmetic at all
            const unsigned tile_byte = ((unsigned)st / kTile) * (4u * kTile * (unsigned)sizeof(T));
            if (l < nwin) issue_entries(tile_byte, e0 * (unsigned)sizeof(T), base, kWin * (unsigned)sizeof(T));
            if (l + kWave < nwin) issue_entries(tile_byte, e1 * (unsigned)sizeof(T), base + kWave * (unsigned)sizeof(T), kWin * (unsigned)sizeof(T));
        } else {
            if (l < nwin) issue_entries(0u, window_offset(st, e0), base, kWin * (unsigned)sizeof(T));
            if (l + kWave < nwin) issue_entries(0u, window_offset(st, e1), base + kWave * (unsigned)sizeof(T), kWin * (unsigned)sizeof(T));
        }
    };
    // A truncated tile: its L <= 128 entries in order, component-major x[128] y[128] m[128] r[128] across BOTH window
    // buffers (2 * 4 * kWin >= 4 * 128 words), so it can only be issued when the wave is done with its current window:
    // after the hand-off of the turn before.
    static_assert(2 * kWin >= kTile, "a whole tile fits the two window buffers");
    float* const whole = &win[w][0][0][0];
    auto issue_truncated = [&](long long st, int L) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds_offset_of(whole));
        if (l < L) issue_entries(0u, window_offset(st, (unsigned)l), base, kTile * (unsigned)sizeof(T));
        if (l + kWave < L) issue_entries(0u, window_offset(st, (unsigned)(l + kWave)), base + kWave * (unsigned)sizeof(T), kTile * (unsigned)sizeof(T));
    };
    // one entry of a window / of the whole truncated tile, as a record (general code only)
    auto window_record = [&](const float* comp0, int stride, int idx) -> Rec<T> {
        return Rec<T>{comp0[idx], comp0[stride + idx], comp0[2 * stride + idx], any_radius ? comp0[3 * stride + idx] : 0.0f};
    };
    // after the loads have landed: are all coordinates of the window bounded, is some radius not +0.0f
    // The collision screen of a fast turn needs an upper bound of the radii the window holds.  The window's bodies are
    // J[st .. st + 127] (wrapped at N): they lie in the aligned tiles st / 128 and st / 128 + 1 and, when wrapped, tile 0;
    // unpack_slots keeps max |radius| per aligned tile.  Scalar loads, issued with the window a turn ahead.
    auto window_rmax = [&](long long st) -> float {
        if (!any_radius) return 0.0f;
        const int ta = __builtin_amdgcn_readfirstlane((int)(st / kTile));
        if (kLog && (ta < 0 || ta + 1 >= NB_RING_LATE(n_tiles))) {
            if (l == 0) atomicAdd(&NB_RING_LATE(ctr)->errors, kIndexError);
            return __builtin_inff();                       // every lane is flagged: the general code decides
        }
        const float ra = tile_rmax[ta], rb = tile_rmax[ta + 1];
        const float rw = (st + kTile > N) ? tile_rmax[0] : 0.0f;
        const float rab = ra > rb ? ra : rb;
        return rab > rw ? rab : rw;
    };
    struct WindowState { bool fast; float rmax; };
    auto check_window = [&](int kind, int kk, int buf, long long st_w) -> WindowState {
        if (kind != 1) return WindowState{false, 0.0f};    // (a truncated tile is waited for where it is read)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // issued a whole turn ago
        __builtin_amdgcn_wave_barrier();
        // the whole replica is bounded (Meta::summary): no scan of the window; all its radii are +0 or not, globally
        if (all_bounded) return WindowState{wave_ok && fast_tile(kk), window_rmax(st_w)};
        Rec<T> r0{0, 0, 0, 0}, r1{0, 0, 0, 0};
        if (l < nwin) r0 = window_record(&win[w][buf][0][0], kWin, l);
        if (l + kWave < nwin) r1 = window_record(&win[w][buf][0][0], kWin, l + kWave);
        const bool bad0 = !((abs_(r0.x) < kCoordBound) && (abs_(r0.y) < kCoordBound));
        const bool bad1 = !((abs_(r1.x) < kCoordBound) && (abs_(r1.y) < kCoordBound));
        WindowState ws;
        ws.fast = __ballot(bad0 || bad1) == 0ull && wave_ok && fast_tile(kk);
        ws.rmax = window_rmax(st_w);
        return ws;
    };
    // the general code on this turn's walk positions, records from the window (kind 1) / the whole tile (kind 2)
    // `bounded`: the window passed the coordinate check (a fast turn redone for a flagged lane): a pair that is no
    // collision and not closer than 2^-40 then takes the scalar form of the fast chain - the same bits as the general
    // code's IEEE square root and reciprocal (nbody_selftest_ieee_f32) at a quarter of the instructions; a flagged
    // lane holds up its whole ring, so this path is worth keeping short.
    auto general_turn = [&](BodyAcc<T>& a, int kind, int kk, long long st, int L, int off0, int buf, bool bounded) {
        const long long i64 = body_index64();
        const int i = (int)(i64 < 0x7fffffff ? i64 : 0x7fffffff);
        const T growth = NB_RING_LATE(p.growth);
        Event* const ev = kLog ? NB_RING_LATE(ev) : nullptr;
        const int ev_cap = kLog ? NB_RING_LATE(ev_cap) : 0;
        Counters* const ctr = kLog ? NB_RING_LATE(ctr) : nullptr;
        const int step = kLog ? meta->step : 0;
        if (kind == 2) {                                   // issued after the previous own turn's hand-off
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        const int hi = off0 + kT < L ? off0 + kT : L;
        int lane_entry = lit ? l : 0;                      // this lane's first window entry.  Computed HERE: hoisted out of
        asm volatile("" : "+v"(lane_entry));               // the turn loop it is one more register the fast path cannot spare
        const int t = (int)(i64 - blk0);                   // threadIdx.x of the body in the reference
#pragma unroll 1
        for (int off = off0; off < hi; ++off) {
            int sidx;
            long long j;
            if (lit) {
                if (kk == 0 && off == 0) continue;                                 // :200-204
                sidx = (L == kTile) ? ((t + off) & (kTile - 1)) : ((t + off) % L); // :207
                j = st + sidx;
                if (j >= N) j %= N;
            } else {
                sidx = off;
                j = st + off;
                if (j == i64) continue;
            }
            const Rec<T> rec = kind == 1 ? window_record(&win[w][buf][0][0], kWin, lane_entry + (off - off0))
                                         : window_record(whole, kTile, sidx);
            if (bounded) {
                const T dx = rec.x - a.xi, dy = rec.y - a.yi;
                const T d2 = (dx * dx) + (dy * dy);
                const T rs = a.ri + rec.r;
                if (!(d2 <= fma_(rs, rs, kFastLo))) {      // no collision, inside the proved domain
                    const T inv = fast_chain(d2).inv;
                    a.fx = a.fx + (dx * rec.m) * inv;
                    a.fy = a.fy + (dy * rec.m) * inv;
                    continue;
                }
            }
            interact<T, kLog>(a, rec, growth, i, (int)j, ev, ev_cap, ctr, step);
        }
    };

    long long st = tile_start_slow(w / kTurnsPerTile);
    int buf = 0;
    int plain_turns = 0;                                   // turns of this wave that took the common path
    bool first_plain = false;                              // ... the first turn of the walk among them (one pair less)
    int kind = turn_kind(w, st);
    if (kind == 1) issue_window(st, buf);
    if (kind == 2) issue_truncated(st, tile_len(w / kTurnsPerTile, st));
    WindowState cur = check_window(kind, w / kTurnsPerTile, buf, st);
    for (int tau = w; tau < nturns; tau += kW) {
        unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0;
        if (kProbe) pt0 = __builtin_readcyclecounter();
        const int kk = tau / kTurnsPerTile;
        const int off0 = (tau % kTurnsPerTile) * kT;
        const int L = tile_len(kk, st);
        const bool fast = cur.fast;
        const long long st_next = round_on(st);
        const int kind_next = turn_kind(tau + kW, st_next);
        if (kind_next == 1) issue_window(st_next, buf ^ 1);                  // in flight for the whole turn
        const bool first = lit && tau == 0;                // walk position 0 is the body itself (:200-204)
        if (kRings > 1) {                                  // behind the furthest ring of the workgroup: evaluate first
            int ahead = *seq_o1;
            if (kRings == 4) {
                const int o2 = *seq_o2, o3 = *seq_o3;
                ahead = ahead > o2 ? ahead : o2;
                ahead = ahead > o3 ? ahead : o3;
            }
            if (*seq_mine < ahead) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // (2) the kT terms of this turn: walk positions 2v, 2v + 1 in the two halves of termx[v] / termy[v]
        V2 termx[kT / 2], termy[kT / 2];
        auto term_x = [&](int r) -> float { return (r & 1) ? termx[r / 2].y : termx[r / 2].x; };
        auto term_y = [&](int r) -> float { return (r & 1) ? termy[r / 2].y : termy[r / 2].x; };
        unsigned long long flag = 0;
        auto evaluate = [&](auto screen_tag) {
            constexpr bool kScreen = decltype(screen_tag)::value;     // false: no screen per pair (nan_screen)
            const float* wx = &win[w][buf][0][lit ? l : 0];
            const float* wy = wx + kWin;
            const float* wm = wx + 2 * kWin;
            const V2 ownx = {xi, xi}, owny = {yi, yi};
            // Collision / tiny-distance screen, half an instruction per pair: a pair may only take the fast chain if it is
            // no collision, d2 > (ri + rj)^2, and d2 > 2^-80 (the proved domain).  With R = |ri| + (largest |radius| the
            // window can hold) every such pair has d2 > fma(R, R, 2^-80), a per-LANE constant of the turn, so the
            // SMALLEST d2 of the lane's kT pairs decides for all of them (d2 is finite here - the coordinates are
            // bounded - so no NaN can hide in the minimum; a NaN radius never collides and is not in the bound).  A
            // lane below the threshold is redone by the general code, which applies the exact predicate.
            const float reach = abs_(ri) + cur.rmax;
            const float threshold = __builtin_fmaf(reach, reach, kFastLo);
            float closest = kFastHi;
            // the reads of a batch are issued one batch ahead (NB_RING_PREFETCH): their latency then runs under the arithmetic
            // of the batch before instead of being exposed at every batch's start
            V2 pxa = {wx[0], wx[1]}, pxb = {wx[2], wx[3]}, pya = {wy[0], wy[1]}, pyb = {wy[2], wy[3]};
            V2 pma = {wm[0], wm[1]}, pmb = {wm[2], wm[3]};
#pragma unroll
            for (int r0 = 0; r0 < kT; r0 += 4) {           // four walk positions per batch of reads: a, a, b, b
                const V2 xa = pxa, xb = pxb, ya = pya, yb = pyb, ma = pma, mb = pmb;
                if (r0 + 4 < kT) {
                    pxa = V2{wx[r0 + 4], wx[r0 + 5]}; pxb = V2{wx[r0 + 6], wx[r0 + 7]};
                    pya = V2{wy[r0 + 4], wy[r0 + 5]}; pyb = V2{wy[r0 + 6], wy[r0 + 7]};
                    pma = V2{wm[r0 + 4], wm[r0 + 5]}; pmb = V2{wm[r0 + 6], wm[r0 + 7]};
                }
                __builtin_amdgcn_sched_barrier(0);
                const V2 dxa = xa - ownx, dxb = xb - ownx;
                const V2 dya = ya - owny, dyb = yb - owny;
                const V2 sxa = dxa * dxa, sxb = dxb * dxb;
                const V2 sya = dya * dya, syb = dyb * dyb;
                const V2 d2a = sxa + sya, d2b = sxb + syb; // three roundings per element (no contraction in this file)
                if (kScreen) {
                    const float first_d2 = (r0 == 0 && first) ? kFastHi : d2a.x;   // the self position is no pair
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(closest) : "v"(closest), "v"(first_d2), "v"(d2a.y));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(closest) : "v"(closest), "v"(d2b.x), "v"(d2b.y));
                }
                V2 inva, invb;
                fast_inv_cube2x2(d2a, d2b, inva, invb);
                const V2 txa = dxa * ma, txb = dxb * mb;
                const V2 tya = dya * ma, tyb = dyb * mb;
                termx[r0 / 2] = txa * inva; termx[r0 / 2 + 1] = txb * invb;
                termy[r0 / 2] = tya * inva; termy[r0 / 2 + 1] = tyb * invb;
                // the terms are finished HERE (hipcc otherwise sinks the last multiply into the blocks that add them up
                // and keeps both factors alive until then: twice the registers, spills)
                asm volatile("" : "+v"(termx[r0 / 2]), "+v"(termx[r0 / 2 + 1]), "+v"(termy[r0 / 2]), "+v"(termy[r0 / 2 + 1]));
            }
            if (kScreen) flag = le_mask(closest, threshold);
            // the reference skips the self position; the sum starts at +0.0f and +0 + +0 = +0: adding a zero term is
            // the same bits (whatever the self "pair" evaluated to - NaN: d2 = 0 - is dropped here)
            if (first) { termx[0].x = 0.0f; termy[0].x = 0.0f; }
        };
        if (fast) {
            if (nan_screen) evaluate(std::false_type{});
            else evaluate(std::true_type{});
        }
        if (kProbe) pt1 = __builtin_readcyclecounter();
        // (3) the state after turn tau - 1.  Polled at raised priority: a poll is one LDS read plus scalar work, it
        // takes next to nothing from the vector pipelines of the waves that are evaluating, and the chain moves on
        // within one LDS round trip of the record being written.
        Int4 h = Int4{0, 0, 0, 0};
        bool timed_out = false;
        __builtin_amdgcn_s_setprio(3);
        if (tau > 0) {
            int spins = 0;
            for (;;) {
                h = *hand_l;                               // one ds_read_b128
                if (__ballot(h.z < tau) == 0ull) break;
                if (++spins > spin_limit) { timed_out = true; break; }
                if (kSleep > 0) __builtin_amdgcn_s_sleep(kSleep);
            }
            if (kProbe) pr_polls += spins + 1;
        }
        if (kProbe) pt2 = __builtin_readcyclecounter();
        // (4) + (5).  The common case - a fast turn, no flagged lane, the sequence number the expected one - is kept as
        // short as the arithmetic allows, because it is the serial part of the whole workgroup: 2 kT dependent adds
        // (x and y chains interleaved: scalar adds need no wait states between dependent instructions, packed ones
        // do), then one LDS write; `flags` passes through untouched.  The last turn publishes too: the epilogue takes the
        // final state from the records.
        bool plain = fast && flag == 0ull && !dead && !timed_out && __ballot(h.z != tau) == 0ull;
        if (__builtin_expect(plain, 1)) {
            float fx = __int_as_float(h.x), fy = __int_as_float(h.y);
#pragma unroll
            for (int r = 0; r < kT; ++r) {
                fx = fx + term_x(r);
                asm("" : "+v"(fx));                        // keeps hipcc from pairing the two adds into one v_pk_add_f32
                fy = fy + term_y(r);
            }
            // nan_screen: a coincident pair (d2 = 0: a collision, :215-226) made its term NaN, and a NaN among the kT terms
            // is a NaN sum.  ONE comparison per turn instead of a v_min3 per two pairs; a lane whose sum was NaN already is
            // flagged in every turn (slow, and right: its collisions still have to be found).
            if (nan_screen) flag = __builtin_amdgcn_fcmpf(fx, fy, 8 /* llvm::CmpInst::FCMP_UNO */);
            if (__builtin_expect(flag == 0ull, 1)) {
                *hand_l = Int4{(int)__float_as_uint(fx), (int)__float_as_uint(fy), tau + 1, h.w};
                __builtin_amdgcn_s_setprio(0);
                plain_turns += 1;                          // a scalar: kT pairs per active lane, added up at the end
                first_plain = first_plain || first;
            } else {
                plain = false;                             // the sums are dropped: the turn is redone below from `h`
            }
        }
        if (__builtin_expect(!plain, 0)) {
            // the lane's whole state, for the general code: position and radius from the registers, the running sum and
            // `deleted` from the record just received, {mnew, rnew, mi} from the rare record (its last writer published
            // it before the sequence number this wave has seen)
            BodyAcc<T> a;
            a.xi = xi; a.yi = yi; a.ri = ri;
            a.fx = __int_as_float(h.x); a.fy = __int_as_float(h.y);
            a.deleted = h.w & 1;
            const Float4 hm = *hand_m_l;
            a.mnew = hm.x; a.rnew = hm.y; a.mi = hm.z;
            timeouts += timed_out ? 1 : 0;
            dead = dead || timed_out || __ballot(h.z >= kRingDeadSeq) != 0ull;
            const unsigned m_before = __float_as_uint(a.mnew), r_before = __float_as_uint(a.rnew);
            if (fast && nan_screen) {                      // (a turn that came here without its sums: every NaN term flags its lane)
#pragma unroll
                for (int r = 0; r < kT; ++r) flag |= __builtin_amdgcn_fcmpf(term_x(r), term_y(r), 8);
            }
            if (fast) {
                // The flagged lanes (a SUPERSET of the lanes with a collision or a tiny distance in this turn) get the exact
                // status of each of their kT pairs, and they get it in parallel: lane r of the wave evaluates walk position r
                // of flagged lane fl's body from the same window.  `hits`: positions that are collisions the reference
                // handles and moves on from (:215-226: no force term); `odd`: some pair inside the guard that is not
                // such a collision (tiny distance, the 2^-80 margin, a NaN mass) - that lane is redone by the general code.
                unsigned hits = 0;
                bool odd = false;
                unsigned long long todo = flag & __ballot(active);
                while (todo != 0ull) {
                    const int fl = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
                    todo &= todo - 1ull;
                    const T xf = __int_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(a.xi), fl));
                    const T yf = __int_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(a.yi), fl));
                    const T mf = __int_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(a.mi), fl));
                    const T rf = __int_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(a.ri), fl));
                    int r = l & (kT - 1);                                       // formed HERE (see general_turn's lane_entry)
                    asm volatile("" : "+v"(r));
                    const Rec<T> rec = window_record(&win[w][buf][0][0], kWin, (lit ? fl : 0) + r);
                    const T dx = rec.x - xf, dy = rec.y - yf;
                    const T d2 = (dx * dx) + (dy * dy);
                    const T rs = rf + rec.r;
                    const bool is_pair = l < kT && !(first && r == 0);          // walk position 0 of tile 0 is the body itself
                    const bool hit = d2 <= rs * rs && (mf >= rec.m || mf < rec.m); // interact(): `hit && (ge || lt)`
                    const bool guarded = d2 <= fma_(rs, rs, kFastLo);              // what the fast chain must not see
                    const unsigned long long hm = __ballot(is_pair && hit);
                    const unsigned long long om = __ballot(is_pair && guarded && !hit);
                    if (l == fl) { hits = (unsigned)hm; odd = om != 0ull; }
                }
                if (active) {
                    if (odd) {
                        general_turn(a, 1, kk, st, L, off0, buf, true);
                    } else {
                        float fx = a.fx, fy = a.fy;
#pragma unroll
                        for (int r = 0; r < kT; ++r) {
                            const float nx = add_unmerged(fx, term_x(r));
                            const float ny = add_unmerged(fy, term_y(r));
                            const bool skip = ((hits >> r) & 1u) != 0u;            // a collision adds no force term (not even +0)
                            fx = skip ? fx : nx;
                            fy = skip ? fy : ny;
                        }
                        a.fx = fx; a.fy = fy;
                        unsigned rest = hits;                                      // the collisions themselves, in walk order
                        int lane_entry = lit ? l : 0;
                        asm volatile("" : "+v"(lane_entry));                       // (see general_turn)
                        const T growth = NB_RING_LATE(p.growth);
                        while (rest != 0u) {
                            const int r = __builtin_ctz(rest);
                            rest &= rest - 1u;
                            const Rec<T> rec = window_record(&win[w][buf][0][0], kWin, lane_entry + r);
                            const bool ge = a.mi >= rec.m;
                            if (ge) {                                              // :215-221
                                a.mnew = a.mnew + rec.m;
                                a.rnew = a.rnew + rec.r * growth;
                            } else {                                               // :222-226
                                a.deleted = 1;
                            }
                            if (kLog) {
                                const long long i64 = body_index64();
                                const int i = (int)(i64 < 0x7fffffff ? i64 : 0x7fffffff);
                                long long j = lit ? st + (((int)(i64 - blk0) + off0 + r) & (kTile - 1)) : st + off0 + r;
                                if (j >= N) j %= N;
                                Counters* const ctr = NB_RING_LATE(ctr);
                                const unsigned long long slot = atomicAdd(&ctr->events, 1ull);
                                if (slot < (unsigned long long)NB_RING_LATE(ev_cap))
                                    NB_RING_LATE(ev)[slot] = Event{meta->step, i, (int)j, ge ? 0 : 1};
                            }
                        }
                    }
                    pairs_rare += kT - (first ? 1 : 0);
                }
            } else if (active) {
                general_turn(a, kind, kk, st, L, off0, buf, false);
                const int hi = off0 + kT < L ? off0 + kT : L;
                if (lit) {
                    if (hi > off0) pairs_rare += (hi - off0) - ((kk == 0 && off0 == 0) ? 1 : 0);
                } else {
                    const long long i64 = body_index64();
                    for (int off = off0; off < hi; ++off) pairs_rare += (st + off != i64) ? 1 : 0;
                }
            }
            if (dead) {                                    // a hand-off wait gave up somewhere before: poison, never a result
                a.fx = a.fy = a.mnew = a.rnew = __builtin_nanf("");
                a.deleted = 0;
            }
            // publish: the rare record first, then the one the next wave polls (a wave's LDS operations execute in order)
            if (__float_as_uint(a.mnew) != m_before || __float_as_uint(a.rnew) != r_before)
                *hand_m_l = Float4{a.mnew, a.rnew, a.mi, 0.0f};
            *hand_l = Int4{(int)__float_as_uint(a.fx), (int)__float_as_uint(a.fy), dead ? kRingDeadSeq : tau + 1,
                           a.deleted & 1};
            __builtin_amdgcn_s_setprio(0);
        }
        if (kProbe) pt3 = __builtin_readcyclecounter();
        // the prefetched window of this wave's next turn
        if (kind_next == 2) issue_truncated(st_next, tile_len((tau + kW) / kTurnsPerTile, st_next));   // both buffers are free now
        buf ^= 1;
        st = st_next;
        kind = kind_next;
        cur = check_window(kind, (tau + kW) / kTurnsPerTile, buf, st);
        if (kProbe) {
            pr_eval += pt1 - pt0; pr_wait += pt2 - pt1; pr_chain += pt3 - pt2;
            pr_check += __builtin_readcyclecounter() - pt3;
        }
    }
    // Nothing above this line in the loop is a memory access the compiler tracks in vmcnt (the window loads and the
    // general code's record loads are inline assembly with their own waits): hipcc therefore places no vmcnt wait in
    // the loop, and the only one there is check_window's, for a prefetch issued a whole turn earlier.
    if (mine && (nturns - 1) % kW == w) {                  // the wave that took the last turn: epilogue, from the records it
        const long long i64 = body_index64();              // has just written (a wave's LDS operations execute in order)
        const int q = (int)(i64 - lo);
        const Int4 hf = *hand_l;
        const Float4 hm = *hand_m_l;
        BodyAcc<T> a;
        a.xi = xi; a.yi = yi; a.ri = ri; a.mi = hm.z;
        a.fx = __int_as_float(hf.x); a.fy = __int_as_float(hf.y);
        a.mnew = hm.x; a.rnew = hm.y; a.deleted = hf.w & 1;
        Rec<T>* const S_J = NB_RING_LATE(S_J);
        Vec2<T>* const S_V = NB_RING_LATE(S_V);
        const bool q_ok = q >= 0 && q < NB_RING_LATE(cap_own);
        const Vec2<T> v = q_ok ? NB_RING_LATE(Vown)[q] : Vec2<T>{0, 0};
        if (!q_ok) {
            atomicAdd(&NB_RING_LATE(ctr)->errors, kIndexError);
        } else if (active) {
            const StepParams<T> p = NB_RING_LATE(p);
            Rec<T> out; Vec2<T> vout;
            finish_body<T>(a, v, p, out, vout);
            S_J[q] = out;
            S_V[q] = vout;
        } else {   // frozen body: no thread exists for it in the reference, state carried over unchanged
            S_J[q] = Rec<T>{a.xi, a.yi, a.mi, a.ri};
            S_V[q] = v;
        }
    }
    Counters* const ctr = NB_RING_LATE(ctr);
    unsigned long long pairs = 0;
    if (active) pairs = (unsigned long long)((long long)plain_turns * kT - (first_plain ? 1 : 0) + pairs_rare);
    if (timeouts != 0 && l == 0) atomicAdd(&ctr->errors, (unsigned long long)timeouts);
    for (int sh = kWave / 2; sh > 0; sh >>= 1) pairs += __shfl_down(pairs, sh, kWave);
    if (l == 0 && pairs) atomicAdd(&ctr->pairs, pairs);
    if (kProbe && l == 0) {
        Event* const ev = NB_RING_LATE(ev);
        const int ev_cap = NB_RING_LATE(ev_cap);
        atomicAdd(&ctr->probe[0], pr_eval); atomicAdd(&ctr->probe[1], pr_wait);
        atomicAdd(&ctr->probe[2], pr_chain); atomicAdd(&ctr->probe[3], pr_check);
        atomicAdd(&ctr->probe[4], pr_polls);
        atomicAdd(&ctr->probe[5], (unsigned long long)((nturns - w + kW - 1) / kW));
        if (wg == 0 && w == 0) {
            ctr->probe[6] = __builtin_readcyclecounter() - pr_t0;   // shader clocks of one wave's life
            ctr->probe[7] = wall_clock64() - pr_r0;                 // the same in 100 MHz ticks
        }
        if (w == 0) {   // where and when this workgroup ran: {start, end (constant-rate ticks), HW_ID, XCC_ID << 20 | wg}
            const unsigned long long slot = atomicAdd(&ctr->events, 1ull);
            if (slot < (unsigned long long)ev_cap)
                ev[slot] = Event{(int)(unsigned)pr_r0, (int)(unsigned)wall_clock64(),
                                 (int)__builtin_amdgcn_s_getreg((31 << 11) | 4),
                                 (int)((__builtin_amdgcn_s_getreg((31 << 11) | 20) << 20) | (unsigned)wg)};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stable compaction of the own range on `mass != 0` (src/nbody.cu:488-510), two small kernels.
// ---------------------------------------------------------------------------------------------------------
constexpr int kCompactBlock = 1024;
// The tiled copy of the fp32 replica (the ring kernel's window source): body i of aligned tile i / 128 at x[i % 128],
// y[...], m[...], r[...] of that tile's 2 KiB.
template <typename T>
__device__ __forceinline__ void store_tiled(float* __restrict__ Jt, int i, const Rec<T>& r) {
    float* t = Jt + (size_t)(i / kTile) * (4 * kTile) + (i % kTile);
    t[0] = (float)r.x; t[kTile] = (float)r.y; t[2 * kTile] = (float)r.m; t[3 * kTile] = (float)r.r;
}
__global__ __launch_bounds__(256) void records_to_tiles_f32(const Rec<float>* __restrict__ J, int n, float* __restrict__ Jt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) store_tiled(Jt, i, J[i]);
}

template <typename T>
__global__ __launch_bounds__(kCompactBlock) void compact_count(const Rec<T>* __restrict__ S_J,
                                                               const Meta* __restrict__ meta,
                                                               int* __restrict__ blk_counts,
                                                               unsigned* __restrict__ tile_rmax, int n_tiles) {
    __shared__ int wsum[kCompactBlock / kWave];
    const int cnt = meta->cnt;
    const int q = blockIdx.x * kCompactBlock + threadIdx.x;
    if (q == 0) const_cast<Meta*>(meta)->summary = 0;      // the force kernel of this step is done with it
    for (int k = q; k < n_tiles; k += gridDim.x * kCompactBlock) tile_rmax[k] = 0u;   // ... and with these: unpack_slots refills
    const bool keep = q < cnt && S_J[q].m != (T)0;
    const unsigned long long bal = __ballot(keep);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kCompactBlock / kWave; ++w) s += wsum[w];
        blk_counts[blockIdx.x] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(kCompactBlock) void compact_scatter(const Rec<T>* __restrict__ S_J,
                                                                 const Vec2<T>* __restrict__ S_V,
                                                                 const Meta* __restrict__ meta,
                                                                 const int* __restrict__ blk_counts, int nblk,
                                                                 SlotHeader* __restrict__ slot_hdr,
                                                                 Rec<T>* __restrict__ slot_recs,
                                                                 Vec2<T>* __restrict__ slot_vels) {
    __shared__ int wsum[kCompactBlock / kWave];
    __shared__ int red[kCompactBlock / kWave];
    __shared__ int base_s;
    const int cnt = meta->cnt;
    // offset of this block = sum of the counts of all lower blocks
    int part = 0;
    for (int bidx = threadIdx.x; bidx < (int)blockIdx.x; bidx += kCompactBlock) part += blk_counts[bidx];
    for (int sh = kWave / 2; sh > 0; sh >>= 1) part += __shfl_down(part, sh, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = part;

    const int q = blockIdx.x * kCompactBlock + threadIdx.x;
    Rec<T> rec{};
    Vec2<T> vel{};
    bool keep = false;
    if (q < cnt) {
        rec = S_J[q];
        vel = S_V[q];
        keep = rec.m != (T)0;
    }
    const unsigned long long bal = __ballot(keep);
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x / kWave;
    if (lane == 0) wsum[wid] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kCompactBlock / kWave; ++w) s += red[w];
        base_s = s;
        if ((int)blockIdx.x == nblk - 1) {
            int tot = s;
            for (int w = 0; w < kCompactBlock / kWave; ++w) tot += wsum[w];
            slot_hdr->count = tot;
            slot_hdr->layout = (int)(reinterpret_cast<const Rec<T>*>(slot_vels) - slot_recs);
        }
    }
    __syncthreads();
    if (keep) {
        int off = base_s;
        for (int w = 0; w < wid; ++w) off += wsum[w];
        off += __popcll(bal & ((1ull << lane) - 1ull));
        slot_recs[off] = rec;
        slot_vels[off] = vel;
    }
}

// Builds step t+1 from the gathered slots of all ranks (global stable order = rank order): the replica of all
// bodies, the velocities of THIS rank's new own range, and the new Meta.  A slot = header {count} | records[cap_own] |
// velocities[cap_own].  grid = (ceil(cap_own / 256), world).
template <typename T>
__global__ __launch_bounds__(256) void unpack_slots(const unsigned char* __restrict__ gather, size_t slot_bytes,
                                                    int cap_own, int world, int rank, Rec<T>* __restrict__ J,
                                                    Vec2<T>* __restrict__ Vown, Meta* __restrict__ meta,
                                                    unsigned* __restrict__ tile_rmax, float* __restrict__ Jt,
                                                    Counters* __restrict__ ctr) {
    const int g = blockIdx.y;
    int off = 0, total = 0;
    bool bad_header = false;
    // A header that does not fit the layout this rank unpacks with - another rank laid its slot out differently, or the
    // gather brought something else - must not become an index: counts are clamped to the layout and the step is
    // reported as failed (Counters::errors, like the ring kernel's index checks).
    auto count_of = [&](int h) -> int {
        const SlotHeader* hd = reinterpret_cast<const SlotHeader*>(gather + (size_t)h * slot_bytes);
        const int c = hd->count;
        if (c < 0 || c > cap_own || hd->layout != cap_own) { bad_header = true; return c < 0 ? 0 : (c > cap_own ? cap_own : c); }
        return c;
    };
    for (int h = 0; h < world; ++h) {
        const int c = count_of(h);
        if (h < g) off += c;
        total += c;
    }
    if (bad_header && g == 0 && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&ctr->errors, kIndexError);
    int lo, cnt;
    nbody_own_range_of(total, rank, world, &lo, &cnt);       // csrc/nbody_partition.h
    const unsigned char* slot = gather + (size_t)g * slot_bytes;
    const int c = count_of(g);
    const Rec<T>* recs = reinterpret_cast<const Rec<T>*>(slot + sizeof(SlotHeader));
    const Vec2<T>* vels = reinterpret_cast<const Vec2<T>*>(slot + sizeof(SlotHeader) + (size_t)cap_own * sizeof(Rec<T>));
    const int q = blockIdx.x * 256 + threadIdx.x;
    int bits = 0;
    unsigned rbits = 0;                                    // bits of |radius| (non-negative floats order like unsigned)
    if (q < c) {
        const Rec<T> r = recs[q];
        const int i = off + q;                             // index of this body in step t+1
        J[i] = r;
        if (Jt != nullptr) store_tiled(Jt, i, r);          // fp32 contexts
        if (i >= lo && i < lo + cnt) Vown[i - lo] = vels[q];
        const bool bounded = abs_(r.x) < FastDomain<T>::coord && abs_(r.y) < FastDomain<T>::coord;
        bits = (bounded ? 0 : kSummaryUnbounded) | (not_plus_zero(r.r) ? kSummaryRadius : 0);
        if (Jt != nullptr) bits |= coord_summary((float)r.x, (float)r.y, (float)r.m) & (kSummarySmall | kSummaryMass);   // fp32 contexts (ring kernel)
        const float ar = (float)abs_(r.r);
        rbits = (ar == ar) ? __float_as_uint(ar) : 0u;      // a NaN radius never collides (the predicate is false): skipped
    }
    // Largest |radius| per aligned 128-body tile of the new replica, for the ring kernel's collision screen.  The 64
    // lanes of a wave hold consecutive bodies, i.e. at most two tiles: one wave reduction per tile, two atomics per wave.
    {
        const int i0 = off + blockIdx.x * 256 + (int)(threadIdx.x & ~(kWave - 1u));   // body of this wave's lane 0
        const int t0 = i0 / kTile;
        const bool in_t0 = (off + q) / kTile == t0;
        unsigned m0 = in_t0 ? rbits : 0u, m1 = in_t0 ? 0u : rbits;
        for (int sh = kWave / 2; sh > 0; sh >>= 1) {
            const unsigned o0 = __shfl_xor(m0, sh, kWave), o1 = __shfl_xor(m1, sh, kWave);
            m0 = o0 > m0 ? o0 : m0;
            m1 = o1 > m1 ? o1 : m1;
        }
        if ((threadIdx.x & (kWave - 1)) == 0) {
            if (m0 != 0u) atomicMax(&tile_rmax[t0], m0);
            if (m1 != 0u) atomicMax(&tile_rmax[t0 + 1], m1);
        }
    }
    const int wave_bits = (__ballot(bits & kSummaryUnbounded) != 0ull ? kSummaryUnbounded : 0) |
                          (__ballot(bits & kSummaryRadius) != 0ull ? kSummaryRadius : 0) |
                          (__ballot(bits & kSummarySmall) != 0ull ? kSummarySmall : 0) |
                          (__ballot(bits & kSummaryMass) != 0ull ? kSummaryMass : 0);
    if (wave_bits != 0 && (threadIdx.x & (kWave - 1)) == 0) atomicOr(&meta->summary, wave_bits);
    if (g == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        // every block has read meta-independent data only, so the in-place update is race-free
        const int step = meta->step;
        meta->n_prev = meta->n;
        meta->n = total;
        meta->lo = lo;
        meta->cnt = cnt;
        meta->step = step + 1;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Reference-shaped kernels on the reference's device block layout (drop-in for the <<<>>> sites
// src/nbody.cu:481-483): velocities updated in place, updatedMasses / updatedRadii scratch written.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTile) void ref_layout_forces_f32(void* bodyData, float* __restrict__ updM,
                                                               float* __restrict__ updR, int N, int nb,
                                                               StepParams<float> p) {
    __shared__ Rec<float> tile[2][kTile];
    const Vec2<float>* P = reinterpret_cast<const Vec2<float>*>(bodyData);    // :147-150
    Vec2<float>* V = reinterpret_cast<Vec2<float>*>(bodyData) + N;
    const float* M = reinterpret_cast<const float*>(V + N);
    const float* R = M + N;
    const int t = threadIdx.x;
    const long long blk0 = (long long)blockIdx.x * kTile;
    const int i = (int)blk0 + t;
    const bool active = i < N;                                                // :143
    BodyAcc<float> a;
    Vec2<float> v{0, 0};
    if (active) {
        const Vec2<float> pi = P[i];
        a.xi = pi.x; a.yi = pi.y; a.mi = M[i]; a.ri = R[i];
        v = V[i];
    } else {
        a.xi = a.yi = a.mi = a.ri = 0;
    }
    a.fx = 0; a.fy = 0; a.mnew = a.mi; a.rnew = a.ri; a.deleted = 0;
    auto load = [&](int k) -> Rec<float> {
        long long src = blk0 + (long long)kTile * k + t;
        if (src >= N) src %= N;
        const Vec2<float> pj = P[src];
        return Rec<float>{pj.x, pj.y, M[src], R[src]};
    };
    if (active) tile[0][t] = load(0);
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        const int cur = k & 1;
        Rec<float> nxt{};
        const bool have_next = k + 1 < nb;
        if (have_next && active) nxt = load(k + 1);
        const int L = (k == nb - 1) ? N % (kTile + 1) : kTile;
        if (active) {
            for (int off = (k == 0 ? 1 : 0); off < L; ++off) {
                const int s = (L == kTile) ? ((t + off) & (kTile - 1)) : ((t + off) % L);
                interact<float, false>(a, tile[cur][s], p.growth, i, 0, nullptr, 0, nullptr, 0);
            }
        }
        if (have_next && active) tile[cur ^ 1][t] = nxt;
        __syncthreads();
    }
    if (active) {
        // finish_body computes the drifted position too; only the :245-264 part is stored here, the drift
        // belongs to the separate MoveBodies launch in this API shape
        Rec<float> out; Vec2<float> vout;
        finish_body<float>(a, v, p, out, vout);
        updM[i] = out.m;
        updR[i] = out.r;
        V[i] = vout;
    }
}

// The reference-shaped launch through the ring kernel (nbody_launch_compute_forces_f32 with the reference's own block
// count): the device block [P|V|M|R] is packed into the {x,y,m,r} replica of a process-wide workspace - with Meta and
// the per-tile radius bounds exactly as nbody_upload / unpack_slots produce them -, the ring kernel reads the
// velocities where they lie in the block, and its staged output is written back in the reference's form:
// velocities in place, updatedMasses / updatedRadii (src/nbody.cu:245-246,264).  meta and tile_rmax are zeroed before.
__global__ __launch_bounds__(256) void ref_layout_pack_f32(const void* bodyData, int N, Rec<float>* __restrict__ J,
                                                           Meta* __restrict__ meta, unsigned* __restrict__ tile_rmax,
                                                           float* __restrict__ Jt) {
    const Vec2<float>* P = reinterpret_cast<const Vec2<float>*>(bodyData);    // :147-150
    const float* M = reinterpret_cast<const float*>(P + 2 * (size_t)N);
    const float* R = M + N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int bits = 0;
    unsigned rbits = 0;
    if (i < N) {
        const Vec2<float> pi = P[i];
        const Rec<float> r{pi.x, pi.y, M[i], R[i]};
        J[i] = r;
        store_tiled(Jt, i, r);
        bits = coord_summary(r.x, r.y, r.m) | (not_plus_zero(r.r) ? kSummaryRadius : 0);
        const float ar = abs_(r.r);
        rbits = (ar == ar) ? __float_as_uint(ar) : 0u;
    }
    for (int sh = kWave / 2; sh > 0; sh >>= 1) {           // a wave's 64 bodies lie in one aligned 128-body tile
        const unsigned o = __shfl_xor(rbits, sh, kWave);
        rbits = o > rbits ? o : rbits;
    }
    const int wave_bits = (__ballot(bits & kSummaryUnbounded) != 0ull ? kSummaryUnbounded : 0) |
                          (__ballot(bits & kSummaryRadius) != 0ull ? kSummaryRadius : 0) |
                          (__ballot(bits & kSummarySmall) != 0ull ? kSummarySmall : 0) |
                          (__ballot(bits & kSummaryMass) != 0ull ? kSummaryMass : 0);
    if ((threadIdx.x & (kWave - 1)) == 0) {
        if (rbits != 0u) atomicMax(&tile_rmax[i / kTile], rbits);
        if (wave_bits != 0) atomicOr(&meta->summary, wave_bits);
    }
    if (i == 0) { meta->n = N; meta->lo = 0; meta->cnt = N; meta->step = 0; meta->n_prev = N; }
}

__global__ __launch_bounds__(256) void ref_layout_finish_f32(void* bodyData, float* __restrict__ updM,
                                                             float* __restrict__ updR, int N, int n_active,
                                                             const Rec<float>* __restrict__ S_J,
                                                             const Vec2<float>* __restrict__ S_V) {
    Vec2<float>* V = reinterpret_cast<Vec2<float>*>(bodyData) + N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_active) {                                    // bodies past the last full block have no thread (:142-143, :473)
        const Rec<float> out = S_J[i];
        updM[i] = out.m;
        updR[i] = out.r;
        V[i] = S_V[i];
    }
}

__global__ __launch_bounds__(kTile) void ref_layout_move_f32(void* bodyData, const float* __restrict__ updM,
                                                             const float* __restrict__ updR, int N, float dt) {
    Vec2<float>* P = reinterpret_cast<Vec2<float>*>(bodyData);                // :283-286
    const Vec2<float>* V = P + N;
    float* M = reinterpret_cast<float*>(P + 2 * (size_t)N);
    float* R = M + N;
    const int j = blockIdx.x * kTile + threadIdx.x;
    if (j < N) {
        Vec2<float> pj = P[j];
        const Vec2<float> vj = V[j];
        pj.x = pj.x + dt * vj.x;                                              // :288
        pj.y = pj.y + dt * vj.y;
        P[j] = pj;
        M[j] = updM[j];                                                       // :289
        R[j] = updR[j];                                                       // :290
    }
}

// ---------------------------------------------------------------------------------------------------------
// Image rasteriser: generateImage, src/nbody.cu:294-348, one lane per body, filled discs of value 0 into an
// image pre-set to 254 (:534); concurrent writers all store 0, so the result is deterministic.  Differences
// from the reference, both asked for by SURVEY.md 8 f3: the missing `i < numBodies` guard is present, and
// bodies are read from the {x,y,m,r} records.  `limit` = 128 * (block count of the reference's launch, :535,
// i.e. the count of the step that produced the state) in literal mode, the body count in clean mode.
// fp64 contexts rasterise the float-rounded coordinates.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void render_discs(const Rec<T>* __restrict__ J, const Meta* __restrict__ meta,
                                                    int literal, unsigned char* __restrict__ img, int width,
                                                    int height, int fieldWidth, int fieldHeight) {
    const int n = meta->n;
    long long limit = n;
    if (literal) {
        const int np = meta->n_prev;
        limit = (long long)(np < kTile ? 1 : np / kTile) * kTile;             // :473, :535
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || i >= limit) return;
    const float px = (float)J[i].x, py = (float)J[i].y;
    const float r = ((float)J[i].r * width) / fieldWidth;                     // :310
    const int doubleFieldWidth = fieldWidth << 1, doubleFieldHeight = fieldHeight << 1;   // :314-315
    const int xc = (int)(((px + fieldWidth) / doubleFieldWidth) * width);     // :318
    const int yc = (int)(((py + fieldHeight) / doubleFieldHeight) * height);  // :319
    const int y_min = yc - r < 0 ? 0 : yc - r;                                // :323
    const int y_max = yc + r >= height ? height : yc + r;                     // :324
    const int x_min = xc - r < 0 ? 0 : xc - r;                                // :325
    const int x_max = xc + r > width ? width : xc + r;                        // :326
    const int r2 = (int)(r * r);
    for (int y = y_min; y < y_max; ++y)                                       // :328-347
        for (int x = x_min; x < x_max; ++x) {
            const int x_sq = (x - xc) * (x - xc), y_sq = (y - yc) * (y - yc);
            if (x_sq + y_sq <= r2) img[(size_t)width * y + x] = 0;
        }
}

// ---------------------------------------------------------------------------------------------------------
// Device self-test of the assumption behind the ring kernel's hand-off: a 16-byte LDS record written by one lane
// with ds_write_b128 is seen by a ds_read_b128 of another wave entirely old or entirely new.  Wave 0 of every
// workgroup rewrites its 64 records {k, k, k, k} for k = 1..iters; the other seven waves poll them, count records
// whose four words differ (torn) or whose k went backwards, and leave when they have seen k = iters.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(8 * kWave) void selftest_lds_record(unsigned long long* out, int iters) {
    __shared__ Int4 rec[kWave];
    const int l = threadIdx.x % kWave, w = threadIdx.x / kWave;
    const LdsInt4Ptr mine = (LdsInt4Ptr)&rec[l];
    if (w == 0) *mine = Int4{0, 0, 0, 0};
    __syncthreads();
    unsigned long long torn = 0, backwards = 0, reads = 0;
    if (w == 0) {
        for (int k = 1; k <= iters; ++k) *mine = Int4{k, k, k, k};
    } else {
        int last = 0;
        for (long long guard = 0; guard < (1ll << 40); ++guard) {
            const Int4 r = *mine;
            ++reads;
            torn += (r.x != r.y) || (r.y != r.z) || (r.z != r.w);
            backwards += r.x < last;
            last = r.x;
            if (__ballot(r.x < iters) == 0ull) break;
        }
    }
    if (torn) atomicAdd(&out[0], torn);
    if (backwards) atomicAdd(&out[1], backwards);
    if (reads) atomicAdd(&out[2], reads);
}

// ---------------------------------------------------------------------------------------------------------
// Device self-test: fp32 sqrt and reciprocal as compiled in this TU vs fp64 evaluation rounded once to fp32
// (innocuous double rounding: 53 >= 2*24+2), on all 2^32 bit patterns.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void selftest_ieee_f32(unsigned long long* mism) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned long long bad_sqrt = 0, bad_rcp = 0, bad_fast = 0;
    for (unsigned long long u = gid; u < (1ull << 32); u += stride) {
        const float x = __uint_as_float((unsigned)u);
        const float s1 = ieee_sqrt<float>(x);
        const float s2 = (float)__builtin_sqrt((double)x);
        const float r1 = 1.0f / x;
        const float r2 = (float)(1.0 / (double)x);
        const bool s_ok = (__float_as_uint(s1) == __float_as_uint(s2)) || (s1 != s1 && s2 != s2);
        const bool r_ok = (__float_as_uint(r1) == __float_as_uint(r2)) || (r1 != r1 && r2 != r2);
        bad_sqrt += !s_ok;
        bad_rcp += !r_ok;
        // the fast chain of the fp32 force kernels against the general code, on its whole guarded domain
        if (x >= kFastLo && x <= kFastHi) {
            const FastChain f = fast_chain(x);
            const float c = (s1 * s1) * s1;
            const float inv = 1.0f / c;
            bad_fast += (__float_as_uint(f.d) != __float_as_uint(s1)) || (__float_as_uint(f.inv) != __float_as_uint(inv));
            // the 2-vector form, with this input in either element
            Pair<float>::type v2;
            v2.x = x; v2.y = 3.0f;
            const Pair<float>::type i2 = fast_inv_cube2(v2);
            v2.x = 0.75f; v2.y = x;
            const Pair<float>::type j2 = fast_inv_cube2(v2);
            bad_fast += (__float_as_uint(i2.x) != __float_as_uint(inv)) || (__float_as_uint(j2.y) != __float_as_uint(inv));
        }
    }
    if (bad_sqrt) atomicAdd(&mism[0], bad_sqrt);
    if (bad_rcp) atomicAdd(&mism[1], bad_rcp);
    if (bad_fast) atomicAdd(&mism[2], bad_fast);
}

// Device self-test of the fp64 fast chain against the compiler's IEEE sqrt and 1/x on its guarded domain: per
// thread `iters` inputs from a counter-based generator.  mode 0: random mantissa, exponent uniform in [-500, 500];
// mode 1: mantissas within 2^12 ulps of a power of two from either side; mode 2: perfect squares +- 4 ulps.
__global__ __launch_bounds__(256) void selftest_chain_f64(unsigned long long* mism, int mode, int iters,
                                                          unsigned long long seed) {
    unsigned long long s = seed + 0x1234567ull * ((unsigned long long)blockIdx.x * 256 + threadIdx.x);
    auto next = [&]() {
        unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    };
    unsigned long long bad_sqrt = 0, bad_inv = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long r = next(), r2 = next();
        const int e = (int)(r2 % 1001) - 500;
        unsigned long long man = r & 0xfffffffffffffull;
        if (mode == 1) man = (r & 1) ? ((r >> 1) & 0xfff) : 0xfffffffffffffull - ((r >> 1) & 0xfff);
        double x = __longlong_as_double((long long)(((unsigned long long)(e + 1023) << 52) | man));
        if (mode == 2) {
            const double k = (double)((r >> 20) | 1);
            x = __longlong_as_double(__double_as_longlong(k * k) + (long long)(r2 % 9) - 4);
        }
        const FastChainD f = fast_chain(x);
        const double d = ieee_sqrt<double>(x);
        const double c = (d * d) * d;
        const double inv = ieee_rcp<double>(c);
        bad_sqrt += __double_as_longlong(f.d) != __double_as_longlong(d);
        bad_inv += __double_as_longlong(f.inv) != __double_as_longlong(inv);
    }
    if (bad_sqrt) atomicAdd(&mism[0], bad_sqrt);
    if (bad_inv) atomicAdd(&mism[1], bad_inv);
}

// Device self-test of the one known exception of the fp64 reciprocal refinements (ieee_rcp): c = (2 - 2^-52) 2^k for every
// exponent of the guarded domain of d^3, [2^-750, 2^750].  Expected: 2^-(k+1) (1 + 2^-52).  out[0]: mismatches of
// ieee_rcp<double> (the general code; must be 0), out[1]: of the compiler's bare 1.0 / c (informational), out[2]: of the fast
// chain's refinement from a seed as good as the kernels' (informational: this is WHY the kernels screen such c),
// out[3]: inputs the screen (low word == 0xffffffff) would have missed (must be 0), out[4]: inputs checked.
__global__ __launch_bounds__(256) void selftest_rcp_ones_f64(unsigned long long* out) {
    const int k = (int)(blockIdx.x * 256 + threadIdx.x) - 750;
    if (k > 750) return;
    const double c = __longlong_as_double((long long)(((unsigned long long)(k + 1023) << 52) | kF64Frac));
    const double want = __longlong_as_double((long long)(((unsigned long long)(1023 - k - 1) << 52) | 1ull));
    const double general = ieee_rcp<double>(c);
    double bare;
    {
        double cc = c;
        asm volatile("" : "+v"(cc));                       // the plain division, not folded
        bare = 1.0 / cc;
    }
    const double d = ::cbrt(c);
    const double fast = fast_rcp_cube(c, 0.5 / d);
    atomicAdd(&out[0], (unsigned long long)(__double_as_longlong(general) != __double_as_longlong(want)));
    atomicAdd(&out[1], (unsigned long long)(__double_as_longlong(bare) != __double_as_longlong(want)));
    atomicAdd(&out[2], (unsigned long long)(__double_as_longlong(fast) != __double_as_longlong(want)));
    atomicAdd(&out[3], (unsigned long long)((unsigned)__double_as_longlong(c) != 0xffffffffu || !significand_all_ones(c)));
    atomicAdd(&out[4], 1ull);
}

}  // namespace nbk
