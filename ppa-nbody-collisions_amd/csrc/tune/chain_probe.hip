// csrc/tune/chain_probe.hip -- development probe (not part of the library): which cheap fp32 evaluation
// chains for   d = RN(sqrt(a)),  c = RN(RN(d*d)*d),  inv = RN(1/c)   are bit-identical to the compiler's
// correctly-rounded sqrt / divide for EVERY fp32 input a, and what the gfx950 issue rates of the
// instructions involved are.  Results are recorded in DESIGN.md; the shipped self-test
// (nbody_selftest_ieee_f32) re-proves the chain that the kernels use.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o chain_probe chain_probe.hip && ./chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#pragma clang fp contract(off)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

struct Chain { float d, inv; };

__device__ __forceinline__ Chain ref_chain(float a) {
    Chain r;
    r.d = __builtin_sqrtf(a);
    const float c = (r.d * r.d) * r.d;
    r.inv = 1.0f / c;
    return r;
}

// sqrt variants ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sqrt_S1(float a, float y) {      // 2 mul + 2 fma
    const float g = a * y, h = 0.5f * y;
    const float e = fma_(-g, g, a);
    return fma_(e, h, g);
}
__device__ __forceinline__ float sqrt_S2(float a, float y, float* h_out) {   // 2 mul + 5 fma
    const float g = a * y, h = 0.5f * y;
    const float r = fma_(-g, h, 0.5f);
    const float g1 = fma_(g, r, g), h1 = fma_(h, r, h);
    const float e = fma_(-g1, g1, a);
    *h_out = h1;
    return fma_(e, h1, g1);
}
// S3: S1 followed by one more exact-residual correction (2 mul + 4 fma)
__device__ __forceinline__ float sqrt_S3(float a, float y) {
    const float g = a * y, h = 0.5f * y;
    const float e = fma_(-g, g, a);
    const float g1 = fma_(e, h, g);
    const float e1 = fma_(-g1, g1, a);
    return fma_(e1, h, g1);
}
// reciprocal variants ------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_N(float c, float y0, int iters) {
    float y = y0;
    for (int k = 0; k < iters; ++k) {
        const float e = fma_(-c, y, 1.0f);
        y = fma_(e, y, y);
    }
    return y;
}

constexpr int NV = 12;
__global__ __launch_bounds__(256) void probe(unsigned long long* bad, unsigned* first_bad, float lo, float hi) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    unsigned long long cnt[NV] = {0};
    for (unsigned long long u = gid; u < (1ull << 31); u += stride) {   // non-negative inputs
        const float a = __uint_as_float((unsigned)u);
        if (!(a >= lo && a <= hi)) continue;
        const Chain R = ref_chain(a);
        const float y = __builtin_amdgcn_rsqf(a);
        float h1;
        const float d1 = sqrt_S1(a, y);
        const float d2 = sqrt_S2(a, y, &h1);
        const float d3 = sqrt_S3(a, y);
        const float dq = __builtin_amdgcn_sqrtf(a);          // raw v_sqrt_f32
        // sqrt correctness
        cnt[0] += __float_as_uint(d1) != __float_as_uint(R.d);
        cnt[1] += __float_as_uint(d2) != __float_as_uint(R.d);
        cnt[2] += __float_as_uint(d3) != __float_as_uint(R.d);
        cnt[3] += __float_as_uint(dq) != __float_as_uint(R.d);
        // reciprocal of c (c from the exact d)
        const float c = (R.d * R.d) * R.d;
        const float y3 = (y * y) * y;
        const float yr = 2.0f * h1;                          // refined 1/sqrt(a)
        const float y3r = (yr * yr) * yr;
        const float r0 = __builtin_amdgcn_rcpf(c);
        cnt[4] += __float_as_uint(rcp_N(c, y3, 1)) != __float_as_uint(R.inv);
        cnt[5] += __float_as_uint(rcp_N(c, y3, 2)) != __float_as_uint(R.inv);
        cnt[6] += __float_as_uint(rcp_N(c, y3r, 1)) != __float_as_uint(R.inv);
        cnt[7] += __float_as_uint(rcp_N(c, y3r, 2)) != __float_as_uint(R.inv);
        cnt[8] += __float_as_uint(rcp_N(c, r0, 1)) != __float_as_uint(R.inv);
        cnt[9] += __float_as_uint(rcp_N(c, r0, 2)) != __float_as_uint(R.inv);
        cnt[10] += __float_as_uint(r0) != __float_as_uint(R.inv);
        // y3 variant where d itself came from S1 (full fast chain S1 + y3 + 2 iterations)
        {
            const float cf = (d1 * d1) * d1;
            const float inv = rcp_N(cf, y3, 2);
            const bool ok = __float_as_uint(d1) == __float_as_uint(R.d) && __float_as_uint(inv) == __float_as_uint(R.inv);
            cnt[11] += !ok;
            if (!ok) atomicMin(first_bad, (unsigned)u);
        }
    }
    for (int k = 0; k < NV; ++k) if (cnt[k]) atomicAdd(&bad[k], cnt[k]);
}

// throughput probes: one wave-instruction stream of independent ops, W waves per SIMD ----------------------
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void tput(float* out, int iters) {
    float2_ a0 = {1.0f + threadIdx.x, 2.0f}, a1 = {3.0f, 4.0f + threadIdx.x}, a2 = {5.0f, 6.0f}, a3 = {7.0f, 8.0f};
    float2_ a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
    const float2_ m = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // 8 independent v_pk_fma_f32
            a0 = __builtin_elementwise_fma(a0, m, b); a1 = __builtin_elementwise_fma(a1, m, b);
            a2 = __builtin_elementwise_fma(a2, m, b); a3 = __builtin_elementwise_fma(a3, m, b);
            a4 = __builtin_elementwise_fma(a4, m, b); a5 = __builtin_elementwise_fma(a5, m, b);
            a6 = __builtin_elementwise_fma(a6, m, b); a7 = __builtin_elementwise_fma(a7, m, b);
        } else if (KIND == 1) {   // 8 independent v_fma_f32
            a0.x = fma_(a0.x, m.x, b.x); a1.x = fma_(a1.x, m.x, b.x); a2.x = fma_(a2.x, m.x, b.x); a3.x = fma_(a3.x, m.x, b.x);
            a4.x = fma_(a4.x, m.x, b.x); a5.x = fma_(a5.x, m.x, b.x); a6.x = fma_(a6.x, m.x, b.x); a7.x = fma_(a7.x, m.x, b.x);
        } else if (KIND == 2) {   // 8 independent v_rsq_f32
            a0.x = __builtin_amdgcn_rsqf(a0.x); a1.x = __builtin_amdgcn_rsqf(a1.x); a2.x = __builtin_amdgcn_rsqf(a2.x); a3.x = __builtin_amdgcn_rsqf(a3.x);
            a4.x = __builtin_amdgcn_rsqf(a4.x); a5.x = __builtin_amdgcn_rsqf(a5.x); a6.x = __builtin_amdgcn_rsqf(a6.x); a7.x = __builtin_amdgcn_rsqf(a7.x);
        } else if (KIND == 3) {   // 2 rsq + 6 pk_fma : is the transcendental unit co-issued with the main VALU?
            a0.x = __builtin_amdgcn_rsqf(a0.x); a1.x = __builtin_amdgcn_rsqf(a1.x);
            a2 = __builtin_elementwise_fma(a2, m, b); a3 = __builtin_elementwise_fma(a3, m, b);
            a4 = __builtin_elementwise_fma(a4, m, b); a5 = __builtin_elementwise_fma(a5, m, b);
            a6 = __builtin_elementwise_fma(a6, m, b); a7 = __builtin_elementwise_fma(a7, m, b);
        } else if (KIND == 4) {   // 8 independent v_pk_mul_f32
            a0 = a0 * m; a1 = a1 * m; a2 = a2 * m; a3 = a3 * m; a4 = a4 * m; a5 = a5 * m; a6 = a6 * m; a7 = a7 * m;
        } else if (KIND == 5) {   // 8 independent v_pk_add_f32
            a0 = a0 + b; a1 = a1 + b; a2 = a2 + b; a3 = a3 + b; a4 = a4 + b; a5 = a5 + b; a6 = a6 + b; a7 = a7 + b;
        }
    }
    const float2_ s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 123.456f) out[0] = s.x;
}

template <int KIND>
int run_tput(const char* name, int ops_per_iter, float* d_out) {
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {   // waves per SIMD: 256-thread blocks = 1 wave on each of the 4 SIMDs
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(tput<KIND>, dim3(256 * wps), dim3(256), 0, 0, d_out, 100);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(tput<KIND>, dim3(256 * wps), dim3(256), 0, 0, d_out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double winstr = (double)iters * ops_per_iter * wps;      // wave-instructions per SIMD
        printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / winstr, ms * 1e6 / winstr * 2.4);
    }
    return 0;
}

int main() {
    unsigned long long* d_bad; unsigned* d_first;
    CK(hipMalloc((void**)&d_bad, NV * sizeof(unsigned long long)));
    CK(hipMalloc((void**)&d_first, sizeof(unsigned)));
    const char* names[NV] = {"sqrt S1 (2mul+2fma)", "sqrt S2 (2mul+5fma)", "sqrt S3 (2mul+4fma)", "raw v_sqrt_f32",
                             "rcp y^3 + 1 iter", "rcp y^3 + 2 iter", "rcp yref^3 + 1 iter", "rcp yref^3 + 2 iter",
                             "rcp v_rcp + 1 iter", "rcp v_rcp + 2 iter", "raw v_rcp_f32", "FULL S1 + y^3 + 2 iter"};
    const float ranges[][2] = {{0x1p-80f, 0x1p80f}, {0x1p-126f, 0x1p-80f}, {0x1p80f, 0x1p127f}, {0.0f, 0x1p-126f}};
    for (auto& r : ranges) {
        CK(hipMemset(d_bad, 0, NV * sizeof(unsigned long long)));
        CK(hipMemset(d_first, 0xff, sizeof(unsigned)));
        hipLaunchKernelGGL(probe, dim3(256 * 32), dim3(256), 0, 0, d_bad, d_first, r[0], r[1]);
        CK(hipDeviceSynchronize());
        unsigned long long h[NV]; unsigned first;
        CK(hipMemcpy(h, d_bad, sizeof(h), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&first, d_first, sizeof(first), hipMemcpyDeviceToHost));
        printf("== a in [%a, %a]\n", r[0], r[1]);
        for (int k = 0; k < NV; ++k) printf("   %-26s mismatches %llu\n", names[k], h[k]);
        printf("   first failing input of FULL chain: 0x%08x\n", first);
    }
    float* d_out; CK(hipMalloc((void**)&d_out, 64));
    run_tput<0>("v_pk_fma_f32 x8", 8, d_out);
    run_tput<1>("v_fma_f32 x8", 8, d_out);
    run_tput<2>("v_rsq_f32 x8", 8, d_out);
    run_tput<3>("2 rsq + 6 pk_fma", 8, d_out);
    run_tput<4>("v_pk_mul_f32 x8", 8, d_out);
    run_tput<5>("v_pk_add_f32 x8", 8, d_out);
    return 0;
}
