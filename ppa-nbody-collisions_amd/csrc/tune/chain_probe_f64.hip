// csrc/tune/chain_probe_f64.hip -- development probe: the fp64 fast chain (hardware rsq/rcp seeds + the Newton /
// correction steps of the compiler's own correctly-rounded expansions, without their range scaling and special-case
// fix-ups) against the compiler's IEEE sqrt and 1/x, on random and structured inputs of the guarded domain.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o chain_probe_f64 chain_probe_f64.hip && ./chain_probe_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Chain { double d, inv; };
__device__ __forceinline__ Chain fast_chain(double x) {
    // sqrt: y ~ 1/sqrt(x); g -> sqrt(x), h -> 1/(2 sqrt(x)), two coupled Newton steps and two corrections
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
    // 1/c: two Newton steps on the rcp seed, then the residual correction
    const double q0 = __builtin_amdgcn_rcp(c);
    const double e0 = __builtin_fma(-c, q0, 1.0);
    const double q1 = __builtin_fma(q0, e0, q0);
    const double e1 = __builtin_fma(-c, q1, 1.0);
    const double q2 = __builtin_fma(q1, e1, q1);
    const double e2 = __builtin_fma(-c, q2, 1.0);
    return Chain{d, __builtin_fma(e2, q2, q2)};
}

__device__ __forceinline__ uint64_t splitmix(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// mode 0: random mantissa, exponent uniform in [-500, 500]; mode 1: mantissas within 2^12 ulps of a power of two
// or of all-ones; mode 2: perfect squares / cubes neighbourhoods (d2 = k*k +- few ulps)
__global__ void probe(unsigned long long* bad, int mode, int iters, unsigned long long seed) {
    uint64_t s = seed + 0x1234567ull * (blockIdx.x * blockDim.x + threadIdx.x);
    unsigned long long bs = 0, bi = 0;
    for (int it = 0; it < iters; ++it) {
        const uint64_t r = splitmix(s), r2 = splitmix(s);
        const int e = (int)(r2 % 1001) - 500;
        uint64_t man = r & 0xfffffffffffffull;
        if (mode == 1) man = (r & 1) ? (r >> 1) & 0xfff : 0xfffffffffffffull - ((r >> 1) & 0xfff);
        double x = __longlong_as_double(((uint64_t)(e + 1023) << 52) | man);
        if (mode == 2) {
            const double k = (double)((r >> 20) | 1);
            const double sq = k * k;
            x = __longlong_as_double(__double_as_longlong(sq) + (long long)(r2 % 9) - 4);
        }
        const Chain f = fast_chain(x);
        const double d = __builtin_sqrt(x);
        const double c = (d * d) * d;
        const double inv = 1.0 / c;
        bs += __double_as_longlong(f.d) != __double_as_longlong(d);
        bi += __double_as_longlong(f.inv) != __double_as_longlong(inv);
    }
    if (bs) atomicAdd(&bad[0], bs);
    if (bi) atomicAdd(&bad[1], bi);
}

int main() {
    unsigned long long* d;
    CK(hipMalloc(&d, 16));
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(d, 0, 16));
        const int blocks = 4096, threads = 256, iters = 4096;
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, mode, iters, 0xabcdefull + mode);
        CK(hipDeviceSynchronize());
        unsigned long long h[2];
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("mode %d: %llu inputs, sqrt mismatches %llu, 1/d^3 mismatches %llu\n", mode,
               (unsigned long long)blocks * threads * iters, h[0], h[1]);
    }
    return 0;
}
