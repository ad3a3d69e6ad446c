// csrc/tune/lds_add_probe.hip -- development probe (not part of the library): can the LDS do the ordered chain?
//   (a) exactness: ds_add_f32 against v_add_f32 on random and structured operand pairs (bitwise, NaN == NaN);
//   (b) chains: 32 dependent ds_add_f32 on one address against 32 dependent v_add_f32;
//   (c) throughput: ds_add_f32 instructions per CU-cycle with 16 waves per CU, alone and mixed with ds_read2_b32.
//   hipcc -O3 --offload-arch=gfx950 -o lds_add_probe lds_add_probe.hip && ./lds_add_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long& s) {
    unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float operand(unsigned long long r, int mode) {
    unsigned u = (unsigned)r;
    if (mode == 0) return __uint_as_float(u);                                  // any bit pattern
    if (mode == 1) return __uint_as_float((u & 0x807fffffu) | ((100u + (unsigned)((r >> 40) % 60)) << 23));   // normal, close exponents
    if (mode == 2) return __uint_as_float(u & 0x807fffffu);                    // denormals and zeros
    if (mode == 3) return __uint_as_float((u & 0x807fffffu) | ((1u + (unsigned)((r >> 40) % 3)) << 23));      // smallest normals
    return __uint_as_float((u & 0x807fffffu) | ((252u + (unsigned)((r >> 40) % 3)) << 23));                    // largest normals
}
__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

__global__ __launch_bounds__(256) void exact(unsigned long long* bad, int mode, int iters) {
    __shared__ float acc[256];
    unsigned long long s = 0x1234567ull * (blockIdx.x * 256 + threadIdx.x) + mode;
    const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)&acc[threadIdx.x];
    unsigned long long nbad = 0, nbad_chain = 0;
    for (int it = 0; it < iters; ++it) {
        const float a = operand(mix(s), mode), b = operand(mix(s), mode == 1 ? 1 : (it & 1 ? mode : 1));
        float want;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(want) : "v"(a), "v"(b));
        acc[threadIdx.x] = a;
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_f32 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(b) : "memory");
        const float got = *(volatile float*)&acc[threadIdx.x];
        nbad += !same(got, want);
        // a chain of 32 adds in a row
        float t[32], w = a;
        for (int k = 0; k < 32; ++k) { t[k] = operand(mix(s), 1) * (k & 1 ? 1.0f : -1.0f); asm volatile("v_add_f32 %0, %0, %1" : "+v"(w) : "v"(t[k])); }
        acc[threadIdx.x] = a;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int k = 0; k < 32; ++k) asm volatile("ds_add_f32 %0, %1" :: "v"(addr), "v"(t[k]) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        nbad_chain += !same(*(volatile float*)&acc[threadIdx.x], w);
    }
    if (nbad) atomicAdd(&bad[0], nbad);
    if (nbad_chain) atomicAdd(&bad[1], nbad_chain);
}

template <int MIX>
__global__ __launch_bounds__(1024) void rate(unsigned long long* out, int trips) {
    __shared__ float acc[1024 * 2];
    acc[threadIdx.x] = 0.f; acc[threadIdx.x + 1024] = 1.f;
    __syncthreads();
    const unsigned addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)&acc[threadIdx.x];
    float v = 1.0f + threadIdx.x;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r = {0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < trips; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            asm volatile("ds_add_f32 %0, %1" :: "v"(addr), "v"(v) : "memory");
            if (MIX == 1) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(r) : "v"(addr) : "memory");
            if (MIX == 2) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(v));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(r.x + v); }
}

int main() {
    unsigned long long* d;
    CK(hipMalloc((void**)&d, 16));
    const char* names[] = {"any bit pattern", "normal, close exponents", "denormals / zeros", "smallest normals", "largest normals"};
    for (int mode = 0; mode < 5; ++mode) {
        CK(hipMemset(d, 0, 16));
        hipLaunchKernelGGL(exact, dim3(1024), dim3(256), 0, 0, d, mode, 2000);
        unsigned long long h[2];
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("exactness, %-24s: %llu single adds and %llu chains of 32 differ from v_add_f32 (of %d each)\n", names[mode], h[0], h[1], 1024 * 256 * 2000);
    }
    const int trips = 2000;
    for (int mixk = 0; mixk < 3; ++mixk) {
        CK(hipMemset(d, 0, 16));
        if (mixk == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(1024), 0, 0, d, trips);
        if (mixk == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(1024), 0, 0, d, trips);
        if (mixk == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(1024), 0, 0, d, trips);
        unsigned long long h[2];
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        const double per = (double)h[0] / (trips * 16.0);
        printf("rate, 16 waves per CU, %s: %.1f cycles per ds_add_f32 as seen by one wave -> %.2f CU-cycles per wave-instruction\n",
               mixk == 0 ? "ds_add_f32 alone" : mixk == 1 ? "1 ds_add_f32 : 1 ds_read2_b32" : "1 ds_add_f32 : 4 v_fma_f32", per, per / 16.0);
    }
    return 0;
}
