// csrc/tune/issue_probe_f64.hip -- development probe: saturated issue cost (4 waves per SIMD, every CU busy) of the fp64
// instructions the fp64 force kernel is made of, and of the conversions an fp32-seeded square root would add.
//   hipcc -O3 --offload-arch=gfx950 -o issue_probe_f64 issue_probe_f64.hip && ./issue_probe_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
enum Kind { FMA64, MUL64, ADD64, RSQ64, RCP64, CVT_F32_F64, CVT_F64_F32, RSQ32, CMP64, NK };
static const char* names[NK] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64", "v_rcp_f64", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_rsq_f32", "v_cmp_le_f64"};
template <int K>
__global__ __launch_bounds__(256) void probe(double* out, int trips) {
    double a[8]; float f[8];
    for (int k = 0; k < 8; ++k) { a[k] = 1.0 + 0.001 * (k + threadIdx.x); f[k] = 1.0f + 0.01f * k; }
    const double m = 1.0000001, b = 1e-7;
    for (int i = 0; i < trips; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (K == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(b));
                if (K == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                if (K == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                if (K == RSQ64) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[k]));
                if (K == RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
                if (K == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[k]) : "v"(a[k]));
                if (K == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f[k]));
                if (K == RSQ32) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[k]));
                if (K == CMP64) asm volatile("v_cmp_le_f64 vcc, %0, %1" :: "v"(a[k]), "v"(b) : "vcc");
            }
        }
    }
    double s = 0; for (int k = 0; k < 8; ++k) s += a[k] + f[k];
    if (s == 123.456) out[0] = s;
}
template <int K> int run(double* d) {
    const int trips = 4000, blocks = 256 * 4;          // 4 waves per SIMD
    hipLaunchKernelGGL(probe<K>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<K>, dim3(blocks), dim3(256), 0, 0, d, trips);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)trips * 32 * 4;     // 4 waves per SIMD
    printf("%-16s %.3f ns per wave-instruction per SIMD (%.1f cycles at 2.3 GHz)\n", names[K], ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.3);
    return 0;
}
int main() {
    double* d; CK(hipMalloc((void**)&d, 64));
    run<FMA64>(d); run<MUL64>(d); run<ADD64>(d); run<RSQ64>(d); run<RCP64>(d); run<CVT_F32_F64>(d); run<CVT_F64_F32>(d); run<RSQ32>(d); run<CMP64>(d);
    return 0;
}
