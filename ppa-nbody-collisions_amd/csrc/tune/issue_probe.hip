// csrc/tune/issue_probe.hip -- development probe (not part of the library): gfx950 issue cost of the VALU
// instructions the force loop is made of, in shader cycles (s_memtime), at 1/2/4 waves per SIMD, every CU busy.
// Each wave runs `REP` copies of an 8-instruction independent block per loop trip, inline asm so the
// compiler cannot change the instruction.  Reported: cycles per wave-instruction as seen by one wave, and
// the same divided by waves/SIMD (= SIMD issue cycles per instruction when the SIMD is saturated).
//
//   hipcc -O3 --offload-arch=gfx950 -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float float2_ __attribute__((ext_vector_type(2)));
typedef float float4_ __attribute__((ext_vector_type(4)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { FMA, MUL, ADD, PK_FMA, PK_MUL, PK_ADD, RSQ, RCP, SQRT, MAX, MAX3, CMP, CNDMASK, MOV_DPP, ADD_DPP,
            MIX_RSQ_FMA, MIX_PKADD_FMA, DSREAD128_FMA, NKINDS };
static const char* kNames[NKINDS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32",
                                     "v_pk_add_f32", "v_rsq_f32", "v_rcp_f32", "v_sqrt_f32", "v_max_f32", "v_max3_f32",
                                     "v_cmp_le_f32 (vcc)", "v_cndmask_b32", "v_mov_b32 dpp row_shr:1", "v_add_f32 dpp row_shr:1",
                                     "1 rsq : 7 fma", "4 pk_add : 4 fma", "1 ds_read_b128 : 7 fma"};

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long* cyc, int trips) {
    __shared__ float4_ lds[256];
    lds[threadIdx.x] = float4_{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    float a[8];
    float2_ p[8];
    float4_ q = {0, 0, 0, 0};
    for (int k = 0; k < 8; ++k) { a[k] = 1.0f + 0.001f * (k + threadIdx.x); p[k] = float2_{a[k], a[k] + 0.5f}; }
    const float m = 1.0000001f, b = 1e-7f;
    const float2_ pm = {m, m}, pb = {b, b};
    const unsigned addr = (threadIdx.x & 63) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; ++i) {
#define X_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(b));
#define X_MUL(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
#define X_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define X_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pm), "v"(pb));
#define X_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pm));
#define X_PKADD(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
#define X_RSQ(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
#define X_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
#define X_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
#define X_MAX(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
#define X_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(m));
#define X_CMP(k) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");
#define X_CND(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
#define X_MOVDPP(k) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
#define X_ADDDPP(k) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(b));
        if (KIND == FMA) { R8(X_FMA) R8(X_FMA) R8(X_FMA) R8(X_FMA) }
        if (KIND == MUL) { R8(X_MUL) R8(X_MUL) R8(X_MUL) R8(X_MUL) }
        if (KIND == ADD) { R8(X_ADD) R8(X_ADD) R8(X_ADD) R8(X_ADD) }
        if (KIND == PK_FMA) { R8(X_PKFMA) R8(X_PKFMA) R8(X_PKFMA) R8(X_PKFMA) }
        if (KIND == PK_MUL) { R8(X_PKMUL) R8(X_PKMUL) R8(X_PKMUL) R8(X_PKMUL) }
        if (KIND == PK_ADD) { R8(X_PKADD) R8(X_PKADD) R8(X_PKADD) R8(X_PKADD) }
        if (KIND == RSQ) { R8(X_RSQ) R8(X_RSQ) R8(X_RSQ) R8(X_RSQ) }
        if (KIND == RCP) { R8(X_RCP) R8(X_RCP) R8(X_RCP) R8(X_RCP) }
        if (KIND == SQRT) { R8(X_SQRT) R8(X_SQRT) R8(X_SQRT) R8(X_SQRT) }
        if (KIND == MAX) { R8(X_MAX) R8(X_MAX) R8(X_MAX) R8(X_MAX) }
        if (KIND == MAX3) { R8(X_MAX3) R8(X_MAX3) R8(X_MAX3) R8(X_MAX3) }
        if (KIND == CMP) { R8(X_CMP) R8(X_CMP) R8(X_CMP) R8(X_CMP) }
        if (KIND == CNDMASK) { R8(X_CND) R8(X_CND) R8(X_CND) R8(X_CND) }
        if (KIND == MOV_DPP) { R8(X_MOVDPP) R8(X_MOVDPP) R8(X_MOVDPP) R8(X_MOVDPP) }
        if (KIND == ADD_DPP) { R8(X_ADDDPP) R8(X_ADDDPP) R8(X_ADDDPP) R8(X_ADDDPP) }
        if (KIND == MIX_RSQ_FMA) {
            for (int r = 0; r < 4; ++r) { X_RSQ(0) X_FMA(1) X_FMA(2) X_FMA(3) X_FMA(4) X_FMA(5) X_FMA(6) X_FMA(7) }
        }
        if (KIND == MIX_PKADD_FMA) {
            for (int r = 0; r < 4; ++r) { X_PKADD(0) X_FMA(1) X_PKADD(2) X_FMA(3) X_PKADD(4) X_FMA(5) X_PKADD(6) X_FMA(7) }
        }
        if (KIND == DSREAD128_FMA) {
            for (int r = 0; r < 4; ++r) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(addr));
                X_FMA(1) X_FMA(2) X_FMA(3) X_FMA(4) X_FMA(5) X_FMA(6) X_FMA(7)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                a[0] += q.x;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;
    if (s == 123.456f) lds[0].x = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) / 64] = t1 - t0;
    if (s == 123.456f) cyc[0] = (unsigned long long)lds[0].x;
}

template <int KIND>
int run(unsigned long long* d_cyc) {
    const int trips = 200000;     // 32 instr per trip -> 6.4 M instr per wave, tens of ms: clocks settle
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, 1000);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, trips);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double ninstr = 32.0 * trips;
        // s_memtime counts at a constant 100 MHz on this part if it is the "realtime" flavour; report both the raw
        // tick ratio and the wall-clock figure so the unit can be identified
        printf("%-26s w/SIMD=%d  wall %.2f ms  ticks/instr(wave) %.3f  ticks/instr/SIMD %.3f  ns/instr/SIMD %.3f\n",
               kNames[KIND], wps, ms, med / ninstr, med / ninstr / wps, ms * 1e6 / ninstr / wps);
    }
    return 0;
}

int main() {
    unsigned long long* d_cyc;
    CK(hipMalloc((void**)&d_cyc, 8 * 4096));
    run<FMA>(d_cyc); run<MUL>(d_cyc); run<ADD>(d_cyc); run<PK_FMA>(d_cyc); run<PK_MUL>(d_cyc); run<PK_ADD>(d_cyc);
    run<RSQ>(d_cyc); run<RCP>(d_cyc); run<SQRT>(d_cyc); run<MAX>(d_cyc); run<MAX3>(d_cyc); run<CMP>(d_cyc);
    run<CNDMASK>(d_cyc); run<MOV_DPP>(d_cyc); run<ADD_DPP>(d_cyc); run<MIX_RSQ_FMA>(d_cyc); run<MIX_PKADD_FMA>(d_cyc);
    run<DSREAD128_FMA>(d_cyc);
    return 0;
}
