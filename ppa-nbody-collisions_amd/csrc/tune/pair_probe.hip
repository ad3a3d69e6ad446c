// csrc/tune/pair_probe.hip -- development probe: cycles per pair of the exact per-pair VALU sequence the
// production kernel's fast path executes (copied from its ISA), without LDS, at 1/2/4 waves per SIMD.
// Variants: SERIAL = one pair after another (what hipcc emits), INTER2 = two pairs interleaved by hand.
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float float2_ __attribute__((ext_vector_type(2)));

// one pair: inputs rec (x,y in a pair register, m in a scalar reg), body (xi,yi pair); in/out F pair, flag sgpr
#define PAIR(R, M, T0, T1, T2, T3)                                                            \
    "v_pk_add_f32 " T0 ", " R ", %[pi] neg_lo:[0,1] neg_hi:[0,1]\n\t"                          \
    "v_pk_mul_f32 " T1 ", " T0 ", " T0 "\n\t"                                                  \
    "s_nop 0\n\t"                                                                             \
    "v_add_f32 %[d2], " T1 "\n\t"

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* cyc, float* out, int trips) {
    float2_ rec = {1.0f + threadIdx.x * 0.001f, 2.0f}, pi = {0.5f, 0.25f}, F = {0.f, 0.f};
    float2_ recb = {1.37f + threadIdx.x * 0.001f, 2.0f};
    float m = 3.0f;
    const float lo = 0x1p-80f;
    unsigned long long flagacc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < trips; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float2_ d, sq, tt;
            float d2, y, g, h, e, dd, c, r;
            if (MODE == 0) {   // the production fast path's C++ (hipcc emits the packed serial chain, 17 VALU)
                asm volatile("" : "+v"(rec), "+v"(m));
                const float dx = rec.x - pi.x, dy = rec.y - pi.y;
                const float dd2 = (dx * dx) + (dy * dy);
                flagacc |= __builtin_amdgcn_fcmpf(dd2, lo, 5);
                const float yy = __builtin_amdgcn_rsqf(dd2);
                const float gg = dd2 * yy, hh = 0.5f * yy;
                const float ee = __builtin_fmaf(-gg, gg, dd2);
                const float dsq = __builtin_fmaf(ee, hh, gg);
                const float cc = (dsq * dsq) * dsq;
                const float rr = __builtin_amdgcn_rcpf(cc);
                const float e2 = __builtin_fmaf(-cc, rr, 1.0f);
                const float inv = __builtin_fmaf(e2, rr, rr);
                F.x = F.x + inv * (m * dx);
                F.y = F.y + inv * (m * dy);
                (void)d; (void)sq; (void)tt; (void)d2; (void)y; (void)g; (void)h; (void)e; (void)dd; (void)c; (void)r;
            } else if (MODE == 2) {   // two pairs at a time: (x,y) packed within a pair, the d2 -> inv chain packed ACROSS the pairs
                if (u & 1) continue;
                asm volatile("" : "+v"(rec), "+v"(recb), "+v"(m));
                const float2_ da = rec - pi, db = recb - pi;
                const float2_ sa = da * da, sb = db * db;
                float2_ q2;
                asm("v_add_f32 %0, %1, %2" : "=v"(q2.x) : "v"(sa.x), "v"(sa.y));   // as add_unmerged() in the kernel
                asm("v_add_f32 %0, %1, %2" : "=v"(q2.y) : "v"(sb.x), "v"(sb.y));
                flagacc |= __builtin_amdgcn_fcmpf(q2.x, lo, 5);
                flagacc |= __builtin_amdgcn_fcmpf(q2.y, lo, 5);
                float2_ yy; yy.x = __builtin_amdgcn_rsqf(q2.x); yy.y = __builtin_amdgcn_rsqf(q2.y);
                const float2_ gg = q2 * yy, hh = yy * 0.5f;
                const float2_ ee = __builtin_elementwise_fma(-gg, gg, q2);
                const float2_ ds = __builtin_elementwise_fma(ee, hh, gg);
                const float2_ cc = (ds * ds) * ds;
                float2_ rr; rr.x = __builtin_amdgcn_rcpf(cc.x); rr.y = __builtin_amdgcn_rcpf(cc.y);
                const float2_ one = {1.0f, 1.0f};
                const float2_ e2 = __builtin_elementwise_fma(-cc, rr, one);
                const float2_ inv = __builtin_elementwise_fma(e2, rr, rr);
                F = F + (da * m) * inv.x;
                F = F + (db * m) * inv.y;
                (void)d; (void)sq; (void)tt; (void)d2; (void)y; (void)g; (void)h; (void)e; (void)dd; (void)c; (void)r;
            } else if (MODE == 3) {   // as MODE 2, reciprocal from the rsq seed: y^3, one Newton step, one correction (no v_rcp)
                if (u & 1) continue;
                asm volatile("" : "+v"(rec), "+v"(recb), "+v"(m));
                const float2_ da = rec - pi, db = recb - pi;
                const float2_ sa = da * da, sb = db * db;
                float2_ q2;
                asm("v_add_f32 %0, %1, %2" : "=v"(q2.x) : "v"(sa.x), "v"(sa.y));
                asm("v_add_f32 %0, %1, %2" : "=v"(q2.y) : "v"(sb.x), "v"(sb.y));
                flagacc |= __builtin_amdgcn_fcmpf(q2.x, lo, 5);
                flagacc |= __builtin_amdgcn_fcmpf(q2.y, lo, 5);
                float2_ yy; yy.x = __builtin_amdgcn_rsqf(q2.x); yy.y = __builtin_amdgcn_rsqf(q2.y);
                const float2_ gg = q2 * yy, hh = yy * 0.5f;
                const float2_ ee = __builtin_elementwise_fma(-gg, gg, q2);
                const float2_ ds = __builtin_elementwise_fma(ee, hh, gg);
                const float2_ cc = (ds * ds) * ds;
                const float2_ r0 = (yy * yy) * yy;
                const float2_ one = {1.0f, 1.0f};
                const float2_ e0 = __builtin_elementwise_fma(-cc, r0, one);
                const float2_ r1 = __builtin_elementwise_fma(e0, r0, r0);
                const float2_ e1 = __builtin_elementwise_fma(-cc, r1, one);
                const float2_ inv = __builtin_elementwise_fma(e1, r1, r1);
                F = F + (da * m) * inv.x;
                F = F + (db * m) * inv.y;
                (void)d; (void)sq; (void)tt; (void)d2; (void)y; (void)g; (void)h; (void)e; (void)dd; (void)c; (void)r;
            } else if (MODE == 4 || MODE == 5) {
                // The round-3 ring kernel's sequence: four walk positions per batch, EVERY instruction packed over two
                // positions (component-major operands), two packed chains side by side, 64 scalar chain adds per 32
                // positions; MODE 5 adds the v_min3 screen of the non-zero-radius case.  8.5 packed + 2 transcendental
                // + 2 adds per pair.
                if (u & 3) continue;
                float2_ xa = {rec.x, rec.x + 0.5f}, xb = {recb.x, recb.x + 0.25f};
                float2_ ya = {rec.y, rec.y + 0.5f}, yb = {recb.y, recb.y - 0.25f};
                float2_ ma = {m, m + 1.0f}, mb = {m + 2.0f, m + 3.0f};
                asm volatile("" : "+v"(xa), "+v"(xb), "+v"(ya), "+v"(yb), "+v"(ma), "+v"(mb));
                const float2_ ownx = {pi.x, pi.x}, owny = {pi.y, pi.y};
                const float2_ dxa = xa - ownx, dxb = xb - ownx, dya = ya - owny, dyb = yb - owny;
                const float2_ sxa = dxa * dxa, sxb = dxb * dxb, sya = dya * dya, syb = dyb * dyb;
                const float2_ d2a = sxa + sya, d2b = sxb + syb;
                if (MODE == 5) {
                    float closest = m;
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(closest) : "v"(closest), "v"(d2a.x), "v"(d2a.y));
                    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(closest) : "v"(closest), "v"(d2b.x), "v"(d2b.y));
                    m = closest * 0.0f + m;
                }
                float2_ ya_, yb_;
                ya_.x = __builtin_amdgcn_rsqf(d2a.x); ya_.y = __builtin_amdgcn_rsqf(d2a.y);
                yb_.x = __builtin_amdgcn_rsqf(d2b.x); yb_.y = __builtin_amdgcn_rsqf(d2b.y);
                const float2_ ga = d2a * ya_, gb = d2b * yb_, ha = ya_ * 0.5f, hb = yb_ * 0.5f;
                const float2_ ea = __builtin_elementwise_fma(-ga, ga, d2a), eb = __builtin_elementwise_fma(-gb, gb, d2b);
                const float2_ da = __builtin_elementwise_fma(ea, ha, ga), db = __builtin_elementwise_fma(eb, hb, gb);
                const float2_ ca = (da * da) * da, cb = (db * db) * db;
                float2_ ra, rb;
                ra.x = __builtin_amdgcn_rcpf(ca.x); ra.y = __builtin_amdgcn_rcpf(ca.y);
                rb.x = __builtin_amdgcn_rcpf(cb.x); rb.y = __builtin_amdgcn_rcpf(cb.y);
                const float2_ one = {1.0f, 1.0f};
                const float2_ fa = __builtin_elementwise_fma(-ca, ra, one), fb = __builtin_elementwise_fma(-cb, rb, one);
                const float2_ inva = __builtin_elementwise_fma(fa, ra, ra), invb = __builtin_elementwise_fma(fb, rb, rb);
                const float2_ txa = (dxa * ma) * inva, txb = (dxb * mb) * invb, tya = (dya * ma) * inva, tyb = (dyb * mb) * invb;
                F.x = F.x + txa.x; asm("" : "+v"(F.x)); F.y = F.y + tya.x;
                F.x = F.x + txa.y; asm("" : "+v"(F.x)); F.y = F.y + tya.y;
                F.x = F.x + txb.x; asm("" : "+v"(F.x)); F.y = F.y + tyb.x;
                F.x = F.x + txb.y; asm("" : "+v"(F.x)); F.y = F.y + tyb.y;
                rec.x += 3e-3f; recb.x += 4e-3f;
                (void)d; (void)sq; (void)tt; (void)d2; (void)y; (void)g; (void)h; (void)e; (void)dd; (void)c; (void)r;
            } else {           // same work with scalar (non-packed) ops only: 22 VALU
                float dx, dy, a2, b2, mx, my, tx, ty;
                asm volatile(
                    "v_sub_f32 %[dx], %[rx], %[px]\n\t"
                    "v_sub_f32 %[dy], %[ry], %[py]\n\t"
                    "v_mul_f32 %[a2], %[dx], %[dx]\n\t"
                    "v_mul_f32 %[b2], %[dy], %[dy]\n\t"
                    "v_add_f32 %[d2], %[a2], %[b2]\n\t"
                    "v_rsq_f32 %[y], %[d2]\n\t"
                    "v_cmp_ge_f32 vcc, %[lo], %[d2]\n\t"
                    "s_or_b64 s[10:11], vcc, s[10:11]\n\t"
                    "v_mul_f32 %[g], %[d2], %[y]\n\t"
                    "v_mul_f32 %[h], 0.5, %[y]\n\t"
                    "v_fma_f32 %[e], -%[g], %[g], %[d2]\n\t"
                    "v_fmac_f32 %[g], %[e], %[h]\n\t"
                    "v_mul_f32 %[dd], %[g], %[g]\n\t"
                    "v_mul_f32 %[c], %[g], %[dd]\n\t"
                    "v_rcp_f32 %[r], %[c]\n\t"
                    "v_mul_f32 %[mx], %[m], %[dx]\n\t"
                    "v_mul_f32 %[my], %[m], %[dy]\n\t"
                    "v_fma_f32 %[e], -%[c], %[r], 1.0\n\t"
                    "v_fmac_f32 %[r], %[e], %[r]\n\t"
                    "v_mul_f32 %[tx], %[r], %[mx]\n\t"
                    "v_mul_f32 %[ty], %[r], %[my]\n\t"
                    "v_add_f32 %[Fx], %[Fx], %[tx]\n\t"
                    "v_add_f32 %[Fy], %[Fy], %[ty]\n\t"
                    : [dx] "=&v"(dx), [dy] "=&v"(dy), [a2] "=&v"(a2), [b2] "=&v"(b2), [d2] "=&v"(d2), [y] "=&v"(y),
                      [g] "=&v"(g), [h] "=&v"(h), [e] "=&v"(e), [dd] "=&v"(dd), [c] "=&v"(c), [r] "=&v"(r),
                      [mx] "=&v"(mx), [my] "=&v"(my), [tx] "=&v"(tx), [ty] "=&v"(ty), [Fx] "+v"(F.x), [Fy] "+v"(F.y)
                    : [rx] "v"(rec.x), [ry] "v"(rec.y), [px] "v"(pi.x), [py] "v"(pi.y), [lo] "v"(lo), [m] "v"(m)
                    : "vcc", "s10", "s11");
            }
            rec.x += 1e-3f;
            if (MODE == 2 || MODE == 3) recb.x += 1e-3f;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (F.x + F.y == 123.456f || flagacc == 12345ull) out[0] = F.x;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * 256 + threadIdx.x) / 64] = t1 - t0;
}

template <int MODE>
int run(const char* name, unsigned long long* d_cyc, float* d_out) {
    const int trips = 20000;
    for (int wps : {1, 2, 4, 6}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_out, 100);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_out, trips);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double pairs = 8.0 * trips;
        printf("%-22s waves/SIMD=%d  %.2f ms  cycles/pair(wave) %.1f  cycles/pair/SIMD %.1f  -> %.3e pairs/s chip-wide\n", name, wps, ms,
               h[h.size() / 2] / pairs, h[h.size() / 2] / pairs / wps, pairs * 64 * blocks * 4 / (ms * 1e-3));
    }
    return 0;
}
int main() {
    unsigned long long* d_cyc; float* d_out;
    CK(hipMalloc((void**)&d_cyc, 8 * 8192)); CK(hipMalloc((void**)&d_out, 64));
    run<0>("packed, serial chain", d_cyc, d_out);
    run<1>("scalar ops only", d_cyc, d_out);
    run<2>("chain packed across 2", d_cyc, d_out);
    run<3>("... and no v_rcp", d_cyc, d_out);
    run<4>("r03 ring sequence", d_cyc, d_out);
    run<5>("r03 ring seq + v_min3", d_cyc, d_out);
    return 0;
}
