// csrc/tune/placement_probe.hip -- development probe: where the dispatcher puts the waves of a grid.
// Every wave records its HW_ID (SIMD, CU, SE) and XCC_ID, then spins so that the whole grid is resident at once;
// the host prints, per workgroup shape, the histogram of "waves on the busiest / idlest SIMD of a CU".
//   hipcc --offload-arch=gfx950 -O2 -o placement_probe placement_probe.hip && ./placement_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int THREADS, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) void probe(unsigned* out, int spin) {
    __shared__ char pad[LDS_BYTES];
    pad[threadIdx.x] = (char)threadIdx.x;
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * THREADS + threadIdx.x) / 64;
        out[2 * w] = hw;
        out[2 * w + 1] = xcc + (unsigned)pad[threadIdx.x & 1] * 0;
    }
}

template <int THREADS, int LDS_BYTES>
int run(int grid, const char* what) {
    const int waves = grid * THREADS / 64;
    unsigned* d;
    CK(hipMalloc(&d, sizeof(unsigned) * 2 * waves));
    hipLaunchKernelGGL((probe<THREADS, LDS_BYTES>), dim3(grid), dim3(THREADS), 0, 0, d, 20000000);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * waves);
    CK(hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * waves, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<int>> per_cu;       // key: xcc, se, sh, cu -> waves per simd
    for (int w = 0; w < waves; ++w) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
        auto& v = per_cu[key];
        if (v.empty()) v.assign(4, 0);
        v[simd]++;
    }
    std::map<std::vector<int>, int> shapes;
    for (auto& kv : per_cu) {
        std::vector<int> v = kv.second;
        std::sort(v.begin(), v.end(), std::greater<int>());
        shapes[v]++;
    }
    printf("%s: grid %d x %d threads, %d waves on %zu CUs; waves per SIMD (sorted) -> number of CUs:\n", what, grid,
           THREADS, waves, per_cu.size());
    for (auto& kv : shapes) printf("    %d-%d-%d-%d : %d\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3], kv.second);
    CK(hipFree(d));
    return 0;
}

int main() {
    if (run<128, 8448>(2048, "v3 K=1, G=1 shape")) return 1;
    if (run<128, 8448>(1024, "v3 K=1, G=2 shape")) return 1;
    if (run<128, 8448>(512, "v3 K=1, G=4 shape")) return 1;
    if (run<256, 8448>(512, "256-thread WGs, 2048 waves")) return 1;
    if (run<256, 8448>(256, "256-thread WGs, 1024 waves")) return 1;
    if (run<512, 73880>(512, "8-wave workgroups, G=8")) return 1;
    if (run<512, 73880>(1024, "8-wave workgroups, G=4")) return 1;
    return 0;
}
