// Phase clocks of the device Jonker-Volgenant solver (boxmot_amd/csrc/lap_jv.hpp) on one workgroup; development tool.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I boxmot_amd/csrc tools/jv_prof.hip -o tools/_build/jv_prof && tools/_build/jv_prof [n_rows n_cols]
// The matrix is DeepOCSORT's recovery round: mostly zeros (no overlap), a few negative IoUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long g_jv_clk[8];
#define BM_JV_PROF(k) do { if (threadIdx.x == 0) g_jv_clk[k] = wall_clock64(); } while (0)
#ifdef JV_FINE
__device__ long long g_jv_fine[8];
#define BM_JV_T(k) do { const long long t_ = clock64(); if (threadIdx.x == 0) { g_jv_fine[k] += t_ - jv_t0; } jv_t0 = clock64(); } while (0)
#define BM_JV_T_DECL long long jv_t0 = clock64();
#endif
#include "lap_jv.hpp"

template <int NTHR>
__global__ __launch_bounds__(NTHR) void k_jv(int nr, int nc, const double* cost, int* x, int* y) {
    extern __shared__ unsigned char dyn[];
    __shared__ int s_int[bm::MAX_WAVES + 1];
    __shared__ double s_dbl[bm::MAX_WAVES];
    const bm::Ctx c = bm::make_ctx(s_int, s_dbl);
    const bm::JvLds L = bm::jv_carve(dyn, nr + nc);
    bm::lap_jv_extended(c, L, nr, nc, [&](int i, int j) { return cost[(long)i * nc + j]; }, false, 0.0, x, y);
    BM_JV_PROF(5);
}

int main(int argc, char** argv) {
    const int nr = argc > 2 ? atoi(argv[1]) : 1, nc = argc > 2 ? atoi(argv[2]) : 385;
    std::vector<double> h((size_t)nr * nc, 0.0);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; if ((s >> 8) % 50 == 0) v = -double((s >> 16) % 100) / 100.0; }
    double* d; int *x, *y;
    hipMalloc(&d, h.size() * 8); hipMalloc(&x, nr * 4); hipMalloc(&y, nc * 4);
    hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = bm::jv_lds_bytes(nr + nc);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_jv<512>, dim3(1), dim3(512), lds, 0, nr, nc, d, x, y);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long clk[8];
        hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_jv_clk), sizeof(clk));
        printf("%d x %d: %.3f ms | us (100 MHz clock): minima+claim %.1f transfer %.1f row-reduction %.1f augmentation %.1f tail %.1f\n", nr, nc, ms,
               (clk[1] - clk[0]) / 100.0, (clk[2] - clk[1]) / 100.0, (clk[3] - clk[2]) / 100.0, (clk[4] - clk[3]) / 100.0, (clk[5] - clk[4]) / 100.0);
    }
#ifdef JV_FINE
    long long fine[8];
    hipMemcpyFromSymbol(fine, HIP_SYMBOL(g_jv_fine), sizeof(fine));
    printf("  row-reduction iteration segments, shader clocks summed over 3 runs: [loop head -> fetch] %lld  scan %lld  reduce %lld  tail %lld\n", fine[0], fine[1], fine[2], fine[3]);
#endif
    return 0;
}
