// Standalone pass timer for the osnet_x1_0 kernel family (osnet_wide.hpp: the layer-per-launch fp16 MFMA kernels of configuration 3).
// Development tool, not part of the library: lets a -D variant of the family be A/B-timed in ~1 GPU-minute without Python / torch
// (tools/config_bench.py measures the same pass inside the tracker; `rocprofv3 --kernel-trace --stats -- tools/_build/wide_prof` gives
// the per-kernel table).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-D<variant switches>] -I boxmot_amd/csrc tools/wide_prof.hip \
//         -o tools/_build/wide_prof[_variant] && tools/_build/wide_prof [n_crops = 1024] [iters = 5]
// Random folded weights and random fp16 crops (only the access pattern and the instruction stream matter); prints the best time of a
// whole forward pass, the algorithmic TFLOP/s (1.958 GFLOP per crop) and an order-independent checksum of the embeddings (equal
// across variants = same results on the device).  Round-3 baseline on MI355X: 10.86 ms per 1024 crops (profiles/r3_wide_prof_baseline.txt).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernel_macros.hpp"
#include "reid_layout.hpp"
#include "gemm_f16.hpp"
#include "osnet_wide.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned g_seed = 2463534242u;
static float rnd() { g_seed ^= g_seed << 13; g_seed ^= g_seed >> 17; g_seed ^= g_seed << 5; return (g_seed >> 8) / 16777216.0f; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 5;
    const int ch[4] = {64, 256, 384, 512};
    const bm::OsnetLayout L = bm::make_osnet_layout(ch, 512);
    std::vector<float> w((size_t)L.total);
    for (auto& v : w) v = rnd() * 0.1f - 0.05f;
    float* d_w32;
    CK(hipMalloc(&d_w32, w.size() * 4));
    CK(hipMemcpy(d_w32, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    std::vector<void*> owned;
    bm::WideOsnet net(w.data(), L, d_w32, n, owned);
    {   // crops: fp16 RGBX with a 3-pixel zero border; random interior in [-2, 2]
        std::vector<_Float16> c((size_t)n * bm::WSTEM_ROWS * bm::WSTEM_COLS * 4, (_Float16)0.f);
        for (int i = 0; i < n; ++i)
            for (int y = 3; y < 259; ++y)
                for (int x = 3; x < 131; ++x)
                    for (int k = 0; k < 3; ++k) c[(((size_t)i * bm::WSTEM_ROWS + y) * bm::WSTEM_COLS + x) * 4 + k] = (_Float16)(rnd() * 4.f - 2.f);
        CK(hipMemcpy(net.crops_buffer(), c.data(), c.size() * 2, hipMemcpyHostToDevice));
    }
    float* d_out;
    CK(hipMalloc(&d_out, (size_t)n * L.feat * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < iters + 1; ++it) {            // the first pass is a warm-up
        CK(hipEventRecord(e0, 0));
        net.forward(n, d_out, nullptr, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    std::vector<float> out((size_t)n * L.feat);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (size_t i = 0; i < out.size(); ++i) { unsigned u; memcpy(&u, &out[i], 4); sum += (unsigned long long)u * (i % 8191 + 1); }
    printf("osnet_x1_0 forward: n=%d best %.3f ms = %.1f TFLOP/s algorithmic (1.9577 GFLOP per crop)\n", n, best, n * 1.957691392e9 / (best * 1e9));
    printf("    output checksum %016llx\n", sum);
    return 0;
}
