// Standalone pass timer for the fp32-grade osnet_x1_0 kernel family (osnet_wide_hp.hpp: chain-fused LightConvs, (hi, lo) GEMMs).
// Development tool, not part of the library (see tools/wide_prof.hip for the fp16 family's twin).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-D<variant switches>] -I boxmot_amd/csrc tools/wide_hp_prof.hip \
//         -o tools/_build/wide_hp_prof[_variant] && tools/_build/wide_hp_prof [n_crops = 1024] [iters = 5]
// Random folded weights and random (hi, lo) crops; prints the best time of a whole forward pass, the algorithmic TFLOP/s
// (1.958 GFLOP per crop) and an order-independent checksum of the embeddings.  `rocprofv3 --kernel-trace --stats -- <this>` gives
// the per-kernel table.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernel_macros.hpp"
#include "reid_layout.hpp"
#include "osnet_wide_hp.hpp"

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
    bm::WideOsnetHP net(w.data(), L, d_w32, n, owned);
    {   // crops: (hi, lo) fp16 RGBX planes with a 3-pixel zero border; random interior in [-2, 2]
        std::vector<_Float16> ch_((size_t)n * bm::WSTEM_ROWS * bm::WSTEM_COLS * 4, (_Float16)0.f), cl_(ch_.size(), (_Float16)0.f);
        for (int i = 0; i < n; ++i)
            for (int y = 3; y < 259; ++y)
                for (int x = 3; x < 131; ++x)
                    for (int k = 0; k < 3; ++k) {
                        const float v = rnd() * 4.f - 2.f;
                        const size_t o = (((size_t)i * bm::WSTEM_ROWS + y) * bm::WSTEM_COLS + x) * 4 + k;
                        ch_[o] = (_Float16)v; cl_[o] = (_Float16)(v - (float)ch_[o]);
                    }
        CK(hipMemcpy(net.crops_h(), ch_.data(), ch_.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(net.crops_l(), cl_.data(), cl_.size() * 2, hipMemcpyHostToDevice));
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
    int bad = 0;
    for (size_t i = 0; i < out.size(); ++i) { unsigned u; memcpy(&u, &out[i], 4); sum += (unsigned long long)u * (i % 8191 + 1); bad += !(out[i] == out[i]); }
    printf("osnet_x1_0 fp32-grade forward: n=%d best %.3f ms = %.1f TFLOP/s algorithmic (1.9577 GFLOP per crop)\n", n, best, n * 1.957691392e9 / (best * 1e9));
    printf("    output checksum %016llx, %d NaNs\n", sum, bad);
#ifdef BM_CHAIN_PROF
    {
        unsigned long long pr[8];
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(bm::g_chain_prof), sizeof(pr)));
        static const char* nm[6] = {"init (LDS clear, weights)", "x1 load", "1x1 (pointwise)", "dw: image write + barrier", "dw: taps", "branch output + band sums"};
        unsigned long long tot = 0;
        for (int k = 0; k < 6; ++k) tot += pr[k];
        printf("    k_chain_hp phase clocks (all chain launches of %d passes, all waves): total %llu\n", iters + 1, tot);
        for (int k = 0; k < 6; ++k) printf("      %-28s %6.1f %%\n", nm[k], 100.0 * pr[k] / (double)tot);
    }
#endif
    return 0;
}
