// Standalone timing + equality check of the two CLIP-ReID attention kernels (clip_kernels.hpp).  Development tool, not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I boxmot_amd/csrc tools/attn_prof.hip -o tools/_build/attn_prof && tools/_build/attn_prof [crops] [iters]
// Random q | k | v rows (fp16) for `crops` crops of ViT-B/16 geometry (129 tokens, 12 heads of 64): k_clip_attention (run-time T, four waves,
// transposed V image) against k_clip_attention_t<129> (three waves, three query tiles per wave, ds_read_b64_tr_b16): every output half must be
// identical; a (crop, head) sample is also checked against a double-precision softmax(q k^T / 8) v on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "clip_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float h2f(uint16_t h) { _Float16 v; memcpy(&v, &h, 2); return (float)v; }
static uint16_t f2h(float f) { _Float16 v = (_Float16)f; uint16_t h; memcpy(&h, &v, 2); return h; }

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 5;
    constexpr int T = 129, D = 768, H = 12;
    const size_t rows = (size_t)n * T;
    std::vector<uint16_t> qkv(rows * 3 * D);
    unsigned s = 12345u;
    for (auto& v : qkv) { s = s * 1664525u + 1013904223u; v = f2h((((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 4.0f); }
    _Float16 *d_qkv, *d_a, *d_b;
    CK(hipMalloc(&d_qkv, qkv.size() * 2)); CK(hipMalloc(&d_a, rows * D * 2)); CK(hipMalloc(&d_b, rows * D * 2));
    CK(hipMemcpy(d_qkv, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(d_a, 0xFF, rows * D * 2)); CK(hipMemset(d_b, 0xEE, rows * D * 2));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_clip_attention), hipFuncAttributeMaxDynamicSharedMemorySize, bm::clip_attn_lds_bytes(T)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_clip_attention_t<T>), hipFuncAttributeMaxDynamicSharedMemorySize, bm::clip_attn_t_lds_bytes<T>()));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) {
        float best = 1e30f;
        for (int i = 0; i < iters + 1; ++i) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (i > 0 && ms < best) best = ms;
        }
        return best;
    };
    const float t_old = time([&] { hipLaunchKernelGGL(bm::k_clip_attention, dim3(n * H), dim3(256), (size_t)bm::clip_attn_lds_bytes(T), 0, d_qkv, d_a, T, D, H); });
    const float t_new = time([&] { hipLaunchKernelGGL((bm::k_clip_attention_t<T>), dim3(n * H), dim3(192), (size_t)bm::clip_attn_t_lds_bytes<T>(), 0, d_qkv, d_b, D, H); });
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> a(rows * D), b(rows * D);
    CK(hipMemcpy(a.data(), d_a, a.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_b, b.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0; double maxd = 0;
    for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { ++diff; const double d = fabs((double)h2f(a[i]) - (double)h2f(b[i])); if (d > maxd || d != d) maxd = d; }
    // host check of (crop n - 1, head 7) and (crop 0, head 0)
    double max_ref = 0;
    for (int which = 0; which < 2; ++which) {
        const int crop = which ? n - 1 : 0, head = which ? 7 : 0;
        for (int q = 0; q < T; q += (which ? 1 : 16)) {
            std::vector<double> p(T); double mx = -1e300;
            for (int k = 0; k < T; ++k) {
                double sc = 0;
                for (int d = 0; d < 64; ++d) sc += (double)h2f(qkv[((size_t)crop * T + q) * 3 * D + head * 64 + d]) * (double)h2f(qkv[((size_t)crop * T + k) * 3 * D + D + head * 64 + d]);
                p[k] = sc * 0.125; mx = p[k] > mx ? p[k] : mx;
            }
            double sum = 0; for (int k = 0; k < T; ++k) { p[k] = exp(p[k] - mx); sum += p[k]; }
            for (int d = 0; d < 64; ++d) {
                double o = 0;
                for (int k = 0; k < T; ++k) o += p[k] * (double)h2f(qkv[((size_t)crop * T + k) * 3 * D + 2 * D + head * 64 + d]);
                const double got = h2f(b[((size_t)crop * T + q) * D + head * 64 + d]), err = fabs(got - o / sum);
                if (err > max_ref || err != err) max_ref = err;
            }
        }
    }
    const double fl = (double)n * H * 2.0 * 2.0 * T * T * 64;
    printf("attention, %d crops x %d heads, T = %d: k_clip_attention %.3f ms (%.0f TFLOP/s) | k_clip_attention_t<%d> %.3f ms (%.0f TFLOP/s) | differing halves %zu (max |d| %.3g) | max |new - host fp64| %.3g\n",
           n, H, T, t_old, fl / t_old / 1e9, T, t_new, fl / t_new / 1e9, diff, maxd, max_ref);
    return diff == 0 && max_ref < 2e-3 ? 0 : 1;
}
