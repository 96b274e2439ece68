// Standalone launch-time / phase profiler for the fp32-grade fused ReID kernels (reid_hp.hpp).  Development tool, not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DBM_OSBLOCK_PROF] [-D<variant switches>] -I boxmot_amd/csrc \
//         tools/hp_prof.hip -o tools/_build/hp_prof[_variant] && tools/_build/hp_prof [n_crops] [iters]
// Random weights and activations (only the access pattern and the instruction stream matter); prints the best launch time of every
// kernel of the family and, in a -DBM_OSBLOCK_PROF build, the shader-clock cycles per wave of each phase of k_osblock_hp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernel_macros.hpp"
#include "reid_hp.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned g_seed = 12345u;
static float rnd() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffff) / 65536.0f; }

template <class T> static T* dev(const std::vector<T>& v) { T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

static std::vector<float> rand_block_weights(const bm::OsnetLayout& L) {
    std::vector<float> w((size_t)L.total);
    for (auto& v : w) v = rnd() * 0.2f - 0.1f;
    return w;
}

static int g_persist = 0;       // argv[3]: persistent launch form with this many workgroups per CU slot (0 = one workgroup per crop)

template <int STAGE, int CIN, bool DOWN, bool TRANS, bool EMIT = false, bool RECON = false>
static void run(const char* name, int n, int iters, const bm::OsnetLayout& L, const std::vector<float>& w, int bi) {
    using G = bm::GeoHP<STAGE>;
    const bm::BlkPackHP bp = bm::make_blk_pack_hp(STAGE, CIN, DOWN);
    std::vector<uint8_t> wb, wq, wt;
    bm::pack_osblock_hp(w.data(), L.block[bi], bp, wb);
    const int bj = EMIT ? bi + 1 : (RECON ? bi - 1 : bi);
    const bm::BlkPackHP bq = bm::make_blk_pack_hp(STAGE, L.block[bj].cin, L.block[bj].cin != L.block[bj].cout);
    bm::pack_osblock_hp(w.data(), L.block[bj], bq, wq);
    bm::pack_pointwise_hp(w.data() + L.trans_w[STAGE < 2 ? STAGE : 0], w.data() + L.trans_b[STAGE < 2 ? STAGE : 0], G::COUT, G::COUT, wt, 0.25f);
    const int cin_mem = RECON ? (STAGE == 0 ? 16 : 64) : CIN;
    std::vector<unsigned short> xh((size_t)n * G::P * cin_mem), xl(xh.size());
    for (size_t i = 0; i < xh.size(); ++i) { const float v = rnd(); bm::split_hl(v, xh[i], xl[i]); }
    std::vector<float> x1((size_t)n * G::P * G::MIDP), x2(x1.size());
    for (auto& v : x1) v = rnd();
    for (auto& v : x2) v = rnd();
    unsigned char *d_w = dev(wb), *d_wq = dev(wq), *d_wt = dev(wt);
    _Float16 *d_xh = reinterpret_cast<_Float16*>(dev(xh)), *d_xl = reinterpret_cast<_Float16*>(dev(xl)), *d_oh, *d_ol;
    float *d_x1 = dev(x1), *d_x2 = dev(x2);
    const size_t out_elems = (size_t)n * G::P * G::COUT;
    CK(hipMalloc(&d_oh, out_elems * 2)); CK(hipMalloc(&d_ol, out_elems * 2));
    CK(hipMemset(d_oh, 0, out_elems * 2)); CK(hipMemset(d_ol, 0, out_elems * 2));
    bm::BlkLinkHP link = EMIT ? bm::BlkLinkHP{d_wq, bq.conv1_a, bq.conv1_b, 0, d_x2}
                              : (RECON ? bm::BlkLinkHP{d_wq, bq.conv3_a, bq.conv3_b, bq.down_a, d_x2} : bm::BlkLinkHP{});
    int grid = n;
    if (g_persist > 0) {
        link.n_crops = n;
        const int slots = 256 * (STAGE == 2 ? 2 : 1) * g_persist;
        grid = n < slots ? n : slots;
    }
    auto kern = bm::k_osblock_hp<STAGE, CIN, DOWN, TRANS, EMIT, RECON>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long zero[8] = {};
    float best = 1e9f;
    for (int it = 0; it < iters; ++it) {
#ifdef BM_OSBLOCK_PROF
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bm::g_osblock_prof), zero, sizeof(zero)));
#endif
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * G::NWAVES), G::LDS_BYTES, 0, (const _Float16*)d_xh, (const _Float16*)d_xl, d_oh, d_ol,
                           (const unsigned char*)d_w, bp, (const int*)nullptr, d_x1, (const unsigned char*)d_wt, link);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%s: n=%d best %.3f ms (%.1f us per crop per CU at 256 CUs)\n", name, n, best, best * 1e3 * 256 / n);
    {   // position-weighted checksum of everything the kernel wrote (output planes; EMIT: the fp32 hand-over tensors): equal across
        // -D variants of the kernel = the same results on the device
        const size_t px_out = TRANS ? G::P / 4 : G::P;
        std::vector<unsigned short> oh((size_t)n * px_out * G::COUT), ol(oh.size());
        std::vector<unsigned> f1((size_t)n * G::P * G::MIDP), f2(f1.size());
        CK(hipMemcpy(oh.data(), d_oh, oh.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ol.data(), d_ol, ol.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(f1.data(), d_x1, f1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f2.data(), d_x2, f2.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long sum = 0;
        for (size_t i = 0; i < oh.size(); ++i) sum += (unsigned long long)(oh[i] * 2654435761u + ol[i] * 40503u) * (i % 8191 + 1);
        for (size_t i = 0; i < f1.size(); ++i) sum += (unsigned long long)(f1[i] * 2246822519u + f2[i] * 3266489917u) * (i % 8191 + 1);
        printf("    output checksum %016llx\n", sum);
    }
#ifdef BM_OSBLOCK_PROF
    unsigned long long acc[8] = {};
    CK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(bm::g_osblock_prof), sizeof(acc)));
    const double waves = (double)n * G::NWAVES;
    static const char* PH[8] = {"init+conv1", "branch in+pw", "image write", "barrier1", "dw3x3(+pw)", "barrier2", "gate", "epilogue"};
    double tot = 0; for (int k = 0; k < 8; ++k) tot += acc[k] / waves;
    printf("    cycles per wave: total %.0f\n", tot);
    for (int k = 0; k < 8; ++k) printf("    %-13s %9.0f  %5.1f%%\n", PH[k], acc[k] / waves, 100.0 * acc[k] / waves / tot);
#else
    (void)zero;
#endif
    for (void* p : {(void*)d_w, (void*)d_wq, (void*)d_wt, (void*)d_xh, (void*)d_xl, (void*)d_oh, (void*)d_ol, (void*)d_x1, (void*)d_x2}) CK(hipFree(p));
}

static void run_stem(int n, int iters, const bm::OsnetLayout& L, const std::vector<float>& w) {
    const int W = 1920, H = 1080;
    std::vector<unsigned char> frame((size_t)W * H * 3);
    for (auto& v : frame) v = (unsigned char)(rnd() * 255.f);
    std::vector<float> boxes((size_t)n * 4);
    for (int i = 0; i < n; ++i) {
        const float x = 10.f + (int)(rnd() * 1800), y = 10.f + (int)(rnd() * 980), bw = 35.f + (int)(rnd() * 36), bh = 36.f + (int)(rnd() * 37);
        boxes[4 * i] = x + 0.3f; boxes[4 * i + 1] = y + 0.6f; boxes[4 * i + 2] = x + bw; boxes[4 * i + 3] = y + bh;
    }
    std::vector<uint8_t> ws;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    bm::pack_stem_hp_fused(w.data() + L.stem_w, w.data() + L.stem_b, mean, stdv, ws);
    unsigned char *d_frame = dev(frame), *d_w = dev(ws);
    const unsigned char** d_frames; CK(hipMalloc(&d_frames, sizeof(void*))); CK(hipMemcpy(d_frames, &d_frame, sizeof(void*), hipMemcpyHostToDevice));
    int* d_cs; CK(hipMalloc(&d_cs, n * 4)); CK(hipMemset(d_cs, 0, n * 4));
    float* d_boxes = dev(boxes);
    _Float16 *d_oh, *d_ol; CK(hipMalloc(&d_oh, (size_t)n * 2048 * 16 * 2)); CK(hipMalloc(&d_ol, (size_t)n * 2048 * 16 * 2));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_stem_resize_fused_hp), hipFuncAttributeMaxDynamicSharedMemorySize, bm::STEM2_LDS_HP));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(bm::k_stem_resize_fused_hp, dim3(n), dim3(512), bm::STEM2_LDS_HP, 0, (const uint8_t* const*)d_frames, (const int*)d_cs,
                           (const float*)d_boxes, 4, W, H, d_oh, d_ol, (const unsigned char*)d_w, (const int*)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("stem_resize_fused_hp: n=%d best %.3f ms\n", n, best);
    {   // order-independent checksum of the (hi, lo) output planes: equal across -D variants of the kernel = same results on the device
        std::vector<unsigned short> oh((size_t)n * 2048 * 16), ol(oh.size());
        CK(hipMemcpy(oh.data(), d_oh, oh.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ol.data(), d_ol, ol.size() * 2, hipMemcpyDeviceToHost));
        unsigned long long sum = 0;
        for (size_t i = 0; i < oh.size(); ++i) sum += (unsigned long long)(oh[i] * 2654435761u + ol[i] * 40503u) * (i % 8191 + 1);
        printf("    output checksum %016llx\n", sum);
    }
#ifdef BM_OSBLOCK_PROF
    unsigned long long zero[8] = {}, acc[8] = {};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(bm::g_osblock_prof), zero, sizeof(zero)));
    hipLaunchKernelGGL(bm::k_stem_resize_fused_hp, dim3(n), dim3(512), bm::STEM2_LDS_HP, 0, (const uint8_t* const*)d_frames, (const int*)d_cs,
                       (const float*)d_boxes, 4, W, H, d_oh, d_ol, (const unsigned char*)d_w, (const int*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(bm::g_osblock_prof), sizeof(acc)));
    static const char* PH[8] = {"prologue", "stage rows", "barrier a", "resample", "barrier b", "conv+pool", "barrier c", "edge fix-up"};
    const double waves = (double)n * 8;
    double tot = 0; for (int k = 0; k < 8; ++k) tot += acc[k] / waves;
    printf("    cycles per wave: total %.0f\n", tot);
    for (int k = 0; k < 8; ++k) printf("    %-12s %9.0f  %5.1f%%\n", PH[k], acc[k] / waves, 100.0 * acc[k] / waves / tot);
#endif
}

static void run_head(int n, int iters, const bm::OsnetLayout& L, const std::vector<float>& w) {
    std::vector<uint8_t> w5, wfc;
    bm::pack_pointwise_hp(w.data() + L.conv5_w, w.data() + L.conv5_b, 128, 128, w5);
    bm::pack_fc_hp(w.data() + L.fc_w, w.data() + L.fc_b, 512, 128, wfc);
    std::vector<unsigned short> xh((size_t)n * 128 * 128), xl(xh.size());
    for (size_t i = 0; i < xh.size(); ++i) bm::split_hl(rnd(), xh[i], xl[i]);
    unsigned char *d5 = dev(w5), *dfc = dev(wfc);
    _Float16 *d_xh = reinterpret_cast<_Float16*>(dev(xh)), *d_xl = reinterpret_cast<_Float16*>(dev(xl));
    float* d_out; CK(hipMalloc(&d_out, (size_t)n * 512 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((bm::k_head_hp<128, 512>), dim3((n + bm::HEAD_NB - 1) / bm::HEAD_NB), dim3(256), 0, 0, (const _Float16*)d_xh, (const _Float16*)d_xl,
                           (const unsigned char*)d5, (const unsigned char*)dfc, d_out, (const int*)nullptr, (const int*)nullptr, n);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("head_hp: n=%d best %.3f ms\n", n, best);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, iters = argc > 2 ? atoi(argv[2]) : 5;
    g_persist = argc > 3 ? atoi(argv[3]) : 0;
    if (g_persist && !BM_HP_PERSIST) { fprintf(stderr, "the persistent form needs a -DBM_HP_PERSIST=1 build\n"); return 1; }
    if (g_persist) printf("# persistent launch form: %d workgroup(s) per CU slot\n", g_persist);
    const int ch[4] = {16, 64, 96, 128};
    const bm::OsnetLayout L = bm::make_osnet_layout(ch, 512);
    const std::vector<float> w = rand_block_weights(L);
    run_stem(n, iters, L, w);
    run<0, 16, true, false, true, false>("osblock_hp<0,16,down,EMIT>", n, iters, L, w, 0);
    run<0, 64, false, true, false, true>("osblock_hp<0,64,trans,RECON>", n, iters, L, w, 1);
    run<1, 64, true, false>("osblock_hp<1,64,down>", n, iters, L, w, 2);
    run<1, 96, false, true>("osblock_hp<1,96,trans>", n, iters, L, w, 3);
    run<2, 96, true, false>("osblock_hp<2,96,down>", n, iters, L, w, 4);
    run<2, 128, false, false>("osblock_hp<2,128>", n, iters, L, w, 5);
    run_head(n, iters, L, w);
    return 0;
}
