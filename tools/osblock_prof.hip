// Standalone phase profiler for the fused OSBlock kernel (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DBM_OSBLOCK_PROF -I boxmot_amd/csrc \
//         tools/osblock_prof.hip -o gpurun_out/osblock_prof && gpurun_out/osblock_prof [n_crops] [iters]
// Prints the launch time (HIP events) and the shader-clock cycles each phase of k_osblock took, averaged per wave.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernel_macros.hpp"
#include "reid_fused.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int STAGE, int CIN, bool DOWN, bool TRANS, bool EMIT = false, bool RECON = false>
static void run(const char* name, int n, int iters) {
    using G = bm::Geo<STAGE>;
    bm::BlkPack bp = bm::make_blk_pack(STAGE, CIN, DOWN);
    std::vector<unsigned short> w(bp.total / 2);
    unsigned s = 12345u;
    for (auto& v : w) { s = s * 1664525u + 1013904223u; v = bm::f32_to_f16_bits(((s >> 8) & 0xffff) / 65536.0f * 0.2f - 0.1f); }
    // fp32 regions (biases, gate fc) get small fp32 values
    auto fill_f32 = [&](long off, long cnt) { float* f = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(w.data()) + off); for (long i = 0; i < cnt; ++i) f[i] = 0.01f * (i % 7); };
    fill_f32(bp.conv1_b, bp.midp); fill_f32(bp.fc1_w, bp.hid * bp.midp); fill_f32(bp.fc1_b, bp.hid);
    fill_f32(bp.fc2_w, bp.midp * bp.hid); fill_f32(bp.fc2_b, bp.midp); fill_f32(bp.conv3_b, bp.cout);
    for (int li = 0; li < 10; ++li) fill_f32(bp.light0 + li * bp.light_bytes + bp.light_b, bp.midp);
    const size_t in_elems = (size_t)n * G::P * (RECON ? (STAGE == 0 ? 16 : 64) : CIN), out_elems = (size_t)n * G::P * G::COUT;
    std::vector<unsigned short> x(in_elems);
    for (auto& v : x) { s = s * 1664525u + 1013904223u; v = bm::f32_to_f16_bits(((s >> 8) & 0xffff) / 65536.0f); }
    unsigned char *d_w, *d_wt; _Float16 *d_in, *d_out, *d_x1;
    CK(hipMalloc(&d_w, bp.total)); CK(hipMalloc(&d_in, in_elems * 2)); CK(hipMalloc(&d_out, out_elems * 2));
    CK(hipMalloc(&d_x1, (size_t)n * G::P * G::MIDP * 2));
    const size_t wt_bytes = (size_t)(G::COUT / 16) * (G::COUT / 32) * 1024 + G::COUT * 4;     // transition fragments + bias
    std::vector<unsigned short> wt(wt_bytes / 2);
    for (auto& v : wt) { s = s * 1664525u + 1013904223u; v = bm::f32_to_f16_bits(((s >> 8) & 0xffff) / 65536.0f * 0.2f - 0.1f); }
    for (int i = 0; i < G::COUT; ++i) reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(wt.data()) + wt_bytes - G::COUT * 4)[i] = 0.01f;
    CK(hipMalloc(&d_wt, wt_bytes)); CK(hipMemcpy(d_wt, wt.data(), wt_bytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, w.data(), bp.total, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_in, x.data(), in_elems * 2, hipMemcpyHostToDevice));
    auto kern = bm::k_osblock<STAGE, CIN, DOWN, TRANS, EMIT, RECON>;
    // EMIT / RECON: the neighbouring block's weights are the same random blob (only the access pattern matters here)
    _Float16* d_x2; CK(hipMalloc(&d_x2, (size_t)n * G::P * G::MIDP * 2)); CK(hipMemset(d_x2, 0, (size_t)n * G::P * G::MIDP * 2)); CK(hipMemset(d_x1, 0, (size_t)n * G::P * G::MIDP * 2));
    const bm::BlkPack bq = bm::make_blk_pack(STAGE, EMIT ? G::COUT : (STAGE == 0 ? 16 : 64), EMIT ? 0 : 1);
    std::vector<unsigned short> wq(bq.total / 2);
    for (auto& v : wq) { s = s * 1664525u + 1013904223u; v = bm::f32_to_f16_bits(((s >> 8) & 0xffff) / 65536.0f * 0.2f - 0.1f); }
    unsigned char* d_wq; CK(hipMalloc(&d_wq, bq.total)); CK(hipMemcpy(d_wq, wq.data(), bq.total, hipMemcpyHostToDevice));
    const bm::BlkLink link = EMIT ? bm::BlkLink{d_wq, bq.conv1_a, bq.conv1_b, 0, d_x2} : (RECON ? bm::BlkLink{d_wq, bq.conv3_a, bq.conv3_b, bq.down_a, d_x2} : bm::BlkLink{});
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long zero[8] = {};
    float best = 1e9f;
    for (int it = 0; it < iters; ++it) {
#ifdef BM_OSBLOCK_PROF
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bm::g_osblock_prof), zero, sizeof(zero)));
#endif
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(n), dim3(64 * G::NWAVES), G::LDS_BYTES, 0, d_in, d_out, d_w, bp, (const int*)nullptr, d_x1, (const unsigned char*)d_wt, link);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    {   // output statistics: the same inputs through a differently built kernel must give (nearly) the same numbers
        const size_t oe = (size_t)n * (TRANS ? G::P / 4 : G::P) * G::COUT;
        std::vector<unsigned short> y(oe);
        CK(hipMemcpy(y.data(), d_out, oe * 2, hipMemcpyDeviceToHost));
        double sum = 0, sabs = 0; size_t bad = 0;
        for (size_t i = 0; i < oe; ++i) {
            _Float16 hv; memcpy(&hv, &y[i], 2); const float f = (float)hv;
            if (!(f == f) || f > 60000.f || f < -60000.f) { ++bad; continue; }
            sum += f * (double)((i % 97) + 1); sabs += f < 0 ? -f : f;
        }
        printf("%s: output weighted sum %.6e, mean |y| %.6e, non-finite %zu\n", name, sum, sabs / oe, bad);
    }
    unsigned long long acc[8] = {};
#ifdef BM_OSBLOCK_PROF
    CK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(bm::g_osblock_prof), sizeof(acc)));
#else
    printf("%s: n=%d best %.3f ms\n", name, n, best); (void)zero;
    CK(hipFree(d_w)); CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(d_x1)); CK(hipFree(d_wt));
    return;
#endif
    const double waves = (double)n * G::NWAVES;
    static const char* PH[8] = {"init", "conv1", "pw+write", "barrier1", "dw3x3", "barrier2", "gate", "conv3"};
    double tot = 0; for (int k = 0; k < 8; ++k) tot += acc[k] / waves;
    printf("%s: n=%d best %.3f ms, waves/crop %d, lds %d B; cycles per wave: total %.0f\n", name, n, best, G::NWAVES, G::LDS_BYTES, tot);
    for (int k = 0; k < 8; ++k) printf("    %-9s %9.0f  %5.1f%%\n", PH[k], acc[k] / waves, 100.0 * acc[k] / waves / tot);
    CK(hipFree(d_w)); CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(d_x1)); CK(hipFree(d_wt));
}

// stem: one 1080p frame of noise, boxes of the bench scenario's size (35-70 px wide and tall)
static void run_stem(int n, int iters) {
    const int W = 1920, H = 1080;
    std::vector<unsigned char> frame((size_t)W * H * 3);
    unsigned s = 777u;
    for (auto& v : frame) { s = s * 1664525u + 1013904223u; v = (unsigned char)(s >> 24); }
    std::vector<float> boxes((size_t)n * 4), lut(768);
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u; const float x = 10.f + (s >> 8) % 1800;
        s = s * 1664525u + 1013904223u; const float y = 10.f + (s >> 8) % 980;
        s = s * 1664525u + 1013904223u; const float w = 35.f + (s >> 8) % 36;
        s = s * 1664525u + 1013904223u; const float h = 36.f + (s >> 8) % 37;
        boxes[4 * i] = x + 0.3f; boxes[4 * i + 1] = y + 0.6f; boxes[4 * i + 2] = x + w; boxes[4 * i + 3] = y + h;
    }
    for (int i = 0; i < 768; ++i) lut[i] = ((i & 255) / 255.f - 0.45f) / 0.225f;
    std::vector<unsigned short> w(7 * 512 + 32);
    for (auto& v : w) { s = s * 1664525u + 1013904223u; v = bm::f32_to_f16_bits(((s >> 8) & 0xffff) / 65536.0f * 0.2f - 0.1f); }
    unsigned char *d_frame, *d_w; const unsigned char** d_frames; int* d_cs; float *d_boxes, *d_lut; _Float16* d_out;
    CK(hipMalloc(&d_frame, frame.size())); CK(hipMemcpy(d_frame, frame.data(), frame.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_frames, sizeof(void*))); CK(hipMemcpy(d_frames, &d_frame, sizeof(void*), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_cs, n * 4)); CK(hipMemset(d_cs, 0, n * 4));
    CK(hipMalloc(&d_boxes, boxes.size() * 4)); CK(hipMemcpy(d_boxes, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_lut, 768 * 4)); CK(hipMemcpy(d_lut, lut.data(), 768 * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_w, w.size() * 2)); CK(hipMemcpy(d_w, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)n * 2048 * 16 * 2));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bm::k_stem_resize_fused), hipFuncAttributeMaxDynamicSharedMemorySize, bm::STEM2_LDS));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(bm::k_stem_resize_fused, dim3(n), dim3(512), bm::STEM2_LDS, 0, (const uint8_t* const*)d_frames, (const int*)d_cs,
                           (const float*)d_boxes, 4, W, H, (const float*)d_lut, d_out, (const unsigned char*)d_w, (const int*)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("stem_resize_fused: n=%d best %.3f ms\n", n, best);
#ifdef BM_OSBLOCK_PROF
    {   // phase clocks of the last launch (a -DBM_OSBLOCK_PROF build): shader-clock cycles per wave
        unsigned long long zero[8] = {}, acc[8] = {};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bm::g_osblock_prof), zero, sizeof(zero)));
        hipLaunchKernelGGL(bm::k_stem_resize_fused, dim3(n), dim3(512), bm::STEM2_LDS, 0, (const uint8_t* const*)d_frames, (const int*)d_cs,
                           (const float*)d_boxes, 4, W, H, (const float*)d_lut, d_out, (const unsigned char*)d_w, (const int*)nullptr);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(acc, HIP_SYMBOL(bm::g_osblock_prof), sizeof(acc)));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(bm::g_osblock_prof), zero, sizeof(zero)));
        static const char* PH[8] = {"prologue", "stage rows", "barrier a", "resample", "barrier b", "conv+pool", "barrier c", "edge fix-up"};
        const double waves = (double)n * 8;
        double tot = 0; for (int k = 0; k < 8; ++k) tot += acc[k] / waves;
        printf("stem_resize_fused phases, cycles per wave: total %.0f\n", tot);
        for (int k = 0; k < 8; ++k) printf("    %-12s %9.0f  %5.1f%%\n", PH[k], acc[k] / waves, 100.0 * acc[k] / waves / tot);
    }
#endif
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, iters = argc > 2 ? atoi(argv[2]) : 5;
    run_stem(n, iters);
    if (argc > 3) return 0;
    run<0, 16, true, false>("osblock<0,16,down>", n, iters);
    run<0, 64, false, true>("osblock<0,64,trans>", n, iters);
    run<0, 16, true, false, true, false>("osblock<0,16,down,EMIT>", n, iters);
    run<0, 64, false, true, false, true>("osblock<0,64,trans,RECON>", n, iters);
    run<1, 64, true, false>("osblock<1,64,down>", n, iters);
    run<1, 96, false, true>("osblock<1,96,trans>", n, iters);
    run<1, 64, true, false, true, false>("osblock<1,64,down,EMIT>", n, iters);          // BM_STAGE1_HANDOVER pair
    run<1, 96, false, true, false, true>("osblock<1,96,trans,RECON>", n, iters);
    run<2, 96, true, false>("osblock<2,96,down>", n, iters);
    run<2, 128, false, false>("osblock<2,128>", n, iters);
    return 0;
}
