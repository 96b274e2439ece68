// TEST-ONLY harness: runs the wide-OSNet kernels (boxmot_amd/csrc/osnet_wide_kernels.hpp + gemm_f16.hpp, the device source unchanged) on
// CPU threads with the emulated MFMA of hip_shim.hpp, in the launch order of WideOsnet::forward.  A reduced architecture
// (channels 32 / 128 / 128 / 128: every width a multiple of 32, like osnet_x1_0's) keeps it to seconds; osnet_x1_0 itself runs
// on the GPU (tests/test_gpu_reid.py).
#include "hip_shim.hpp"

#include <cstdlib>
#include <functional>
#include <vector>

#include "../../boxmot_amd/csrc/osnet_wide_kernels.hpp"
#include "../../boxmot_amd/csrc/osnet_wide_pack.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuDim3 gridDim;
EmuBlock* g_emu_block = nullptr;
unsigned char* g_emu_dynamic_lds = nullptr;
EmuMfmaBuf* g_emu_mfma = nullptr;

namespace {
struct TA { const std::function<void()>* fn; int tid, bx, by; };
void* tmain(void* p) {
    TA* a = static_cast<TA*>(p);
    threadIdx.x = a->tid; blockIdx.x = a->bx; blockIdx.y = a->by;
    (*a->fn)();
    return nullptr;
}
void launch(long gx, int gy, int nthr, const std::function<void()>& fn) {
    static EmuBlock block;
    static EmuMfmaBuf mf;
    static std::vector<unsigned char> lds(200000 + 64);
    g_emu_block = &block; g_emu_mfma = &mf;
    g_emu_dynamic_lds = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~uintptr_t(15));
    blockDim.x = nthr; gridDim.x = (unsigned)gx; gridDim.y = gy;
    block.block_barrier.init(nthr);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
    for (int by = 0; by < gy; ++by)
        for (long bx = 0; bx < gx; ++bx) {
            std::memset(lds.data(), 0xFF, lds.size());
            std::vector<TA> ta(nthr);
            for (int t = 0; t < nthr; ++t) ta[t] = TA{&fn, t, (int)bx, by};
            emu_run_threads(nthr, tmain, ta.data(), sizeof(ta[0]), 1 << 20);
        }
}
}  // namespace

// crops: normalised fp32 NHWC [n][256][128][3]; feats [rows][feat]; stages (optional): fp32 copies of the six block outputs
extern "C" int emu_wide_forward(const float* blob, long n_floats, const float* crops, int n, const int* rows, float* feats, float** stages) {
    using namespace bm;
    const int32_t* h = reinterpret_cast<const int32_t*>(blob);
    if (h[0] != REID_MAGIC) return -1;
    const int ch[4] = {h[1], h[2], h[3], h[4]};
    const OsnetLayout L = make_osnet_layout(ch, h[5]);
    if (n_floats != REID_HEADER_INTS + L.total) return -2;
    if (!wide_osnet_supports(L)) return -3;
    const float* W32 = blob + REID_HEADER_INTS;
    const WideW16 pk = wide_pack_w16(W32, L);
    std::vector<_Float16> w16(pk.data.size());
    std::memcpy(w16.data(), pk.data.data(), pk.data.size() * 2);
    const _Float16* W16 = w16.data();
    const int c0 = ch[0];
    const _Float16* stem16 = W16 + pk.stem;
    const size_t N = (size_t)n;
    // the stem's input layout: fp16 RGBX with a 3-pixel zero border (what k_crop_resize_rgbx writes on the device)
    std::vector<_Float16> crops16(N * WSTEM_ROWS * WSTEM_COLS * 4, (_Float16)0.f);
    for (size_t i = 0; i < N; ++i)
        for (int y = 0; y < REID_IN_H; ++y)
            for (int x = 0; x < REID_IN_W; ++x)
                for (int c = 0; c < 3; ++c)
                    crops16[((i * WSTEM_ROWS + y + 3) * WSTEM_COLS + x + 3) * 4 + c] = (_Float16)crops[((i * REID_IN_H + y) * REID_IN_W + x) * 3 + c];
    size_t blk = N * 2048 * (size_t)c0, mid = 0;
    { int P = 2048; for (int s = 0; s < 3; ++s, P /= 4) { blk = std::max(blk, N * P * (size_t)ch[s + 1]); mid = std::max(mid, N * P * (size_t)(ch[s + 1] / 4)); } }
    std::vector<_Float16> act_a(blk), act_b(blk);
    std::vector<std::vector<_Float16>> midb(8, std::vector<_Float16>(mid));
    std::vector<float> gap_part(4 * N * 8 * 128);
    std::vector<float> bsum[6];

    auto gemm = [&](const _Float16* X, const _Float16* Wt, const float* bias, _Float16* out, const _Float16* res, long M, int Nn, int K, int relu,
                    GemmExt ext = GemmExt{}) {
        const long gx = (M + GEMM_BM - 1) / GEMM_BM;
        if (ext.pool_w == 32) launch(gx * (Nn / 128), 1, 256, [=]() { k_gemm_f16_glds<5, 32>(X, Wt, bias, out, res, (int)M, Nn, K, relu, ext); });
        else if (ext.pool_w == 16) launch(gx * (Nn / 128), 1, 256, [=]() { k_gemm_f16_glds<6, 32>(X, Wt, bias, out, res, (int)M, Nn, K, relu, ext); });
        else if (Nn % 128 == 0) launch(gx * (Nn / 128), 1, 256, [=]() { k_gemm_f16_glds<4, 32>(X, Wt, bias, out, res, (int)M, Nn, K, relu, ext); });
        else if (Nn % 96 == 0) launch(gx * (Nn / 96), 1, 256, [=]() { k_gemm_f16<4, 96>(X, Wt, bias, out, res, (int)M, Nn, K, relu); });
        else if (Nn % 64 == 0) launch(gx * (Nn / 64), 1, 256, [=]() { k_gemm_f16<4, 64>(X, Wt, bias, out, res, (int)M, Nn, K, relu); });
        else launch(gx * (Nn / 32), 1, 256, [=]() { k_gemm_f16<4, 32>(X, Wt, bias, out, res, (int)M, Nn, K, relu); });
    };
    auto light = [&](int C, const _Float16* in, const LightW& lw, _Float16* out, float* gap, int H, int W) {
        const _Float16* pw = W16 + pk.of(lw.pw); const float* dw = W32 + lw.dw; const float* b = W32 + lw.b;
        if (C == 32) launch(H / WIDE_BAND, n, 256, [=]() { k_light_fused<32>(in, pw, dw, b, out, gap, H, W); });
        else if (C == 64) launch(H / WIDE_BAND, n, 256, [=]() { k_light_fused<64>(in, pw, dw, b, out, gap, H, W); });
        else if (C == 96) launch(H / WIDE_BAND, n, 256, [=]() { k_light_fused<96>(in, pw, dw, b, out, gap, H, W); });
        else launch(H / WIDE_BAND, n, 256, [=]() { k_light_fused<128>(in, pw, dw, b, out, gap, H, W); });
    };
    auto osblock = [&](const BlockW& B, const _Float16* x, _Float16* out, int H, int W) {
        const long n_pix = (long)n * H * W;
        const int nbands = H / WIDE_BAND, P = H * W;
        _Float16* x1 = midb[0].data();
        _Float16* brs[4] = {midb[1].data(), midb[2].data(), midb[3].data(), midb[4].data()};
        _Float16* tmp[2] = {midb[5].data(), midb[6].data()};
        gemm(x, W16 + pk.of(B.conv1_w), W32 + B.conv1_b, x1, nullptr, n_pix, B.mid, B.cin, 1);
        int li = 0;
        for (int br = 0; br < 4; ++br) {
            const _Float16* cur = x1;
            const int Lc = br + 1;
            for (int k = 0; k < Lc; ++k) {
                const bool last = k + 1 == Lc;
                _Float16* dst = last ? brs[br] : tmp[k & 1];
                light(B.mid, cur, B.light[li + k], dst, last ? gap_part.data() + (long)br * n * nbands * B.mid : nullptr, H, W);
                cur = dst;
            }
            li += Lc;
        }
        _Float16* x2 = midb[7].data();
        const float* gp = gap_part.data();
        const float *f1w = W32 + B.fc1_w, *f1b = W32 + B.fc1_b, *f2w = W32 + B.fc2_w, *f2b = W32 + B.fc2_b;
        const int ppb = 128;
        const long nn = n;
        _Float16 *ba = brs[0], *bb = brs[1], *bc = brs[2], *bd = brs[3];
        if (B.mid == 32) launch(n, (P + ppb - 1) / ppb, 256, [=]() { k_gate_sum4<32>(ba, bb, bc, bd, gp, f1w, f1b, f2w, f2b, x2, P, nbands, nn, ppb); });
        else if (B.mid == 64) launch(n, (P + ppb - 1) / ppb, 256, [=]() { k_gate_sum4<64>(ba, bb, bc, bd, gp, f1w, f1b, f2w, f2b, x2, P, nbands, nn, ppb); });
        else if (B.mid == 96) launch(n, (P + ppb - 1) / ppb, 256, [=]() { k_gate_sum4<96>(ba, bb, bc, bd, gp, f1w, f1b, f2w, f2b, x2, P, nbands, nn, ppb); });
        else launch(n, (P + ppb - 1) / ppb, 256, [=]() { k_gate_sum4<128>(ba, bb, bc, bd, gp, f1w, f1b, f2w, f2b, x2, P, nbands, nn, ppb); });
        if (B.down_w >= 0) {        // conv3(x2) + downsample(x) as one GEMM over two operand pairs, summed biases
            std::vector<float>& bs = bsum[&B - L.block];
            bs.resize(B.cout);
            for (int c = 0; c < B.cout; ++c) bs[c] = W32[B.conv3_b + c] + W32[B.down_b + c];
            GemmExt two;
            two.X2 = x; two.W2 = W16 + pk.of(B.down_w); two.K2 = B.cin;
            gemm(x2, W16 + pk.of(B.conv3_w), bs.data(), out, nullptr, n_pix, B.cout, B.mid, 1, two);
        } else
            gemm(x2, W16 + pk.of(B.conv3_w), W32 + B.conv3_b, out, x, n_pix, B.cout, B.mid, 1);
    };

    { const _Float16* c = crops16.data(); _Float16* o = act_a.data(); const float* sb = W32 + L.stem_b;
      if (c0 == 64) launch(64 / WSTEM_PBAND, n, 256, [=]() { k_wide_stem<64>(c, stem16, sb, o); });
      else launch(64 / WSTEM_PBAND, n, 256, [=]() { k_wide_stem<32>(c, stem16, sb, o); }); }
    _Float16 *cur = act_a.data(), *other = act_b.data();
    int H = 64, W = 32;
    for (int s = 0; s < 3; ++s) {
        for (int k = 0; k < 2; ++k) {
            osblock(L.block[s * 2 + k], cur, other, H, W);
            std::swap(cur, other);
            if (stages && stages[s * 2 + k]) {
                const long cnt = (long)n * H * W * L.block[s * 2 + k].cout;
                for (long i = 0; i < cnt; ++i) stages[s * 2 + k][i] = (float)cur[i];
            }
        }
        if (s < 2) {
            const int c = ch[s + 1];
            GemmExt pool;
            pool.pool_w = W;
            gemm(cur, W16 + pk.of(L.trans_w[s]), W32 + L.trans_b[s], other, nullptr, (long)n * H * W, c, c, 1, pool);
            std::swap(cur, other);
            H /= 2; W /= 2;
        }
    }
    const int c3 = ch[3];
    gemm(cur, W16 + pk.of(L.conv5_w), W32 + L.conv5_b, other, nullptr, (long)n * H * W, c3, c3, 1);
    {   // head: GAP -> FC GEMM (bias + ReLU) -> L2 + scatter
        const int P = H * W, F = L.feat;
        std::vector<_Float16> gap16((size_t)n * c3);
        std::vector<float> fc32((size_t)n * F);
        const long g8 = (long)n * (c3 / 8);
        { const _Float16* i = other; _Float16* o = gap16.data(); launch((g8 + 255) / 256, 1, 256, [=]() { k_wide_gap(i, o, P, c3, g8); }); }
        { const _Float16* x = gap16.data(); const _Float16* w = W16 + pk.of(L.fc_w); const float* b = W32 + L.fc_b; float* o = fc32.data();
          const long mt = ((long)n + GEMM_BM - 1) / GEMM_BM;
          launch(mt * (F / 128), 1, 256, [=]() { k_gemm_f16_glds<3, 32>(x, w, b, o, nullptr, n, F, c3, 1, GemmExt{}); }); }
        { const float* v = fc32.data(); launch((n + 3) / 4, 1, 256, [=]() { k_wide_l2(v, feats, rows, (long)n, F); }); }
    }
    return 0;
}
