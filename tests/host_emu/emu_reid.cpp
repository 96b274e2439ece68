// TEST-ONLY harness: runs the fused fp16 ReID kernels (boxmot_amd/csrc/reid_fused.hpp, the device
// source unchanged) on CPU threads with an emulated MFMA.  Built with the ROCm clang in host mode
// (needs _Float16).  Validates weight packing, fragment layouts and the data flow against the torch
// oracle before any GPU time is spent; not a product path (see hip_shim.hpp).
#include "hip_shim.hpp"

#include <cstdlib>
#include <functional>
#include <vector>

#include "../../boxmot_amd/csrc/reid_fused.hpp"
#include "../../boxmot_amd/csrc/reid_hp.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuDim3 gridDim;
EmuBlock* g_emu_block = nullptr;
unsigned char* g_emu_dynamic_lds = nullptr;
static int g_stage1_handover = 0;
static int g_head_per_crop = 0;
EmuMfmaBuf* g_emu_mfma = nullptr;

namespace {

struct TA { const std::function<void()>* fn; int tid; int bx, by; };

void* tmain(void* p) {
    TA* a = static_cast<TA*>(p);
    threadIdx.x = a->tid; blockIdx.x = a->bx; blockIdx.y = a->by;
    (*a->fn)();
    return nullptr;
}

// run `fn` as a grid of gx x gy workgroups of nthr threads (workgroups sequential)
void launch(int gx, int gy, int nthr, const std::function<void()>& fn) {
    static EmuBlock block;
    static EmuMfmaBuf mf;
    static std::vector<unsigned char> lds(400000 + 64);
    g_emu_block = &block; g_emu_mfma = &mf;
    g_emu_dynamic_lds = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~uintptr_t(15));
    blockDim.x = nthr; gridDim.x = (unsigned)gx; gridDim.y = (unsigned)gy;
    block.block_barrier.init(nthr);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx) {
            // LDS is NOT zero on the device when a workgroup starts: poison it (fp16/fp32 NaN patterns) so that a
            // kernel relying on stale or zero LDS fails here too
            std::memset(lds.data(), 0xFF, lds.size());
            std::vector<TA> ta(nthr);
            for (int t = 0; t < nthr; ++t) ta[t] = TA{&fn, t, bx, by};
            emu_run_threads(nthr, tmain, ta.data(), sizeof(ta[0]), 1 << 20);
        }
}

// L-layout fp16 tensor [n][P][C] -> fp32 natural NHWC
void unpack_act(const _Float16* src, float* dst, long n_pix, int C) {
    for (long p = 0; p < n_pix; ++p)
        for (int c = 0; c < C; ++c) {
            const int ct = c / 16, g = (c % 16) / 4, r = c % 4;
            dst[p * C + c] = (float)src[p * C + g * (C / 4) + 4 * ct + r];
        }
}

}  // namespace

extern "C" {

void emu_reid_set_stage1_handover(int on) { g_stage1_handover = on; }
void emu_reid_set_head_per_crop(int on) { g_head_per_crop = on; }

// crops: normalised fp32 NHWC (n, 256, 128, 3).  stage_out[k] (may be null) receives the fp32 NHWC
// activation after: 0 stem+maxpool, 1..2 stage-1 blocks, 3 transition, 4..5 blocks, 6 transition,
// 7..8 blocks.  feats: (n, 512) L2-normalised.
// Fused crop+resize+stem kernel alone: frame (H, W, 3) uint8 BGR + boxes (n, 4) -> stem output fp32 NHWC (n, 2048, 16)
int emu_stem_from_frame(const float* blob, long n_floats, const uint8_t* frame, int W, int H, const float* boxes, int n,
                        float* stage0) {
    using namespace bm;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(blob);
    if (hdr[0] != REID_MAGIC || hdr[1] != 16) return -1;
    const int ch[4] = {hdr[1], hdr[2], hdr[3], hdr[4]};
    const OsnetLayout L = make_osnet_layout(ch, hdr[5]);
    if (n_floats != REID_HEADER_INTS + L.total) return -2;
    const float* w = blob + REID_HEADER_INTS;
    std::vector<uint8_t> wst;
    pack_stem(w + L.stem_w, w + L.stem_b, wst);
    float lut[768];
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float a = (float)v / 255.0f;
            volatile float b = a - mean[c];
            lut[c * 256 + v] = b / stdv[c];
        }
    std::vector<_Float16> A((size_t)n * 2048 * 16);
    std::vector<int> streams(n, 0);
    const uint8_t* frames[1] = {frame};
    {
        const uint8_t* const* fr = frames; const int* cs = streams.data(); _Float16* out = A.data();
        const unsigned char* wp = wst.data(); const float* lp = lut;
        launch(n, 1, 512, [=]() { k_stem_resize_fused(fr, cs, boxes, 4, W, H, lp, out, wp, nullptr); });
    }
    unpack_act(A.data(), stage0, (long)n * 2048, 16);
    return 0;
}

// The standalone crop kernel (k_crop_resize<float>): frame (H, W, 3) uint8 BGR + boxes (n, 4) -> normalised fp32 NHWC
// crops (n, 256, 128, 3); pad = 0 "resize", 1 "resize_pad".
int emu_crop_resize(const uint8_t* frame, int W, int H, const float* boxes, int n, int pad, float* out) {
    using namespace bm;
    float lut[768];
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float a = (float)v / 255.0f;
            volatile float b = a - mean[c];
            lut[c * 256 + v] = b / stdv[c];
        }
    std::vector<int> streams(n, 0);
    const uint8_t* frames[1] = {frame};
    const uint8_t* const* fr = frames; const int* cs = streams.data(); const float* lp = lut;
    launch(n, REID_IN_H / 16, REID_IN_W, [=]() { k_crop_resize<float>(fr, cs, boxes, 4, W, H, lp, out, 16, pad); });
    return 0;
}

// Oriented boxes: k_crop_resize_obb<float, false> with the per-box geometry [out_w, out_h, inverse map] the C ABI computes on the host
int emu_crop_resize_obb(const uint8_t* frame, int W, int H, const double* geo, int n, int pad, float* out) {
    using namespace bm;
    float lut[768];
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float a = (float)v / 255.0f;
            volatile float b = a - mean[c];
            lut[c * 256 + v] = b / stdv[c];
        }
    std::vector<int> streams(n, 0);
    const uint8_t* frames[1] = {frame};
    const uint8_t* const* fr = frames; const int* cs = streams.data(); const float* lp = lut;
    launch(n, REID_IN_H / 16, REID_IN_W, [=]() { k_crop_resize_obb<float, false>(fr, cs, geo, W, H, lp, out, 16, pad); });
    return 0;
}

// Head kernels alone on given stage-2 activations (fp32 natural NHWC, (n, 128, 128)): the batched head (16 crops per
// workgroup, FC on the matrix pipe) and the per-crop head of round 1, with an optional output-row map and device-style count.
int emu_head_pair(const float* blob, long n_floats, const float* act, int n, int count, const int* rows, float* feats_batched,
                  float* feats_per_crop) {
    using namespace bm;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(blob);
    if (hdr[0] != REID_MAGIC || hdr[1] != 16) return -1;
    const int ch[4] = {hdr[1], hdr[2], hdr[3], hdr[4]};
    const OsnetLayout L = make_osnet_layout(ch, hdr[5]);
    if (n_floats != REID_HEADER_INTS + L.total) return -2;
    const float* w = blob + REID_HEADER_INTS;
    std::vector<_Float16> A((size_t)n * 128 * 128);
    for (long p = 0; p < (long)n * 128; ++p)
        for (int c = 0; c < 128; ++c) {
            const int ct = c / 16, g = (c % 16) / 4, r = c % 4;
            A[p * 128 + g * 32 + 4 * ct + r] = (_Float16)act[p * 128 + c];
        }
    std::vector<uint8_t> w5, wfc;
    pack_pointwise(w + L.conv5_w, w + L.conv5_b, 128, 128, w5);
    pack_fc(w + L.fc_w, w + L.fc_b, 512, 128, wfc);
    const _Float16* in = A.data(); const unsigned char* p5 = w5.data(); const unsigned char* pf = wfc.data();
    const int* cnt = count >= 0 ? &count : nullptr;
    launch(n, 1, 128, [=]() { k_head_fused<128, 512>(in, p5, pf, feats_per_crop, rows, cnt); });
    launch((n + HEAD_NB - 1) / HEAD_NB, 1, 256, [=]() { k_head_batched<128, 512>(in, p5, pf, feats_batched, rows, cnt, n); });
    return 0;
}

int emu_reid_forward(const float* blob, long n_floats, const float* crops, int n, float* feats, float** stage_out) {
    using namespace bm;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(blob);
    if (hdr[0] != REID_MAGIC || hdr[1] != 16) return -1;
    const int ch[4] = {hdr[1], hdr[2], hdr[3], hdr[4]};
    const OsnetLayout L = make_osnet_layout(ch, hdr[5]);
    if (n_floats != REID_HEADER_INTS + L.total) return -2;
    const float* w = blob + REID_HEADER_INTS;
    // inputs: fp16 RGBX with a 3-pixel zero border
    std::vector<_Float16> cx((size_t)n * STEM_ROWS * STEM_COLS * 4, (_Float16)0.f);
    for (int i = 0; i < n; ++i)
        for (int y = 0; y < 256; ++y)
            for (int x = 0; x < 128; ++x)
                for (int c = 0; c < 3; ++c)
                    cx[(((size_t)i * STEM_ROWS + y + 3) * STEM_COLS + x + 3) * 4 + c] =
                        (_Float16)crops[(((size_t)i * 256 + y) * 128 + x) * 3 + c];
    std::vector<_Float16> A((size_t)n * 2048 * 64), B((size_t)n * 2048 * 64);
    std::vector<uint8_t> wst;
    pack_stem(w + L.stem_w, w + L.stem_b, wst);
    {
        const _Float16* in = cx.data(); _Float16* out = A.data(); const unsigned char* wp = wst.data();
        launch(n, 1, 512, [=]() { k_stem_fused(in, out, wp, nullptr); });
    }
    if (stage_out && stage_out[0]) unpack_act(A.data(), stage_out[0], (long)n * 2048, 16);
    _Float16* cur = A.data();
    _Float16* nxt = B.data();
    int dump = 1;
    std::vector<_Float16> x1s((size_t)n * 2048 * 16), x2s((size_t)n * 2048 * 16);
    _Float16* x1p = x1s.data();
    _Float16* x2p = x2s.data();
    // `trans` >= 0: the stage's transition is fused into the block (the product configuration); the block's own
    // output then never exists in memory and its dump slot is skipped.
    auto run_block = [&](int bi, auto kernel, int stage, int cin, int down, int nthr, int P, int cout, int trans) {
        const BlkPack bp = make_blk_pack(stage, cin, down);
        std::vector<uint8_t> wb, wt;
        pack_osblock(w, L.block[bi], bp, wb);
        if (trans >= 0) pack_pointwise(w + L.trans_w[trans], w + L.trans_b[trans], cout, cout, wt, 0.25f);
        const _Float16* in = cur; _Float16* out = nxt; const unsigned char* wp = wb.data();
        const unsigned char* wtp = trans >= 0 ? wt.data() : nullptr;
        launch(n, 1, nthr, [=]() { kernel(in, out, wp, bp, nullptr, x1p, wtp, BlkLink{}); });
        std::swap(cur, nxt);
        if (trans >= 0) { ++dump; P /= 4; }
        if (stage_out && stage_out[dump]) unpack_act(cur, stage_out[dump], (long)n * P, cout);
        ++dump;
    };
    {   // stage 0 as the engine runs it: EMIT (block 1) -> RECON (block 2); block 1's 64-channel output is never stored
        const BlkPack bp0 = make_blk_pack(0, 16, 1), bp1 = make_blk_pack(0, 64, 0);
        std::vector<uint8_t> wb0, wb1, wt;
        pack_osblock(w, L.block[0], bp0, wb0);
        pack_osblock(w, L.block[1], bp1, wb1);
        pack_pointwise(w + L.trans_w[0], w + L.trans_b[0], 64, 64, wt, 0.25f);
        const _Float16* in = cur; _Float16* out = nxt;
        const unsigned char *w0 = wb0.data(), *w1 = wb1.data(), *wtp = wt.data();
        const BlkLink emit{w1, bp1.conv1_a, bp1.conv1_b, 0, x2p}, recon{w0, bp0.conv3_a, bp0.conv3_b, bp0.down_a, x2p};
        launch(n, 1, 64 * Geo<0>::NWAVES, [=]() { k_osblock<0, 16, true, false, true, false>(in, nullptr, w0, bp0, nullptr, x1p, nullptr, emit); });
        launch(n, 1, 64 * Geo<0>::NWAVES, [=]() { k_osblock<0, 64, false, true, false, true>(in, out, w1, bp1, nullptr, x1p, wtp, recon); });
        std::swap(cur, nxt);
        dump = 3;
        if (stage_out && stage_out[dump]) unpack_act(cur, stage_out[dump], (long)n * 512, 64);
        ++dump;
    }
    if (g_stage1_handover) {   // stage 1 the same way (engine: BM_STAGE1_HANDOVER)
        const BlkPack bp0 = make_blk_pack(1, 64, 1), bp1 = make_blk_pack(1, 96, 0);
        std::vector<uint8_t> wb0, wb1, wt;
        pack_osblock(w, L.block[2], bp0, wb0);
        pack_osblock(w, L.block[3], bp1, wb1);
        pack_pointwise(w + L.trans_w[1], w + L.trans_b[1], 96, 96, wt, 0.25f);
        const _Float16* in = cur; _Float16* out = nxt;
        const unsigned char *w0 = wb0.data(), *w1 = wb1.data(), *wtp = wt.data();
        const BlkLink emit{w1, bp1.conv1_a, bp1.conv1_b, 0, x2p}, recon{w0, bp0.conv3_a, bp0.conv3_b, bp0.down_a, x2p};
        launch(n, 1, 64 * Geo<1>::NWAVES, [=]() { k_osblock<1, 64, true, false, true, false>(in, nullptr, w0, bp0, nullptr, x1p, nullptr, emit); });
        launch(n, 1, 64 * Geo<1>::NWAVES, [=]() { k_osblock<1, 96, false, true, false, true>(in, out, w1, bp1, nullptr, x1p, wtp, recon); });
        std::swap(cur, nxt);
        dump = 6;
        if (stage_out && stage_out[dump]) unpack_act(cur, stage_out[dump], (long)n * 128, 96);
        ++dump;
    } else {
        run_block(2, k_osblock<1, 64, true, false>, 1, 64, 1, 64 * Geo<1>::NWAVES, 512, 96, -1);
        run_block(3, k_osblock<1, 96, false, true>, 1, 96, 0, 64 * Geo<1>::NWAVES, 512, 96, 1);
    }
    run_block(4, k_osblock<2, 96, true, false>, 2, 96, 1, 64 * Geo<2>::NWAVES, 128, 128, -1);
    run_block(5, k_osblock<2, 128, false, false>, 2, 128, 0, 64 * Geo<2>::NWAVES, 128, 128, -1);
    std::vector<uint8_t> w5, wfc;
    pack_pointwise(w + L.conv5_w, w + L.conv5_b, 128, 128, w5);
    pack_fc(w + L.fc_w, w + L.fc_b, 512, 128, wfc);
    {
        const _Float16* in = cur; const unsigned char* p5 = w5.data(); const unsigned char* pf = wfc.data();
        if (g_head_per_crop) launch(n, 1, 128, [=]() { k_head_fused<128, 512>(in, p5, pf, feats, nullptr, nullptr); });
        else launch((n + HEAD_NB - 1) / HEAD_NB, 1, 256, [=]() { k_head_batched<128, 512>(in, p5, pf, feats, nullptr, nullptr, n); });
    }
    return 0;
}


// fp32-grade family (reid_hp.hpp, mode 2) as the engine runs it: frame + boxes -> (hi, lo) crops -> stem -> EMIT / RECON stage 0 ->
// stage 1 -> stage 2 -> head.  stage_out[k] (may be null) receives hi + lo as fp32 natural NHWC after: 0 stem+maxpool, 3 stage-0
// transition, 4 conv3.0, 6 stage-1 transition, 7 conv4.0, 8 conv4.1.
int emu_reid_forward_hp(const float* blob, long n_floats, const uint8_t* frame, int W, int H, const float* boxes, int n, float* feats,
                        float** stage_out, int fused_stem) {
    using namespace bm;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(blob);
    if (hdr[0] != REID_MAGIC || hdr[1] != 16) return -1;
    const int ch[4] = {hdr[1], hdr[2], hdr[3], hdr[4]};
    const OsnetLayout L = make_osnet_layout(ch, hdr[5]);
    if (n_floats != REID_HEADER_INTS + L.total) return -2;
    const float* w = blob + REID_HEADER_INTS;
    float lut[768];
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int c = 0; c < 3; ++c)
        for (int v = 0; v < 256; ++v) {
            volatile float a = (float)v / 255.0f;
            volatile float b = a - mean[c];
            lut[c * 256 + v] = b / stdv[c];
        }
    std::vector<_Float16> cxh((size_t)n * STEM_ROWS * STEM_COLS * 4, (_Float16)0.f), cxl(cxh.size(), (_Float16)0.f);
    std::vector<int> streams(n, 0);
    const uint8_t* frames[1] = {frame};
    {
        const uint8_t* const* fr = frames; const int* cs = streams.data(); const float* lp = lut;
        _Float16 *oh = cxh.data(), *ol = cxl.data();
        launch(n, REID_IN_H / 16, REID_IN_W, [=]() { k_crop_resize_rgbx_hl(fr, cs, boxes, 4, W, H, lp, oh, ol, 16, nullptr, 0); });
    }
    std::vector<_Float16> Ah((size_t)n * 2048 * 32), Al(Ah.size()), Bh(Ah.size()), Bl(Ah.size());
    std::vector<float> x1s((size_t)n * 2048 * 16), x2s((size_t)n * 2048 * 16);
    std::vector<uint8_t> wst;
    if (fused_stem) {       // crop + resize + stem in one kernel on raw pixel values (normalisation folded into the weights)
        pack_stem_hp_fused(w + L.stem_w, w + L.stem_b, mean, stdv, wst);
        const uint8_t* const* fr = frames; const int* cs = streams.data();
        _Float16 *oh = Ah.data(), *ol = Al.data(); const unsigned char* wp = wst.data();
        launch(n, 1, 512, [=]() { k_stem_resize_fused_hp(fr, cs, boxes, 4, W, H, oh, ol, wp, nullptr); });
    } else {
        pack_stem_hp(w + L.stem_w, w + L.stem_b, wst);
        const _Float16 *ih = cxh.data(), *il = cxl.data(); _Float16 *oh = Ah.data(), *ol = Al.data(); const unsigned char* wp = wst.data();
        launch(n, 1, 512, [=]() { k_stem_hp(ih, il, oh, ol, wp, nullptr); });
    }
    auto dump = [&](int slot, const _Float16* h, const _Float16* l, long n_pix, int C) {
        if (!stage_out || !stage_out[slot]) return;
        std::vector<float> a((size_t)n_pix * C), b((size_t)n_pix * C);
        unpack_act(h, a.data(), n_pix, C);
        unpack_act(l, b.data(), n_pix, C);
        for (size_t k = 0; k < a.size(); ++k) stage_out[slot][k] = a[k] + b[k];
    };
    dump(0, Ah.data(), Al.data(), (long)n * 2048, 16);
    static const int stage[6] = {0, 0, 1, 1, 2, 2}, cin[6] = {16, 64, 64, 96, 96, 128}, down[6] = {1, 0, 1, 0, 1, 0};
    BlkPackHP bp[6];
    std::vector<uint8_t> wb[6], wt[2];
    for (int b = 0; b < 6; ++b) { bp[b] = make_blk_pack_hp(stage[b], cin[b], down[b]); pack_osblock_hp(w, L.block[b], bp[b], wb[b]); }
    pack_pointwise_hp(w + L.trans_w[0], w + L.trans_b[0], 64, 64, wt[0], 0.25f);
    pack_pointwise_hp(w + L.trans_w[1], w + L.trans_b[1], 96, 96, wt[1], 0.25f);
    float* x1p = x1s.data();
    auto blk = [&](auto kernel, const _Float16* ih, const _Float16* il, _Float16* oh, _Float16* ol, int b, const unsigned char* wtr, BlkLinkHP link) {
        const unsigned char* wp = wb[b].data(); const BlkPackHP bpb = bp[b];
        const int nthr = 64 * (stage[b] == 0 ? GeoHP<0>::NWAVES : (stage[b] == 1 ? GeoHP<1>::NWAVES : GeoHP<2>::NWAVES));
        // EMU_HP_PERSIST_GRID = G > 0: the persistent launch form (BlkLinkHP::n_crops): G workgroups loop over the crops
        const char* pg = std::getenv("EMU_HP_PERSIST_GRID");
        const int grid = pg && std::atoi(pg) > 0 ? (std::atoi(pg) < n ? std::atoi(pg) : n) : n;
        if (grid != n || (pg && std::atoi(pg) > 0)) link.n_crops = n;
        launch(grid, 1, nthr, [=]() { kernel(ih, il, oh, ol, wp, bpb, nullptr, x1p, wtr, link); });
    };
    blk(k_osblock_hp<0, 16, true, false, true, false>, Ah.data(), Al.data(), nullptr, nullptr, 0, nullptr,
        BlkLinkHP{wb[1].data(), bp[1].conv1_a, bp[1].conv1_b, 0, x2s.data()});
    blk(k_osblock_hp<0, 64, false, true, false, true>, Ah.data(), Al.data(), Bh.data(), Bl.data(), 1, wt[0].data(),
        BlkLinkHP{wb[0].data(), bp[0].conv3_a, bp[0].conv3_b, bp[0].down_a, x2s.data()});
    dump(3, Bh.data(), Bl.data(), (long)n * 512, 64);
    blk(k_osblock_hp<1, 64, true, false>, Bh.data(), Bl.data(), Ah.data(), Al.data(), 2, nullptr, BlkLinkHP{});
    dump(4, Ah.data(), Al.data(), (long)n * 512, 96);
    blk(k_osblock_hp<1, 96, false, true>, Ah.data(), Al.data(), Bh.data(), Bl.data(), 3, wt[1].data(), BlkLinkHP{});
    dump(6, Bh.data(), Bl.data(), (long)n * 128, 96);
    blk(k_osblock_hp<2, 96, true, false>, Bh.data(), Bl.data(), Ah.data(), Al.data(), 4, nullptr, BlkLinkHP{});
    dump(7, Ah.data(), Al.data(), (long)n * 128, 128);
    blk(k_osblock_hp<2, 128, false, false>, Ah.data(), Al.data(), Bh.data(), Bl.data(), 5, nullptr, BlkLinkHP{});
    dump(8, Bh.data(), Bl.data(), (long)n * 128, 128);
    std::vector<uint8_t> w5, wfc;
    pack_pointwise_hp(w + L.conv5_w, w + L.conv5_b, 128, 128, w5);
    pack_fc_hp(w + L.fc_w, w + L.fc_b, 512, 128, wfc);
    {
        const _Float16 *ih = Bh.data(), *il = Bl.data(); const unsigned char* p5 = w5.data(); const unsigned char* pf = wfc.data();
        launch((n + HEAD_NB - 1) / HEAD_NB, 1, 256, [=]() { k_head_hp<128, 512>(ih, il, p5, pf, feats, nullptr, nullptr, n); });
    }
    return 0;
}

}  // extern "C"
