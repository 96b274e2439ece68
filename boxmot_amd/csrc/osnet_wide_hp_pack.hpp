// Host-side weight preparation of the fp32-grade wide-OSNet kernel family (osnet_wide_hp_kernels.hpp): every matrix-pipe operand
// of an OSN1 blob split into fp16 (hi, lo) parts -- w = hi + lo, hi = fp16(w), lo = fp16(w - hi) -- and laid out the way its kernel
// reads it; every tensor starts on a 16-byte boundary.  Shared by WideOsnetHP (osnet_wide_hp.hpp) and the emulation harness
// (tests/host_emu/emu_wide_hp.cpp).
//   GEMM operands (conv1, conv3, downsample, transitions, conv5, FC): two planes [N][K], the k columns in the paired order of the
//   activation tensors (k_gemm_hp)
//   stem: A fragment pairs [ky][channel tile] (k_wide_stem_hp)
//   LightConv chains: ten records per block (k_chain_hp): 1x1 pairs [out tile][k-step] with the k-slots in accumulator order,
//   depthwise taps fp32 [channel tile][g][tap][4], bias fp32 [C]
//   biases fp32; a block with a downsample carries conv3's and the downsample's biases summed (one GEMM over two operand pairs)
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "reid_hp_pack.hpp"
#include "reid_layout.hpp"

namespace bm {

// "paired" storage order of the family's activation tensors (osnet_wide_hp_kernels.hpp): inside each block of 32 channels, logical
// channel 16 p + 4 g + r sits at position 8 g + 4 p + r
constexpr int hp_paired_pos(int c) { return (c & ~31) + 8 * ((c >> 2) & 3) + 4 * ((c >> 4) & 1) + (c & 3); }

struct GemmWHp { long wh = -1, wl = -1, bias = -1; int n = 0, k = 0; };
struct BlockHp {
    GemmWHp conv1, conv3, down;         // conv3.bias = conv3_b (+ down_b when the block has a downsample)
    long chain = -1;                    // ten records of chain_rec_bytes(mid)
};
struct WideHpPack {
    std::vector<uint8_t> data;
    long stem_a = -1, stem_b = -1;
    BlockHp block[6];
    GemmWHp trans[2], conv5, fc;
};

inline int chain_rec_bytes(int C) { return C * C * 4 + C * 9 * 4 + C * 4; }

inline WideHpPack wide_pack_hp(const float* w, const OsnetLayout& L) {
    WideHpPack P;
    auto take = [&](long bytes) {
        while (P.data.size() % 16) P.data.push_back(0);
        const long o = (long)P.data.size();
        P.data.resize(P.data.size() + (size_t)bytes, 0);
        return o;
    };
    auto planes = [&](long off, int n, int k, float scale = 1.0f) {
        GemmWHp gw;
        gw.n = n; gw.k = k;
        gw.wh = take((long)n * k * 2);
        gw.wl = take((long)n * k * 2);
        uint16_t* dh = reinterpret_cast<uint16_t*>(P.data.data() + gw.wh);
        uint16_t* dl = reinterpret_cast<uint16_t*>(P.data.data() + gw.wl);
        for (int r = 0; r < n; ++r)            // k columns in the order the consumed tensor is stored in
            for (int c = 0; c < k; ++c) split_hl(w[off + (long)r * k + c] * scale, dh[(long)r * k + hp_paired_pos(c)], dl[(long)r * k + hp_paired_pos(c)]);
        return gw;
    };
    auto f32s = [&](const float* src, const float* add, int n) {
        const long o = take((long)n * 4);
        float* d = reinterpret_cast<float*>(P.data.data() + o);
        for (int i = 0; i < n; ++i) d[i] = src[i] + (add ? add[i] : 0.f);
        return o;
    };
    // stem A pairs [ky][ct]: lane (co = 16 ct + (lane & 15), g), k-slot j -> tap kx = 2 g + (j >> 2), channel j & 3 of the RGBX pixel
    const int c0 = L.c[0], nct0 = c0 / 16;
    P.stem_a = take((long)7 * nct0 * HP_FRAG_PAIR);
    for (int ky = 0; ky < 7; ++ky)
        for (int ct = 0; ct < nct0; ++ct) {
            uint16_t* dh = reinterpret_cast<uint16_t*>(P.data.data() + P.stem_a + (long)(ky * nct0 + ct) * HP_FRAG_PAIR);
            uint16_t* dl = dh + 512;
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = 16 * ct + (lane & 15), kx = 2 * (lane >> 4) + (j >> 2), c = j & 3;
                    const float v = (kx < 7 && c < 3) ? w[L.stem_w + ((long)(co * 7 + ky) * 7 + kx) * 3 + c] : 0.f;
                    split_hl(v, dh[lane * 8 + j], dl[lane * 8 + j]);
                }
        }
    P.stem_b = f32s(w + L.stem_b, nullptr, c0);
    for (int b = 0; b < 6; ++b) {
        const BlockW& B = L.block[b];
        BlockHp& H = P.block[b];
        H.conv1 = planes(B.conv1_w, B.mid, B.cin);
        H.conv1.bias = f32s(w + B.conv1_b, nullptr, B.mid);
        const int C = B.mid, CT = C / 16, KS = C / 32, rec = chain_rec_bytes(C);
        H.chain = take(10L * rec);
        for (int l = 0; l < 10; ++l) {
            uint8_t* base = P.data.data() + H.chain + (long)l * rec;
            for (int co = 0; co < CT; ++co)
                for (int s = 0; s < KS; ++s)
                    pack_a_frag_hl(base + (long)(co * KS + s) * HP_FRAG_PAIR, w + B.light[l].pw, C, C, C, co,
                                   [&](int g, int j) { return 16 * (2 * s + (j >> 2)) + 4 * g + (j & 3); });
            float* dw = reinterpret_cast<float*>(base + (long)C * C * 4);          // [ct][g][tap][r], channel 16 ct + 4 g + r
            for (int ct = 0; ct < CT; ++ct)
                for (int g = 0; g < 4; ++g)
                    for (int tap = 0; tap < 9; ++tap)
                        for (int r = 0; r < 4; ++r) dw[((ct * 4 + g) * 9 + tap) * 4 + r] = w[B.light[l].dw + (long)(16 * ct + 4 * g + r) * 9 + tap];
            float* bs = dw + (long)C * 9;
            for (int c = 0; c < C; ++c) bs[c] = w[B.light[l].b + c];
        }
        H.conv3 = planes(B.conv3_w, B.cout, B.mid);
        if (B.down_w >= 0) {
            H.down = planes(B.down_w, B.cout, B.cin);
            H.conv3.bias = f32s(w + B.conv3_b, w + B.down_b, B.cout);
        } else
            H.conv3.bias = f32s(w + B.conv3_b, nullptr, B.cout);
    }
    for (int s = 0; s < 2; ++s) {
        P.trans[s] = planes(L.trans_w[s], L.c[s + 1], L.c[s + 1]);
        P.trans[s].bias = f32s(w + L.trans_b[s], nullptr, L.c[s + 1]);
    }
    P.conv5 = planes(L.conv5_w, L.c[3], L.c[3]);
    P.conv5.bias = f32s(w + L.conv5_b, nullptr, L.c[3]);
    P.fc = planes(L.fc_w, L.feat, L.c[3]);
    P.fc.bias = f32s(w + L.fc_b, nullptr, L.feat);
    while (P.data.size() % 16) P.data.push_back(0);
    return P;
}

// widths this kernel family takes (k_gemm_hp's feature tiles, k_chain_hp's instantiations): osnet_x1_0 and reduced test networks
inline bool wide_hp_supports(const OsnetLayout& L) {
    if ((L.c[0] != 32 && L.c[0] != 64) || L.feat % 128 != 0 || L.feat > 512) return false;
    for (int b = 0; b < 6; ++b) {
        const BlockW& B = L.block[b];
        if (B.cin % 32 || B.cout % 128 || B.mid % 32) return false;
        // k_chain_hp instantiations per stage (register file / LDS budget of a band's tensor): 64 x 32 images up to 64 middle channels,
        // 32 x 16 up to 96, 16 x 8 up to 128
        if (B.mid > (b < 2 ? 64 : (b < 4 ? 96 : 128))) return false;
    }
    return true;
}

}  // namespace bm
