// Weight packing for the fp32-grade fused ReID kernels (reid_hp.hpp, ReID "mode 2").
//
// Every 1x1 convolution of OSNet-x0.25 (osnet.py:212-260, 380-405) runs on the fp16 matrix pipe with BOTH operands carried as
// an fp16 (hi, lo) pair: w = hi + lo with hi = fp16(w), lo = fp16(w - hi) -- 22 significant bits, products exact in the fp32
// accumulator.  A K = 32 product tile is three MFMAs (Wh.xh + Wh.xl + Wl.xh; the dropped Wl.xl term is 2^-22 relative); a
// K = 16 layer fills the K = 32 shape with the (hi, lo) pair of the SAME 16 channels, B = [xh | xl], against the duplicated
// fragments A = [Wh | Wh] and [Wl | Wl]: two MFMAs give all four terms.  Only the v_mfma_f32_16x16x32_f16 shape is used.
// Weights are split here, once, on the host; activations are split in registers where they are produced.
//
// Activations between kernels: two fp16 planes (hi, lo), each in the lane-group-major NHWC layout of reid_pack.hpp, so a
// consumer's B fragments are plain 16-byte loads from the two planes.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "reid_layout.hpp"
#include "reid_pack.hpp"

namespace bm {

inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) x = sign;
        else {          // subnormal: normalise
            int e = -1;
            do { ++e; man <<= 1; } while ((man & 0x400u) == 0);
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) x = sign | 0x7f800000u | (man << 13);
    else x = sign | ((exp - 15 + 127) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

// (hi, lo) fp16 parts of w
inline void split_hl(float w, uint16_t& hi, uint16_t& lo) {
    hi = f32_to_f16_bits(w);
    lo = f32_to_f16_bits(w - f16_bits_to_f32(hi));
}

// One pair of A fragments (hi then lo, 1 KiB each): rows co = 16 ct + (lane & 15), 8 k-slots per lane; chan(g, j) = input channel of slot j
template <class ChanFn>
inline void pack_a_frag_hl(uint8_t* dst, const float* W, int m_real, int k_real, int ld, int ct, ChanFn chan, float scale = 1.0f) {
    uint16_t* dh = reinterpret_cast<uint16_t*>(dst);
    uint16_t* dl = reinterpret_cast<uint16_t*>(dst + 1024);
    for (int lane = 0; lane < 64; ++lane) {
        const int co = 16 * ct + (lane & 15), g = lane >> 4;
        for (int j = 0; j < 8; ++j) {
            const int ci = chan(g, j);
            const float v = (co < m_real && ci < k_real) ? W[(long)co * ld + ci] * scale : 0.f;
            split_hl(v, dh[lane * 8 + j], dl[lane * 8 + j]);
        }
    }
}

constexpr long HP_FRAG_PAIR = 2048;      // bytes of one (hi, lo) fragment pair

// Stage 0 (mid width 16): the linear 1x1 of a LightConv (osnet.py:127-155) on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, four per
// 16-pixel tile) instead of a (hi, lo) split of the activations + two fp16 MFMAs.  The block kernels are bound by vector-instruction
// ISSUE (profiles/r6_valu_lds_microbench.txt: LDS reads and MFMAs aside, a SIMD retires one wave-instruction per ~4.4 cycles and its two
// waves' depthwise taps, ReLUs and splits fill it), the matrix pipe is ~20 % busy: the fp32 form trades the split's 8 vector instructions
// per tile and layer for 95 more matrix-pipe cycles that run beside the next rows' taps.  The layer's weights are then stored as fp32
// A fragments in the record's `light_pw` region: [lane] f4 = W[out = lane & 15][in = 4 (lane >> 4) + 0..3]  (MFMA kk takes component kk;
// its k index is the lane group, i.e. input channel 4 g + kk -- the channel component kk of the lane's activation f4 is the B operand).
#ifndef BM_HP_PW32
#define BM_HP_PW32 0
#endif

struct BlkPackHP {
    int stage, cin, down;
    int mid, kt, midp, cout, nct, hid, kin_steps;
    long conv1_a, conv1_b;                      // [ks][ct] pairs, fp32 bias[midp]
    long light0, light_bytes, light_pw, light_dw, light_b;      // per light: [ct] pairs, fp32 dw [ct][g][tap][4], fp32 bias[midp]
    long fc1_w, fc1_b, fc2_w, fc2_b;            // fp32 (as reid_pack.hpp)
    long conv3_a, conv3_b;                      // [nct] pairs, fp32 bias[cout] (conv3 + downsample biases)
    long down_a;                                // [nct][ks] pairs
    long total;
};

inline BlkPackHP make_blk_pack_hp(int stage, int cin, int down) {
    static const int MID[3] = {16, 24, 32}, COUT[3] = {64, 96, 128};
    BlkPackHP b{};
    b.stage = stage; b.cin = cin; b.down = down;
    b.mid = MID[stage]; b.kt = (b.mid + 15) / 16; b.midp = 16 * b.kt; b.cout = COUT[stage]; b.nct = b.cout / 16;
    b.hid = b.mid / 16;
    b.kin_steps = cin == 16 ? 1 : cin / 32;
    long off = 0;
    auto take = [&](long n) { long o = off; off += (n + 15) / 16 * 16; return o; };
    b.conv1_a = take((long)b.kin_steps * b.kt * HP_FRAG_PAIR);
    b.conv1_b = take(b.midp * 4);
    b.light_pw = 0;
    b.light_dw = b.kt * HP_FRAG_PAIR;
    b.light_b = b.light_dw + (long)b.midp * 9 * 4;
    b.light_bytes = (b.light_b + b.midp * 4 + 15) / 16 * 16;
    b.light0 = take(10 * b.light_bytes);
    b.fc1_w = take(b.hid * b.midp * 4); b.fc1_b = take(b.hid * 4);
    b.fc2_w = take(b.midp * b.hid * 4); b.fc2_b = take(b.midp * 4);
    b.conv3_a = take((long)b.nct * HP_FRAG_PAIR);
    b.conv3_b = take(b.cout * 4);
    b.down_a = take(down ? (long)b.nct * b.kin_steps * HP_FRAG_PAIR : 0);
    b.total = off;
    return b;
}

// k-slot -> channel maps.  Input tensors in memory (L-layout, C channels): K = 32 steps as reid_pack.hpp's chan_mem_slot; a
// 16-channel tensor (or a 16-channel mid width) is the duplicated form: slots j and j + 4 both hold channel 4 g + (j & 3).
inline int chan_in_hp(int c_total, int ks, int g, int j) { return c_total == 16 ? 4 * g + (j & 3) : 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3); }
inline int chan_mid_hp(int kt, int g, int j) { return kt == 1 ? 4 * g + (j & 3) : 16 * (j >> 2) + 4 * g + (j & 3); }

inline void pack_osblock_hp(const float* w, const BlockW& B, const BlkPackHP& P, std::vector<uint8_t>& out) {
    out.assign((size_t)P.total, 0);
    auto frag_in = [&](long off, const float* W, int m_real, int ct, int ks) {
        pack_a_frag_hl(out.data() + off, W, m_real, P.cin, P.cin, ct, [&](int g, int j) { return chan_in_hp(P.cin, ks, g, j); });
    };
    auto frag_mid = [&](long off, const float* W, int m_real, int ct) {
        pack_a_frag_hl(out.data() + off, W, m_real, P.mid, P.mid, ct, [&](int g, int j) { return chan_mid_hp(P.kt, g, j); });
    };
    for (int ks = 0; ks < P.kin_steps; ++ks)
        for (int ct = 0; ct < P.kt; ++ct) frag_in(P.conv1_a + (long)(ks * P.kt + ct) * HP_FRAG_PAIR, w + B.conv1_w, P.mid, ct, ks);
    put_f32(out, P.conv1_b, w + B.conv1_b, P.mid, P.midp);
    for (int l = 0; l < 10; ++l) {
        const long base = P.light0 + l * P.light_bytes;
        if (BM_HP_PW32 && P.stage == 0) {
            float* a32 = reinterpret_cast<float*>(out.data() + base + P.light_pw);
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) a32[lane * 4 + r] = w[B.light[l].pw + (long)(lane & 15) * P.mid + 4 * (lane >> 4) + r];
        } else
        for (int ct = 0; ct < P.kt; ++ct) frag_mid(base + P.light_pw + (long)ct * HP_FRAG_PAIR, w + B.light[l].pw, P.mid, ct);
        float* dw = reinterpret_cast<float*>(out.data() + base + P.light_dw);      // [ct][g][tap][r], channel 16 ct + 4 g + r
        for (int ct = 0; ct < P.kt; ++ct)
            for (int g = 0; g < 4; ++g)
                for (int tap = 0; tap < 9; ++tap)
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * ct + 4 * g + r;
                        dw[((ct * 4 + g) * 9 + tap) * 4 + r] = c < P.mid ? w[B.light[l].dw + (long)c * 9 + tap] : 0.f;
                    }
        put_f32(out, base + P.light_b, w + B.light[l].b, P.mid, P.midp);
    }
    for (int h = 0; h < P.hid; ++h) put_f32(out, P.fc1_w + 4L * h * P.midp, w + B.fc1_w + (long)h * P.mid, P.mid, P.midp);
    put_f32(out, P.fc1_b, w + B.fc1_b, P.hid, P.hid);
    put_f32(out, P.fc2_w, w + B.fc2_w, P.mid * P.hid, P.midp * P.hid);
    put_f32(out, P.fc2_b, w + B.fc2_b, P.mid, P.midp);
    for (int ct = 0; ct < P.nct; ++ct) frag_mid(P.conv3_a + (long)ct * HP_FRAG_PAIR, w + B.conv3_w, P.cout, ct);
    std::vector<float> bias(w + B.conv3_b, w + B.conv3_b + P.cout);
    if (P.down) {
        for (int i = 0; i < P.cout; ++i) bias[i] += w[B.down_b + i];
        for (int ct = 0; ct < P.nct; ++ct)
            for (int ks = 0; ks < P.kin_steps; ++ks)
                frag_in(P.down_a + (long)(ct * P.kin_steps + ks) * HP_FRAG_PAIR, w + B.down_w, P.cout, ct, ks);
    }
    put_f32(out, P.conv3_b, bias.data(), P.cout, P.cout);
}

// 1x1 conv C -> M over an L-layout tensor: pairs [ct][ks] + fp32 bias[M]; `scale` (a power of two) as in pack_pointwise
inline void pack_pointwise_hp(const float* W, const float* bias, int M, int C, std::vector<uint8_t>& out, float scale = 1.0f) {
    const int nct = M / 16, ks_n = C / 32;
    out.assign((size_t)nct * ks_n * HP_FRAG_PAIR + (size_t)M * 4, 0);
    for (int ct = 0; ct < nct; ++ct)
        for (int ks = 0; ks < ks_n; ++ks)
            pack_a_frag_hl(out.data() + ((long)ct * ks_n + ks) * HP_FRAG_PAIR, W, M, C, C, ct,
                           [&](int g, int j) { return chan_in_hp(C, ks, g, j); }, scale);
    float* b = reinterpret_cast<float*>(out.data() + (size_t)nct * ks_n * HP_FRAG_PAIR);
    for (int i = 0; i < M; ++i) b[i] = bias[i] * scale;
}

// stem 7x7/2: seven pairs (one per ky), k-slots as pack_stem; + fp32 bias[16]
inline void pack_stem_hp(const float* W /*[16][7][7][3]*/, const float* bias, std::vector<uint8_t>& out) {
    out.assign(7 * HP_FRAG_PAIR + 64, 0);
    for (int ky = 0; ky < 7; ++ky) {
        uint16_t* dh = reinterpret_cast<uint16_t*>(out.data() + ky * HP_FRAG_PAIR);
        uint16_t* dl = dh + 512;
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 15, g = lane >> 4;
            for (int j = 0; j < 8; ++j) {
                const int kx = 2 * g + (j >> 2), c = j & 3;
                const float v = (kx < 7 && c < 3) ? W[((co * 7 + ky) * 7 + kx) * 3 + c] : 0.f;
                split_hl(v, dh[lane * 8 + j], dl[lane * 8 + j]);
            }
        }
    }
    std::memcpy(out.data() + 7 * HP_FRAG_PAIR, bias, 64);
}


// stem for the fused crop + resize + stem kernel of the fp32-grade family (k_stem_resize_fused_hp): the operand is the RAW pixel
// [R, G, B, 1] (exact in fp16), the normalisation x = (v / 255 - mean_c) / std_c = a_c v + b_c is folded into the weights:
//   colour slots:  W[co][ky][kx][c] * a_c          fourth slot:  sum_c W[co][ky][kx][c] * b_c   (0 in the zero padding)
// per ky a (hi, lo) fragment pair; the lo fragment is stored times 2^11 (the kernel multiplies the operand by 2^-11, both exact),
// which keeps the residual in fp16's normal range.  + fp32 bias[16].
inline void pack_stem_hp_fused(const float* W /*[16][7][7][3]*/, const float* bias, const float mean[3], const float stdv[3],
                               std::vector<uint8_t>& out) {
    out.assign(14 * 1024 + 64, 0);
    for (int ky = 0; ky < 7; ++ky) {
        uint16_t* dh = reinterpret_cast<uint16_t*>(out.data() + (2 * ky) * 1024);
        uint16_t* dl = reinterpret_cast<uint16_t*>(out.data() + (2 * ky + 1) * 1024);
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 15, g = lane >> 4;
            for (int j = 0; j < 8; ++j) {
                const int kx = 2 * g + (j >> 2), c = j & 3;
                double v = 0.0;
                if (kx < 7) {
                    const float* w3 = W + ((co * 7 + ky) * 7 + kx) * 3;
                    if (c < 3) v = (double)w3[c] / (255.0 * (double)stdv[c]);
                    else for (int cc = 0; cc < 3; ++cc) v -= (double)w3[cc] * (double)mean[cc] / (double)stdv[cc];
                }
                const uint16_t hi = f32_to_f16_bits((float)v);
                dh[lane * 8 + j] = hi;
                dl[lane * 8 + j] = f32_to_f16_bits((float)((v - (double)f16_bits_to_f32(hi)) * 2048.0));
            }
        }
    }
    std::memcpy(out.data() + 14 * 1024, bias, 64);
}

// head FC [feat][C], input channels in L-layout memory order: fp16 hi [feat][C], fp16 lo [feat][C], fp32 bias
inline void pack_fc_hp(const float* W, const float* bias, int feat, int C, std::vector<uint8_t>& out) {
    out.assign((size_t)feat * C * 4 + (size_t)feat * 4, 0);
    uint16_t* dh = reinterpret_cast<uint16_t*>(out.data());
    uint16_t* dl = dh + (size_t)feat * C;
    for (int f = 0; f < feat; ++f)
        for (int c = 0; c < C; ++c) {
            const int ct = c / 16, g = (c % 16) / 4, r = c % 4;
            const long o = (long)f * C + g * (C / 4) + 4 * ct + r;
            split_hl(W[(long)f * C + c], dh[o], dl[o]);
        }
    std::memcpy(out.data() + (size_t)feat * C * 4, bias, (size_t)feat * 4);
}

}  // namespace bm
