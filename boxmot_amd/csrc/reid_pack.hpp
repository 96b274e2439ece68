// Weight packing for the fused fp16 MFMA ReID kernels (reid_fused.hpp).
//
// Activations between kernels are fp16, "lane-group-major" NHWC:
//   pixel p of a crop, C channels (C % 16 == 0): channel c = 16*ct + 4*g + r
//   (ct = 16-channel tile, g = lane>>4 of the MFMA wave, r = accumulator row)
//   lives at half offset  (crop*P + p)*C + g*(C/4) + 4*ct + r.
// With that order the 4 accumulator rows a lane holds after an MFMA
// (D[co][px]: px = lane&15, co = 16*ct + 4*(lane>>4) + r) are contiguous in
// memory, and 8 consecutive halves are exactly the B fragment of a
// v_mfma_f32_16x16x32_f16 k-step (k-slot j of lane group g  <->  channel
// 16*(2*ks + (j>>2)) + 4*g + (j&3)).  Weights ("A" operands, rows = output
// channels) are pre-permuted to the same k order here, once, on the host.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "reid_layout.hpp"

namespace bm {

// fp32 -> fp16 bits, round to nearest even (host side, no compiler fp16 support needed)
inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t man = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - exp;
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)exp << 10) | (man >> 13);
    const uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(sign | h);
}

// byte offsets inside one packed OSBlock weight region
struct BlkPack {
    int stage, cin, down;        // stage 0..2
    int mid, kt, midp, cout, nct, hid;
    int kin_steps, kin_h;        // k-steps over the block input, halves per lane per fragment (4: K=16, 8: K=32)
    int mid_h;                   // halves per lane of a mid-channel fragment (4 or 8)
    long conv1_a, conv1_b;       // [ks][ct] fragments, fp32 bias[midp]
    long light0, light_bytes;    // per light: pw fragments [ct], dw fp16 [ct][g][tap][4], bias fp32 [midp]
    long light_pw, light_dw, light_b;   // offsets inside one light record
    long fc1_w, fc1_b, fc2_w, fc2_b;
    long conv3_a, conv3_b;       // [nct] fragments, fp32 bias[cout] (conv3 + downsample biases)
    long down_a;                 // [nct][ks] fragments
    long dense0, dense_bytes;    // stage 0 only: per light, the 1x1 and the depthwise 3x3 composed into one dense 3x3 (pack_light_dense)
    long total;
};
constexpr long DENSE_LIGHT_BYTES = 3 * 1024 + 3 * 512;

inline BlkPack make_blk_pack(int stage, int cin, int down) {
    static const int MID[3] = {16, 24, 32}, COUT[3] = {64, 96, 128};
    BlkPack b{};
    b.stage = stage; b.cin = cin; b.down = down;
    b.mid = MID[stage]; b.kt = (b.mid + 15) / 16; b.midp = 16 * b.kt; b.cout = COUT[stage]; b.nct = b.cout / 16;
    b.hid = b.mid / 16;
    b.kin_steps = cin == 16 ? 1 : cin / 32;
    b.kin_h = cin == 16 ? 4 : 8;
    b.mid_h = b.kt == 1 ? 4 : 8;
    long off = 0;
    auto take = [&](long n) { long o = off; off += (n + 15) / 16 * 16; return o; };
    const long frag_in = 64L * b.kin_h * 2, frag_mid = 64L * b.mid_h * 2;
    b.conv1_a = take(b.kin_steps * b.kt * frag_in);
    b.conv1_b = take(b.midp * 4);
    b.light_pw = 0;
    b.light_dw = b.kt * frag_mid;
    b.light_b = b.light_dw + b.midp * 9 * 2;
    b.light_bytes = (b.light_b + b.midp * 4 + 15) / 16 * 16;
    b.light0 = take(10 * b.light_bytes);
    b.fc1_w = take(b.hid * b.midp * 4); b.fc1_b = take(b.hid * 4);
    b.fc2_w = take(b.midp * b.hid * 4); b.fc2_b = take(b.midp * 4);
    b.conv3_a = take(b.nct * frag_mid);
    b.conv3_b = take(b.cout * 4);
    b.down_a = take(down ? b.nct * b.kin_steps * frag_in : 0);
    b.dense_bytes = DENSE_LIGHT_BYTES;
    b.dense0 = stage == 0 ? take(10 * DENSE_LIGHT_BYTES) : 0;
    b.total = off;
    return b;
}

// channel held in k-slot j of lane group g
inline int chan_mid_slot(int kt, int g, int j) { return kt == 1 ? 4 * g + j : 16 * (j >> 2) + 4 * g + (j & 3); }
inline int chan_mem_slot(int c_total, int ks, int g, int j) {
    return c_total == 16 ? 4 * g + j : 16 * (2 * ks + (j >> 2)) + 4 * g + (j & 3);
}

// One A fragment: rows co = 16*ct + (lane&15); `chan(g, j)` gives the input channel of k-slot j.
template <class ChanFn>
inline void pack_a_frag(uint16_t* dst, const float* W, int m_real, int k_real, int ld, int ct, int halves, ChanFn chan) {
    for (int lane = 0; lane < 64; ++lane) {
        const int co = 16 * ct + (lane & 15), g = lane >> 4;
        for (int j = 0; j < halves; ++j) {
            const int ci = chan(g, j);
            const float v = (co < m_real && ci < k_real) ? W[(long)co * ld + ci] : 0.f;
            dst[lane * halves + j] = f32_to_f16_bits(v);
        }
    }
}

inline void put_f32(std::vector<uint8_t>& buf, long off, const float* src, int n_real, int n_padded) {
    for (int i = 0; i < n_padded; ++i) {
        const float v = i < n_real ? src[i] : 0.f;
        std::memcpy(buf.data() + off + 4L * i, &v, 4);
    }
}

// LightConv3x3 = 1x1 (linear, mid -> mid) then depthwise 3x3 (+ folded BN): out[co][p] = sum_tap wd[co][tap] * sum_ci W1[co][ci] *
// x[ci][p + tap], i.e. ONE dense 3x3 convolution with weights wd[co][tap] * W1[co][ci].  For a 16-channel mid width its
// K = 9 x 16 runs on the matrix pipe as, per image row dy of the window, one K=32 fragment "P" over the taps (x - 1, x) and one
// K=16 fragment "R" over the tap x + 1 (the kernel keeps the pixel-shifted copies of a row as [left | centre] and [right]):
//   P_dy: lane (co = lane & 15, g), k-slot j -> tap dx = j >> 2, input channel 4 g + (j & 3)
//   R_dy: lane (co, g), k-slot j -> tap dx = 2, input channel 4 g + j
// The products are formed in fp32 and rounded to fp16 once.
inline void pack_light_dense(const float* w1 /*[16][16]*/, const float* wd /*[16][9]*/, uint8_t* dst) {
    for (int dy = 0; dy < 3; ++dy) {
        uint16_t* P = reinterpret_cast<uint16_t*>(dst + dy * 1024);
        uint16_t* R = reinterpret_cast<uint16_t*>(dst + 3 * 1024 + dy * 512);
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 15, g = lane >> 4;
            for (int j = 0; j < 8; ++j)
                P[lane * 8 + j] = f32_to_f16_bits(wd[co * 9 + dy * 3 + (j >> 2)] * w1[co * 16 + 4 * g + (j & 3)]);
            for (int j = 0; j < 4; ++j)
                R[lane * 4 + j] = f32_to_f16_bits(wd[co * 9 + dy * 3 + 2] * w1[co * 16 + 4 * g + j]);
        }
    }
}

// Pack one OSBlock from the folded fp32 blob (reid_layout.hpp offsets) into `out`.
inline void pack_osblock(const float* w, const BlockW& B, const BlkPack& P, std::vector<uint8_t>& out) {
    out.assign((size_t)P.total, 0);
    auto frag_in = [&](long off, const float* W, int m_real, int ct, int ks) {
        pack_a_frag(reinterpret_cast<uint16_t*>(out.data() + off), W, m_real, P.cin, P.cin, ct, P.kin_h,
                    [&](int g, int j) { return chan_mem_slot(P.cin, ks, g, j); });
    };
    auto frag_mid = [&](long off, const float* W, int m_real, int ct) {
        pack_a_frag(reinterpret_cast<uint16_t*>(out.data() + off), W, m_real, P.mid, P.mid, ct, P.mid_h,
                    [&](int g, int j) { return chan_mid_slot(P.kt, g, j); });
    };
    const long fin = 64L * P.kin_h * 2, fmid = 64L * P.mid_h * 2;
    for (int ks = 0; ks < P.kin_steps; ++ks)
        for (int ct = 0; ct < P.kt; ++ct) frag_in(P.conv1_a + (ks * P.kt + ct) * fin, w + B.conv1_w, P.mid, ct, ks);
    put_f32(out, P.conv1_b, w + B.conv1_b, P.mid, P.midp);
    for (int l = 0; l < 10; ++l) {
        const long base = P.light0 + l * P.light_bytes;
        for (int ct = 0; ct < P.kt; ++ct) frag_mid(base + P.light_pw + ct * fmid, w + B.light[l].pw, P.mid, ct);
        {   // depthwise taps as fp16, grouped per lane group: [ct][g][tap][r] with channel 16ct + 4g + r
            uint16_t* dw = reinterpret_cast<uint16_t*>(out.data() + base + P.light_dw);
            for (int ct = 0; ct < P.kt; ++ct)
                for (int g = 0; g < 4; ++g)
                    for (int tap = 0; tap < 9; ++tap)
                        for (int r = 0; r < 4; ++r) {
                            const int c = 16 * ct + 4 * g + r;
                            const float v = c < P.mid ? w[B.light[l].dw + (long)c * 9 + tap] : 0.f;
                            dw[((ct * 4 + g) * 9 + tap) * 4 + r] = f32_to_f16_bits(v);
                        }
        }
        put_f32(out, base + P.light_b, w + B.light[l].b, P.mid, P.midp);
        if (P.stage == 0) pack_light_dense(w + B.light[l].pw, w + B.light[l].dw, out.data() + P.dense0 + l * P.dense_bytes);
    }
    // gate: fc1 [hid][midp] (padded columns zero), fc2 [midp][hid]
    for (int h = 0; h < P.hid; ++h) put_f32(out, P.fc1_w + 4L * h * P.midp, w + B.fc1_w + (long)h * P.mid, P.mid, P.midp);
    put_f32(out, P.fc1_b, w + B.fc1_b, P.hid, P.hid);
    put_f32(out, P.fc2_w, w + B.fc2_w, P.mid * P.hid, P.midp * P.hid);
    put_f32(out, P.fc2_b, w + B.fc2_b, P.mid, P.midp);
    for (int ct = 0; ct < P.nct; ++ct) frag_mid(P.conv3_a + ct * fmid, w + B.conv3_w, P.cout, ct);
    std::vector<float> bias(w + B.conv3_b, w + B.conv3_b + P.cout);
    if (P.down) {
        for (int i = 0; i < P.cout; ++i) bias[i] += w[B.down_b + i];
        for (int ct = 0; ct < P.nct; ++ct)
            for (int ks = 0; ks < P.kin_steps; ++ks)
                frag_in(P.down_a + (ct * P.kin_steps + ks) * fin, w + B.down_w, P.cout, ct, ks);
    }
    put_f32(out, P.conv3_b, bias.data(), P.cout, P.cout);
}

// 1x1 conv C -> M over an L-layout tensor: fragments [ct][ks] (x32) + fp32 bias[M]
// `scale` (a power of two) multiplies weights and bias: the stage transitions fold their 2x2 average's 1/4 into the
// conv (ReLU is positively homogeneous), which is exact in fp16/fp32
inline void pack_pointwise(const float* W, const float* bias, int M, int C, std::vector<uint8_t>& out, float scale = 1.0f) {
    const int nct = M / 16, ks_n = C / 32;
    out.assign((size_t)nct * ks_n * 1024 + (size_t)M * 4, 0);
    std::vector<float> ws(W, W + (size_t)M * C), bs(bias, bias + M);
    for (auto& v : ws) v *= scale;
    for (auto& v : bs) v *= scale;
    for (int ct = 0; ct < nct; ++ct)
        for (int ks = 0; ks < ks_n; ++ks)
            pack_a_frag(reinterpret_cast<uint16_t*>(out.data() + ((long)ct * ks_n + ks) * 1024), ws.data(), M, C, C, ct, 8,
                        [&](int g, int j) { return chan_mem_slot(C, ks, g, j); });
    std::memcpy(out.data() + (size_t)nct * ks_n * 1024, bs.data(), (size_t)M * 4);
}

// stem 7x7/2: seven fragments (one per ky), k-slot j of group g = input pixel 2*cx + 2g + (j>>2)
// relative to the window start (kx = 2g + (j>>2)), colour j&3 (RGBX, X weight 0); + fp32 bias[16]
inline void pack_stem(const float* W /*[16][7][7][3]*/, const float* bias, std::vector<uint8_t>& out) {
    out.assign(7 * 1024 + 64, 0);
    for (int ky = 0; ky < 7; ++ky) {
        uint16_t* dst = reinterpret_cast<uint16_t*>(out.data() + ky * 1024);
        for (int lane = 0; lane < 64; ++lane) {
            const int co = lane & 15, g = lane >> 4;
            for (int j = 0; j < 8; ++j) {
                const int kx = 2 * g + (j >> 2), c = j & 3;
                const float v = (kx < 7 && c < 3) ? W[((co * 7 + ky) * 7 + kx) * 3 + c] : 0.f;
                dst[lane * 8 + j] = f32_to_f16_bits(v);
            }
        }
    }
    std::memcpy(out.data() + 7 * 1024, bias, 64);
}

// head FC [feat][C] with input channels in L-layout memory order -> fp16 [feat][C] (position-major), fp32 bias
inline void pack_fc(const float* W, const float* bias, int feat, int C, std::vector<uint8_t>& out) {
    out.assign((size_t)feat * C * 2 + (size_t)feat * 4, 0);
    uint16_t* dst = reinterpret_cast<uint16_t*>(out.data());
    for (int f = 0; f < feat; ++f)
        for (int c = 0; c < C; ++c) {
            const int ct = c / 16, g = (c % 16) / 4, r = c % 4;
            dst[(long)f * C + g * (C / 4) + 4 * ct + r] = f32_to_f16_bits(W[(long)f * C + c]);
        }
    std::memcpy(out.data() + (size_t)feat * C * 2, bias, (size_t)feat * 4);
}

}  // namespace bm
