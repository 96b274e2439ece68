// OSNet weight blob layout shared by the Python packer (boxmot_amd/reid_weights.py)
// and the device engine.  All tensors fp32 with BatchNorm already folded
// (eval mode, eps = 1e-5; boxmot/reid/backbones/osnet.py:21-155):
//   conv + BN            -> W' = W * g/sqrt(var+eps),  b' = beta - mean * g/sqrt(var+eps)
//   LightConv3x3         -> 1x1 (linear, kept as is) ; depthwise 3x3 with its BN folded in
//   Linear + BatchNorm1d -> folded the same way
// Blob = header (16 int32) followed by the tensors in this order:
//   stem   W[c0][7][7][3] (co,ky,kx,ci with ci in RGB order), b[c0]
//   for stage s in 0..2, block k in 0..1   (osnet.py:296-309)
//       conv1      W[mid][cin], b[mid]
//       branches a,b,c,d with 1,2,3,4 LightConv3x3 each (osnet.py:223-241), per LightConv:
//                  pw W[mid][mid], dw W[mid][3][3], b[mid]
//       gate       fc1 W[hid][mid], b[hid], fc2 W[mid][hid], b[mid]      (osnet.py:161-209)
//       conv3      W[cout][mid], b[cout]
//       downsample W[cout][cin], b[cout]       (only when cin != cout)
//     after block 1 of stages 0 and 1: transition W[cout][cout], b[cout]   (+ 2x2 avg pool)
//   conv5  W[c3][c3], b[c3]
//   fc     W[feat][c3], b[feat]
#pragma once

#include <cstdint>
#include <vector>

namespace bm {

constexpr int REID_MAGIC = 0x4f534e31;   // "OSN1"
constexpr int REID_HEADER_INTS = 16;
constexpr int REID_IN_H = 256, REID_IN_W = 128;

struct LightW { long pw, dw, b; };
struct BlockW {
    int cin, cout, mid, hid;
    long conv1_w, conv1_b;
    LightW light[10];          // a0, b0, b1, c0..c2, d0..d3
    long fc1_w, fc1_b, fc2_w, fc2_b;
    long conv3_w, conv3_b;
    long down_w, down_b;       // -1 when identity
};
struct OsnetLayout {
    int c[4];
    int feat;
    long stem_w, stem_b;
    BlockW block[6];
    long trans_w[2], trans_b[2];
    long conv5_w, conv5_b;
    long fc_w, fc_b;
    long total;
};

// `like`: take "this block has a downsample convolution" from another layout's blocks instead of from cin != cout (a zero-padded
// copy of a network can have cin == cout where the network itself projects, see osnet_pad_weights)
inline OsnetLayout make_osnet_layout(const int channels[4], int feat, const BlockW* like = nullptr) {
    OsnetLayout L;
    long off = 0;
    auto take = [&](long n) { long o = off; off += n; return o; };
    for (int i = 0; i < 4; ++i) L.c[i] = channels[i];
    L.feat = feat;
    L.stem_w = take((long)channels[0] * 7 * 7 * 3);
    L.stem_b = take(channels[0]);
    int cin = channels[0];
    for (int s = 0; s < 3; ++s) {
        const int cout = channels[s + 1];
        for (int k = 0; k < 2; ++k) {
            BlockW& B = L.block[s * 2 + k];
            B.cin = cin; B.cout = cout; B.mid = cout / 4; B.hid = B.mid / 16;
            B.conv1_w = take((long)B.mid * cin); B.conv1_b = take(B.mid);
            for (int l = 0; l < 10; ++l) {
                B.light[l].pw = take((long)B.mid * B.mid);
                B.light[l].dw = take((long)B.mid * 9);
                B.light[l].b = take(B.mid);
            }
            B.fc1_w = take((long)B.hid * B.mid); B.fc1_b = take(B.hid);
            B.fc2_w = take((long)B.mid * B.hid); B.fc2_b = take(B.mid);
            B.conv3_w = take((long)cout * B.mid); B.conv3_b = take(cout);
            if (like ? like[s * 2 + k].down_w >= 0 : cin != cout) { B.down_w = take((long)cout * cin); B.down_b = take(cout); }
            else { B.down_w = -1; B.down_b = -1; }
            cin = cout;
        }
        if (s < 2) { L.trans_w[s] = take((long)cout * cout); L.trans_b[s] = take(cout); }
    }
    L.conv5_w = take((long)channels[3] * channels[3]); L.conv5_b = take(channels[3]);
    L.fc_w = take((long)feat * channels[3]); L.fc_b = take(feat);
    L.total = off;
    return L;
}

// ---------------------------------------------------------------------------
// Zero-padded copy of a network for the matrix-pipe kernel families (osnet_wide.hpp, osnet_wide_hp.hpp), which are built for
// stems of 32 / 64 channels and middle widths of 32 / 64 / 96 / 128 (at most 64 in stage 0 and 96 in stage 1): osnet_x0_5
// (32, 128, 192, 256: middle 32 / 48 / 64) runs as (32, 128, 256, 256), osnet_x0_75 (48, 192, 288, 384: 48 / 72 / 96) as
// (64, 256, 384, 384) (boxmot/reid/backbones/osnet.py:503-530).  Every added channel has zero weights and a zero bias on both sides:
// it carries ReLU(0) = 0 through the block, its gate multiplies 0, and a sum that includes it adds an exact 0 -- the real channels
// compute what they computed before.
// ---------------------------------------------------------------------------
inline bool osnet_padded_channels(const OsnetLayout& L, int (&cp)[4]) {
    if (L.c[0] > 64) return false;
    cp[0] = L.c[0] <= 32 ? 32 : 64;
    for (int s = 0; s < 3; ++s) {
        if (L.c[s + 1] % 4) return false;
        const int mid = (L.c[s + 1] / 4 + 31) / 32 * 32;
        if (mid > (s == 0 ? 64 : (s == 1 ? 96 : 128))) return false;
        cp[s + 1] = 4 * mid;
    }
    return L.feat % 128 == 0 && L.feat <= 512;
}
inline OsnetLayout osnet_padded_layout(const OsnetLayout& L, const int (&cp)[4]) { return make_osnet_layout(cp, L.feat, L.block); }
inline std::vector<float> osnet_pad_weights(const float* w, const OsnetLayout& L, const OsnetLayout& Lp) {
    std::vector<float> o((size_t)Lp.total, 0.f);
    // [rows][cols] at `src` -> the top-left corner of [.][cols_p] at `dst`
    auto mat = [&](long src, long dst, int rows, int cols, int cols_p) {
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) o[(size_t)(dst + (long)r * cols_p + c)] = w[src + (long)r * cols + c];
    };
    mat(L.stem_w, Lp.stem_w, L.c[0], 7 * 7 * 3, 7 * 7 * 3);
    mat(L.stem_b, Lp.stem_b, 1, L.c[0], Lp.c[0]);
    for (int b = 0; b < 6; ++b) {
        const BlockW &A = L.block[b], &P = Lp.block[b];
        mat(A.conv1_w, P.conv1_w, A.mid, A.cin, P.cin); mat(A.conv1_b, P.conv1_b, 1, A.mid, P.mid);
        for (int l = 0; l < 10; ++l) {
            mat(A.light[l].pw, P.light[l].pw, A.mid, A.mid, P.mid);
            mat(A.light[l].dw, P.light[l].dw, A.mid, 9, 9);
            mat(A.light[l].b, P.light[l].b, 1, A.mid, P.mid);
        }
        mat(A.fc1_w, P.fc1_w, A.hid, A.mid, P.mid); mat(A.fc1_b, P.fc1_b, 1, A.hid, P.hid);
        mat(A.fc2_w, P.fc2_w, A.mid, A.hid, P.hid); mat(A.fc2_b, P.fc2_b, 1, A.mid, P.mid);
        mat(A.conv3_w, P.conv3_w, A.cout, A.mid, P.mid); mat(A.conv3_b, P.conv3_b, 1, A.cout, P.cout);
        if (A.down_w >= 0) { mat(A.down_w, P.down_w, A.cout, A.cin, P.cin); mat(A.down_b, P.down_b, 1, A.cout, P.cout); }
    }
    for (int s = 0; s < 2; ++s) {
        mat(L.trans_w[s], Lp.trans_w[s], L.c[s + 1], L.c[s + 1], Lp.c[s + 1]);
        mat(L.trans_b[s], Lp.trans_b[s], 1, L.c[s + 1], Lp.c[s + 1]);
    }
    mat(L.conv5_w, Lp.conv5_w, L.c[3], L.c[3], Lp.c[3]); mat(L.conv5_b, Lp.conv5_b, 1, L.c[3], Lp.c[3]);
    mat(L.fc_w, Lp.fc_w, L.feat, L.c[3], Lp.c[3]); mat(L.fc_b, Lp.fc_b, 1, L.feat, L.feat);
    return o;
}

}  // namespace bm
