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

inline OsnetLayout make_osnet_layout(const int channels[4], int feat) {
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
            if (cin != cout) { B.down_w = take((long)cout * cin); B.down_b = take(cout); }
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

}  // namespace bm
