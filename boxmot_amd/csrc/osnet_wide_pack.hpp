// Host-side weight preparation of the wide-OSNet kernel family: fp16 copies (round to nearest even) of every matrix-pipe
// operand of an OSN1 blob -- the 1x1 convolutions and the 7x7 stem as MFMA A fragments -- each tensor starting on
// a 16-byte boundary (the kernels fetch operands as 16-byte fragments; tensor offsets inside the fp32 blob are only 4-byte
// aligned).  Shared by WideOsnet (osnet_wide.hpp) and the emulation harness (tests/host_emu/emu_wide.cpp).
#pragma once

#include <cstdint>
#include <unordered_map>
#include <vector>

#include "reid_layout.hpp"
#include "reid_pack.hpp"

namespace bm {

struct WideW16 {
    std::vector<uint16_t> data;                  // fp16 bit patterns
    std::unordered_map<long, long> at;           // fp32-blob offset of a tensor -> index of its fp16 copy in `data`
    long stem = 0;                               // [7][c0 / 16][64][8]
    long of(long blob_off) const { return at.at(blob_off); }
};

inline WideW16 wide_pack_w16(const float* w, const OsnetLayout& L) {
    WideW16 P;
    auto align = [&]() { while (P.data.size() % 8) P.data.push_back(0); };
    auto add = [&](long off, long n) {
        align();
        P.at[off] = (long)P.data.size();
        for (long i = 0; i < n; ++i) P.data.push_back(f32_to_f16_bits(w[off + i]));
    };
    // stem A fragments [ky][channel tile][lane][8] (k_wide_stem): lane (co = lane & 15, g), k-slot j -> tap kx = 2 g + (j >> 2),
    // channel j & 3 of the RGBX pixel
    P.stem = 0;
    const int nct = L.c[0] / 16;
    for (int ky = 0; ky < 7; ++ky)
        for (int ct = 0; ct < nct; ++ct)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int co = 16 * ct + (lane & 15), kx = 2 * (lane >> 4) + (j >> 2), c = j & 3;
                    P.data.push_back((kx < 7 && c < 3) ? f32_to_f16_bits(w[L.stem_w + ((long)(co * 7 + ky) * 7 + kx) * 3 + c]) : (uint16_t)0);
                }
    for (int b = 0; b < 6; ++b) {
        const BlockW& B = L.block[b];
        add(B.conv1_w, (long)B.mid * B.cin);
        for (int l = 0; l < 10; ++l) add(B.light[l].pw, (long)B.mid * B.mid);
        add(B.conv3_w, (long)B.cout * B.mid);
        if (B.down_w >= 0) add(B.down_w, (long)B.cout * B.cin);
    }
    for (int s = 0; s < 2; ++s) add(L.trans_w[s], (long)L.c[s + 1] * L.c[s + 1]);
    add(L.conv5_w, (long)L.c[3] * L.c[3]);
    add(L.fc_w, (long)L.feat * L.c[3]);
    align();
    return P;
}

}  // namespace bm
