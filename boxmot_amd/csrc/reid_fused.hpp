// Fused fp16 MFMA ReID kernels for OSNet-x0.25 (gfx950, wave64).
//
// Reference computation: OSNet.forward, boxmot/reid/backbones/osnet.py:380-405
// (OSBlock :212-260, LightConv3x3 :127-155, ChannelGate :161-209), eval mode
// with BatchNorm folded into the convolutions (reid_pack.hpp).
//
// Design (DESIGN.md "ReID kernels"):
//  * One workgroup owns one crop for a whole OSBlock; every 1x1 convolution is
//    an MFMA with output channels as M (weights = A operand, pre-permuted on
//    the host) and 16 consecutive pixels as N.  The accumulator layout of one
//    MFMA (lane = pixel + 16*channel-group) IS the B-operand layout of the
//    next, so the conv1 -> (1x1 -> dw3x3)^k -> gate -> conv3 chain stays in
//    registers; fp16 operands, fp32 accumulation.
//  * The depthwise 3x3 needs spatial neighbours held by other lanes/waves: the
//    1x1 output is written once to an LDS image of the crop (zero halo, padded
//    pixel stride for bank spread) and read back as 9 taps of 8 bytes.
//  * ChannelGate needs the crop-wide average of each branch: per-wave partial
//    sums meet in LDS; the gated branches accumulate in fp32 registers and feed
//    conv3 (+ the 1x1 downsample or the identity) straight from registers.
//  * HBM traffic per block = block input + block output in fp16
//    (lane-group-major NHWC, reid_pack.hpp), weights from L2.
#pragma once

#include "reid_kernels_v1.hpp"
#include "reid_pack.hpp"

namespace bm {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Unmeasured variants for the next round (validated in emulation, off by default): epilogue operands of the stage-2
// blocks through LDS as in stages 0-1, and a <= 128-VGPR build of them (4 instead of 3 workgroups per CU).
#ifndef BM_STAGE2_EPI_LDS
#define BM_STAGE2_EPI_LDS 0
#endif
#ifndef BM_STAGE2_OCC4
#define BM_STAGE2_OCC4 0
#endif
// Memory-latency hiding in the OSBlock kernels (bit-identical results), measured per kernel at 4096 crops
// (tools/osblock_prof.hip, profiles/r2_osblock_variants.txt):
//   BM_PREFETCH_CONV1: conv1 issues all of its input loads before its first MFMA             -> -1.7 % over the six kernels
//   BM_PREFETCH_EPI:   the epilogue loads tile i + 1's operands (block input for the shortcut / downsample, the hand-over
//                      tensors) while tile i computes: -6 % on the stage-1 second block, but +3 % / +7 % on the stage-0 RECON
//                      and stage-2 first blocks (they are at the 128-register budget: the second operand set spills), so it
//                      is compiled in for that one kernel only (value 2 = "where it pays"; 0 / 1 = off / everywhere for A/B).
#ifndef BM_PREFETCH_CONV1
#define BM_PREFETCH_CONV1 1
#endif
#ifndef BM_PREFETCH_EPI
#define BM_PREFETCH_EPI 2
#endif

// Stage-0 LightConv3x3 as ONE dense 3x3 on the matrix pipe (reid_pack.hpp pack_light_dense): no LDS image, no depthwise
// VALU loop -- per 16-pixel tile 3 K=32 + 3 K=16 MFMAs on pixel-shifted copies of the rows (DPP row shifts inside a tile, the
// neighbour tile's edge pixel by a row rotate), rows of other waves through a 2 x 16 KiB LDS halo exchange (one barrier per
// layer).  0 = the round-1 layer (1x1 MFMA -> LDS image -> sliding-window depthwise on packed-fp16 FMAs).
#ifndef BM_DENSE_LIGHT
#define BM_DENSE_LIGHT 0
#endif

template <int STAGE>
struct Geo {
    static constexpr int H = 64 >> STAGE, W = 32 >> STAGE, P = H * W;
    static constexpr int MID = STAGE == 0 ? 16 : (STAGE == 1 ? 24 : 32);
    static constexpr int KT = STAGE == 0 ? 1 : 2;          // 16-channel tiles of the (padded) mid width
    static constexpr int MIDP = 16 * KT;
    static constexpr int HID = MID / 16;
    static constexpr int COUT = STAGE == 0 ? 64 : (STAGE == 1 ? 96 : 128);
    static constexpr int NCT = COUT / 16;
    // Stage 0 runs TWO workgroups (different crops) per CU so that one crop's LDS/barrier phases overlap the
    // other's VALU/MFMA phases: 8 waves x 16 tiles, <= 128 VGPRs (conv1 is recomputed per branch instead of
    // being held), and an unpadded 72 KB LDS image whose bank spread comes from an XOR swizzle.
    static constexpr int NWAVES = STAGE == 0 ? 8 : (16 >> STAGE);   // 8, 8, 4 waves per crop
    static constexpr int NT = P / 16 / NWAVES;              // 16, 4, 2 pixel tiles per wave
    static constexpr bool SWZ = STAGE == 0;
    static constexpr bool RECOMP = STAGE == 0;
    static constexpr int WG_PER_CU_WAVES = (STAGE == 2 && !BM_STAGE2_OCC4) ? 1 : 4;     // __launch_bounds__ 2nd argument (waves per SIMD)
    static constexpr int PXB = SWZ ? 32 : (KT == 1 ? 40 : 72);      // bytes per pixel of the LDS image
    static constexpr int ROWB = (W + 2) * PXB;
    static constexpr int IMG = (H + 2) * ROWB;
    // stage 2 with LDS-staged epilogue operands: conv3 (8 KiB) + bias + downsample 96 -> 128 (24 KiB) outgrow the image
    static constexpr bool DENSE = BM_DENSE_LIGHT && STAGE == 0;
    static constexpr int HALO = 2 * NWAVES * 2 * 2 * 64 * 8;       // [layer parity][wave][first / last row][half][lane] h4
    static constexpr int TBUF = DENSE ? HALO : ((STAGE == 2 && BM_STAGE2_EPI_LDS && IMG < 33280) ? 33280 : IMG);
    static constexpr int LDS_BYTES = TBUF + 4 * NWAVES * HID * 4;          // image + per-branch gate partials
};

// (y, x) of lane l16 in pixel tile q: tiles are 16 consecutive pixels in row-major order
template <int STAGE>
__device__ inline int tile_lds_offset(int q, int l16) {
    using G = Geo<STAGE>;
    int y, x;
    if (STAGE == 0) { y = q >> 1; x = (q & 1) * 16 + l16; }
    else if (STAGE == 1) { y = q; x = l16; }
    else { y = 2 * q + (l16 >> 3); x = l16 & 7; }
    return (y + 1) * G::ROWB + (x + 1) * G::PXB;
}

// Optional in-kernel phase clocks for tools/osblock_prof.hip (never defined in the product build)
#ifdef BM_OSBLOCK_PROF
__device__ unsigned long long g_osblock_prof[8];
#define BM_PROF_DECL() unsigned long long prof_t = clock64(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BM_PROF(k) do { unsigned long long t_ = clock64(); prof_acc[k] += t_ - prof_t; prof_t = t_; } while (0)
#define BM_PROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_osblock_prof[k_], prof_acc[k_]); } while (0)
#else
#define BM_PROF_DECL() ((void)0)
#define BM_PROF(k) ((void)0)
#define BM_PROF_FLUSH() ((void)0)
#endif

__device__ inline h4 to_h4(f4 v) { return h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; }
__device__ inline h4 residual_h4(f4 v, h4 h) { return h4{(_Float16)(v[0] - (float)h[0]), (_Float16)(v[1] - (float)h[1]), (_Float16)(v[2] - (float)h[2]), (_Float16)(v[3] - (float)h[3])}; }
__device__ inline f4 relu4(f4 v) { return f4{BM_RELU_F32(v[0]), BM_RELU_F32(v[1]), BM_RELU_F32(v[2]), BM_RELU_F32(v[3])}; }
__device__ inline h4 relu_h4(h4 v) { return __builtin_elementwise_max(v, (h4)(_Float16)0.f); }
__device__ inline h4 fma_h4(h4 a, h4 b, h4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ inline h8 cat8(h4 a, h4 b) { return h8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// ---------------------------------------------------------------------------
// OSBlock: in [n][P][CIN] -> out [n][P][COUT], fp16 lane-group-major NHWC
// ---------------------------------------------------------------------------
//
// TRANS fuses the stage's transition (Conv1x1 C->C + BN + ReLU + AvgPool 2x2, osnet.py:344-350) into the
// epilogue: the block output is consumed from registers and only the pooled [P/4][COUT] tensor is written
// (`wtr`: fragments [ct][ks] (1 KiB each) then fp32 bias[COUT], reid_pack.hpp pack_pointwise).
// `x1s` ([n][P][MIDP] halves) is where a recomputing stage with a wide input parks conv1's output instead of
// re-reading CIN channels per branch.
//
// EMIT / RECON split a stage's block pair (stage 0; stage 1 behind BM_STAGE1_HANDOVER, reid_engine.hpp) so that the
// output of block 1 (stage 0: 64 channels x 2048 px = 256 KiB per crop, the largest tensor of the network) never exists in memory:
//   * EMIT (block 1): per tile, the block output stays in registers and feeds the NEXT block's conv1 (weights `link.w`
//     at link.a0 / bias link.a1); what is stored is that conv1's 16-channel result (`x1s`) and this block's 16-channel
//     gated branch sum (`link.x2s`) -- 2 x 64 KiB instead of 256 KiB.
//   * RECON (block 2): takes conv1 from `x1s` for every branch and, where the identity shortcut needs the block input,
//     recomputes it per tile from the PREVIOUS block's branch sum and input (`in` is the previous block's input;
//     conv3 / bias / downsample fragments of the previous block at link.a0 / a1 / a2 of `link.w`).  Same operations
//     in the same order on the same fp16 values: bit-identical to storing and re-reading the tensor.
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
// Value of the pixel to the left (x - 1) / right (x + 1) of every lane's pixel in a 16-pixel row tile, as DPP moves inside the
// rows of 16 lanes (lane = pixel + 16 * channel group: a row of lanes is one channel group of the tile's 16 pixels).  The
// tile's outer lane takes the facing edge pixel of the neighbour tile `edge` (row rotate), or zero at the image border.
template <bool HAS_EDGE>
__device__ inline h4 pixel_left(h4 c, h4 edge) {
    const u2v u = __builtin_bit_cast(u2v, c), e = __builtin_bit_cast(u2v, edge);
    u2v r;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if constexpr (HAS_EDGE) r[d] = BM_DPP_U32(BM_DPP_U32(0u, e[d], 0x121, true), u[d], 0x111, false);      // row_ror:1, then row_shr:1 keeping lane 0
        else r[d] = BM_DPP_U32(0u, u[d], 0x111, true);
    }
    return __builtin_bit_cast(h4, r);
}
template <bool HAS_EDGE>
__device__ inline h4 pixel_right(h4 c, h4 edge) {
    const u2v u = __builtin_bit_cast(u2v, c), e = __builtin_bit_cast(u2v, edge);
    u2v r;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if constexpr (HAS_EDGE) r[d] = BM_DPP_U32(BM_DPP_U32(0u, e[d], 0x12F, true), u[d], 0x101, false);      // row_ror:15, then row_shl:1 keeping lane 15
        else r[d] = BM_DPP_U32(0u, u[d], 0x101, true);
    }
    return __builtin_bit_cast(h4, r);
}

struct BlkLink {
    const unsigned char* w = nullptr;
    long a0 = 0, a1 = 0, a2 = 0;
    _Float16* x2s = nullptr;
};

template <int STAGE, int CIN, bool DOWN, bool TRANS, bool EMIT = false, bool RECON = false>
__global__ void __launch_bounds__(64 * Geo<STAGE>::NWAVES, Geo<STAGE>::WG_PER_CU_WAVES)
k_osblock(const _Float16* __restrict__ in, _Float16* __restrict__ out, const unsigned char* __restrict__ wts, BlkPack bp,
          const int* __restrict__ count, _Float16* __restrict__ x1s, const unsigned char* __restrict__ wtr, BlkLink link) {
    static_assert(!(Geo<STAGE>::KT == 1 && DOWN && CIN != 16),
                  "conv3 (K=16 MFMA) and a K=32 downsample would share an accumulator: mixed-shape chains are wrong on gfx950");
    static_assert(!EMIT || (STAGE <= 1 && CIN == (STAGE == 0 ? 16 : 64) && DOWN && !TRANS), "EMIT: first block of stage 0 or 1");
    static_assert(!RECON || (STAGE <= 1 && CIN == Geo<STAGE>::COUT && !DOWN && TRANS), "RECON: second block of stage 0 or 1");
    constexpr int PREV_CIN = STAGE == 0 ? 16 : 64;       // RECON: input width of the previous block (= the stage input)
    constexpr int KINP = PREV_CIN == 16 ? 1 : PREV_CIN / 32;
    using G = Geo<STAGE>;
    if (count && (int)blockIdx.x >= *count) return;     // device-side crop count (no host round trip)
    constexpr int KT = G::KT, NT = G::NT, MIDP = G::MIDP, NCT = G::NCT, COUT = G::COUT, P = G::P;
    constexpr int KIN = CIN == 16 ? 1 : CIN / 32;
    constexpr bool STASH = G::RECOMP && CIN != 16 && !RECON;      // conv1 output parked in global scratch (L2-resident re-reads)
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    unsigned char* tbuf = lds;
    float* gap_part = reinterpret_cast<float*>(lds + G::TBUF);      // [4 branches][NWAVES][HID]

    // (a scalar wave index, as in the stem, was measured here and is 5-15 % slower: more scalar traffic, no VALU saved)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const long crop = blockIdx.x;
    const _Float16* xin = in + crop * P * (RECON ? PREV_CIN : CIN);      // RECON: the previous block's input
    _Float16* yout = out + crop * (TRANS ? P / 4 : P) * COUT;
    BM_PROF_DECL();

    // zero the LDS image once: the halo ring stays zero (= the dw conv's zero padding)
    if constexpr (!G::DENSE)
        for (int e = tid * 8; e < G::IMG; e += 64 * G::NWAVES * 8) *reinterpret_cast<unsigned long long*>(tbuf + e) = 0ull;

    // ---- conv1: 1x1 CIN -> MID, + bias, ReLU (osnet.py:248) ----
    auto conv1_into = [&](h4 (&x1)[NT][KT]) {
        unsigned xo = 0;            // opaque zero: the recomputing stage must re-read its input per branch,
        if constexpr (G::RECOMP) BM_OPAQUE_U32(xo);   // not hoist 128 registers of it out of the branch loop
        f4 bias[KT];
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) bias[ct] = *reinterpret_cast<const f4*>(wts + bp.conv1_b + (16 * ct + 4 * g) * 4);
        if constexpr (CIN == 16) {
            const h4 a = *reinterpret_cast<const h4*>(wts + bp.conv1_a + lane * 8);
#if BM_PREFETCH_CONV1
            // all input loads first, into the registers the results will live in: one memory round trip per call instead
            // of one per group of four tiles (the phase is latency-bound: 16 waves per CU, 8-byte loads)
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int p = (wave * NT + i) * 16 + l16;
                x1[i][0] = *reinterpret_cast<const h4*>(xin + (xo + (unsigned)(p * CIN + g * 4)));
            }
            BM_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < NT; ++i) x1[i][0] = relu_h4(to_h4(BM_MFMA_F16_K16(a, x1[i][0], bias[0])));
#else
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int p = (wave * NT + i) * 16 + l16;
                const h4 b = *reinterpret_cast<const h4*>(xin + (xo + (unsigned)(p * CIN + g * 4)));
                x1[i][0] = relu_h4(to_h4(BM_MFMA_F16_K16(a, b, bias[0])));
                if ((i & 3) == 3) BM_SCHED_FENCE();
            }
#endif
        } else {
            h8 a[KIN][KT];
#pragma unroll
            for (int ks = 0; ks < KIN; ++ks)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct)
                    a[ks][ct] = *reinterpret_cast<const h8*>(wts + bp.conv1_a + ((ks * KT + ct) * 64 + lane) * 16);
#if BM_PREFETCH_CONV1
            constexpr int GRP = NT * KIN <= 12 ? NT : 4;        // tiles whose loads are in flight together (<= 48 registers)
#pragma unroll
            for (int i0 = 0; i0 < NT; i0 += GRP) {
                h8 braw[GRP][KIN];
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int p = (wave * NT + i0 + i) * 16 + l16;
#pragma unroll
                    for (int ks = 0; ks < KIN; ++ks)
                        braw[i][ks] = *reinterpret_cast<const h8*>(xin + (xo + (unsigned)(p * CIN + g * (CIN / 4) + 8 * ks)));
                }
                BM_SCHED_FENCE();
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    f4 acc[KT];
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) acc[ct] = bias[ct];
#pragma unroll
                    for (int ks = 0; ks < KIN; ++ks)
#pragma unroll
                        for (int ct = 0; ct < KT; ++ct) acc[ct] = BM_MFMA_F16_K32(a[ks][ct], braw[i][ks], acc[ct]);
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) x1[i0 + i][ct] = relu_h4(to_h4(acc[ct]));
                }
                BM_SCHED_FENCE();
            }
#else
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int p = (wave * NT + i) * 16 + l16;
                f4 acc[KT];
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) acc[ct] = bias[ct];
#pragma unroll
                for (int ks = 0; ks < KIN; ++ks) {
                    const h8 b = *reinterpret_cast<const h8*>(xin + (xo + (unsigned)(p * CIN + g * (CIN / 4) + 8 * ks)));
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) acc[ct] = BM_MFMA_F16_K32(a[ks][ct], b, acc[ct]);
                }
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) x1[i][ct] = relu_h4(to_h4(acc[ct]));
                if (i & 1) BM_SCHED_FENCE();
            }
#endif
        }
    };
    h4 x1[NT][KT];
    if constexpr (!G::RECOMP) conv1_into(x1);
    _Float16* x1w = nullptr;        // this lane's slots in the scratch: [(tile, ct)][lane] h4 = 512 B per wave access
    if constexpr (STASH || EMIT || RECON) x1w = x1s + crop * (P * MIDP) + (long)(wave * NT * KT * 64 + lane) * 4;
    _Float16* x2w = nullptr;        // same slot layout for the branch sum handed from EMIT to RECON
    if constexpr (EMIT || RECON) x2w = link.x2s + crop * (P * MIDP) + (long)(wave * NT * KT * 64 + lane) * 4;

    // ---- four branches of 1..4 LightConv3x3, each gated and accumulated (osnet.py:249-253) ----
    h4 x2[NT][KT];          // gated sum of the four branches (packed fp16 accumulate: 4 terms)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) x2[i][ct] = (h4)(_Float16)0.f;
    // LDS addressing: `tap_base[d]` = byte address of the (y-1, x-1+d) neighbour of this lane's pixel in tile 0,
    // channel group included; every tap is tap_base[dx+1] + compile-time constant (DS immediate offsets).
    // Swizzled stage: the 8-byte channel-group slot is g ^ 2*bit3(column), which spreads the 16 pixels of a
    // row tile over all 64 banks without padding (a +16 column step of the second tile leaves bit 3 alone).
    int tap_base[3];
    {
        const int base00 = tile_lds_offset<STAGE>(wave * NT, l16) - G::ROWB - G::PXB;
        const int col0 = (STAGE == 0 ? l16 : (STAGE == 1 ? l16 : (l16 & 7)));     // column of this lane (+1 halo, -1 tap)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int gslot = G::SWZ ? (g ^ ((((col0 + d) >> 3) & 1) << 1)) * 8 : g * (KT * 8);
            tap_base[d] = base00 + d * G::PXB + gslot;
        }
    }
    auto tile_off = [](int i) constexpr {
        return STAGE == 0 ? (i >> 1) * G::ROWB + (i & 1) * 16 * G::PXB : (STAGE == 1 ? i * G::ROWB : 2 * i * G::ROWB);
    };
    __syncthreads();
    BM_PROF(0);

    int li = 0;
#pragma unroll 1
    for (int br = 0; br < 4; ++br) {
        h4 cur[NT][KT];
        if constexpr (RECON) {
            unsigned xo = 0;            // opaque zero: form the 16 addresses here, per branch, instead of keeping (and
            BM_OPAQUE_U32(xo);          // spilling) 16 loop-invariant 64-bit pointers across the branch loop
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) cur[i][ct] = *reinterpret_cast<const h4*>(x1w + (xo + (unsigned)((i * KT + ct) * 256)));
        } else if constexpr (STASH) {
            if (br == 0) {
                conv1_into(cur);
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) *reinterpret_cast<h4*>(x1w + (i * KT + ct) * 256) = cur[i][ct];
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) cur[i][ct] = *reinterpret_cast<const h4*>(x1w + (i * KT + ct) * 256);
            }
        } else if constexpr (G::RECOMP) conv1_into(cur);
        else {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) cur[i][ct] = x1[i][ct];
        }
        BM_PROF(1);
#pragma unroll 1
        for (int k = 0; k <= br; ++k, ++li) {
            const unsigned char* lw = wts + bp.light0 + (long)li * bp.light_bytes;
            if constexpr (G::DENSE) {
                // ---- LightConv3x3 as a dense 3x3 on the matrix pipe (pack_light_dense) ----
                static_assert(!G::DENSE || (STAGE == 0 && KT == 1 && NT == 16), "dense LightConv: stage 0 geometry");
                const unsigned char* ld = wts + bp.dense0 + (long)li * bp.dense_bytes;
                h8 aP[3];
                h4 aR[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    aP[dy] = *reinterpret_cast<const h8*>(ld + dy * 1024 + lane * 16);
                    aR[dy] = *reinterpret_cast<const h4*>(ld + 3072 + dy * 512 + lane * 8);
                }
                const f4 lbias = *reinterpret_cast<const f4*>(lw + bp.light_b + (4 * g) * 4);
                // rows of the neighbouring waves: every wave publishes its first and last row, one barrier, then reads the row
                // above its strip (last row of wave - 1) and below it (first row of wave + 1); the two buffers alternate by
                // layer, so a wave that is one layer ahead never overwrites rows a slower wave still has to read
                unsigned char* hb = tbuf + (li & 1) * (G::HALO / 2);
                auto hslot = [&](int w, int which, int hf) { return hb + (((w * 2 + which) * 2 + hf) * 64 + lane) * 8; };
                *reinterpret_cast<h4*>(hslot(wave, 0, 0)) = cur[0][0];
                *reinterpret_cast<h4*>(hslot(wave, 0, 1)) = cur[1][0];
                *reinterpret_cast<h4*>(hslot(wave, 1, 0)) = cur[NT - 2][0];
                *reinterpret_cast<h4*>(hslot(wave, 1, 1)) = cur[NT - 1][0];
                BM_PROF(2);
                __syncthreads();
                BM_PROF(3);
                const h4 zero4 = (h4)(_Float16)0.f;
                h4 up[2] = {zero4, zero4}, dn[2] = {zero4, zero4};
                if (wave > 0) { up[0] = *reinterpret_cast<const h4*>(hslot(wave - 1, 1, 0)); up[1] = *reinterpret_cast<const h4*>(hslot(wave - 1, 1, 1)); }
                if (wave < G::NWAVES - 1) { dn[0] = *reinterpret_cast<const h4*>(hslot(wave + 1, 0, 0)); dn[1] = *reinterpret_cast<const h4*>(hslot(wave + 1, 0, 1)); }
                struct RowOps { h8 P[2]; h4 R[2]; };
                auto build = [&](h4 c0, h4 c1) {
                    RowOps o;
                    o.P[0] = cat8(pixel_left<false>(c0, c0), c0);
                    o.P[1] = cat8(pixel_left<true>(c1, c0), c1);
                    o.R[0] = pixel_right<true>(c0, c1);
                    o.R[1] = pixel_right<false>(c1, c1);
                    return o;
                };
                RowOps w0 = build(up[0], up[1]), w1 = build(cur[0][0], cur[1][0]);
#pragma unroll
                for (int r = 0; r < NT / 2; ++r) {
                    const RowOps w2 = r + 1 < NT / 2 ? build(cur[2 * r + 2][0], cur[2 * r + 3][0]) : build(dn[0], dn[1]);
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        f4 accP = lbias, accR = f4{0.f, 0.f, 0.f, 0.f};      // one MFMA shape per accumulator
                        accP = BM_MFMA_F16_K32(aP[0], w0.P[hf], accP);
                        accR = BM_MFMA_F16_K16(aR[0], w0.R[hf], accR);
                        accP = BM_MFMA_F16_K32(aP[1], w1.P[hf], accP);
                        accR = BM_MFMA_F16_K16(aR[1], w1.R[hf], accR);
                        accP = BM_MFMA_F16_K32(aP[2], w2.P[hf], accP);
                        accR = BM_MFMA_F16_K16(aR[2], w2.R[hf], accR);
                        cur[2 * r + hf][0] = relu_h4(to_h4(f4{accP[0] + accR[0], accP[1] + accR[1], accP[2] + accR[2], accP[3] + accR[3]}));
                    }
                    w0 = w1; w1 = w2;
                    BM_SCHED_FENCE();
                }
                BM_PROF(4);
                BM_PROF(5);
                continue;
            }
            // 1x1 (linear): t = W_pw . cur  -> LDS image (fp16)
            if constexpr (KT == 1) {
                const h4 a = *reinterpret_cast<const h4*>(lw + bp.light_pw + lane * 8);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const f4 t = BM_MFMA_F16_K16(a, cur[i][0], (f4{0.f, 0.f, 0.f, 0.f}));
                    *reinterpret_cast<h4*>(tbuf + tap_base[1] + tile_off(i) + G::ROWB) = to_h4(t);
                }
            } else {
                h8 a[KT];
#pragma unroll
                for (int ct = 0; ct < KT; ++ct) a[ct] = *reinterpret_cast<const h8*>(lw + bp.light_pw + (ct * 64 + lane) * 16);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const h8 b = cat8(cur[i][0], cur[i][1]);
#pragma unroll
                    for (int ct = 0; ct < KT; ++ct) {
                        const f4 t = BM_MFMA_F16_K32(a[ct], b, (f4{0.f, 0.f, 0.f, 0.f}));
                        *reinterpret_cast<h4*>(tbuf + tap_base[1] + tile_off(i) + G::ROWB + ct * 8) = to_h4(t);
                    }
                }
            }
            BM_PROF(2);
            __syncthreads();
            BM_PROF(3);
            // depthwise 3x3 (pad 1) + bias + ReLU, channel tile by channel tile.  A wave's tiles form NSEQ column
            // strips of consecutive image rows, so the 3x3 window slides down a strip: only the RS new rows are
            // read from LDS per tile (3 reads instead of 9 for stages 0/1) -- the dw phase is LDS-bandwidth bound.
            constexpr int NSEQ = STAGE == 0 ? 2 : 1, SEQ_LEN = NT / NSEQ, RS = STAGE == 2 ? 2 : 1;
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                h4 wd[9];
                const h4* wsrc = reinterpret_cast<const h4*>(lw + bp.light_dw) + (ct * 4 + g) * 9;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) wd[tap] = wsrc[tap];
                const h4 bias = to_h4(*reinterpret_cast<const f4*>(lw + bp.light_b + (16 * ct + 4 * g) * 4));
#pragma unroll
                for (int sq = 0; sq < NSEQ; ++sq) {
                    h4 win[3][3];
#pragma unroll
                    for (int j = 0; j < SEQ_LEN; ++j) {
                        const int i = STAGE == 0 ? 2 * j + sq : j;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            if (j > 0 && dy + RS < 3) {
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx) win[dy][dx] = win[dy + RS][dx];
                            } else {
#pragma unroll
                                for (int dx = 0; dx < 3; ++dx)
                                    win[dy][dx] = *reinterpret_cast<const h4*>(tbuf + tap_base[dx] + tile_off(i) + ct * 8 + dy * G::ROWB);
                            }
                        }
                        // 9 taps x 4 channels as packed fp16 FMAs (v_pk_fma_f16); the 9-term fp16 accumulation
                        // is inside the error budget (tests/test_reid_emu.py, test_gpu_reid.py)
                        h4 o = bias;
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) o = fma_h4(win[tap / 3][tap % 3], wd[tap], o);
                        cur[i][ct] = relu_h4(o);
                        if (j & 1) BM_SCHED_FENCE();
                    }
                }
            }
            BM_PROF(4);
            __syncthreads();
            BM_PROF(5);
        }
        // ChannelGate (osnet.py:194-209): crop-wide average -> fc1 -> ReLU -> fc2 -> sigmoid -> scale
        // fc1 is linear in the pooled vector, so every wave reduces its own share of sum_c W1[h][c] * sum_p x[c][p]
        // to HID scalars (one 64-lane butterfly each); the waves' partials meet in LDS.
        float* part = gap_part + br * (G::NWAVES * G::HID);
        if constexpr (G::HID == 1 && KT == 1) {
            // one hidden unit, one channel tile: the pooled fc1 product is row 0 of W1 . cur summed over the wave's
            // pixels -- accumulate it on the matrix pipe (W1 as fp16 hi + lo parts: fp32-grade weights), then add up
            // the 16 pixel columns of row 0 (lanes 0..15; rows 4, 8, 12 of the other lane groups are zero)
            const f4 w1 = *reinterpret_cast<const f4*>(wts + bp.fc1_w + 4 * (4 * g));
            const h4 w1h = to_h4(w1);
            const f4 w1r = f4{w1[0] - (float)w1h[0], w1[1] - (float)w1h[1], w1[2] - (float)w1h[2], w1[3] - (float)w1h[3]};
            const h4 zero4 = (h4)(_Float16)0.f;
            const h4 a_hi = l16 == 0 ? w1h : zero4, a_lo = l16 == 0 ? to_h4(w1r) : zero4;
            f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                acc = BM_MFMA_F16_K16(a_hi, cur[i][0], acc);
                acc = BM_MFMA_F16_K16(a_lo, cur[i][0], acc);
            }
            float v = acc[0];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (lane == 0) part[wave] = v;
        } else {
            float ph[G::HID];
#pragma unroll
            for (int h = 0; h < G::HID; ++h) ph[h] = 0.f;
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                f4 s = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[r] += (float)cur[i][ct][r];
#pragma unroll
                for (int h = 0; h < G::HID; ++h) {
                    const f4 w1 = *reinterpret_cast<const f4*>(wts + bp.fc1_w + 4 * (h * MIDP + 16 * ct + 4 * g));
#pragma unroll
                    for (int r = 0; r < 4; ++r) ph[h] += w1[r] * s[r];
                }
            }
#pragma unroll
            for (int h = 0; h < G::HID; ++h) {
                float v = ph[h];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (lane == 0) part[wave * G::HID + h] = v;
            }
        }
        __syncthreads();
        float hidv[G::HID];
#pragma unroll
        for (int h = 0; h < G::HID; ++h) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < G::NWAVES; ++w) sum += part[w * G::HID + h];
            const float z = *reinterpret_cast<const float*>(wts + bp.fc1_b + 4 * h) + sum * (1.0f / P);
            hidv[h] = z > 0.f ? z : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) {
            const f4 zb = *reinterpret_cast<const f4*>(wts + bp.fc2_b + 4 * (16 * ct + 4 * g));
            f4 gate;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * ct + 4 * g + r;
                float z = zb[r];
#pragma unroll
                for (int h = 0; h < G::HID; ++h) z += *reinterpret_cast<const float*>(wts + bp.fc2_w + 4 * (c * G::HID + h)) * hidv[h];
                gate[r] = 1.f / (1.f + BM_EXPF(-z));
            }
            const h4 gate_h = to_h4(gate);
#pragma unroll
            for (int i = 0; i < NT; ++i) x2[i][ct] = fma_h4(gate_h, cur[i][ct], x2[i][ct]);
        }
        BM_PROF(6);
    }

    // ---- conv3 (1x1 MID -> COUT, linear) + downsample(x) or identity, ReLU (osnet.py:254-260) ----
    // stage 0 keeps the epilogue's operands in the registers the branch loop has released (16 tiles re-use them);
    // the wider stages have too many fragments for that and re-read them per tile from L1
    // Epilogue operands that do not fit in registers are staged ONCE per workgroup into LDS (the image is dead after the
    // last layer): 16 waves re-reading 10-30 KB of fragments per tile pair from the vector L1 (64 B/clk) was a sizeable
    // share of the block time; LDS delivers 128 B/clk and leaves the L1 to the activations.
    //   [0, EPI_A)      this block's conv3 fragments, bias, downsample fragments (contiguous in the packed blob)
    //   [EPI_A, +EPI_T) fused transition: fragments + bias        [.., +EPI_P) RECON: previous block's conv3/bias/down
    constexpr bool EPI = STAGE <= 1 || BM_STAGE2_EPI_LDS;
    constexpr int KS3E = COUT / 32;
    constexpr int EPI_T = TRANS ? NCT * KS3E * 1024 + COUT * 4 : 0;
    const int epi_a = (EPI && STAGE >= 1 && !RECON) ? (int)(bp.total - bp.conv3_a) : 0;     // (stage-1 RECON: no room left)
    const unsigned char* ew = wts + bp.conv3_a;          // conv3_a-relative base of this block's epilogue operands
    const unsigned char* etr = wtr;
    const unsigned char* epv = RECON ? link.w + link.a0 : nullptr;      // a0-relative base of the previous block's operands
    if constexpr (EPI) {
        auto stage_in = [&](const unsigned char* src, int bytes, int off) {
            for (int e = tid * 16; e < bytes; e += 64 * G::NWAVES * 16)
                *reinterpret_cast<f4*>(tbuf + off + e) = *reinterpret_cast<const f4*>(src + e);
        };
        if (epi_a) { stage_in(ew, epi_a, 0); ew = tbuf; }
        if constexpr (TRANS) { stage_in(wtr, EPI_T, epi_a); etr = tbuf + epi_a; }
        if constexpr (RECON) {
            const int bytes = (int)(link.a2 - link.a0) + NCT * (PREV_CIN == 16 ? 512 : KINP * 1024);   // conv3_a .. end of down_a
            stage_in(epv, bytes, epi_a + EPI_T);
            epv = tbuf + epi_a + EPI_T;
        }
        __syncthreads();
    }
    const long o3a = 0, o3b = bp.conv3_b - bp.conv3_a, oda = bp.down_a - bp.conv3_a;      // offsets from `ew`
    const long p3b = link.a1 - link.a0, pda = link.a2 - link.a0;                            // offsets from `epv`
    h4 eye;                 // A fragment of the 16x16 identity: row l16, k-slots 4g..4g+3
#pragma unroll
    for (int j = 0; j < 4; ++j) eye[j] = (_Float16)(l16 == 4 * g + j ? 1.f : 0.f);
    constexpr bool W3REG = STAGE == 0;
    h4 w3r[NCT], wdr[NCT];
    f4 b3r[NCT];
    if constexpr (W3REG) {
#pragma unroll
        for (int co = 0; co < NCT; ++co) {
            w3r[co] = *reinterpret_cast<const h4*>(ew + o3a + (co * 64 + lane) * 8);
            b3r[co] = *reinterpret_cast<const f4*>(ew + o3b + (16 * co + 4 * g) * 4);
            if constexpr (DOWN && CIN == 16) wdr[co] = *reinterpret_cast<const h4*>(ew + oda + (co * 64 + lane) * 8);
        }
    }
    // Operands a tile's epilogue takes from memory: loaded one tile ahead of their use (BM_PREFETCH) -- with two workgroups
    // per CU the round trip of these loads (L2 or further) was exposed once per tile.
    constexpr bool EPI_PF = BM_PREFETCH_EPI == 1 || (BM_PREFETCH_EPI == 2 && STAGE == 1 && TRANS && !RECON);
    struct TileIn {
        h8 bx[KIN];             // DOWN, wide input: the block input (downsample operand)
        h4 bx4;                 // DOWN, 16-channel input
        h4 x2p[KT], xp;         // RECON: previous block's branch sum and (16-channel) input
        h8 xp8[KINP];           // RECON: previous block's wide input
        h4 idn[NCT];            // identity shortcut read from memory (neither DOWN nor RECON)
    };
    auto tile_loads = [&](int i, TileIn& t) {
        unsigned p = (wave * NT + i) * 16 + l16;
        if constexpr (STASH || RECON) BM_OPAQUE_U32(p);  // addresses are formed at the use, not precomputed and spilled
        if constexpr (RECON) {
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) t.x2p[ct] = *reinterpret_cast<const h4*>(x2w + (i * KT + ct) * 256);
            if constexpr (PREV_CIN == 16) t.xp = *reinterpret_cast<const h4*>(xin + (unsigned)(p * 16 + g * 4));
            else {
#pragma unroll
                for (int ks = 0; ks < KINP; ++ks)
                    t.xp8[ks] = *reinterpret_cast<const h8*>(xin + (unsigned)(p * PREV_CIN + g * (PREV_CIN / 4) + 8 * ks));
            }
        }
        if constexpr (DOWN) {
            if constexpr (CIN == 16) t.bx4 = *reinterpret_cast<const h4*>(xin + (unsigned)(p * CIN + g * 4));
            else {
#pragma unroll
                for (int ks = 0; ks < KIN; ++ks) t.bx[ks] = *reinterpret_cast<const h8*>(xin + (unsigned)(p * CIN + g * (CIN / 4) + 8 * ks));
            }
        } else if constexpr (!RECON) {
#pragma unroll
            for (int co = 0; co < NCT; ++co) t.idn[co] = *reinterpret_cast<const h4*>(xin + (unsigned)(p * CIN + g * (CIN / 4) + 4 * co));
        }
    };
    auto conv3_compute = [&](int i, const TileIn& t, h4 (&y)[NCT]) {
#pragma unroll
        for (int co = 0; co < NCT; ++co) {
            f4 acc;
            if constexpr (W3REG) acc = b3r[co];
            else acc = *reinterpret_cast<const f4*>(ew + o3b + (16 * co + 4 * g) * 4);
            if constexpr (W3REG) {
                acc = BM_MFMA_F16_K16(w3r[co], x2[i][0], acc);
            } else if constexpr (KT == 1) {
                acc = BM_MFMA_F16_K16(*reinterpret_cast<const h4*>(ew + o3a + (co * 64 + lane) * 8), x2[i][0], acc);
            } else {
                acc = BM_MFMA_F16_K32(*reinterpret_cast<const h8*>(ew + o3a + (co * 64 + lane) * 16), cat8(x2[i][0], x2[i][1]), acc);
            }
            if constexpr (DOWN) {
                if constexpr (CIN == 16 && W3REG) {
                    acc = BM_MFMA_F16_K16(wdr[co], t.bx4, acc);
                } else if constexpr (CIN == 16) {
                    acc = BM_MFMA_F16_K16(*reinterpret_cast<const h4*>(ew + oda + (co * 64 + lane) * 8), t.bx4, acc);
                } else {
#pragma unroll
                    for (int ks = 0; ks < KIN; ++ks)
                        acc = BM_MFMA_F16_K32(*reinterpret_cast<const h8*>(ew + oda + ((co * KIN + ks) * 64 + lane) * 16), t.bx[ks], acc);
                }
            } else {
                // identity shortcut: the 4 input channels this lane holds are a K=16 B fragment, so adding them is one
                // MFMA with the 16x16 identity (exact: products by 1, fp32 accumulate) instead of 4 converts + 4 adds;
                // only where conv3 is itself a K=16 MFMA: a dependent chain that mixes the 16x16x16 and 16x16x32 shapes
                // on one accumulator returns wrong sums on gfx950 / ROCm 7.2 unless >= 8 wait states separate the links
                // (tools/mfma_chain_test.hip modes 11-13)
                h4 idn;
                if constexpr (RECON) {      // block input = ReLU(conv3_prev . x2_prev + down_prev . x_prev + bias), as EMIT computed it
                    f4 ap = *reinterpret_cast<const f4*>(epv + p3b + (16 * co + 4 * g) * 4);
                    if constexpr (KT == 1) {        // one MFMA shape per accumulator: K=16 in stage 0, K=32 in stage 1
                        ap = BM_MFMA_F16_K16(*reinterpret_cast<const h4*>(epv + (co * 64 + lane) * 8), t.x2p[0], ap);
                        ap = BM_MFMA_F16_K16(*reinterpret_cast<const h4*>(epv + pda + (co * 64 + lane) * 8), t.xp, ap);
                    } else {
                        ap = BM_MFMA_F16_K32(*reinterpret_cast<const h8*>(epv + (co * 64 + lane) * 16), cat8(t.x2p[0], t.x2p[1]), ap);
#pragma unroll
                        for (int ks = 0; ks < KINP; ++ks)
                            ap = BM_MFMA_F16_K32(*reinterpret_cast<const h8*>(epv + pda + ((co * KINP + ks) * 64 + lane) * 16), t.xp8[ks], ap);
                    }
                    idn = relu_h4(to_h4(ap));
                } else {
                    idn = t.idn[co];
                }
                if constexpr (KT == 1) acc = BM_MFMA_F16_K16(eye, idn, acc);       // same shape as conv3's own MFMA
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += (float)idn[r];
                }
            }
            y[co] = relu_h4(to_h4(acc));
        }
    };
    // tile i's operands are in tin[i & 1]: loaded while tile i - 1 computes (BM_PREFETCH) or right before the use
    TileIn tin[2];
    auto epi_begin = [&](int i_first) { if constexpr (EPI_PF) tile_loads(i_first, tin[i_first & 1]); };
    auto epi_tile = [&](int i, int i_next, h4 (&y)[NCT]) {          // i_next < 0: no further tile
        if constexpr (EPI_PF) { if (i_next >= 0) tile_loads(i_next, tin[i_next & 1]); }
        else tile_loads(i, tin[i & 1]);
        conv3_compute(i, tin[i & 1], y);
    };
    if constexpr (EMIT) {
        // next block's conv1 (COUT -> 16, + bias, ReLU) on the in-register block output: its accumulator layout is the
        // K=32 B operand (k-slot j <-> channel tile 2ks + (j >> 2)), exactly as the fused transition consumes it
        constexpr int KSN = COUT / 32;
        h8 wn[KSN][KT];
        f4 bn[KT];
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) {
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) wn[ks][ct] = *reinterpret_cast<const h8*>(link.w + link.a0 + ((ks * KT + ct) * 64 + lane) * 16);
            bn[ct] = *reinterpret_cast<const f4*>(link.w + link.a1 + (16 * ct + 4 * g) * 4);
        }
        epi_begin(0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            h4 y[NCT];
            epi_tile(i, i + 1 < NT ? i + 1 : -1, y);
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) {
                f4 an = bn[ct];
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) an = BM_MFMA_F16_K32(wn[ks][ct], cat8(y[2 * ks], y[2 * ks + 1]), an);
                *reinterpret_cast<h4*>(x1w + (i * KT + ct) * 256) = relu_h4(to_h4(an));
                *reinterpret_cast<h4*>(x2w + (i * KT + ct) * 256) = x2[i][ct];
            }
            BM_SCHED_FENCE();
        }
    } else if constexpr (!TRANS) {
        epi_begin(0);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int p = (wave * NT + i) * 16 + l16;
            h4 y[NCT];
            epi_tile(i, i + 1 < NT ? i + 1 : -1, y);
#pragma unroll
            for (int co = 0; co < NCT; ++co) *reinterpret_cast<h4*>(yout + (unsigned)(p * COUT + g * (COUT / 4) + 4 * co)) = y[co];
            BM_SCHED_FENCE();
        }
    } else {
        // two vertically adjacent tiles at a time: transition conv on the in-register block output (its MFMA
        // accumulator layout is the K32 B operand: k-slot j <-> channel tile 2ks + (j >> 2)), ReLU, 2x2 average
        // (vertical = the two tiles, horizontal = lane ^ 1), even lanes store the pooled pixel.
        static_assert(!TRANS || (STAGE < 2 && COUT % 32 == 0), "fused transition: stages 0 and 1");
        constexpr int KS3 = COUT / 32, WP = G::W / 2;
        const unsigned char* tbias = etr + NCT * KS3 * 1024;
        auto pair_i0 = [](int pr) constexpr { return STAGE == 0 ? (pr >> 1) * 4 + (pr & 1) : 2 * pr; };
        auto pair_i1 = [](int pr) constexpr { return (STAGE == 0 ? (pr >> 1) * 4 + (pr & 1) : 2 * pr) + (STAGE == 0 ? 2 : 1); };
        TileIn tp[2][2];        // [pair parity][tile of the pair]: the operands of pair pr + 1 load while pair pr computes
        if constexpr (EPI_PF) { tile_loads(pair_i0(0), tp[0][0]); tile_loads(pair_i1(0), tp[0][1]); }
#pragma unroll
        for (int pr = 0; pr < NT / 2; ++pr) {
            const int i0 = pair_i0(pr), i1 = pair_i1(pr);
            h4 y0[NCT], y1[NCT];
            if constexpr (EPI_PF) {
                if (pr + 1 < NT / 2) { tile_loads(pair_i0(pr + 1), tp[(pr + 1) & 1][0]); tile_loads(pair_i1(pr + 1), tp[(pr + 1) & 1][1]); }
            } else { tile_loads(i0, tp[pr & 1][0]); tile_loads(i1, tp[pr & 1][1]); }
            conv3_compute(i0, tp[pr & 1][0], y0);
            conv3_compute(i1, tp[pr & 1][1], y1);
            const int row = STAGE == 0 ? wave * (NT / 2) + (i0 >> 1) : wave * NT + i0;     // even image row of tile i0
            const int po = (row >> 1) * WP + (STAGE == 0 ? (i0 & 1) * 8 : 0) + (l16 >> 1);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f4 bv = *reinterpret_cast<const f4*>(tbias + (16 * ct + 4 * g) * 4);
                f4 a0 = bv, a1 = bv;
#pragma unroll
                for (int ks = 0; ks < KS3; ++ks) {
                    const h8 a = *reinterpret_cast<const h8*>(etr + ((ct * KS3 + ks) * 64 + lane) * 16);
                    a0 = BM_MFMA_F16_K32(a, cat8(y0[2 * ks], y0[2 * ks + 1]), a0);
                    a1 = BM_MFMA_F16_K32(a, cat8(y1[2 * ks], y1[2 * ks + 1]), a1);
                }
                a0 = relu4(a0); a1 = relu4(a1);
                f4 sp;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = a0[r] + a1[r];
                    sp[r] = v + BM_QUAD_SWAP1_F32(v);       // the average's 1/4 is folded into `wtr` (reid_pack.hpp)
                }
                if ((l16 & 1) == 0) *reinterpret_cast<h4*>(yout + (unsigned)(po * COUT + g * (COUT / 4) + 4 * ct)) = to_h4(sp);
            }
            BM_SCHED_FENCE();
        }
    }
    BM_PROF(7);
    BM_PROF_FLUSH();
}

// ---------------------------------------------------------------------------
// head: conv5 (1x1 C->C + ReLU) -> GAP -> FC(C->F) + BN1d + ReLU -> L2 norm
// (osnet.py:310-315, 393-396; base_backend.py:206).  in [n][128 px][C]; one
// workgroup (2 waves) per crop.  wts5: fragments [ct][ks] + bias; wfc: fp16 [F][C] (memory order) + fp32 bias.
// ---------------------------------------------------------------------------
template <int C, int F>
__global__ void __launch_bounds__(128) k_head_fused(const _Float16* __restrict__ in, const unsigned char* __restrict__ wts5,
                                                    const unsigned char* __restrict__ wfc, float* __restrict__ out_base,
                                                    const int* __restrict__ out_rows, const int* __restrict__ count) {
    if (count && (int)blockIdx.x >= *count) return;
    constexpr int NCT = C / 16, KS = C / 32, P = 128, NT = 4;
    __shared__ float s_gap[2][C];
    __shared__ float s_v[C];
    __shared__ float s_red[2];
    // (a scalar wave index, as in the stem, was measured here and is 5-15 % slower: more scalar traffic, no VALU saved)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const long crop = blockIdx.x;
    const _Float16* xin = in + crop * P * C;
    const unsigned char* bias5 = wts5 + NCT * KS * 1024;
    f4 sum[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) sum[ct] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int i = 0; i < NT; ++i) {
        const int p = (wave * NT + i) * 16 + l16;
        h8 b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const h8*>(xin + (long)p * C + g * (C / 4) + 8 * ks);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f4 acc = *reinterpret_cast<const f4*>(bias5 + (16 * ct + 4 * g) * 4);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = BM_MFMA_F16_K32(*reinterpret_cast<const h8*>(wts5 + ((ct * KS + ks) * 64 + lane) * 16), b[ks], acc);
            acc = relu4(acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[ct][r] += acc[r];
        }
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = sum[ct][r];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (l16 == 0) s_gap[wave][g * (C / 4) + 4 * ct + r] = v;      // memory (L-layout) channel order
        }
    __syncthreads();
    for (int c = tid; c < C; c += 128) s_v[c] = (s_gap[0][c] + s_gap[1][c]) * (1.0f / P);
    __syncthreads();
    const float* fcb = reinterpret_cast<const float*>(wfc + (long)F * C * 2);
    const _Float16* fcw = reinterpret_cast<const _Float16*>(wfc);
    float* out = out_base + (out_rows ? (long)out_rows[crop] : crop) * F;
    float vals[F / 128];
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < F / 128; ++k) {
        const int f = tid + k * 128;
        float a = fcb[f];
        const h8* wr = reinterpret_cast<const h8*>(fcw + (long)f * C);
        for (int c8 = 0; c8 < C / 8; ++c8) {
            const h8 w8 = wr[c8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a += (float)w8[j] * s_v[c8 * 8 + j];
        }
        a = a > 0.f ? a : 0.f;
        vals[k] = a;
        sq += a * a;
    }
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if (lane == 0) s_red[wave] = sq;
    __syncthreads();
    const float nrm = sqrtf(s_red[0] + s_red[1]);
#pragma unroll
    for (int k = 0; k < F / 128; ++k) out[tid + k * 128] = vals[k] / nrm;
}

// ---------------------------------------------------------------------------
// Batched head (the shipped one): the same computation as k_head_fused for NB = 16 crops per workgroup of 4 waves.
//   * conv5: wave w owns output-channel tiles 2w, 2w+1 for every pixel tile and keeps their 8 A fragments in registers for
//     all 16 crops; a crop's input (32 KiB) is staged once into LDS (16-byte slots XOR-swizzled by the pixel so that the
//     16 pixels of a B-fragment read spread over the banks) while the previous crop computes (loads issued before, LDS
//     writes after the MFMAs); bias + ReLU + global average in fp32 registers.
//   * FC 128 -> 512 for the 16 crops at once on the matrix pipe: M = feature, N = crop, K = channel; the pooled vector
//     enters as an fp16 hi + lo pair (two MFMAs per k-step: fp32-grade operand, the weights are the fp16 ones of
//     k_head_fused), + bias, ReLU, L2 normalisation over the 512 features (per crop = per accumulator column).
// k_head_fused keeps 2 waves per crop at 256 + 110 registers (one wave per SIMD, 0.64 ms per 16384 crops); this one runs
// two workgroups per CU with the FC as a real 16-column MFMA tile.
// ---------------------------------------------------------------------------
constexpr int HEAD_NB = 16;
template <int C>
struct HeadGeo {
    static constexpr int CROP_BYTES = 128 * C * 2;
    static constexpr int VSTRIDE = C + 4;                       // floats per pooled vector in LDS (+4: bank spread, 16-byte rows)
    static constexpr int LDS_BYTES = 2 * CROP_BYTES + HEAD_NB * VSTRIDE * 4 + 4 * HEAD_NB * 4;
};

template <int C, int F>
__global__ void __launch_bounds__(256, 2) k_head_batched(const _Float16* __restrict__ in, const unsigned char* __restrict__ wts5,
                                                         const unsigned char* __restrict__ wfc, float* __restrict__ out_base,
                                                         const int* __restrict__ out_rows, const int* __restrict__ count, int n_total) {
    static_assert(C == 128 && F % 64 == 0, "head: 128 channels in, a multiple of 64 features out");
    constexpr int NCT = C / 16, KS = C / 32, P = 128, NB = HEAD_NB, CROP_BYTES = HeadGeo<C>::CROP_BYTES, VS = HeadGeo<C>::VSTRIDE;
    constexpr int FT_PER_WAVE = F / 16 / 4;
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    float* vbuf = reinterpret_cast<float*>(lds + 2 * CROP_BYTES);          // [NB][VS] pooled vectors (L-layout channel order)
    float* red = vbuf + NB * VS;                                            // [4 waves][NB]
    int n_eff = n_total;
    if (count) { const int c = *count; n_eff = c < n_total ? c : n_total; }
    const long crop0 = (long)blockIdx.x * NB;
    if (crop0 >= n_eff) return;
    const int nb = n_eff - crop0 < NB ? (int)(n_eff - crop0) : NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const unsigned char* bias5 = wts5 + NCT * KS * 1024;
    h8 a[2][KS];
    f4 bias[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        const int ct = 2 * wave + c2;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[c2][ks] = *reinterpret_cast<const h8*>(wts5 + ((ct * KS + ks) * 64 + lane) * 16);
        bias[c2] = *reinterpret_cast<const f4*>(bias5 + (16 * ct + 4 * g) * 4);
    }
    for (int e = tid; e < NB * VS; e += 256) vbuf[e] = 0.f;                 // rows of absent crops stay zero
    // staging: thread t moves the 16-byte slots t, t + 256, ... of a crop; slot q = (pixel q >> 4, slot q & 15)
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 st[CROP_BYTES / 16 / 256];
    auto stage_load = [&](int k) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(in + (crop0 + k) * (long)(P * C));
#pragma unroll
        for (int j = 0; j < CROP_BYTES / 16 / 256; ++j) st[j] = *reinterpret_cast<const u4*>(src + (size_t)(tid + 256 * j) * 16);
    };
    auto stage_store = [&](int b) {
#pragma unroll
        for (int j = 0; j < CROP_BYTES / 16 / 256; ++j) {
            const int q = tid + 256 * j, p = q >> 4, slot = q & 15;
            *reinterpret_cast<u4*>(lds + b * CROP_BYTES + p * (C * 2) + ((slot ^ (p & 15)) << 4)) = st[j];
        }
    };
    stage_load(0);
    stage_store(0);
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < nb; ++k) {
        if (k + 1 < nb) stage_load(k + 1);
        const unsigned char* buf = lds + (k & 1) * CROP_BYTES;
        f4 sum[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < P / 16; ++i) {
            const int p = i * 16 + l16;
            h8 b[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[ks] = *reinterpret_cast<const h8*>(buf + p * (C * 2) + (((g * KS + ks) ^ (p & 15)) << 4));
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                f4 acc = bias[c2];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = BM_MFMA_F16_K32(a[c2][ks], b[ks], acc);
                acc = relu4(acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[c2][r] += acc[r];
            }
        }
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = sum[c2][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (l16 == 0) vbuf[k * VS + g * (C / 4) + 4 * (2 * wave + c2) + r] = v * (1.0f / P);
            }
        if (k + 1 < nb) stage_store((k + 1) & 1);
        __syncthreads();
    }
    // ---- FC + BN1d (folded) + ReLU for the NB crops: D[f][crop] = sum_c W[f][c] * v[crop][c] ----
    const float* fcb = reinterpret_cast<const float*>(wfc + (long)F * C * 2);
    const _Float16* fcw = reinterpret_cast<const _Float16*>(wfc);
    h8 vh[KS], vl[KS];          // this lane's k-slots of crop l16: channels 32 ks + 8 g + j (memory order), hi + lo parts
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const f4 v0 = *reinterpret_cast<const f4*>(vbuf + l16 * VS + 32 * ks + 8 * g);
        const f4 v1 = *reinterpret_cast<const f4*>(vbuf + l16 * VS + 32 * ks + 8 * g + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? v0[j & 3] : v1[j & 3];
            const _Float16 hi = (_Float16)v;
            vh[ks][j] = hi;
            vl[ks][j] = (_Float16)(v - (float)hi);
        }
    }
    f4 vals[FT_PER_WAVE];
    float sq = 0.f;
#pragma unroll
    for (int fi = 0; fi < FT_PER_WAVE; ++fi) {
        const int ft = wave * FT_PER_WAVE + fi;
        f4 acc = *reinterpret_cast<const f4*>(fcb + 16 * ft + 4 * g);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const h8 w = *reinterpret_cast<const h8*>(fcw + (long)(16 * ft + l16) * C + 32 * ks + 8 * g);
            acc = BM_MFMA_F16_K32(w, vh[ks], acc);
            acc = BM_MFMA_F16_K32(w, vl[ks], acc);
        }
        acc = relu4(acc);
        vals[fi] = acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) sq += acc[r] * acc[r];
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    if (lane < 16) red[wave * NB + l16] = sq;
    __syncthreads();
    const float nrm = sqrtf(red[l16] + red[NB + l16] + red[2 * NB + l16] + red[3 * NB + l16]);
    if (l16 < nb) {
        float* out = out_base + (out_rows ? (long)out_rows[crop0 + l16] : crop0 + l16) * F;
#pragma unroll
        for (int fi = 0; fi < FT_PER_WAVE; ++fi) {
            const int ft = wave * FT_PER_WAVE + fi;
            f4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = vals[fi][r] / nrm;
            *reinterpret_cast<f4*>(out + 16 * ft + 4 * g) = o;
        }
    }
}

// ---------------------------------------------------------------------------
// stem: conv 7x7 stride 2 pad 3 (3 -> 16) + BN + ReLU + maxpool 3x3 stride 2 pad 1
// (osnet.py:294-295).  Input: fp16 RGBX crops with a 3-pixel zero border,
// [n][262][136][4]; output [n][64*32][16] lane-group-major.  One workgroup
// (8 waves) per crop; wave w produces pooled rows 8w..8w+7.  The 7x7x3 window is
// 7 MFMA k-steps (one per kernel row): 8 input pixels x RGBX = 32 halves = one
// 16-byte load per lane (k-slot order fixed by pack_stem).
// ---------------------------------------------------------------------------
constexpr int STEM_ROWS = 262, STEM_COLS = 136;

__device__ inline f4 max4(f4 a, f4 b) {
    return f4{a[0] > b[0] ? a[0] : b[0], a[1] > b[1] ? a[1] : b[1], a[2] > b[2] ? a[2] : b[2], a[3] > b[3] ? a[3] : b[3]};
}

__global__ void __launch_bounds__(512) k_stem_fused(const _Float16* __restrict__ crops, _Float16* __restrict__ out,
                                                    const unsigned char* __restrict__ wts, const int* __restrict__ count) {
    if (count && (int)blockIdx.x >= *count) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const long crop = blockIdx.x;
    const _Float16* img = crops + crop * (STEM_ROWS * STEM_COLS * 4);
    _Float16* yout = out + crop * (64 * 32) * 16;
    h8 a[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) a[ky] = *reinterpret_cast<const h8*>(wts + (ky * 64 + lane) * 16);
    const f4 bias = *reinterpret_cast<const f4*>(wts + 7 * 1024 + 4 * g * 4);

    // conv row cy (0..127), 4 tiles of 16 conv pixels; rows outside the image contribute 0 (post-ReLU >= 0)
    auto conv_row = [&](int cy, f4 (&row)[4]) {
        if (cy < 0 || cy > 127) {
#pragma unroll
            for (int t = 0; t < 4; ++t) row[t] = f4{0.f, 0.f, 0.f, 0.f};
            return;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 acc = bias;
            const int cx = t * 16 + l16;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const h8 b = *reinterpret_cast<const h8*>(img + ((long)(2 * cy + ky) * STEM_COLS + 2 * cx + 2 * g) * 4);
                acc = BM_MFMA_F16_K32(a[ky], b, acc);
            }
            row[t] = relu4(acc);
        }
    };
    f4 prev[4], mid[4], next[4];
    const int oy0 = wave * 8;
    conv_row(2 * oy0 - 1, prev);
#pragma unroll 1
    for (int oy = oy0; oy < oy0 + 8; ++oy) {
        conv_row(2 * oy, mid);
        conv_row(2 * oy + 1, next);
        // vertical max, then horizontal max over (cx-1, cx, cx+1); pooled pixel ox = cx/2 sits on even lanes
        f4 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = max4(max4(prev[t], mid[t]), next[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 m = v[t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {      // all values are >= 0 (post-ReLU), so a missing neighbour reads as 0
                const float right = BM_ROW_SHL1_F32(v[t][r]);
                float left = BM_ROW_SHR1_F32(v[t][r]);
                const float left_prev_tile = BM_ROW_ROR1_F32(v[t > 0 ? t - 1 : 0][r]);
                if (l16 == 0) left = t > 0 ? left_prev_tile : 0.f;
                float mm = m[r] > right ? m[r] : right;
                m[r] = mm > left ? mm : left;
            }
            if ((l16 & 1) == 0) {
                const int p = oy * 32 + t * 8 + (l16 >> 1);
                *reinterpret_cast<h4*>(yout + (long)p * 16 + g * 4) = to_h4(m);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) prev[t] = next[t];
    }
}

// ---------------------------------------------------------------------------
// Fused crop -> resize -> normalise -> stem conv 7x7/2 -> maxpool 3x3/2: the resized crop never
// exists in HBM.  One workgroup (8 waves) per crop walks the crop in 8 bands of 8 pooled rows:
//   1. the source rows the band needs are staged from the frame into LDS with coalesced dword loads
//      (crop pixels are read from HBM exactly once; boxes too large for the staging area take direct loads),
//   2. cv2.resize's fixed-point bilinear + the normalisation table produce 32 new fp16 RGBX input rows
//      into a 40-row LDS ring (rows are reused by the next band),
//   3. wave w computes pooled row 8*band + w: three conv rows as MFMA k-steps over LDS B fragments
//      (one ds_read_b128 per MFMA), vertical + horizontal max, store.
// Same arithmetic (and bit-identical fp16 output) as k_crop_resize_rgbx + k_stem_fused.
// ---------------------------------------------------------------------------
// BM_STEM_STREAM = 1: in the conv phase a wave owns one 16-pixel column strip and FOUR pooled rows (nine conv rows) and
// streams the 23 input rows of the strip once: one ds_read_b128 per input row feeds the 3-4 conv rows that row contributes to
// (kernel rows ky of the same parity), each into its own accumulator in ascending ky -- the same sums in the same order as
// BM_STEM_STREAM = 0 (a wave owns one pooled row across the width: 84 reads and 84 MFMAs per band against 23 and 63 here; the
// phase was bound by the LDS read bandwidth, 1 KiB per MFMA).  The pooled pixel on a strip's first column needs the last conv
// column of the strip to its left, which another wave computes: it crosses through a 4 KiB LDS edge buffer (double-buffered
// by band parity) under the barrier that ends the band.
#ifndef BM_STEM_STREAM
#define BM_STEM_STREAM 1
#endif
// BM_STEM_ASYNC = 1: the source rows of band b + 1 are requested by asynchronous global -> LDS copies (global_load_lds_dword)
// right after the barrier that ends band b's resampling -- the staging area is idle from there to band b + 1's resampling -- and
// land under band b's convolution phase: no registers, no exposed load latency per band (a register prefetch had cost 1.5 %).
#ifndef BM_STEM_ASYNC
#define BM_STEM_ASYNC 1
#endif
constexpr int RING_ROWS = 40;
constexpr int RING_ROW_BYTES = STEM_COLS * 8;                    // 136 px * RGBX fp16
constexpr int SRC_STAGE_BYTES = 20 * 1024;
constexpr int STEM_EDGE_BYTES = BM_STEM_STREAM ? 2 * 2 * 4 * 4 * 16 * 4 : 0;   // [band parity][half][strip][pooled row][channel] fp32
constexpr int STEM_LUT_BYTES = 768 * 4;                          // normalisation table, 4-byte entries (fp16 in the low half)
constexpr int STEM2_LDS = RING_ROWS * RING_ROW_BYTES + SRC_STAGE_BYTES + STEM_LUT_BYTES + 256 * 8 + STEM_EDGE_BYTES;   // ring, staging, LUT, y table, edges
// the fp32-grade stem keeps its seven lo fragments in LDS (one ds_read_b128 per lo MFMA) instead of 28 registers: at <= 128
// registers and 78.5 KiB two workgroups share a CU, like the fp16 stem (BM_HP_STEM_LDS_LO = 0: registers, one workgroup per CU)
#ifndef BM_HP_STEM_LDS_LO
#define BM_HP_STEM_LDS_LO 1
#endif
constexpr int STEM2_LDS_HP = STEM2_LDS + (BM_HP_STEM_LDS_LO ? 7 * 1024 : 0);

// HP (the fp32-grade family, reid_hp.hpp): a pixel byte v is EXACT in fp16, and the normalisation (v / 255 - mean) / std = a v + b
// is linear, so it is folded into the weights on the host (pack_stem_hp_fused): the ring holds [R, G, B, 1] raw values (the
// fourth channel carries the bias term b of real pixels and is 0 in the zero padding, which reproduces the reference's zero
// padding of the NORMALISED image exactly), the weights are fp16 (hi, lo) pairs -- the lo fragment pre-scaled by 2^11 and
// multiplied with the operand scaled by 2^-11 (both exact) so that it stays in fp16's normal range -- two MFMAs per k-step into
// the same fp32 accumulator; the pooled result leaves as an fp16 (hi, lo) pair of planes.  One workgroup per CU (the second
// fragment set does not fit 128 registers).
template <bool HP>
__device__ __forceinline__ void stem_resize_fused_body(const uint8_t* const* frames, const int* crop_stream,
                                                           const float* boxes, int box_stride, int W, int H,
                                                           const float* lut, _Float16* __restrict__ out, _Float16* __restrict__ out_lo,
                                                           const unsigned char* __restrict__ wts,
                                                           const int* __restrict__ count) {
    if (count && (int)blockIdx.x >= *count) return;
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    unsigned char* ring = lds;
    unsigned char* stage = lds + RING_ROWS * RING_ROW_BYTES;
    _Float16* lut_h = reinterpret_cast<_Float16*>(stage + SRC_STAGE_BYTES);
    unsigned* ytab = reinterpret_cast<unsigned*>(stage + SRC_STAGE_BYTES + STEM_LUT_BYTES);   // [256]{s0 | s1 << 16, a0 | a1 << 16}
    const int tid = threadIdx.x, lane = tid & 63, wave = BM_UNIFORM_I32(tid >> 6), g = lane >> 4, l16 = lane & 15;
    const long crop = blockIdx.x;
    const uint8_t* frame = frames[crop_stream[crop]];
    const CropRect r = crop_rect(boxes + crop * box_stride, W, H);
    const long row_stride = (long)W * 3;
    _Float16* yout = out + crop * (64 * 32) * 16;
    _Float16* yout_lo = HP ? out_lo + crop * (64 * 32) * 16 : nullptr;
    for (int e = tid; e < 768; e += 512) { lut_h[2 * e] = HP ? (_Float16)(float)(e & 255) : (_Float16)lut[e]; lut_h[2 * e + 1] = (_Float16)0.f; }      // 4-byte entries
    if (tid < REID_IN_H) {      // vertical taps of every resized row, once per crop
        const ResizeAxis ay = resize_axis_y(tid, REID_IN_H, r.h > 0 ? r.h : 1);
        ytab[2 * tid] = (unsigned)ay.s0 | ((unsigned)ay.s1 << 16);
        ytab[2 * tid + 1] = (unsigned)ay.a0 | ((unsigned)ay.a1 << 16);
    }
    for (int e = tid * 8; e < RING_ROWS * RING_ROW_BYTES; e += 512 * 8) *reinterpret_cast<unsigned long long*>(ring + e) = 0ull;
    constexpr bool LO_LDS = HP && BM_HP_STEM_LDS_LO;
    h8 a[7], a_lo[(HP && !LO_LDS) ? 7 : 1];
    unsigned char* alo_lds = stage + SRC_STAGE_BYTES + STEM_LUT_BYTES + 256 * 8 + STEM_EDGE_BYTES;      // [ky][lane] 16-byte fragments
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        a[ky] = *reinterpret_cast<const h8*>(wts + ((HP ? 2 * ky : ky) * 64 + lane) * 16);
        if constexpr (HP && !LO_LDS) a_lo[ky] = *reinterpret_cast<const h8*>(wts + ((2 * ky + 1) * 64 + lane) * 16);
    }
    if constexpr (LO_LDS) {
        if (tid < 7 * 64) *reinterpret_cast<h8*>(alo_lds + tid * 16) = *reinterpret_cast<const h8*>(wts + ((2 * (tid >> 6) + 1) * 64 + (tid & 63)) * 16);
    }
    const f4 bias = *reinterpret_cast<const f4*>(wts + (HP ? 14 : 7) * 1024 + 4 * g * 4);
    // resampling role of this thread: output column dx, row phase rp (4 threads per column)
    const int dx = tid & 127, rp = tid >> 7;
    const ResizeAxis ax = resize_axis_x(dx, REID_IN_W, r.w > 0 ? r.w : 1);
    const bool identity = r.w == REID_IN_W && r.h == REID_IN_H;
    const bool area2 = r.w == 2 * REID_IN_W && r.h == 2 * REID_IN_H;
    __syncthreads();
    BM_PROF_DECL();
    BM_PROF(0);

    // geometry of a band: padded input rows [pr0, pr1) are new in it (padded row = resized row + 3) = resized rows [dy0, dy1),
    // which read source rows [sy_lo, sy_hi] of the crop, staged at `pitch` bytes per row if they fit the staging area
    struct BandGeom { int pr0, pr1, dy0, dy1, sy_lo, sy_hi, pitch; bool staged; };
    auto band_geom = [&](int band) {
        BandGeom b;
        b.pr0 = band == 0 ? 0 : 32 * band + 5; b.pr1 = 32 * band + 37 < STEM_ROWS ? 32 * band + 37 : STEM_ROWS;
        b.dy0 = b.pr0 - 3 > 0 ? b.pr0 - 3 : 0; b.dy1 = b.pr1 - 3 < REID_IN_H ? b.pr1 - 3 : REID_IN_H;
        b.sy_lo = 0; b.sy_hi = -1; b.pitch = 0; b.staged = false;
        if (r.w > 0 && b.dy1 > b.dy0) {
            if (identity) { b.sy_lo = b.dy0; b.sy_hi = b.dy1 - 1; }
            else if (area2) { b.sy_lo = 2 * b.dy0; b.sy_hi = 2 * b.dy1 - 1; }
            else { b.sy_lo = (int)(ytab[2 * b.dy0] & 0xffffu); b.sy_hi = (int)(ytab[2 * (b.dy1 - 1)] >> 16); }      // the table of vertical taps
            b.pitch = ((3 + r.w * 3 + 3) / 4) * 4;          // room for the per-row alignment offset (0..3)
            b.staged = (long)(b.sy_hi - b.sy_lo + 1) * b.pitch <= SRC_STAGE_BYTES;
        }
        return b;
    };
    // ---- 1. stage the source rows of a band ----
    auto stage_rows = [&](const BandGeom& b) {
        if (!b.staged) return;
        // wave w moves rows w, w + 8, ...: row addresses are wave-uniform (scalar unit), lanes take consecutive dwords
        const long byte0 = (long)r.x1 * 3;
        const int ndw = b.pitch / 4, nrow = b.sy_hi - b.sy_lo + 1;
        const uint8_t* frame_end = frame + (long)H * row_stride;
        for (int rr = wave; rr < nrow; rr += 8) {
            const uint8_t* rowp = frame + (long)(r.y1 + b.sy_lo + rr) * row_stride + byte0;
            const uint8_t* src0 = rowp - (reinterpret_cast<uintptr_t>(rowp) & 3);            // aligned dwords
            unsigned char* dst0 = stage + rr * b.pitch;
            for (int cw = lane; cw < ndw; cw += 64) {
                const uint8_t* src = src0 + 4 * cw;
                if (src + 4 <= frame_end) {
                    if constexpr (BM_STEM_ASYNC) BM_GLDS4(src, dst0 + 4 * (cw - lane), lane);
                    else *reinterpret_cast<unsigned*>(dst0 + 4 * cw) = *reinterpret_cast<const unsigned*>(src);
                } else {                                  // tail of the frame: stay inside the allocation
                    unsigned v = 0;
                    for (int q = 0; q < 4 && src + q < frame_end; ++q) v |= (unsigned)src[q] << (8 * q);
                    *reinterpret_cast<unsigned*>(dst0 + 4 * cw) = v;
                }
            }
        }
    };
    if constexpr (BM_STEM_ASYNC) stage_rows(band_geom(0));
#pragma unroll 1
    for (int band = 0; band < 8; ++band) {
        const BandGeom bg = band_geom(band);
        const int pr0 = bg.pr0, pr1 = bg.pr1, sy_lo = bg.sy_lo, pitch = bg.pitch;
        const bool staged = bg.staged;
        if constexpr (BM_STEM_ASYNC) BM_WAIT_VM0();         // this wave's copies of the band's rows have landed
        else stage_rows(bg);
        BM_PROF(1);
        __syncthreads();
        BM_PROF(2);
        // ---- 2. resample the new rows into the ring ----
        if (staged && !identity && !area2) {
            // Bilinear fast path.  A thread owns one output column and a run of consecutive rows: the horizontal
            // interpolations S0/S1 of the two source rows are kept while the vertical tap pair is unchanged (an
            // upscaled crop repeats it for several rows) and S1 becomes S0 when the pair advances by one row.
            // A pixel's three bytes come from one 8-byte LDS read + funnel shift at any byte alignment.
            const int nrows = pr1 - pr0, run = (nrows + 3) >> 2;
            const unsigned abase = (unsigned)reinterpret_cast<uintptr_t>(frame + (long)r.y1 * row_stride + (long)r.x1 * 3);
            const unsigned rs = (unsigned)row_stride;
            const int xs0 = ax.s0 * 3, xs1 = ax.s1 * 3;
            // Every product here has operands below 2^24 (bytes x 11-bit weights, row indices x pitches), so the multiplies are
            // the full-rate 24-bit ones.  T = (S >> 4) << 8 is kept instead of S: with the vertical weight also shifted left by
            // 8, (a * (S >> 4)) >> 16 is one v_mul_hi_u32_u24.
            auto hrow = [&](int sy, unsigned (&T)[3]) {
                const int o = (int)BM_MUL24(sy - sy_lo, pitch) + (int)((abase + BM_MUL24(sy, rs)) & 3u);
                unsigned w[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int A = o + (k ? xs1 : xs0);
                    const unsigned* q = reinterpret_cast<const unsigned*>(stage + (A & ~3));
                    const unsigned long long two = ((unsigned long long)q[1] << 32) | q[0];
                    w[k] = (unsigned)(two >> (8 * (A & 3)));
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {       // output channel c (RGB) = source byte 2 - c (BGR)
                    const unsigned S = BM_MUL24((w[0] >> (8 * (2 - c))) & 255u, ax.a0) + BM_MUL24((w[1] >> (8 * (2 - c))) & 255u, ax.a1);
                    T[c] = (S >> 4) << 8;
                }
            };
            unsigned T0[3] = {0, 0, 0}, T1[3] = {0, 0, 0};
            int have0 = -1, have1 = -1;
            int pr = pr0 + rp * run;
            int slot = pr % RING_ROWS;             // once per run; the ring slot then advances with the row
            unsigned char* dst = ring + slot * RING_ROW_BYTES + (dx + 3) * 8;
            const unsigned* lut_w = reinterpret_cast<const unsigned*>(lut_h);      // 4-byte entries (fp16 in the low half)
            for (int k = 0; k < run && pr < pr1; ++k, ++pr) {
                const int dy = pr - 3;
                unsigned lo = 0, hi = 0;            // h4 {R, G, B, 0} as two dwords
                if (dy >= 0 && dy < REID_IN_H) {
                    const unsigned ys = ytab[2 * dy], ya = ytab[2 * dy + 1];
                    const int s0 = (int)(ys & 0xffffu), s1 = (int)(ys >> 16);
                    const unsigned a0 = (ya & 0xffffu) << 8, a1 = (ya >> 16) << 8;
                    if (s0 != have0) {
                        if (s0 == have1) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) T0[c] = T1[c];
                        } else hrow(s0, T0);
                        have0 = s0;
                    }
                    if (s1 != have1) {
                        if (s1 == s0) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) T1[c] = T0[c];
                        } else hrow(s1, T1);
                        have1 = s1;
                    }
                    unsigned px3[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const unsigned v4 = (BM_MULHI24(a0, T0[c]) + BM_MULHI24(a1, T1[c]) + 2u) & ~3u;       // 4 v: byte offset of the table entry
                        px3[c] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(lut_w + c * 256) + v4);
                    }
                    lo = (px3[0] & 0xffffu) | (px3[1] << 16);
                    hi = (px3[2] & 0xffffu) | (HP ? 0x3C000000u : 0u);          // HP: fourth channel = 1.0 on real pixels
                }
                *reinterpret_cast<unsigned long long*>(dst) = ((unsigned long long)hi << 32) | lo;
                dst += RING_ROW_BYTES;
                if (++slot == RING_ROWS) { slot = 0; dst -= RING_ROWS * RING_ROW_BYTES; }
            }
        } else
        for (int pr = pr0 + rp; pr < pr1; pr += 4) {
            const int dy = pr - 3;
            h4 px = h4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (dy >= 0 && dy < REID_IN_H) {
                const ResizeAxis ay = resize_axis_y(dy, REID_IN_H, r.h > 0 ? r.h : 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    int v;
                    if (r.w == 0) v = 0;
                    else if (staged) {
                        // staged row sy starts at the byte offset its frame address has inside a dword
                        auto row_ptr = [&](int sy) {
                            const uintptr_t addr = reinterpret_cast<uintptr_t>(frame + (long)(r.y1 + sy) * row_stride + (long)r.x1 * 3);
                            return stage + (sy - sy_lo) * pitch + (int)(addr & 3) + (2 - c);
                        };
                        if (identity) v = row_ptr(dy)[dx * 3];
                        else {                      // area2: exact 2x shrink -> box filter
                            const unsigned char* p0 = row_ptr(2 * dy) + 6 * dx;
                            const unsigned char* p1 = row_ptr(2 * dy + 1) + 6 * dx;
                            v = (p0[0] + p0[3] + p1[0] + p1[3] + 2) >> 2;
                        }
                    } else {
                        v = resize_sample(frame + (long)r.y1 * row_stride + r.x1 * 3, row_stride, r, ax, ay, dy, dx, 2 - c,
                                          REID_IN_W, REID_IN_H);
                    }
                    px[c] = lut_h[2 * (c * 256 + v)];
                }
                if constexpr (HP) px[3] = (_Float16)1.f;
            }
            *reinterpret_cast<h4*>(ring + (pr % RING_ROWS) * RING_ROW_BYTES + (dx + 3) * 8) = px;
        }
        BM_PROF(3);
        __syncthreads();
        BM_PROF(4);
        if constexpr (BM_STEM_ASYNC) { if (band + 1 < 8) stage_rows(band_geom(band + 1)); }     // the staging area is idle until then
#if BM_STEM_STREAM
        // ---- 3. conv rows + pooling: strip t (16 conv pixels), pooled rows oy0 .. oy0 + 3 ----
        {
            const int t = wave & 3, half = wave >> 2;
            const int oy0 = 8 * band + 4 * half;
            const int rbase = 4 * oy0 - 2;                 // padded input row of (conv row 2 oy0 - 1, ky 0); conv row j of 9 is 2 oy0 - 1 + j
            const unsigned char* bcol = ring + (2 * (t * 16 + l16) + 2 * g) * 8;
            f4 acc[9];
#pragma unroll
            for (int i = 0; i < 23; ++i) {                 // input row rbase + i feeds conv row j with ky = i - 2 j
                const int jlo = i > 6 ? (i - 5) >> 1 : 0, jhi = (i >> 1) < 8 ? (i >> 1) : 8;
                if (oy0 == 0 && jlo == 0 && jhi == 0) continue;         // only conv row -1 (outside the image) reads these rows
                const int slot = (rbase + i + RING_ROWS) % RING_ROWS;
                const h8 b = *reinterpret_cast<const h8*>(bcol + slot * RING_ROW_BYTES);
                h8 b_s;
                if constexpr (HP) b_s = b * (_Float16)0.00048828125f;       // 2^-11: exact on byte values and on 1.0
#pragma unroll
                for (int j = jlo; j <= jhi; ++j) {
                    if (j == 0 && oy0 == 0) continue;                   // wave-uniform
                    const int ky = i - 2 * j;
                    acc[j] = BM_MFMA_F16_K32(a[ky], b, ky == 0 ? bias : acc[j]);
                    if constexpr (LO_LDS) acc[j] = BM_MFMA_F16_K32(*reinterpret_cast<const h8*>(alo_lds + (ky * 64 + lane) * 16), b_s, acc[j]);
                    else if constexpr (HP) acc[j] = BM_MFMA_F16_K32(a_lo[ky], b_s, acc[j]);
                }
            }
            float* edge = reinterpret_cast<float*>(stage + SRC_STAGE_BYTES + STEM_LUT_BYTES + 256 * 8) + (((band & 1) * 2 + half) * 4) * (4 * 16);
            f4 m[4];
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                f4 vm = max4(relu4(acc[2 * p4 + 1]), relu4(acc[2 * p4 + 2]));
                if (!(oy0 == 0 && p4 == 0)) vm = max4(vm, relu4(acc[2 * p4]));          // conv row -1 does not exist (wave-uniform)
                if (l16 == 15) *reinterpret_cast<f4*>(edge + (t * 4 + p4) * 16 + g * 4) = vm;
                m[p4] = vm;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {   // all values are >= 0 (post-ReLU), so a missing neighbour reads as 0
                    const float right = BM_ROW_SHL1_F32(vm[rr]);
                    float left = BM_ROW_SHR1_F32(vm[rr]);
                    if (l16 == 0) left = 0.f;
                    const float mm = m[p4][rr] > right ? m[p4][rr] : right;
                    m[p4][rr] = mm > left ? mm : left;
                }
                if ((l16 & 1) == 0 && l16 != 0) {
                    const int p = (oy0 + p4) * 32 + t * 8 + (l16 >> 1);
                    const h4 hh = to_h4(m[p4]);
                    *reinterpret_cast<h4*>(yout + (long)p * 16 + g * 4) = hh;
                    if constexpr (HP) *reinterpret_cast<h4*>(yout_lo + (long)p * 16 + g * 4) = residual_h4(m[p4], hh);
                }
            }
            BM_PROF(5);
            __syncthreads();
            BM_PROF(6);
            if (l16 == 0) {
#pragma unroll
                for (int p4 = 0; p4 < 4; ++p4) {
                    if (t > 0) m[p4] = max4(m[p4], *reinterpret_cast<const f4*>(edge + ((t - 1) * 4 + p4) * 16 + g * 4));
                    const int p = (oy0 + p4) * 32 + t * 8;
                    const h4 hh = to_h4(m[p4]);
                    *reinterpret_cast<h4*>(yout + (long)p * 16 + g * 4) = hh;
                    if constexpr (HP) *reinterpret_cast<h4*>(yout_lo + (long)p * 16 + g * 4) = residual_h4(m[p4], hh);
                }
            }
        }
#else
        static_assert(!HP, "the fp32-grade stem exists for the strip-wise conv phase (BM_STEM_STREAM)");
        // ---- 3. conv rows + pooling for pooled row oy ----
        const int oy = 8 * band + wave;
        f4 vprev = f4{0.f, 0.f, 0.f, 0.f};       // vertical max of the previous tile (for the left neighbour of lane 0)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f4 vm = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const int cy = 2 * oy + dc;
                if (cy < 0 || cy > 127) continue;          // wave-uniform
                f4 acc = bias;
                const int cx = t * 16 + l16;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    const int slot = (2 * cy + ky) % RING_ROWS;
                    const h8 b = *reinterpret_cast<const h8*>(ring + slot * RING_ROW_BYTES + (2 * cx + 2 * g) * 8);
                    acc = BM_MFMA_F16_K32(a[ky], b, acc);
                }
                vm = max4(vm, relu4(acc));
            }
            f4 m = vm;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {   // all values are >= 0 (post-ReLU), so a missing neighbour reads as 0
                const float right = BM_ROW_SHL1_F32(vm[rr]);
                float left = BM_ROW_SHR1_F32(vm[rr]);
                const float left_prev = BM_ROW_ROR1_F32(vprev[rr]);
                if (l16 == 0) left = t > 0 ? left_prev : 0.f;
                const float mm = m[rr] > right ? m[rr] : right;
                m[rr] = mm > left ? mm : left;
            }
            if ((l16 & 1) == 0) {
                const int p = oy * 32 + t * 8 + (l16 >> 1);
                *reinterpret_cast<h4*>(yout + (long)p * 16 + g * 4) = to_h4(m);
            }
            vprev = vm;
        }
        __syncthreads();
#endif
        BM_PROF(7);
    }
    BM_PROF_FLUSH();
}

__global__ void __launch_bounds__(512, 4) k_stem_resize_fused(const uint8_t* const* frames, const int* crop_stream, const float* boxes,
                                                           int box_stride, int W, int H, const float* lut, _Float16* __restrict__ out,
                                                           const unsigned char* __restrict__ wts, const int* __restrict__ count) {
    stem_resize_fused_body<false>(frames, crop_stream, boxes, box_stride, W, H, lut, out, nullptr, wts, count);
}
// the fp32-grade family's stem: (hi, lo) output planes, weights from pack_stem_hp_fused
__global__ void __launch_bounds__(512, BM_HP_STEM_LDS_LO ? 4 : 2) k_stem_resize_fused_hp(const uint8_t* const* frames, const int* crop_stream, const float* boxes,
                                                              int box_stride, int W, int H, _Float16* __restrict__ out_hi,
                                                              _Float16* __restrict__ out_lo, const unsigned char* __restrict__ wts,
                                                              const int* __restrict__ count) {
    stem_resize_fused_body<true>(frames, crop_stream, boxes, box_stride, W, H, nullptr, out_hi, out_lo, wts, count);
}

// crop -> resize -> normalise into the stem's fp16 RGBX layout (interior only; the 3-pixel border and the
// X channel of the buffer stay zero from allocation).  Same integer pipeline as k_crop_resize.
__global__ void k_crop_resize_rgbx(const uint8_t* const* frames, const int* crop_stream, const float* boxes,
                                   int box_stride, int W, int H, const float* lut, _Float16* out, int rows_per_block,
                                   const int* count, int pad) {
    if (count && (int)blockIdx.x >= *count) return;
    const int i = blockIdx.x;
    const int dx = threadIdx.x;
    const uint8_t* frame = frames[crop_stream[i]];
    const CropRect r = crop_rect(boxes + (long)i * box_stride, W, H);
    const long row_stride = (long)W * 3;
    const uint8_t* src = frame + (long)r.y1 * row_stride + r.x1 * 3;
    const ResizeAxis ax = resize_axis_x(dx, REID_IN_W, r.w > 0 ? r.w : 1);
    const PadGeom g = pad_geom(r, REID_IN_W, REID_IN_H);
    const int y0 = blockIdx.y * rows_per_block;
    for (int dy = y0; dy < y0 + rows_per_block && dy < REID_IN_H; ++dy) {
        h4 px;
        for (int c = 0; c < 3; ++c) {
            const int v = preprocess_sample(src, row_stride, r, g, pad, ax, dy, dx, 2 - c, REID_IN_W, REID_IN_H);
            px[c] = (_Float16)lut[c * 256 + v];
        }
        px[3] = (_Float16)0.f;
        *reinterpret_cast<h4*>(out + (((long)i * STEM_ROWS + dy + 3) * STEM_COLS + dx + 3) * 4) = px;
    }
}

}  // namespace bm
