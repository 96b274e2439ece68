// fp32-grade kernels for the wide OSNets (osnet_x1_0: 64 / 256 / 384 / 512 channels, BASELINE.json configuration 3's backbone):
// ReID "mode 2" at widths that are multiples of 32.
//
// Reference computation: OSNet.forward (eval), boxmot/reid/backbones/osnet.py:380-405 (OSBlock :212-260, LightConv3x3 :127-155,
// ChannelGate :161-209) in fp32 on the CPU (reid/backends/base_backend.py:197-217) -- north_star's comparator, tolerance 1e-3.
// The fp16 layer-per-launch family (osnet_wide_kernels.hpp, mode 1) meets that tolerance on the reference's own initialisation
// only (6e-3 on BatchNorm-calibrated weights).  This family carries EVERY matrix-pipe operand as an fp16 (hi, lo) pair --
// w = hi + lo, hi = fp16(w), lo = fp16(w - hi), ~22 significant bits -- three v_mfma_f32_16x16x32_f16 per K = 32 product tile
// (Wh.xh + Wh.xl + Wl.xh; the dropped lo.lo term is below fp32 rounding), fp32 accumulation, and keeps everything between the
// matrix products (depthwise 3x3, gates, pools, shortcuts, biases) in fp32 -- the recipe of the x0.25 family (reid_hp.hpp).
//
// Activations between kernels: two fp16 planes (hi, lo), each plain NHWC [crop][pixel][channel] (same bytes as fp32, and a
// consumer's MFMA B fragments are plain 16-byte loads from the two planes).
//
// Launch structure per OSBlock (4 launches instead of the fp16 family's 15):
//   k_gemm_hp          conv1 (cin -> mid, + bias, ReLU) as a GEMM over all pixels of the pass
//   k_chain_hp         ALL TEN LightConv3x3 layers of the block: a workgroup owns a band of image rows of one crop (plus a 4-row
//                      halo: the longest chain has four 3x3 layers), keeps the band's tensor in REGISTERS in the MFMA accumulator
//                      layout (which -- with the k-slots of the next layer's weights permuted to match -- is the next 1x1's B
//                      operand: a chain never leaves the register file), runs each 1x1 on (hi, lo) operands with the weight
//                      fragments staged in LDS, and each depthwise 3x3 through a double-buffered 16-channel LDS image in fp32;
//                      emits the four branch outputs of the band's rows and the band's channel sums (the gates' average pool)
//   k_gate_sum4_hp     the four ChannelGates (crop-wide, hence a launch boundary) and the gated branch sum
//   k_gemm_hp          conv3 + (downsample as a second operand pair | identity shortcut as a residual) + ReLU; the stage
//                      transitions carry their 2x2 average pool in the epilogue
// Algorithmic HBM bytes per crop: DESIGN.md section 4.6b.
#pragma once

#include <stdint.h>

#include "gemm_f16.hpp"
#include "osnet_wide_kernels.hpp"
#include "osnet_wide_hp_pack.hpp"
#include "reid_hp.hpp"

namespace bm {

// ---------------------------------------------------------------------------------------------------------------------------
// Stem: conv 7x7 / 2 (3 -> C0) + folded BN + ReLU + max pool 3x3 / 2 (osnet.py:294-295), k_wide_stem's structure on (hi, lo) operands.
//   crops_h / crops_l  fp16 RGBX planes with a 3-pixel zero border, [n][262][136][4] (k_crop_resize_rgbx_hl)
//   wts                A fragment pairs [ky][channel tile] (2 KiB each: hi then lo), k-slot j of lane group g -> tap kx = 2 g + (j >> 2),
//                      channel j & 3 of the RGBX pixel (pack_wide_stem_hp)
//   out_h / out_l      [n][64 * 32][C0], channels in the paired order (below)
// grid (16 bands of 4 pooled rows, crops), 256 threads; the band's 23 input rows of BOTH planes are staged once into LDS.
// ---------------------------------------------------------------------------------------------------------------------------
template <int C0>
__global__ void __launch_bounds__(256) k_wide_stem_hp(const _Float16* __restrict__ crops_h, const _Float16* __restrict__ crops_l,
                                                      const unsigned char* __restrict__ wts, const float* __restrict__ bias,
                                                      _Float16* __restrict__ out_h, _Float16* __restrict__ out_l) {
    static_assert(C0 == 32 || C0 == 64, "stem width");
    constexpr int NCT = C0 / 16, NRG = 4 / NCT, PR = WSTEM_PBAND / NRG;
    constexpr int PLANE_HALVES = WSTEM_IN_ROWS * WSTEM_COLS * 4;
    __shared__ __attribute__((aligned(16))) _Float16 sIn[2 * PLANE_HALVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const int band = blockIdx.x;
    const long crop = blockIdx.y;
    const int in0 = 16 * band - 2;
    const _Float16* imh = crops_h + crop * WSTEM_ROWS * (long)(WSTEM_COLS * 4);
    const _Float16* iml = crops_l + crop * WSTEM_ROWS * (long)(WSTEM_COLS * 4);
    for (int e = tid; e < WSTEM_IN_ROWS * WSTEM_COLS / 2; e += 256) {            // 16-byte chunks = 2 pixels
        const int r = e / (WSTEM_COLS / 2), row = in0 + r;
        cu4 vh = cu4{0u, 0u, 0u, 0u}, vl = cu4{0u, 0u, 0u, 0u};
        if (row >= 0 && row < WSTEM_ROWS) {
            const long o = (long)row * (WSTEM_COLS * 4) + (e - r * (WSTEM_COLS / 2)) * 8;
            vh = *reinterpret_cast<const cu4*>(imh + o);
            vl = *reinterpret_cast<const cu4*>(iml + o);
        }
        *reinterpret_cast<cu4*>(sIn + e * 8) = vh;
        *reinterpret_cast<cu4*>(sIn + PLANE_HALVES + e * 8) = vl;
    }
    const int ct = wave % NCT, rg = wave / NCT;
    h8 ah[7], al[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
        const unsigned char* a = wts + (long)(ky * NCT + ct) * HP_FRAG_PAIR;
        ah[ky] = *reinterpret_cast<const h8*>(a + lane * 16);
        al[ky] = *reinterpret_cast<const h8*>(a + 1024 + lane * 16);
    }
    const f4 bv = *reinterpret_cast<const f4*>(bias + 16 * ct + 4 * g);
    __syncthreads();
    auto conv_row = [&](int cy, f4 (&row)[4]) {
        if (cy < 0 || cy > 127) {
#pragma unroll
            for (int t = 0; t < 4; ++t) row[t] = f4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        const int lr = 2 * cy - in0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cx = 16 * t + l16;
            f4 acc = bv;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const int o = ((lr + ky) * WSTEM_COLS + 2 * cx + 2 * g) * 4;
                const h8 bh = *reinterpret_cast<const h8*>(sIn + o), bl = *reinterpret_cast<const h8*>(sIn + PLANE_HALVES + o);
                acc = mm3r(ah[ky], al[ky], bh, bl, acc);
            }
            row[t] = relu4(acc);
        }
    };
    const int py0 = WSTEM_PBAND * band + rg * PR;
    f4 prev[4], mid[4], next[4];
    conv_row(2 * py0 - 1, prev);
#pragma unroll 1
    for (int py = py0; py < py0 + PR; ++py) {
        conv_row(2 * py, mid);
        conv_row(2 * py + 1, next);
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
                const float mm = m[r] > right ? m[r] : right;
                m[r] = mm > left ? mm : left;
            }
            if ((l16 & 1) == 0) {
                const long p = (long)py * 32 + t * 8 + (l16 >> 1);
                h4 hh, ll;
                split4(m, hh, ll);
                const long o = (crop * 2048 + p) * C0 + 32 * (ct >> 1) + 8 * g + 4 * (ct & 1);       // paired order (hp_paired_pos)
                *reinterpret_cast<h4*>(out_h + o) = hh;
                *reinterpret_cast<h4*>(out_l + o) = ll;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) prev[t] = next[t];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_gemm_hp: C[m][n] = sum_k X[m][k] * Wt[n][k] (+ bias) with BOTH operands as (hi, lo) fp16 planes -- the 1x1 convolutions.
//   Xh / Xl [M][K], Wh / Wl [N][K] row-major fp16;  N % BN == 0, BN in {32, 64, 96, 128};  K % 32 == 0;  any M.  The k order of X
//   and W only has to agree; outputs (and the residual) are in the paired order below (EPI 4: fp32 in logical order).
// k_gemm_f16_glds's structure (gemm_f16.hpp): 128 rows x BN features x 32 k-tiles, operands HBM -> LDS by global_load_lds into a
// swizzled image (chunk c of row r at slot c ^ ((r >> 1) & 3): conflict-free 16-byte fragment reads), two LDS buffers, ONE barrier
// per k-tile, 4 waves as 2 x 2 of 64 rows x BN / 2 features, weight rows as the MFMA A operand.  A k-tile is four operand tiles
// (Wh, Wl, Xh, Xl) and 3 MFMAs per fragment pair.
//   EPI 0  (ReLU when relu) -> (hi, lo) planes [M][N]
//   EPI 1  + (hi, lo) residual [M][N], (ReLU when relu) -> (hi, lo) planes             (conv3 + identity shortcut)
//   EPI 2 / 3  ReLU + 2 x 2 average pool over an image of width 32 / 16 -> pooled (hi, lo) planes [M / 4][N]      (transitions)
//   EPI 4  (ReLU when relu) -> fp32 [M][N]                                              (the head's FC)
// ext.X2h .. K2: a second operand pair accumulated into the same tile (conv3(x2) + downsample(x): the shortcut tensor never exists).
// ---------------------------------------------------------------------------------------------------------------------------
// Storage order of every (hi, lo) activation tensor this family writes from MFMA accumulators ("paired order"): inside each block
// of 32 channels, logical channel 16 p + 4 g + r (p = 0, 1: the two 16-channel accumulator tiles, g = lane group, r = register) sits
// at position 8 g + 4 p + r -- the eight values a lane holds of a tile PAIR are contiguous: one 16-byte store per plane instead of
// two 8-byte ones (the store path is issue-bound: 16-byte stores run at about twice the bytes per clock of 8-byte ones), and as an
// MFMA B operand chunk g of a k-tile is again one 16-byte load with k-slot j <-> channel 16 (j >> 2) + 4 g + (j & 3) (the weights'
// k columns are permuted to match when they are packed, osnet_wide_hp_pack.hpp).  Elementwise kernels never notice.
// (hp_paired_pos: osnet_wide_hp_pack.hpp)
__device__ inline void hp_store_pair(_Float16* ph, _Float16* pl, f4 v0, f4 v1) {
    h4 h0, l0, h1, l1;
    split4(v0, h0, l0);
    split4(v1, h1, l1);
    *reinterpret_cast<h8*>(ph) = cat8(h0, h1);
    *reinterpret_cast<h8*>(pl) = cat8(l0, l1);
}
__device__ inline void hp_load_pair(const _Float16* ph, const _Float16* pl, f4& v0, f4& v1) {
    const h8 hh = *reinterpret_cast<const h8*>(ph), ll = *reinterpret_cast<const h8*>(pl);
#pragma unroll
    for (int r = 0; r < 4; ++r) { v0[r] = (float)hh[r] + (float)ll[r]; v1[r] = (float)hh[4 + r] + (float)ll[4 + r]; }
}

struct GemmHpExt {
    const _Float16 *X2h = nullptr, *X2l = nullptr, *W2h = nullptr, *W2l = nullptr;
    int K2 = 0;
};

// DB = 2: two LDS operand buffers, one barrier per k-tile (deep products: the transitions, conv5); DB = 1: one buffer, two barriers
// per k-tile, half the LDS and <= 128 registers -- FOUR workgroups per CU instead of two: the 1x1 convolutions of the OSBlocks have
// 2 .. 12 k-tiles and move 128 x (K + N) x 4 bytes per tile; they wait on HBM, and what hides that is more tiles in flight per CU
template <int BN, int DB = 2>
__host__ __device__ constexpr int gemm_hp_lds_bytes() { return DB * (2 * BN * 32 + 2 * 128 * 32) * 2; }

template <int EPI, int BN, int DB = 2>
__global__ void __launch_bounds__(256, DB == 1 ? 4 : 1) k_gemm_hp(const _Float16* __restrict__ Xh, const _Float16* __restrict__ Xl,
                                                 const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl,
                                                 const float* __restrict__ bias, void* __restrict__ Oh, void* __restrict__ Ol,
                                                 const _Float16* __restrict__ res_h, const _Float16* __restrict__ res_l, int M, int N, int K,
                                                 int relu, GemmHpExt ext) {
    static_assert(BN == 32 || BN == 64 || BN == 96 || BN == 128, "feature tile");
    // waves as WM x WN: 2 x 2 of 64 rows x BN / 2 features when BN / 2 is a whole number of 32-channel storage blocks, else 4 x 1 of
    // 32 rows x BN features -- a wave always owns PAIRS of 16-feature tiles (hp_store_pair)
    constexpr int WN = BN % 64 == 0 ? 2 : 1, WM = 4 / WN;
    constexpr int RT = 8 / WM;                            // 16-row tiles per wave
    constexpr int NTW = BN / WN / 16;                     // 16-feature MFMA tiles per wave
    static_assert(NTW % 2 == 0, "tile pairs");
    static_assert(EPI < 2 || EPI > 3 || WN == 2, "the pooling epilogues fold row tiles of one wave");
    constexpr int TW = BN * 32, TX = 128 * 32;            // halves per operand tile
    constexpr int BUF = 2 * TW + 2 * TX;
    constexpr int WBLK = BN / 16;                         // 16-row copy blocks of a weight tile (an activation tile has 8)
    BM_DYNAMIC_LDS_T(unsigned char, lds_raw);
    _Float16* lds = reinterpret_cast<_Float16*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = BM_UNIFORM_I32(tid >> 6), g = lane >> 4, l16 = lane & 15;
    const int wn = WN == 2 ? wave >> 1 : 0, wm = WN == 2 ? wave & 1 : wave;
    const int row_w = wm * (128 / WM), col_w = wn * (BN / WN);
    int mt, nt;
    gemm_tile_of_block((M + GEMM_BM - 1) / GEMM_BM, N / BN, mt, nt);
    const long m0 = (long)mt * GEMM_BM;
    const int n0 = nt * BN;
    f4 acc[NTW][RT];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < RT; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    // a copy moves one 16-row block of an operand tile (64 lanes x 16 bytes): lane -> (row lr of the block, chunk slot lc); the
    // swizzle is applied to the SOURCE chunk (the destination is lane-linear)
    const int lr = lane >> 2, lc = (lane & 3) ^ ((lr >> 1) & 3);
    const int nk1 = K / 32, nk = nk1 + ext.K2 / 32;
    auto issue = [&](int kt, int buf) {
        _Float16* d = lds + buf * BUF;
        const bool second = kt >= nk1;
        const int kk = second ? ext.K2 : K;
        const long k0 = (long)(second ? kt - nk1 : kt) * 32 + 8 * lc;
        const _Float16* xh = second ? ext.X2h : Xh;
        const _Float16* xl = second ? ext.X2l : Xl;
        const _Float16* wh = second ? ext.W2h : Wh;
        const _Float16* wl = second ? ext.W2l : Wl;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                     // activation blocks 2 wave, 2 wave + 1 of both planes
            const int blk = 2 * wave + j;
            long m = m0 + blk * 16 + lr;
            if (m >= M) m = M - 1;
            BM_GLDS16(xh + m * kk + k0, d + 2 * TW + blk * 512, lane);
            BM_GLDS16(xl + m * kk + k0, d + 2 * TW + TX + blk * 512, lane);
        }
#pragma unroll
        for (int j = 0; j < (WBLK + 3) / 4; ++j) {        // weight blocks wave, wave + 4
            const int blk = wave + 4 * j;
            if (WBLK % 4 == 0 || blk < WBLK) {
                const long o = (long)(n0 + blk * 16 + lr) * kk + k0;
                BM_GLDS16(wh + o, d + blk * 512, lane);
                BM_GLDS16(wl + o, d + TW + blk * 512, lane);
            }
        }
    };
    issue(0, 0);
    // the shortcut operand of the epilogue does not depend on the product: fetch it now, its latency hides behind the k-loop
    // (DB = 1: the other workgroups of the CU hide it, and 64 registers of prefetch would not fit the 128 of that form)
    constexpr bool RES_EARLY = EPI == 1 && DB == 2;
    h8 rrh[RES_EARLY ? NTW / 2 : 1][RT], rrl[RES_EARLY ? NTW / 2 : 1][RT];
    if constexpr (RES_EARLY) {
#pragma unroll
        for (int q = 0; q < NTW / 2; ++q)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                long m = m0 + row_w + t * 16 + l16;
                if (m >= M) m = M - 1;
                const long o = m * N + n0 + col_w + 32 * q + 8 * g;
                rrh[q][t] = *reinterpret_cast<const h8*>(res_h + o);
                rrl[q][t] = *reinterpret_cast<const h8*>(res_l + o);
            }
    }
    for (int kt = 0; kt < nk; ++kt) {
        BM_WAIT_VM0();                      // this wave's copies of tile kt have landed ...
        __syncthreads();                    // ... everybody's have, and buffer (kt + 1) & 1 is free
        if (DB == 2 && kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const _Float16* sWh = lds + (DB == 2 ? (kt & 1) * BUF : 0);
        const _Float16* sWl = sWh + TW;
        const _Float16* sXh = sWh + 2 * TW;
        const _Float16* sXl = sXh + TX;
        h8 ah[NTW], al[NTW], bh[RT], bl[RT];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int ra = col_w + t * 16 + l16, o = ra * 32 + 8 * (g ^ ((ra >> 1) & 3));
            ah[t] = *reinterpret_cast<const h8*>(sWh + o);
            al[t] = *reinterpret_cast<const h8*>(sWl + o);
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const int rb = row_w + t * 16 + l16, o = rb * 32 + 8 * (g ^ ((rb >> 1) & 3));
            bh[t] = *reinterpret_cast<const h8*>(sXh + o);
            bl[t] = *reinterpret_cast<const h8*>(sXl + o);
        }
#pragma unroll
        for (int p = 0; p < NTW; ++p)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[p][t] = mm3r(ah[p], al[p], bh[t], bl[t], acc[p][t]);
        if (DB == 1 && kt + 1 < nk) {
            __syncthreads();                // every wave has read the tile out of the single buffer
            issue(kt + 1, 0);
        }
    }
    if constexpr (EPI == 2 || EPI == 3) {
        // transition layer: ReLU(conv + bias) then the 2 x 2 average pool, on the accumulators (k_gemm_f16_glds EPI 5 / 6)
        constexpr int POOL_W = EPI == 2 ? 32 : 16, dt = POOL_W / 16, sh = EPI == 2 ? 5 : 4;
#pragma unroll
        for (int q = 0; q < NTW / 2; ++q) {
            const int n = n0 + col_w + 32 * q;
            const f4 bv0 = *reinterpret_cast<const f4*>(bias + n + 4 * g), bv1 = *reinterpret_cast<const f4*>(bias + n + 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if ((t / dt) & 1) continue;                      // the lower image row of a pair: folded into its upper one
                const long m = m0 + row_w + t * 16 + l16;
                f4 o[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float b = p ? bv1[r] : bv0[r];
                        const float top = acc[2 * q + p][t][r] + b, bot = acc[2 * q + p][(t + dt) % RT][r] + b;
                        float v = BM_RELU_F32(top) + BM_RELU_F32(bot);
                        v = v + BM_QUAD_SWAP1_F32(v);
                        o[p][r] = v * 0.25f;
                    }
                if (m < M && (l16 & 1) == 0) {
                    const long qq = m >> sh, x = m & (POOL_W - 1);                    // qq = crop * H + y (y even)
                    const long off = ((qq >> 1) * (POOL_W / 2) + (x >> 1)) * N + n + 8 * g;
                    hp_store_pair(static_cast<_Float16*>(Oh) + off, static_cast<_Float16*>(Ol) + off, o[0], o[1]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < NTW / 2; ++q) {
        const int n = n0 + col_w + 32 * q;
        f4 bv0 = f4{0.f, 0.f, 0.f, 0.f}, bv1 = bv0;
        if (bias) { bv0 = *reinterpret_cast<const f4*>(bias + n + 4 * g); bv1 = *reinterpret_cast<const f4*>(bias + n + 16 + 4 * g); }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const long m = m0 + row_w + t * 16 + l16;
            if (m >= M) continue;
            f4 v0 = acc[2 * q][t] + bv0, v1 = acc[2 * q + 1][t] + bv1;
            if constexpr (EPI == 1) {
                h8 rh, rl;
                if constexpr (RES_EARLY) { rh = rrh[q][t]; rl = rrl[q][t]; }
                else {
                    const long o = m * N + n + 8 * g;
                    rh = *reinterpret_cast<const h8*>(res_h + o); rl = *reinterpret_cast<const h8*>(res_l + o);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v0[r] += (float)rh[r] + (float)rl[r];
                    v1[r] += (float)rh[4 + r] + (float)rl[4 + r];
                }
            }
            if (relu) { v0 = relu4(v0); v1 = relu4(v1); }
            if constexpr (EPI == 4) {                  // fp32, logical channel order
                *reinterpret_cast<f4*>(static_cast<float*>(Oh) + m * N + n + 4 * g) = v0;
                *reinterpret_cast<f4*>(static_cast<float*>(Oh) + m * N + n + 16 + 4 * g) = v1;
            } else {
                const long off = m * N + n + 8 * g;
                hp_store_pair(static_cast<_Float16*>(Oh) + off, static_cast<_Float16*>(Ol) + off, v0, v1);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_chain_hp: the four LightConv chains of an OSBlock (osnet.py:223-241, 249-252: conv2a .. conv2d applied to x1) for a band of
// image rows of one crop.
//   x1h / x1l   (hi, lo) [n][H * W][C]: conv1's output (channels in the paired order, as every tensor here)
//   wts         ten records (LightConv a0, b0, b1, c0 .. c2, d0 .. d3), each: 1x1 A fragment pairs [out tile][k-step] with the k-slots
//               in the accumulator order (slot j of lane group g, step s <-> channel 16 (2 s + (j >> 2)) + 4 g + (j & 3)), then the
//               depthwise taps fp32 [channel tile][g][tap][4] with BN folded, then the fp32 bias [C]   (pack_chain_hp)
//   yh / yl     (hi, lo) [4 branches][n][H * W][C]: the branch outputs (post ReLU of each chain's last LightConv)
//   gap_part    fp32 [4][n][bands][C]: sums over the band's pixels of each branch output (the gates' average pool)
// grid (H / R bands, crops), 512 threads (8 waves).  The window of a band is WR = R + 2 HALO rows (HALO = 4 when the image does not
// fit one workgroup's registers: the longest chain is four 3x3 layers; rows outside the image are zero -- the zero padding of
// every depthwise layer).  Tile = 16 consecutive pixels of the window in row-major order; a wave owns NT consecutive tiles; a
// lane holds 4 channels (16 ct + 4 g + r) of its pixel (l16) per channel tile: `cur[NT][CT]`.
//   1x1 (linear):  B = the (hi, lo) split of `cur` in registers, A pairs from LDS (staged per layer by asynchronous copies issued a
//                  layer ahead), 3 MFMAs per (out tile, k-step); results overwrite `cur`.
//   depthwise 3x3 + bias + ReLU: per 16-channel slice, `cur` -> LDS image (fp32, planes [g][row][x] of 16-byte pixels with a zero
//                  halo, plane stride a multiple of 256 bytes: conflict-free b128 accesses) -> barrier -> each lane reads its 3 x 3
//                  neighbourhood (a sliding 3-row window down the wave's rows: 3 reads per input row).  Two slice buffers: the
//                  write of slice ct + 1 and the reads of slice ct share a barrier interval.
// ---------------------------------------------------------------------------------------------------------------------------
// tools/wide_hp_prof -DBM_CHAIN_PROF: shader-clock cycles per wave of each phase of k_chain_hp, summed over all waves
#ifdef BM_CHAIN_PROF
__device__ unsigned long long g_chain_prof[8];
#define BM_CPROF_DECL() unsigned long long cp_t = clock64(), cp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define BM_CPROF(k) do { unsigned long long t_ = clock64(); cp_acc[k] += t_ - cp_t; cp_t = t_; } while (0)
#define BM_CPROF_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_chain_prof[k_], cp_acc[k_]); } while (0)
#else
#define BM_CPROF_DECL() ((void)0)
#define BM_CPROF(k) ((void)0)
#define BM_CPROF_FLUSH() ((void)0)
#endif
#ifndef BM_CHAIN_TGP
#define BM_CHAIN_TGP 2              // tiles whose B operands are resident while the A pairs stream by (A/B switch)
#endif

template <int C, int W, int WR, int HALO>
struct ChainGeo {
    static constexpr int CT = C / 16, KS = C / 32, R = WR - 2 * HALO;
    static constexpr int NW = 8, NTILES = WR * W / 16, NT = NTILES / NW;
    static constexpr int TGP = NT % BM_CHAIN_TGP == 0 ? BM_CHAIN_TGP : 1;                 // tiles whose B operands are resident while the A pairs stream by
    static constexpr int ROWP = (W + 2) * 16;
    static constexpr int PLANE = ((WR + 2) * ROWP + 255) / 256 * 256;
    static constexpr int SLICE = 4 * PLANE;
    static constexpr int PW_BYTES = C * C * 4;                       // CT x KS pairs of 2 KiB
    static constexpr int DW_BYTES = C * 9 * 4 + C * 4;
    static constexpr int LREC = PW_BYTES + DW_BYTES;
    static constexpr int OFF_PW = 2 * SLICE, OFF_DW = OFF_PW + PW_BYTES, OFF_GS = OFF_DW + 2 * DW_BYTES;
    static constexpr int LDS_BYTES = OFF_GS + NW * C * 4;
    static_assert(C % 32 == 0 && NTILES % NW == 0 && CT % 2 == 0, "tiling");
    static_assert(W == 32 || W == 16 || (W == 8 && NT == 1), "a wave's tiles form vertical runs (or it owns a single tile)");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int C, int W, int WR, int HALO>
__global__ void __launch_bounds__(512, 2) k_chain_hp(const _Float16* __restrict__ x1h, const _Float16* __restrict__ x1l,
                                                     const unsigned char* __restrict__ wts, _Float16* __restrict__ yh,
                                                     _Float16* __restrict__ yl, float* __restrict__ gap_part, int H, long n_crops) {
    using G = ChainGeo<C, W, WR, HALO>;
    constexpr int CT = G::CT, KS = G::KS, NT = G::NT, TGP = G::TGP, R = G::R;
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = BM_UNIFORM_I32(tid >> 6), g = lane >> 4, l16 = lane & 15;
    const int band = blockIdx.x, nbands = gridDim.x;
    const long crop = blockIdx.y;
    const int P = H * W;
    const int y0 = band * R - HALO;                      // image row of window row 0
    unsigned char* wpw = lds + G::OFF_PW;
    float* gs = reinterpret_cast<float*>(lds + G::OFF_GS);

    // window position of this lane's pixel in tile i of the wave
    auto q_of = [&](int i) { return (wave * NT + i) * 16 + l16; };
    auto wr_of = [&](int i) { return q_of(i) / W; };
    auto x_of = [&](int i) { return q_of(i) % W; };
    auto in_image = [&](int i) { return (unsigned)(y0 + wr_of(i)) < (unsigned)H; };
    auto pix = [&](int i) { return g * G::PLANE + (wr_of(i) + 1) * G::ROWP + (x_of(i) + 1) * 16; };

    // weights of layer l: the 1x1 pairs -> wpw, taps + bias -> the dw buffer (l & 1); asynchronous, 1 KiB per wave and copy
    auto stage_weights = [&](int l) {
        const unsigned char* src = wts + (long)l * G::LREC;
        for (int c = wave * 1024; c < G::PW_BYTES; c += G::NW * 1024) BM_GLDS16(src + c + lane * 16, wpw + c, lane);
        unsigned char* wd = lds + G::OFF_DW + (l & 1) * G::DW_BYTES;
        for (int c = wave * 1024; c < G::DW_BYTES; c += G::NW * 1024)
            if (c + lane * 16 < G::DW_BYTES) BM_GLDS16(src + G::PW_BYTES + c + lane * 16, wd + c, lane);
    };
    BM_CPROF_DECL();
    stage_weights(0);
    for (int e = tid * 16; e < 2 * G::SLICE; e += 512 * 16) *reinterpret_cast<f4*>(lds + e) = f4{0.f, 0.f, 0.f, 0.f};   // halo = zero padding
    BM_WAIT_VM0();
    __syncthreads();
    BM_CPROF(0);

    int li = 0;
#pragma unroll 1
    for (int br = 0; br < 4; ++br) {
        f4 cur[NT][CT];
        // branch input: x1 (rows outside the image are the zero padding); paired storage order: a tile pair per 16-byte load
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const bool ok = in_image(i);
            const long o = (crop * P + (long)(ok ? y0 + wr_of(i) : 0) * W + x_of(i)) * C + 8 * g;
#pragma unroll
            for (int q = 0; q < CT / 2; ++q) {
                f4 v0, v1;
                hp_load_pair(x1h + o + 32 * q, x1l + o + 32 * q, v0, v1);
                const f4 z = f4{0.f, 0.f, 0.f, 0.f};
                cur[i][2 * q] = ok ? v0 : z;
                cur[i][2 * q + 1] = ok ? v1 : z;
            }
        }
        BM_CPROF(1);
#pragma unroll 1
        for (int k = 0; k <= br; ++k, ++li) {
            // ---- 1x1 (linear, C -> C) ----
#pragma unroll
            for (int i0 = 0; i0 < NT; i0 += TGP) {
                h8 bh[TGP][KS], bl[TGP][KS];
#pragma unroll
                for (int t = 0; t < TGP; ++t)
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        h4 h0, l0, h1, l1;
                        split4(cur[i0 + t][2 * s], h0, l0);
                        split4(cur[i0 + t][2 * s + 1], h1, l1);
                        bh[t][s] = cat8(h0, h1); bl[t][s] = cat8(l0, l1);
                    }
#pragma unroll
                for (int co = 0; co < CT; ++co) {
                    f4 acc[TGP];
#pragma unroll
                    for (int t = 0; t < TGP; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const unsigned char* a = wpw + (long)(co * KS + s) * HP_FRAG_PAIR;
                        const h8 ah = *reinterpret_cast<const h8*>(a + lane * 16), al = *reinterpret_cast<const h8*>(a + 1024 + lane * 16);
#pragma unroll
                        for (int t = 0; t < TGP; ++t) acc[t] = mm3r(ah, al, bh[t][s], bl[t][s], acc[t]);
                    }
#pragma unroll
                    for (int t = 0; t < TGP; ++t) cur[i0 + t][co] = acc[t];
                }
                BM_SCHED_FENCE();
            }
            BM_CPROF(2);
            // ---- depthwise 3x3 (pad 1) + bias + ReLU, slice by slice ----
            const unsigned char* wdl = lds + G::OFF_DW + (li & 1) * G::DW_BYTES;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                unsigned char* img = lds + (ct & 1) * G::SLICE;
#pragma unroll
                for (int i = 0; i < NT; ++i) *reinterpret_cast<f4*>(img + pix(i)) = cur[i][ct];
                if (ct == CT - 1) BM_WAIT_VM0();            // the next layer's weights (requested below, at slice 0) have landed
                __syncthreads();
                BM_CPROF(3);
                if (ct == 0 && li + 1 < 10) stage_weights(li + 1);       // every wave is past this layer's 1x1: wpw is free
                f4 wd[9];
                const f4* wsrc = reinterpret_cast<const f4*>(wdl) + (ct * 4 + g) * 9;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) wd[tap] = wsrc[tap];
                const f4 bias = *reinterpret_cast<const f4*>(wdl + C * 9 * 4 + (16 * ct + 4 * g) * 4);
                if constexpr (W == 8) {
                    const unsigned char* cb = img + pix(0);
                    f4 o = bias;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap)
                        o = fma_f4(wd[tap], *reinterpret_cast<const f4*>(cb + (tap / 3 - 1) * G::ROWP + (tap % 3 - 1) * 16), o);
                    cur[0][ct] = in_image(0) ? relu4(o) : f4{0.f, 0.f, 0.f, 0.f};
                } else {
                    constexpr int NSEQ = W / 16, L = NT / NSEQ;          // column strips of the wave, rows per strip
#pragma unroll
                    for (int sq = 0; sq < NSEQ; ++sq) {
                        const unsigned char* cb = img + pix(sq);          // first row of the strip
                        f4 acc[3];
#pragma unroll
                        for (int rr = 0; rr < L + 2; ++rr) {              // input row (first row of the strip) - 1 + rr
                            const unsigned char* rp = cb + (rr - 1) * G::ROWP;
                            const f4 v0 = *reinterpret_cast<const f4*>(rp - 16), v1 = *reinterpret_cast<const f4*>(rp),
                                     v2 = *reinterpret_cast<const f4*>(rp + 16);
                            if (rr >= 2) {                                  // completes output row rr - 2
                                f4 a = acc[(rr - 2) % 3];
                                a = fma_f4(wd[6], v0, a); a = fma_f4(wd[7], v1, a); a = fma_f4(wd[8], v2, a);
                                const int i = (rr - 2) * NSEQ + sq;
                                cur[i][ct] = in_image(i) ? relu4(a) : f4{0.f, 0.f, 0.f, 0.f};
                            }
                            if (rr >= 1 && rr <= L) {
                                f4 a = acc[(rr - 1) % 3];
                                a = fma_f4(wd[3], v0, a); a = fma_f4(wd[4], v1, a); a = fma_f4(wd[5], v2, a);
                                acc[(rr - 1) % 3] = a;
                            }
                            if (rr <= L - 1) {
                                f4 a = fma_f4(wd[0], v0, bias);
                                a = fma_f4(wd[1], v1, a); a = fma_f4(wd[2], v2, a);
                                acc[rr % 3] = a;
                            }
                        }
                    }
                }
                BM_CPROF(4);
            }
        }
        // ---- branch output: the band's own rows -> (hi, lo) planes; channel sums of those rows for the gate ----
        _Float16* oh = yh + ((long)br * n_crops + crop) * P * C;
        _Float16* ol = yl + ((long)br * n_crops + crop) * P * C;
        f4 sum[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) sum[ct] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int wr = wr_of(i);
            const bool own = wr >= HALO && wr < HALO + R;
            if (own) {
                const long o = ((long)(y0 + wr) * W + x_of(i)) * C + 8 * g;
#pragma unroll
                for (int q = 0; q < CT / 2; ++q) {
                    hp_store_pair(oh + o + 32 * q, ol + o + 32 * q, cur[i][2 * q], cur[i][2 * q + 1]);
                    sum[2 * q] += cur[i][2 * q];
                    sum[2 * q + 1] += cur[i][2 * q + 1];
                }
            }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = sum[ct][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (l16 == 0) gs[wave * C + 16 * ct + 4 * g + r] = v;
            }
        __syncthreads();
        if (tid < C) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < G::NW; ++w) s += gs[w * C + tid];
            gap_part[(((long)br * n_crops + crop) * nbands + band) * C + tid] = s;
        }
        __syncthreads();
        BM_CPROF(5);
    }
    BM_CPROF_FLUSH();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Unified aggregation gate over the four branches (osnet.py:247-258, ChannelGate :161-209) on (hi, lo) planes:
//   x2[p][c] = sum_b y_b[p][c] * sigmoid(fc2(relu(fc1(mean_p y_b))))[c]
// grid (crops, pixel blocks), 256 threads; gap_part fp32 [4][n][bands][C] from k_chain_hp; y (hi, lo) [4][n][P][C].
// ---------------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_gate_sum4_hp(const _Float16* __restrict__ yh, const _Float16* __restrict__ yl,
                                                      const float* __restrict__ gap_part, const float* __restrict__ fc1_w,
                                                      const float* __restrict__ fc1_b, const float* __restrict__ fc2_w,
                                                      const float* __restrict__ fc2_b, _Float16* __restrict__ oh, _Float16* __restrict__ ol,
                                                      int P, int nbands, long n_crops, int pix_per_block) {
    constexpr int HID = C / 16, CG = C / 8;
    __shared__ float s_mean[4][C];
    __shared__ float s_h[4][HID];
    __shared__ __attribute__((aligned(16))) float s_g[4][C];            // indexed by the STORED position of the channel (paired order)
    const long n = blockIdx.x;
    const int tid = threadIdx.x;
    for (int e = tid; e < 4 * C; e += 256) {
        const int b = e / C, c = e - b * C;
        const float* gp = gap_part + ((long)b * n_crops + n) * nbands * C + c;
        float s = 0.f;
        for (int k = 0; k < nbands; ++k) s += gp[(long)k * C];
        s_mean[b][c] = s / (float)P;
    }
    __syncthreads();
    if (tid < 4 * HID) {
        const int b = tid / HID, k = tid - b * HID;
        float h = fc1_b[k];
        for (int c = 0; c < C; ++c) h += fc1_w[k * C + c] * s_mean[b][c];
        s_h[b][k] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < 4 * C; e += 256) {
        const int b = e / C, c = e - b * C;
        float v = fc2_b[c];
        for (int k = 0; k < HID; ++k) v += fc2_w[c * HID + k] * s_h[b][k];
        s_g[b][hp_paired_pos(c)] = 1.f / (1.f + BM_EXPF(-v));
    }
    __syncthreads();
    const long p0 = (long)blockIdx.y * pix_per_block;
    const int items = pix_per_block * CG;
    const long bstride = n_crops * P * (long)C;
    for (int e = tid; e < items; e += 256) {
        const int pl = e / CG, cg = e - pl * CG;
        if (p0 + pl >= P) break;
        const long off = ((n * P + p0 + pl) * C) + cg * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const h8 a = *reinterpret_cast<const h8*>(yh + b * bstride + off), c = *reinterpret_cast<const h8*>(yl + b * bstride + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_fmaf((float)a[j] + (float)c[j], s_g[b][cg * 8 + j], v[j]);
        }
        h8 hh, ll;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            hh[j] = (_Float16)v[j];
            ll[j] = (_Float16)(v[j] - (float)hh[j]);
        }
        *reinterpret_cast<h8*>(oh + off) = hh;
        *reinterpret_cast<h8*>(ol + off) = ll;
    }
}

// head: global average pool of the (hi, lo) conv5 output -> (hi, lo) [n][C] (the FC GEMM's activation operand)
__global__ void __launch_bounds__(256) k_wide_gap_hp(const _Float16* __restrict__ in_h, const _Float16* __restrict__ in_l,
                                                     _Float16* __restrict__ out_h, _Float16* __restrict__ out_l, int P, int C, long total8) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (crop, 8-channel group)
    if (e >= total8) return;
    const int C8 = C / 8;
    const long n = e / C8;
    const int cg = (int)(e - n * C8);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    const long base = n * P * (long)C + cg * 8;
    for (int k = 0; k < P; ++k) {
        const h8 a = *reinterpret_cast<const h8*>(in_h + base + (long)k * C), b = *reinterpret_cast<const h8*>(in_l + base + (long)k * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)a[j] + (float)b[j];
    }
    h8 hh, ll;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = s[j] / (float)P;
        hh[j] = (_Float16)v;
        ll[j] = (_Float16)(v - (float)hh[j]);
    }
    *reinterpret_cast<h8*>(out_h + e * 8) = hh;
    *reinterpret_cast<h8*>(out_l + e * 8) = ll;
}

}  // namespace bm
