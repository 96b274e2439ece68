// fp16 GEMM on the gfx950 matrix pipe, shared by the two GEMM-shaped ReID backbones: the linear layers of CLIP-ReID
// (clip_kernels.hpp) and the 1x1 convolutions of the wide OSNets (osnet_wide.hpp; NHWC activations make a 1x1 convolution a
// plain [pixels][cin] x [cin][cout] product).
//
//   C[m][n] = sum_k X[m][k] * Wt[n][k]  (+ bias[n])      fp16 operands, fp32 accumulation
//   X  [M][K] fp16 row-major (token / pixel rows), Wt [N][K] fp16 row-major (nn.Linear / conv weight layout as stored)
//   N % BN == 0 with BN in {32, 64, 96, 128}; K % 32 == 0; any M.
//
// C^T tiles on v_mfma_f32_16x16x32_f16 with the WEIGHT rows as the MFMA A operand and the activation rows as B, so that a lane
// ends up with 4 consecutive output features of one row: bias / activation / residual run in the epilogue on those and the
// store is 8 (fp16) or 16 (fp32) contiguous bytes per lane.  128 (rows) x BN (features) x 32 tiles, 4 waves as 2 x 2 of
// 64 rows x BN/2 features, operands staged through LDS (rows padded to 40 halves: conflict-free 16-byte fragment reads), the
// next k-tile's global loads in flight while the current one multiplies.
//   EPI 0: fp16 store                          EPI 1: QuickGELU, fp16 store          EPI 2: fp32 C += result
//   EPI 3: (ReLU when relu != 0) fp32 store    EPI 4: (+ fp16 residual[m][n]) (ReLU when relu != 0), fp16 store
//   EPI 5 / 6 (k_gemm_f16_glds only): ReLU + 2 x 2 average pool over an image of width 32 / 16, fp16 store of the pooled tensor
#pragma once

#include <stdint.h>

#include "kernel_macros.hpp"

namespace bm {

typedef _Float16 ch4 __attribute__((ext_vector_type(4)));
typedef _Float16 ch8 __attribute__((ext_vector_type(8)));
typedef float cf4 __attribute__((ext_vector_type(4)));
typedef unsigned int cu4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32, GEMM_LD = 40;       // LDS row = 32 halves + 8 of padding (80 bytes)

// Workgroup -> output tile.  The grid is 1-D (M tiles x N tiles workgroups).  Workgroups are dealt round-robin to the 8 XCDs
// (each with its own L2), so the linear id is first remapped such that one XCD receives a contiguous range of tile ids
// (bijective for any count), and inside that range the N tiles of one M tile are neighbours: the activation rows of an M tile
// are fetched into that XCD's L2 once and reused by all of its N tiles instead of being re-read from HBM N / BN times.
__device__ inline void gemm_tile_of_block(int m_tiles, int n_tiles, int& mt, int& nt) {
    const int total = m_tiles * n_tiles, orig = (int)blockIdx.x;
    const int q = total / 8, r = total % 8, xcd = orig % 8, idx = orig / 8;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    nt = wg % n_tiles;
    mt = wg / n_tiles;
}

template <int EPI, int BN = GEMM_BN>
__global__ void __launch_bounds__(256) k_gemm_f16(const _Float16* __restrict__ X, const _Float16* __restrict__ Wt,
                                                  const float* __restrict__ bias, void* __restrict__ Cout,
                                                  const _Float16* __restrict__ res, int M, int N, int K, int relu) {
    static_assert(BN == 32 || BN == 64 || BN == 96 || BN == 128, "feature tile");
    constexpr int NT = BN / 32;                              // 16-feature MFMA tiles per wave
    constexpr int WCH = BN * 4, WJ = (WCH + 255) / 256;      // 16-byte chunks of the weight tile, chunks per thread
    __shared__ __attribute__((aligned(16))) _Float16 sW[BN * GEMM_LD];
    __shared__ __attribute__((aligned(16))) _Float16 sX[GEMM_BM * GEMM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const int wn = wave >> 1, wm = wave & 1;                 // this wave: BN/2 output features x 64 rows
    int mt, nt;
    gemm_tile_of_block((M + GEMM_BM - 1) / GEMM_BM, N / BN, mt, nt);
    const long m0 = (long)mt * GEMM_BM;
    const int n0 = nt * BN;
    cf4 acc[NT][4];                                          // [feature tile][row tile]
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = cf4{0.f, 0.f, 0.f, 0.f};
    // staging: chunk q = (row q >> 2, k offset 8 (q & 3)); 512 chunks of the activation tile, WCH of the weight tile
    cu4 rw[WJ], rx[2];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int q = tid + 256 * j, r = q >> 2, c = (q & 3) * 8;
            if (WCH % 256 == 0 || q < WCH) rw[j] = *reinterpret_cast<const cu4*>(Wt + (long)(n0 + r) * K + k0 + c);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + 256 * j, r = q >> 2, c = (q & 3) * 8;
            const long m = m0 + r;
            rx[j] = m < M ? *reinterpret_cast<const cu4*>(X + m * K + k0 + c) : cu4{0u, 0u, 0u, 0u};
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int q = tid + 256 * j, r = q >> 2, c = (q & 3) * 8;
            if (WCH % 256 == 0 || q < WCH) *reinterpret_cast<cu4*>(sW + r * GEMM_LD + c) = rw[j];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + 256 * j, r = q >> 2, c = (q & 3) * 8;
            *reinterpret_cast<cu4*>(sX + r * GEMM_LD + c) = rx[j];
        }
    };
    // the shortcut operand of the epilogue does not depend on the product: fetch it now, its latency hides behind the k-loop
    ch4 rres[NT][4];
    if constexpr (EPI == 4) {
        if (res) {
#pragma unroll
            for (int p = 0; p < NT; ++p)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const long m = m0 + wm * 64 + t * 16 + l16;
                    rres[p][t] = m < M ? *reinterpret_cast<const ch4*>(res + m * N + n0 + wn * (BN / 2) + p * 16 + 4 * g) : ch4{0, 0, 0, 0};
                }
        }
    }
    load_tiles(0);
    for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
        __syncthreads();                    // the previous tile's fragment reads are done
        store_tiles();
        __syncthreads();
        if (k0 + GEMM_BK < K) load_tiles(k0 + GEMM_BK);
        ch8 a[NT], b[4];
#pragma unroll
        for (int t = 0; t < NT; ++t) a[t] = *reinterpret_cast<const ch8*>(sW + (wn * (BN / 2) + t * 16 + l16) * GEMM_LD + 8 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const ch8*>(sX + (wm * 64 + t * 16 + l16) * GEMM_LD + 8 * g);
#pragma unroll
        for (int p = 0; p < NT; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[p][t] = BM_MFMA_F16_K32(a[p], b[t], acc[p][t]);
    }
    // epilogue: D[row = feature 4 g + r][col = activation row l16]
#pragma unroll
    for (int p = 0; p < NT; ++p) {
        const int n = n0 + wn * (BN / 2) + p * 16 + 4 * g;
        cf4 bv = cf4{0.f, 0.f, 0.f, 0.f};
        if (bias) {                         // scalar loads: bias tensors inside a weight blob are only 4-byte aligned
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bias[n + r];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long m = m0 + wm * 64 + t * 16 + l16;
            if (m >= M) continue;
            cf4 v = acc[p][t];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
            if constexpr (EPI == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + BM_EXPF(-1.702f * v[r]));        // x * sigmoid(1.702 x), clip/model.py:181-183
            }
            if constexpr (EPI == 4) {
                if (res) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rres[p][t][r];
                }
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
            }
            if constexpr (EPI == 0 || EPI == 1 || EPI == 4) {
                ch4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
                *reinterpret_cast<ch4*>(static_cast<_Float16*>(Cout) + m * N + n) = o;
            } else if constexpr (EPI == 2) {
                float* c = static_cast<float*>(Cout) + m * N + n;
                cf4 old = *reinterpret_cast<const cf4*>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) old[r] += v[r];
                *reinterpret_cast<cf4*>(c) = old;
            } else {
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
                *reinterpret_cast<cf4*>(static_cast<float*>(Cout) + m * N + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same product for N % 128 == 0, K % BK == 0: 128 x 128 x BK tiles whose operands go HBM -> LDS by global_load_lds (no
// staging registers, no ds_write pass), two LDS buffers, ONE barrier per k-tile: the copies of tile t + 1 are issued right after
// the barrier that publishes tile t and fly while tile t multiplies.  BK = 64 (64 KiB of LDS, 32 MFMAs per wave per barrier) for
// the compute-bound shapes (CLIP-ReID's K = 768 / 3072), BK = 32 (32 KiB: twice the resident workgroups) for the short-K,
// store-bound 1x1 convolutions of the wide OSNets.
// LDS image: rows of BK halves = BK / 8 chunks of 16 bytes, chunk c of row r stored at slot c ^ swz(r) with swz(r) = r & 7
// (BK = 64) or (r >> 1) & 3 (BK = 32) -- the copy's destination is lane-linear, so the swizzle is applied to the per-lane SOURCE
// address; with ds_read_b128's lane groups ({0-3, 12-15, 20-27}, ...) every 16-byte fragment read is then bank-conflict free.
// Rows beyond M re-read row M - 1 (their results are never stored).
// ---------------------------------------------------------------------------
// optional extras of k_gemm_f16_glds:
//   X2 / W2 / K2   a second (activation, weight) pair whose product is added into the same accumulators (K2 % BK == 0): two 1x1
//                  convolutions with a common output -- an OSBlock's conv3(x2) + downsample(x) -- as ONE launch, the sum never in HBM
//   pool_w         EPI 5 (image width 32) / EPI 6 (image width 16): the epilogue applies ReLU and the 2 x 2 average pool of the
//                  transition layers (osnet.py:349) on the accumulators and stores the POOLED tensor [rows / 4][N]
struct GemmExt {
    const _Float16* X2 = nullptr;
    const _Float16* W2 = nullptr;
    int K2 = 0;
    int pool_w = 0;
};

template <int BK>
__host__ __device__ constexpr int gemm_glds_lds_bytes() { return 2 * 2 * 128 * BK * 2; }

template <int EPI, int BK>
__global__ void __launch_bounds__(256) k_gemm_f16_glds(const _Float16* __restrict__ X, const _Float16* __restrict__ Wt,
                                                       const float* __restrict__ bias, void* __restrict__ Cout,
                                                       const _Float16* __restrict__ res, int M, int N, int K, int relu, GemmExt ext) {
    static_assert(BK == 32 || BK == 64, "k-tile");
    BM_DYNAMIC_LDS_T(unsigned char, lds_raw);
    _Float16* lds = reinterpret_cast<_Float16*>(lds_raw);
    constexpr int TILE = 128 * BK, CH = BK / 8, NI = CH / 2;      // halves per operand tile, chunks per row, copies per wave and operand
    const int tid = threadIdx.x, lane = tid & 63, wave = BM_UNIFORM_I32(tid >> 6), g = lane >> 4, l16 = lane & 15;
    const int wn = wave >> 1, wm = wave & 1;
    int mt, nt;
    gemm_tile_of_block((M + GEMM_BM - 1) / GEMM_BM, N / GEMM_BN, mt, nt);
    const long m0 = (long)mt * GEMM_BM;
    const int n0 = nt * GEMM_BN;
    auto swz = [](int r) { return BK == 64 ? (r & 7) : ((r >> 1) & 3); };
    cf4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = cf4{0.f, 0.f, 0.f, 0.f};
    // copy i of this wave moves LDS chunks p = (NI wave + i) * 64 + lane of an operand tile: row p / CH, slot p % CH
    const _Float16 *gw[NI], *gx[NI], *gw2[NI], *gx2[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = (NI * wave + i) * 64 + lane, r = p / CH, c = (p % CH) ^ swz(r);
        long m = m0 + r;
        if (m >= M) m = M - 1;
        gw[i] = Wt + (long)(n0 + r) * K + 8 * c;
        gx[i] = X + m * K + 8 * c;
        gw2[i] = ext.W2 + (long)(n0 + r) * ext.K2 + 8 * c;
        gx2[i] = ext.X2 + m * ext.K2 + 8 * c;
    }
    const int nk1 = K / BK, nk = nk1 + ext.K2 / BK;
    auto issue = [&](int kt, int buf) {          // k-tile kt: one of the first operand pair's nk1 tiles, then the second pair's
        _Float16* dW = lds + buf * 2 * TILE;
        _Float16* dX = dW + TILE;
        const bool second = kt >= nk1;
        const int k0 = (second ? kt - nk1 : kt) * BK;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            BM_GLDS16((second ? gw2[i] : gw[i]) + k0, dW + (NI * wave + i) * 512, lane);
            BM_GLDS16((second ? gx2[i] : gx[i]) + k0, dX + (NI * wave + i) * 512, lane);
        }
    };
    issue(0, 0);
    // the shortcut operand of the epilogue does not depend on the product: fetch it now, its latency hides behind the k-loop
    ch4 rres[4][4];
    if constexpr (EPI == 4) {
        if (res) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    long m = m0 + wm * 64 + t * 16 + l16;
                    if (m >= M) m = M - 1;
                    rres[p][t] = *reinterpret_cast<const ch4*>(res + m * N + n0 + wn * 64 + p * 16 + 4 * g);
                }
        }
    }
    for (int kt = 0; kt < nk; ++kt) {
        BM_WAIT_VM0();                      // this wave's copies of tile kt have landed (explicit: the compiler orders a global -> LDS copy
        __syncthreads();                    // against this wave's own LDS reads, not against the barrier) ... and buffer (kt + 1) & 1 is free
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const _Float16* sW = lds + (kt & 1) * 2 * TILE;
        const _Float16* sX = sW + TILE;
        // every fragment of the tile is requested before the first MFMA: the reads of k-step 1 land under the MFMAs of k-step 0
        ch8 a[BK / 32][4], b[BK / 32][4];
#pragma unroll
        for (int s = 0; s < BK / 32; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ra = wn * 64 + t * 16 + l16, rb = wm * 64 + t * 16 + l16;
                a[s][t] = *reinterpret_cast<const ch8*>(sW + ra * BK + 8 * ((4 * s + g) ^ swz(ra)));
                b[s][t] = *reinterpret_cast<const ch8*>(sX + rb * BK + 8 * ((4 * s + g) ^ swz(rb)));
            }
#pragma unroll
        for (int s = 0; s < BK / 32; ++s)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[p][t] = BM_MFMA_F16_K32(a[s][p], b[s][t], acc[p][t]);
    }
    if constexpr (EPI == 5 || EPI == 6) {
        // transition layer: ReLU(conv + bias) then the 2 x 2 average pool, on the accumulators.  A lane holds 4 features of pixel
        // (row tile t, l16); its horizontal partner is lane l16 ^ 1 (quad swap), its vertical partner -- one image row = pool_w
        // pixels further -- is row tile t + pool_w / 16 of the same lane (a 128-row tile holds whole pairs of image rows).
        constexpr int POOL_W = EPI == 5 ? 32 : 16, dt = POOL_W / 16, sh = EPI == 5 ? 5 : 4;     // compile-time: acc[][t + dt] must stay in registers
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int n = n0 + wn * 64 + p * 16 + 4 * g;
            float bv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bias ? bias[n + r] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if ((t / dt) & 1) continue;                      // the lower image row of a pair: folded into its upper one
                const long m = m0 + wm * 64 + t * 16 + l16;
                ch4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float top = acc[p][t][r] + bv[r], bot = acc[p][(t + dt) & 3][r] + bv[r];
                    float v = (top > 0.f ? top : 0.f) + (bot > 0.f ? bot : 0.f);
                    v = v + BM_QUAD_SWAP1_F32(v);
                    o[r] = (_Float16)(v * 0.25f);
                }
                if (m < M && (l16 & 1) == 0) {
                    const long q = m >> sh, x = m & (POOL_W - 1);                    // q = crop * H + y (y even)
                    *reinterpret_cast<ch4*>(static_cast<_Float16*>(Cout) + ((q >> 1) * (POOL_W / 2) + (x >> 1)) * N + n) = o;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int n = n0 + wn * 64 + p * 16 + 4 * g;
        cf4 bv = cf4{0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bias[n + r];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long m = m0 + wm * 64 + t * 16 + l16;
            if (m >= M) continue;
            cf4 v = acc[p][t];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
            if constexpr (EPI == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + BM_EXPF(-1.702f * v[r]));
            }
            if constexpr (EPI == 4) {
                if (res) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rres[p][t][r];
                }
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
            }
            if constexpr (EPI == 0 || EPI == 1 || EPI == 4) {
                ch4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
                *reinterpret_cast<ch4*>(static_cast<_Float16*>(Cout) + m * N + n) = o;
            } else if constexpr (EPI == 2) {
                float* c = static_cast<float*>(Cout) + m * N + n;
                cf4 old = *reinterpret_cast<const cf4*>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) old[r] += v[r];
                *reinterpret_cast<cf4*>(c) = old;
            } else {
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
                *reinterpret_cast<cf4*>(static_cast<float*>(Cout) + m * N + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_gemm_f16_256: the same product on 256 (features) x 256 (rows) x 64 tiles for the large GEMMs (CLIP-ReID's linear layers:
// N a multiple of 256, tens of thousands of rows), one 8-wave workgroup per CU.
//
//   * wave (wr, wc) = (wave >> 2, wave & 3) owns 128 features x 64 rows: 8 x 4 accumulator tiles (128 registers), computed as four
//     quadrants of 4 x 2 tiles x 2 k-steps = 16 MFMAs -- one quadrant per PHASE, four phases per k-tile:
//         P1  read A-lo (8 fragments) + B-0 (4)   multiply (A-lo, B-0)
//         P2  read A-hi (8)                       multiply (A-hi, B-0)
//         P3  read B-1 (4)                        multiply (A-hi, B-1)
//         P4  --                                  multiply (A-lo, B-1)
//     128 + 96 registers: a wave reads (128 + 64) x 64 halves per 1 MFLOP instead of (64 + 64) x 64 per 0.5 (k_gemm_f16_glds).
//   * the two wave groups (wr = 0 / 1; one wave of each per SIMD) run HALF A PHASE APART: a phase is  [reads, copies, waits]
//     barrier [16 MFMAs] barrier, group 1 passes one extra barrier at the start, so one group's fragment reads and address work
//     run under the other group's MFMAs on the same SIMD.
//   * operands HBM -> LDS by global_load_lds in HALF tiles (128 rows x 64 halves = 16 KB = two copies per thread), the LDS image of
//     k_gemm_f16_glds (chunk c of row r at slot c ^ (r & 7), swizzle on the source address), two buffers of four half tiles
//     (128 KB).  The copies of k-tile T are issued 3-5 phases ahead -- A-0 in P3 and A-1 + B-0 in P4 of tile T - 2, B-1 in P1 of
//     tile T - 1 -- each into a slot whose last fragment read finished (lgkmcnt(0) before the reading phase's first barrier) a
//     phase earlier, and they are waited for ONCE per k-tile with a counted `s_waitcnt vmcnt(6)` in P4 of tile T - 1 (the six
//     younger copies stay in flight), one phase before the first read.  Barriers are bare s_barrier: no vmcnt(0) drain anywhere
//     in the loop.
// N % 256 == 0, K % 64 == 0, any M (rows beyond M are clamped on load, skipped on store).  EPI 0 / 1 / 2 / 3 / 4 as above.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int GEMM256_EPI_LD = 136;                        // halves per row of a wave's epilogue region (128 + 8 of padding)
constexpr int GEMM256_LDS_BYTES = 8 * 64 * GEMM256_EPI_LD * 2;       // 136 KB: the epilogue regions; the k-loop uses 2 x 4 half tiles = 128 KB of it

#ifndef BM_GEMM_EPI2_DEEP
#define BM_GEMM_EPI2_DEEP 1
#endif
#ifndef BM_GEMM_PHASE_SYNC
#define BM_GEMM_PHASE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define BM_GEMM_BARRIER() asm volatile("s_barrier" ::: "memory")
#define BM_GEMM_WAIT_COPIES_6() asm volatile("s_waitcnt vmcnt(6)" ::: "memory")
#define BM_GEMM_WAIT_COPIES_0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

template <int EPI>
__global__ void __launch_bounds__(512) k_gemm_f16_256(const _Float16* __restrict__ X, const _Float16* __restrict__ Wt,
                                                      const float* __restrict__ bias, void* __restrict__ Cout,
                                                      const _Float16* __restrict__ res, int M, int N, int K, int relu) {
    BM_DYNAMIC_LDS_T(unsigned char, lds_raw);
    _Float16* lds = reinterpret_cast<_Float16*>(lds_raw);
    constexpr int HALF = 128 * 64;                          // halves per half tile
    const int tid = threadIdx.x, lane = tid & 63, wave = BM_UNIFORM_I32(tid >> 6), g = lane >> 4, l16 = lane & 15;
    const int wr = wave >> 2, wc = wave & 3;
    int mt, nt;
    gemm_tile_of_block((M + 255) / 256, N / 256, mt, nt);
    const long m0 = (long)mt * 256;
    const int n0 = nt * 256;
    // copy j of this wave moves LDS chunks p = (2 wave + j) * 64 + lane of a half tile: row p / 8, slot p % 8
    const _Float16 *gA[2], *gB[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = (2 * wave + j) * 64 + lane, r = p >> 3, c = (p & 7) ^ (r & 7);
        gA[j] = Wt + (long)(n0 + r) * K + 8 * c;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            long m = m0 + h * 128 + r;
            if (m >= M) m = M - 1;
            gB[h][j] = X + m * K + 8 * c;
        }
    }
    const long a_half = 128L * K;
    // slot s of buffer b: A-0, A-1, B-0, B-1
    auto slot = [&](int b, int s) { return lds + (b * 4 + s) * HALF; };
    auto copy_a = [&](int T, int h) {
        _Float16* d = slot(T & 1, h);
#pragma unroll
        for (int j = 0; j < 2; ++j) BM_GLDS16(gA[j] + h * a_half + (long)T * 64, d + (2 * wave + j) * 512, lane);
    };
    auto copy_b = [&](int T, int h) {
        _Float16* d = slot(T & 1, 2 + h);
#pragma unroll
        for (int j = 0; j < 2; ++j) BM_GLDS16(gB[h][j] + (long)T * 64, d + (2 * wave + j) * 512, lane);
    };
    cf4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = cf4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / 64;
    // prologue: tile 0 whole, of tile 1 what the steady state has issued by the end of a P4
    copy_a(0, 0); copy_a(0, 1); copy_b(0, 0); copy_b(0, 1);
    if (nk > 1) { copy_a(1, 0); copy_a(1, 1); copy_b(1, 0); BM_GEMM_WAIT_COPIES_6(); }
    else BM_GEMM_WAIT_COPIES_0();
    BM_GEMM_BARRIER();
    if (wr == 1) BM_GEMM_BARRIER();                         // group 1 runs half a phase behind group 0
    ch8 alo[2][4], ahi[2][4], b0[2][2], b1[2][2];
    const int brow = (wc & 1) * 64;
    auto read_a = [&](const _Float16* sA, int p0, ch8 (&f)[2][4]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ra = (p0 + t) * 16 + l16;
                f[s][t] = *reinterpret_cast<const ch8*>(sA + ra * 64 + 8 * ((4 * s + g) ^ (ra & 7)));
            }
    };
    auto read_b = [&](const _Float16* sB, int t0, ch8 (&f)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rb = brow + (t0 + t) * 16 + l16;
                f[s][t] = *reinterpret_cast<const ch8*>(sB + rb * 64 + 8 * ((4 * s + g) ^ (rb & 7)));
            }
    };
    auto quadrant = [&](const ch8 (&fa)[2][4], int p0, const ch8 (&fb)[2][2], int t0) {
        BM_SETPRIO(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[p0 + p][t0 + t] = BM_MFMA_F16_K32(fa[s][p], fb[s][t], acc[p0 + p][t0 + t]);
        BM_SETPRIO(0);
    };
    for (int T = 0; T < nk; ++T) {
        const _Float16* sA = slot(T & 1, wr);
        const _Float16* sB = slot(T & 1, 2 + (wc >> 1));
        // P1
        read_a(sA, 0, alo);
        read_b(sB, 0, b0);
        if (T + 1 < nk) copy_b(T + 1, 1);
        BM_GEMM_PHASE_SYNC();
        quadrant(alo, 0, b0, 0);
        BM_GEMM_BARRIER();
        // P2
        read_a(sA, 4, ahi);
        BM_GEMM_PHASE_SYNC();
        quadrant(ahi, 4, b0, 0);
        BM_GEMM_BARRIER();
        // P3
        read_b(sB, 2, b1);
        if (T + 2 < nk) copy_a(T + 2, 0);
        BM_GEMM_PHASE_SYNC();
        quadrant(ahi, 4, b1, 2);
        BM_GEMM_BARRIER();
        // P4: tile T + 1 must have landed before the next phase reads it
        if (T + 2 < nk) { copy_a(T + 2, 1); copy_b(T + 2, 0); BM_GEMM_WAIT_COPIES_6(); }
        else BM_GEMM_WAIT_COPIES_0();
        BM_GEMM_PHASE_SYNC();
        quadrant(alo, 0, b1, 2);
        BM_GEMM_BARRIER();
    }
    if (wr == 0) BM_GEMM_BARRIER();                         // pairs with group 1's extra barrier
    if constexpr (EPI == 0 || EPI == 1 || EPI == 4) {
        // fp16 results leave through LDS (the operand slots are dead after the last barrier): a lane holds 4 features of 16 different
        // rows, i.e. 8-byte pieces of 16 lines per store; transposed through a private 64 x 128 region (row stride 272 bytes: 2-way
        // bank conflicts on the 8-byte writes, none on the 16-byte reads) every store instruction writes 4 rows x 256 contiguous bytes.
        _Float16* reg = lds + wave * (64 * GEMM256_EPI_LD);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int n = n0 + wr * 128 + p * 16 + 4 * g;
            cf4 bv = cf4{0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = bias[n + r];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                cf4 v = acc[p][t];
#ifdef BM_GEMM256_NO_STORE
                if (v[0] != 12345.678f) continue;            // profiling build: the main loop alone
#endif
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bv[r];
                if constexpr (EPI == 1) {    // QuickGELU; the quotient as a reciprocal (1 ulp of fp32, the result is rounded to fp16): the IEEE
#pragma unroll                           // division sequence was a quarter of this kernel's epilogue
                    for (int r = 0; r < 4; ++r) v[r] = v[r] * BM_RCPF(1.0f + BM_EXPF(-1.702f * v[r]));
                }
                if constexpr (EPI == 4) {
                    if (res) {              // (no prefetch here: the accumulators and fragments fill the register file)
                        long m = m0 + wc * 64 + t * 16 + l16;
                        if (m >= M) m = M - 1;
                        const ch4 rv = *reinterpret_cast<const ch4*>(res + m * N + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                    }
                    if (relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                    }
                }
                ch4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
                *reinterpret_cast<ch4*>(reg + (t * 16 + l16) * GEMM256_EPI_LD + p * 16 + 4 * g) = o;
            }
        }
        BM_WAVE_LDS_SYNC();
#ifndef BM_GEMM256_NO_STORE
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = i * 4 + g;
            const long m = m0 + wc * 64 + row;
            const ch8 v = *reinterpret_cast<const ch8*>(reg + row * GEMM256_EPI_LD + l16 * 8);
            if (m < M) *reinterpret_cast<ch8*>(static_cast<_Float16*>(Cout) + m * N + n0 + wr * 128 + l16 * 8) = v;
        }
#endif
        return;
    }
    if constexpr (EPI == 2) {
        // fp32 C += result, the same way in two halves of 64 features: 4 rows x 256 contiguous bytes per load / store instruction
        // instead of 16-byte pieces of 16 rows (the residual stream is read and rewritten once per projection: 2 x 101 MB at 256 crops).
        // BM_GEMM_EPI2_DEEP: the sixteen loads of a half's old values are all requested BEFORE the half's accumulators go through LDS
        // (the fragment registers are dead: 64 more fit), so a half waits for HBM once instead of twice and the wait runs under the
        // transposition; 0: two batches of eight, each requested and waited for at its use.
        float* reg = reinterpret_cast<float*>(lds_raw) + wave * (64 * 68);
        float* c = static_cast<float*>(Cout);
        constexpr int NB = BM_GEMM_EPI2_DEEP ? 16 : 8;          // loads in flight
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            cf4 old[NB];
            auto request = [&](int i0) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    long m = m0 + wc * 64 + (i0 + i) * 4 + g;
                    if (m >= M) m = M - 1;
                    old[i] = *reinterpret_cast<const cf4*>(c + m * N + n0 + wr * 128 + hp * 64 + l16 * 4);
                }
            };
            if constexpr (BM_GEMM_EPI2_DEEP) request(0);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int p = hp * 4 + pp, n = n0 + wr * 128 + p * 16 + 4 * g;
                cf4 bv = cf4{0.f, 0.f, 0.f, 0.f};
                if (bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = bias[n + r];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    cf4 v = acc[p][t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bv[r];
                    *reinterpret_cast<cf4*>(reg + (t * 16 + l16) * 68 + pp * 16 + 4 * g) = v;
                }
            }
            BM_WAVE_LDS_SYNC();
#pragma unroll
            for (int i0 = 0; i0 < 16; i0 += NB) {
                if constexpr (!BM_GEMM_EPI2_DEEP) request(i0);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int row = (i0 + i) * 4 + g;
                    const long m = m0 + wc * 64 + row;
                    const cf4 v = *reinterpret_cast<const cf4*>(reg + row * 68 + l16 * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) old[i][r] += v[r];
#ifndef BM_GEMM256_NO_STORE
                    if (m < M) *reinterpret_cast<cf4*>(c + m * N + n0 + wr * 128 + hp * 64 + l16 * 4) = old[i];
#endif
                }
            }
            BM_WAVE_LDS_SYNC();
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int n = n0 + wr * 128 + p * 16 + 4 * g;
        cf4 bv = cf4{0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = bias[n + r];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long m = m0 + wc * 64 + t * 16 + l16;
            if (m >= M) continue;
#ifdef BM_GEMM256_NO_STORE
            if (acc[p][t][0] != 12345.678f) continue;        // profiling build: the main loop alone
#endif
            cf4 v = acc[p][t];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
            if constexpr (EPI == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + BM_EXPF(-1.702f * v[r]));
            }
            if constexpr (EPI == 4) {
                if (res) {                  // (no prefetch here: the accumulators and fragments fill the register file)
                    const ch4 rv = *reinterpret_cast<const ch4*>(res + m * N + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                }
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
            }
            if constexpr (EPI == 0 || EPI == 1 || EPI == 4) {
                ch4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
                *reinterpret_cast<ch4*>(static_cast<_Float16*>(Cout) + m * N + n) = o;
            } else if constexpr (EPI == 2) {
                float* c = static_cast<float*>(Cout) + m * N + n;
                cf4 old = *reinterpret_cast<const cf4*>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) old[r] += v[r];
                *reinterpret_cast<cf4*>(c) = old;
            } else {
                if (relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
                }
                *reinterpret_cast<cf4*>(static_cast<float*>(Cout) + m * N + n) = v;
            }
        }
    }
}

}  // namespace bm
