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
        __syncthreads();                    // tile kt has landed (the barrier drains the copies) and buffer (kt + 1) & 1 is free
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

}  // namespace bm
