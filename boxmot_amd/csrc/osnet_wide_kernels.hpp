// Wide OSNets (osnet_x1_0: 64 / 256 / 384 / 512 channels, the backbone of BASELINE.json configuration 3) on the gfx950 matrix
// pipe.  Reference computation: OSNet.forward (eval), boxmot/reid/backbones/osnet.py:380-405, OSBlock :212-260,
// LightConv3x3 :141-155, ChannelGate :161-209; weights: the same "OSN1" blob as the other two OSNet kernel families.
//
// The x0.25 kernels (reid_fused.hpp) keep a whole block of one crop inside one workgroup because its 16...32-channel layers
// are too thin to fill MFMA tiles on their own; at x1.0 the middle widths are 64 / 96 / 128, every 1x1 convolution is a real
// [pixels][cin] x [cin][cout] GEMM over ALL crops of the batch, and the layer-per-launch shape below keeps the matrix pipe
// fed without per-crop workgroups:
//   * activations: fp16 NHWC [crop][pixel][channel] in HBM, fp32 accumulation everywhere, fp32 folded-BN biases
//   * every 1x1 convolution (conv1, conv3 + shortcut + ReLU, downsample, transitions, conv5) is one
//     launch of k_gemm_f16 / k_gemm_f16_glds (gemm_f16.hpp) with bias / residual / ReLU in its epilogue -- conv3 + downsample as ONE
//     launch over two operand pairs, the transitions with their 2x2 average pool in the epilogue; the 7x7 stem is an
//     implicit GEMM over an LDS-staged band of the RGBX crop with the 3x3 max pool on its accumulators (k_wide_stem)
//   * LightConv3x3 = 1x1 (linear) -> depthwise 3x3 + BN + ReLU is ONE kernel (k_light_fused): a workgroup owns a band of 8
//     image rows of one crop, runs the 1x1 of the band + a one-row halo on the matrix pipe straight from global memory
//     (weights are <= 32 KB: L1 / L2 resident) into an LDS tile, and the depthwise 3x3 reads that tile with a sliding
//     3-row window: the intermediate tensor never reaches HBM.  The last LightConv of a branch also emits the band's
//     per-channel sums, so the gate's global average pool costs no extra pass.
//   * the four gated branches are summed by one kernel (k_gate_sum4), each workgroup recomputing the four tiny gate MLPs.
// Algorithmic HBM bytes per crop (every layer input read once, every layer output written once): see DESIGN.md section 4.6.
#pragma once

#include <stdint.h>

#include "gemm_f16.hpp"
#include "reid_layout.hpp"

namespace bm {

#ifndef BM_WIDE_BAND
#define BM_WIDE_BAND 8
#endif
constexpr int WIDE_BAND = BM_WIDE_BAND;       // image rows per k_light_fused workgroup (8 or 4: must divide the smallest image height, 16)

// widths this kernel family takes: GEMM k-steps of 32, 16-channel MFMA tiles, k_light_fused's thread mapping
inline bool wide_osnet_supports(const OsnetLayout& L) {
    if ((L.c[0] != 32 && L.c[0] != 64) || L.feat > 512 || L.c[3] > 512) return false;
    for (int b = 0; b < 6; ++b) {
        const BlockW& B = L.block[b];
        if (B.cin % 32 || B.cout % 32 || B.mid % 32 || B.mid > 128) return false;
        if ((32 >> (b / 2)) * (B.mid / 8) > 256) return false;      // k_light_fused: one thread per (column, 8 channels) of an image row
    }
    return true;
}

// ---------------------------------------------------------------------------
// Stem: conv 7x7, stride 2, pad 3 (3 -> C0) + folded BN + ReLU + max pool 3x3, stride 2, pad 1 (osnet.py:294-295) in one launch: an
// implicit GEMM on the matrix pipe with the pool on the accumulators (the 128 x 64 x C0 convolution output never reaches HBM).
//   crops  fp16 RGBX with a 3-pixel zero border, [n][262][136][4] (k_crop_resize_rgbx): 8 input pixels x RGBX = 32 halves = one
//          MFMA k-step per kernel row, and the B fragment of conv pixel cx, lane group g is the 16 bytes at pixel 2 cx + 2 g
//   wts    fp16 A fragments [ky][channel tile][lane][8]: lane (co = lane & 15, g), k-slot j -> tap kx = 2 g + (j >> 2), channel
//          j & 3 (zero for kx = 7 and the X channel), packed by wide_pack_w16
//   out    fp16 NHWC [n][64 * 32][C0]
// grid (16 bands of 4 pooled rows, crops), 256 threads.  The 23 input rows of a band (its 9 conv rows: 8 + the one above) are
// staged once into LDS (16-byte copies); wave w owns channel tile w % (C0 / 16) -- its seven A fragments stay in registers -- and
// a run of consecutive pooled rows, sliding over the conv rows: 7 x (ds_read_b128 + MFMA) per 16 conv pixels, vertical max in
// registers, horizontal max by DPP row shifts (conv values are >= 0 after the ReLU, so a missing neighbour reads as 0).
// ---------------------------------------------------------------------------
constexpr int WSTEM_ROWS = 262, WSTEM_COLS = 136, WSTEM_PBAND = 4;
constexpr int WSTEM_IN_ROWS = 4 * WSTEM_PBAND + 7;                   // conv rows 8 b - 1 .. 8 b + 7 read input rows 16 b - 2 .. 16 b + 20

template <int C0>
__global__ void __launch_bounds__(256) k_wide_stem(const _Float16* __restrict__ crops, const _Float16* __restrict__ wts,
                                                   const float* __restrict__ bias, _Float16* __restrict__ out) {
    static_assert(C0 == 32 || C0 == 64, "stem width");
    constexpr int NCT = C0 / 16, NRG = 4 / NCT, PR = WSTEM_PBAND / NRG;          // channel tiles, row groups, pooled rows per wave
    __shared__ __attribute__((aligned(16))) _Float16 sIn[WSTEM_IN_ROWS * WSTEM_COLS * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const int band = blockIdx.x;
    const long crop = blockIdx.y;
    const int in0 = 16 * band - 2;                                    // first staged (padded) input row; negative rows do not exist
    const _Float16* img = crops + crop * WSTEM_ROWS * (long)(WSTEM_COLS * 4);
    for (int e = tid; e < WSTEM_IN_ROWS * WSTEM_COLS / 2; e += 256) {            // 16-byte chunks = 2 pixels
        const int r = e / (WSTEM_COLS / 2), row = in0 + r;
        cu4 v = cu4{0u, 0u, 0u, 0u};
        if (row >= 0 && row < WSTEM_ROWS) v = *reinterpret_cast<const cu4*>(img + (long)row * (WSTEM_COLS * 4) + (e - r * (WSTEM_COLS / 2)) * 8);
        *reinterpret_cast<cu4*>(sIn + e * 8) = v;
    }
    const int ct = wave % NCT, rg = wave / NCT;
    ch8 a[7];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) a[ky] = *reinterpret_cast<const ch8*>(wts + ((long)(ky * NCT + ct) * 64 + lane) * 8);
    cf4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * ct + 4 * g + r];
    __syncthreads();
    // conv row cy (global index) for the 4 tiles of 16 conv pixels; rows outside the image contribute 0 to the pool
    auto conv_row = [&](int cy, cf4 (&row)[4]) {
        if (cy < 0 || cy > 127) {
#pragma unroll
            for (int t = 0; t < 4; ++t) row[t] = cf4{0.f, 0.f, 0.f, 0.f};
            return;
        }
        const int lr = 2 * cy - in0;                                  // staged row of kernel row 0
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int cx = 16 * t + l16;
            cf4 acc = bv;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const ch8 b = *reinterpret_cast<const ch8*>(sIn + ((lr + ky) * WSTEM_COLS + 2 * cx + 2 * g) * 4);
                acc = BM_MFMA_F16_K32(a[ky], b, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) row[t][r] = acc[r] > 0.f ? acc[r] : 0.f;
        }
    };
    const int py0 = WSTEM_PBAND * band + rg * PR;
    cf4 prev[4], mid[4], next[4];
    conv_row(2 * py0 - 1, prev);
#pragma unroll 1
    for (int py = py0; py < py0 + PR; ++py) {
        conv_row(2 * py, mid);
        conv_row(2 * py + 1, next);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ch4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = prev[t][r] > mid[t][r] ? prev[t][r] : mid[t][r];
                v = v > next[t][r] ? v : next[t][r];
                float vp = 0.f;                                        // the same vertical max in the previous tile (for lane 0's left neighbour)
                if (t > 0) {
                    vp = prev[t - 1][r] > mid[t - 1][r] ? prev[t - 1][r] : mid[t - 1][r];
                    vp = vp > next[t - 1][r] ? vp : next[t - 1][r];
                }
                const float right = BM_ROW_SHL1_F32(v);
                float left = BM_ROW_SHR1_F32(v);
                const float left_prev_tile = BM_ROW_ROR1_F32(vp);
                if (l16 == 0) left = left_prev_tile;
                float m = v > right ? v : right;
                m = m > left ? m : left;
                o[r] = (_Float16)m;
            }
            if ((l16 & 1) == 0) {                                      // pooled pixel ox = cx / 2 sits on the even lanes
                const long p = (long)py * 32 + t * 8 + (l16 >> 1);
                *reinterpret_cast<ch4*>(out + (crop * 2048 + p) * C0 + 16 * ct + 4 * g) = o;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) prev[t] = next[t];
    }
}

// ---------------------------------------------------------------------------
// LightConv3x3 (osnet.py:141-155) = 1x1 (linear) -> depthwise 3x3 + folded BN + ReLU, fused: a workgroup owns a band of 8 image rows
// of one crop.  Two building blocks:
//   light_pw  the 1x1 of a run of 16-pixel tiles on the matrix pipe: B fragments from a caller-supplied source (global image rows
//             or an LDS tile), A = the [C][C] weights straight from global (<= 32 KB: L1 / L2 resident), result into an LDS tile T
//             (pixel stride C + 8 halves: 16-byte aligned rows, the 8-byte accumulator writes at most 2-way conflicted).  Rows
//             outside the image are ZERO: the 1x1 has no bias, so the zero padding of the depthwise input is exactly a zero row.
//   light_dw  depthwise 3x3 + bias + ReLU from T: thread = (column, 8-channel group), sliding 3 x 3 window down its rows, fp32.
// k_light_fused (256 threads) = pw -> dw for one LightConv.  (Two chained LightConvs in one launch -- pw -> dw -> pw -> dw with a
// two-row halo, the intermediate tensor only in LDS, half the HBM traffic -- was built from the same two blocks and measured
// 1.5x slower than two launches: 101 KB of LDS leave one workgroup per CU, whose four phases overlap with nothing;
// profiles/r2_c3_fusion_ab.txt.)
//   in / out  fp16 NHWC [n][H][W][C];  pw fp16 [C][C] (out-major, as stored);  dw fp32 [C][9] with BN folded;  bias fp32 [C]
//   gap_part  nullptr, or fp32 [n][bands][C]: sum over the band's pixels of the output (for the channel gate's average pool)
// ---------------------------------------------------------------------------
template <int C>
__host__ __device__ constexpr int light_lds_bytes(int W) { return (WIDE_BAND + 2) * W * (C + 8) * 2; }

template <int C, class LoadB>
__device__ inline void light_pw(int n_ptiles, int wave, int nwaves, int lane, const _Float16* __restrict__ pw, _Float16* T, LoadB loadb) {
    constexpr int LD = C + 8, KS = C / 32, CT = C / 16;
    const int g = lane >> 4, l16 = lane & 15;
    // (requesting all of a wave's B fragments before the first MFMA was measured: no gain at C = 64, 12-19 % slower at C = 96 / 128)
    for (int pt = wave; pt < n_ptiles; pt += nwaves) {
        const int px = pt * 16 + l16;
        ch8 b[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) b[s] = loadb(px, s);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            cf4 acc = cf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const ch8 a = *reinterpret_cast<const ch8*>(pw + (long)(16 * ct + l16) * C + 32 * s + 8 * g);
                acc = BM_MFMA_F16_K32(a, b[s], acc);
            }
            ch4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (_Float16)acc[r];
            *reinterpret_cast<ch4*>(T + px * LD + 16 * ct + 4 * g) = o;
        }
    }
}

// output rows lr0 .. lr1 - 1 (tile rows lr .. lr + 2 each) of column x, channels 8 cg .. 8 cg + 7; store(lr, o) takes each result
template <int C, class Store>
__device__ inline void light_dw(const _Float16* T, int W, int x, int cg, int lr0, int lr1, const float* __restrict__ dw,
                                const float* __restrict__ bias, Store store) {
    constexpr int LD = C + 8;
    if (lr0 >= lr1) return;
    float w[9][8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bv[j] = bias[cg * 8 + j];
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k][j] = dw[(cg * 8 + j) * 9 + k];
    }
    const ch8 zero = ch8{0, 0, 0, 0, 0, 0, 0, 0};
    auto load_row = [&](int trow, ch8& l, ch8& m, ch8& r) {
        const _Float16* p = T + (trow * W + x) * LD + cg * 8;
        m = *reinterpret_cast<const ch8*>(p);
        l = x > 0 ? *reinterpret_cast<const ch8*>(p - LD) : zero;
        r = x < W - 1 ? *reinterpret_cast<const ch8*>(p + LD) : zero;
    };
    ch8 t0, t1, t2, m0, m1, m2, b0, b1, b2;
    load_row(lr0, t0, t1, t2);
    load_row(lr0 + 1, m0, m1, m2);
    for (int lr = lr0; lr < lr1; ++lr) {
        load_row(lr + 2, b0, b1, b2);
        ch8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = (float)t0[j] * w[0][j];
            a += (float)t1[j] * w[1][j]; a += (float)t2[j] * w[2][j];
            a += (float)m0[j] * w[3][j]; a += (float)m1[j] * w[4][j]; a += (float)m2[j] * w[5][j];
            a += (float)b0[j] * w[6][j]; a += (float)b1[j] * w[7][j]; a += (float)b2[j] * w[8][j];
            a += bv[j];
            a = a > 0.f ? a : 0.f;
            o[j] = (_Float16)a;
        }
        store(lr, o);
        t0 = m0; t1 = m1; t2 = m2; m0 = b0; m1 = b1; m2 = b2;
    }
}

// thread -> (row group, column, channel group) of the depthwise phase: NG row groups of TPR = W * C / 8 threads
struct LightMap { int x, cg, grp, ng; bool active; };
template <int C>
__device__ inline LightMap light_map(int tid, int nthreads, int W, int max_groups) {
    constexpr int CG = C / 8;
    const int TPR = W * CG;
    int ng = 1;
    while (ng * 2 <= max_groups && ng * 2 * TPR <= nthreads) ng *= 2;
    LightMap m;
    m.ng = ng;
    m.active = tid < ng * TPR && TPR <= nthreads;
    m.grp = m.active ? tid / TPR : 0;
    const int rem = tid - m.grp * TPR;
    m.x = m.active ? rem / CG : 0;
    m.cg = m.active ? rem - m.x * CG : 0;
    return m;
}

// band sums of the output for the channel gate: per-thread sums -> LDS -> one thread per channel adds them in a fixed order
template <int C>
__device__ inline void light_gap(const float (&gsum)[8], const LightMap& m, int W, float* S, float* __restrict__ dst) {
    __syncthreads();
    if (m.active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) S[(m.grp * W + m.x) * C + m.cg * 8 + j] = gsum[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float s = 0.f;
        for (int k = 0; k < m.ng * W; ++k) s += S[k * C + threadIdx.x];
        dst[threadIdx.x] = s;
    }
}

template <int C>
__global__ void __launch_bounds__(256) k_light_fused(const _Float16* __restrict__ in, const _Float16* __restrict__ pw,
                                                     const float* __restrict__ dw, const float* __restrict__ bias,
                                                     _Float16* __restrict__ out, float* __restrict__ gap_part, int H, int W) {
    static_assert(C % 32 == 0 && C <= 128, "middle width");
    BM_DYNAMIC_LDS_T(unsigned char, lds_raw);
    _Float16* T = reinterpret_cast<_Float16*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    const int band = blockIdx.x, nbands = gridDim.x;
    const long crop = blockIdx.y;
    const int r0 = band * WIDE_BAND;
    const _Float16* img = in + crop * H * W * (long)C;
    light_pw<C>((WIDE_BAND + 2) * W / 16, wave, 4, lane, pw, T, [&](int px, int s) {
        const int lrow = px / W, col = px - lrow * W, grow = r0 - 1 + lrow;
        ch8 b = ch8{0, 0, 0, 0, 0, 0, 0, 0};
        if (grow >= 0 && grow < H) b = *reinterpret_cast<const ch8*>(img + ((long)grow * W + col) * C + 32 * s + 8 * g);
        return b;
    });
    __syncthreads();
    const LightMap m = light_map<C>(tid, 256, W, 4);        // <= 4 row groups: the band sums below reuse the tile's LDS
    const int rpg = WIDE_BAND / m.ng;
    float gsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gsum[j] = 0.f;
    if (m.active)
        light_dw<C>(T, W, m.x, m.cg, m.grp * rpg, (m.grp + 1) * rpg, dw, bias, [&](int lr, const ch8& o) {
            *reinterpret_cast<ch8*>(out + ((crop * H + r0 + lr) * W + m.x) * (long)C + m.cg * 8) = o;
#pragma unroll
            for (int j = 0; j < 8; ++j) gsum[j] += (float)o[j];     // the gate pools the tensor the next layer sees (fp16-rounded)
        });
    if (gap_part) light_gap<C>(gsum, m, W, reinterpret_cast<float*>(lds_raw), gap_part + (crop * nbands + band) * C);
}

// ---------------------------------------------------------------------------
// Unified aggregation gate over the four branches (osnet.py:247-258, ChannelGate :161-209):
//   out[p][c] = sum_b x_b[p][c] * sigmoid(fc2(relu(fc1(mean_p x_b))))[c]
// grid (crops, pixel blocks), 256 threads.  gap_part fp32 [4][n][bands][C] from k_light_fused.
// ---------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_gate_sum4(const _Float16* __restrict__ xa, const _Float16* __restrict__ xb,
                                                   const _Float16* __restrict__ xc, const _Float16* __restrict__ xd,
                                                   const float* __restrict__ gap_part, const float* __restrict__ fc1_w,
                                                   const float* __restrict__ fc1_b, const float* __restrict__ fc2_w,
                                                   const float* __restrict__ fc2_b, _Float16* __restrict__ out, int P, int nbands,
                                                   long n_crops, int pix_per_block) {
    constexpr int HID = C / 16, CG = C / 8;
    __shared__ float s_mean[4][C];
    __shared__ float s_h[4][HID];
    __shared__ __attribute__((aligned(16))) float s_g[4][C];
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
        s_g[b][c] = 1.f / (1.f + BM_EXPF(-v));
    }
    __syncthreads();
    const long p0 = (long)blockIdx.y * pix_per_block;
    const int items = pix_per_block * CG;
    for (int e = tid; e < items; e += 256) {
        const int pl = e / CG, cg = e - pl * CG;
        if (p0 + pl >= P) break;
        const long off = ((n * P + p0 + pl) * C) + cg * 8;
        const ch8 a = *reinterpret_cast<const ch8*>(xa + off), b = *reinterpret_cast<const ch8*>(xb + off);
        const ch8 c = *reinterpret_cast<const ch8*>(xc + off), d = *reinterpret_cast<const ch8*>(xd + off);
        ch8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = cg * 8 + j;
            float v = (float)a[j] * s_g[0][ch];
            v += (float)b[j] * s_g[1][ch];
            v += (float)c[j] * s_g[2][ch];
            v += (float)d[j] * s_g[3][ch];
            o[j] = (_Float16)v;
        }
        *reinterpret_cast<ch8*>(out + off) = o;
    }
}

// head (osnet.py:393-396 + base_backend.py:206): global average pool -> Linear + folded BatchNorm1d -> ReLU -> L2, batched over
// the crops of the pass: k_wide_gap (fp16 [n][C]), the FC as one k_gemm_f16_glds launch over all crops (fp32 out, ReLU in the
// epilogue), k_wide_l2 (row norm + scatter to the caller's rows).  A per-crop head re-read the 1 MB FC matrix once per crop.
__global__ void __launch_bounds__(256) k_wide_gap(const _Float16* __restrict__ in, _Float16* __restrict__ out, int P, int C, long total8) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (crop, 8-channel group)
    if (e >= total8) return;
    const int C8 = C / 8;
    const long n = e / C8;
    const int cg = (int)(e - n * C8);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    const _Float16* p = in + n * P * (long)C + cg * 8;
    for (int k = 0; k < P; ++k) {
        const ch8 v = *reinterpret_cast<const ch8*>(p + (long)k * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
    ch8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (_Float16)(s[j] / (float)P);
    *reinterpret_cast<ch8*>(out + e * 8) = o;
}

// one wavefront per crop: out row (out_rows ? out_rows[n] : n) = v / |v|
__global__ void __launch_bounds__(256) k_wide_l2(const float* __restrict__ v, float* __restrict__ out_base, const int* __restrict__ out_rows,
                                                 long n_rows, int F) {
    const int lane = threadIdx.x & 63;
    const long n = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= n_rows) return;
    const float* src = v + n * F;
    float sq = 0.f;
    for (int f = lane; f < F; f += 64) sq += src[f] * src[f];
    for (int m = 32; m > 0; m >>= 1) sq += __shfl_xor(sq, m, 64);
    const float nrm = sqrtf(sq);
    float* out = out_base + (out_rows ? (long)out_rows[n] : n) * F;
    for (int f = lane; f < F; f += 64) out[f] = src[f] / nrm;
}

}  // namespace bm
