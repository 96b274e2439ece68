// CLIP-ReID (ViT-B/16) kernels for gfx950 (wave64): the second ReID backbone of the path (BASELINE.json configuration 5,
// 1280-d embeddings).
//
// Reference computation: build_transformer.forward (ViT-B-16, eval, TEST.NECK_FEAT = "after"),
// boxmot/reid/backbones/clip/make_model.py:95-139, over VisionTransformer.forward / ResidualAttentionBlock,
// boxmot/reid/backbones/clip/clip/model.py:169-211, 265-295.  Weights: the "CLP1" blob of boxmot_amd/clip_weights.py.
//
// Data layout in HBM, per pass of n crops (T = grid_h * grid_w + 1 tokens, D = width, token rows r = crop * T + t):
//   x     fp32 [n T][D]      the residual stream (LayerNorm, residual adds and the necks stay fp32, as the reference's
//                            LayerNorm subclass computes in fp32)
//   h16   fp16 [n T][D]      LayerNorm output / attention output = the activations operand of the next GEMM
//   qkv16 fp16 [n T][3 D]    q | k | v projections;   mlp16 fp16 [n T][4 D]   QuickGELU(c_fc(.))
// Every linear layer is one launch of k_gemm_f16: C^T tiles on the matrix pipe (v_mfma_f32_16x16x32_f16, fp32 accumulate)
// with the WEIGHT rows as the MFMA A operand and the token rows as B, so that a lane ends up with 4 consecutive output
// features of one token: bias / QuickGELU / residual add run in the epilogue on those and the store is 8 (fp16) or 16 (fp32)
// contiguous bytes per lane.  128 x 128 x 32 tiles, 4 waves of 64 x 64, operands staged through LDS (rows padded to 40 halves:
// conflict-free 16-byte fragment reads), the next k-tile's global loads in flight while the current one multiplies.
// Attention (129 tokens, 0.2 % of the FLOPs): one workgroup per (crop, head), both products on the matrix pipe with the softmax
// in registers between them (k_clip_attention).
#pragma once

#include <stdint.h>

#include "kernel_macros.hpp"
#include "gemm_f16.hpp"

namespace bm {


constexpr int CLIP_WAVE = 64;
constexpr float CLIP_LN_EPS = 1e-5f;

__device__ inline float clip_wave_sum(float v) {
    v += __shfl_xor(v, 32, CLIP_WAVE); v += __shfl_xor(v, 16, CLIP_WAVE); v += __shfl_xor(v, 8, CLIP_WAVE);
    v += __shfl_xor(v, 4, CLIP_WAVE); v += __shfl_xor(v, 2, CLIP_WAVE); v += __shfl_xor(v, 1, CLIP_WAVE);
    return v;
}
__device__ inline float clip_wave_max(float v) {
    for (int m = 32; m > 0; m >>= 1) { const float o = __shfl_xor(v, m, CLIP_WAVE); v = o > v ? o : v; }
    return v;
}

// ---------------------------------------------------------------------------
// Patch gather: normalised crops fp32 NHWC [n][H][W][3] -> fp16 rows [n * gh * gw][patch * patch * 3], k = (ky, kx, c):
// a patch row of `patch` pixels x 3 channels is contiguous in the crop, so every (ky) slice is one contiguous run.
// ---------------------------------------------------------------------------
__global__ void k_clip_patches(const float* __restrict__ crops, _Float16* __restrict__ out, int n, int H, int W, int patch, int gh, int gw) {
    const int run = patch * 3, K = patch * run;
    const long total = (long)n * gh * gw * K;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e % K);
        const long row = e / K;
        const int px = (int)(row % gw), py = (int)((row / gw) % gh);
        const long crop = row / ((long)gw * gh);
        const int ky = k / run, rest = k % run;
        out[e] = (_Float16)crops[((crop * H + py * patch + ky) * W + px * patch) * 3 + rest];
    }
}

// ---------------------------------------------------------------------------
// Token assembly + ln_pre (model.py:266-281): x[crop][0] = class_embedding + pos[0]; x[crop][1 + p] = patch_embed[crop][p] + pos[1 + p];
// then LayerNorm in place.  One wavefront per token row.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_clip_tokens_lnpre(const float* __restrict__ pe /*[n (T-1)][D]*/, const float* __restrict__ cls,
                                                            const float* __restrict__ pos, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ x, long rows, int T, int D) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int t = (int)(r % T);
    const long crop = r / T;
    const float* src = t == 0 ? cls : pe + (crop * (T - 1) + (t - 1)) * D;
    float s = 0.f, ss = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = src[c] + pos[(long)t * D + c]; s += v; }
    const float mean = clip_wave_sum(s) / D;
    for (int c = lane; c < D; c += 64) { const float v = src[c] + pos[(long)t * D + c] - mean; ss += v * v; }
    const float rstd = 1.0f / sqrtf(clip_wave_sum(ss) / D + CLIP_LN_EPS);
    for (int c = lane; c < D; c += 64) x[r * D + c] = (src[c] + pos[(long)t * D + c] - mean) * rstd * gamma[c] + beta[c];
}

// LayerNorm of fp32 rows -> fp16 rows (ln_1 / ln_2: the operand of the next GEMM).  One wavefront per row.
__global__ void __launch_bounds__(256) k_clip_layernorm_f16(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, _Float16* __restrict__ out, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* src = x + r * D;
    float s = 0.f, ss = 0.f;
    if (D == 768) {
        // ViT-B: the row (3 x 16 bytes per lane) is read ONCE and held in registers for the three passes (same operations in the same order as
        // the generic loop below: identical bits) -- the launch moves 304 MB at 512 crops and is bound by HBM, not by the re-reads' L2 hits alone
        cf4 v[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const cf4*>(src + lane * 4 + 256 * i);
#pragma unroll
        for (int i = 0; i < 3; ++i) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        const float mean = clip_wave_sum(s) / D;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) ss += (v[i][j] - mean) * (v[i][j] - mean);
        const float rstd = 1.0f / sqrtf(clip_wave_sum(ss) / D + CLIP_LN_EPS);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int c = lane * 4 + 256 * i;
            const cf4 gm = *reinterpret_cast<const cf4*>(gamma + c), bt = *reinterpret_cast<const cf4*>(beta + c);
            ch4 o;
            for (int j = 0; j < 4; ++j) o[j] = (_Float16)((v[i][j] - mean) * rstd * gm[j] + bt[j]);
            *reinterpret_cast<ch4*>(out + r * D + c) = o;
        }
        return;
    }
    for (int c = lane * 4; c < D; c += 256) { const cf4 v = *reinterpret_cast<const cf4*>(src + c); s += v[0] + v[1] + v[2] + v[3]; }
    const float mean = clip_wave_sum(s) / D;
    for (int c = lane * 4; c < D; c += 256) {
        const cf4 v = *reinterpret_cast<const cf4*>(src + c);
        for (int j = 0; j < 4; ++j) ss += (v[j] - mean) * (v[j] - mean);
    }
    const float rstd = 1.0f / sqrtf(clip_wave_sum(ss) / D + CLIP_LN_EPS);
    for (int c = lane * 4; c < D; c += 256) {
        const cf4 v = *reinterpret_cast<const cf4*>(src + c), gm = *reinterpret_cast<const cf4*>(gamma + c), bt = *reinterpret_cast<const cf4*>(beta + c);
        ch4 o;
        for (int j = 0; j < 4; ++j) o[j] = (_Float16)((v[j] - mean) * rstd * gm[j] + bt[j]);
        *reinterpret_cast<ch4*>(out + r * D + c) = o;
    }
}

// ---------------------------------------------------------------------------
// Multi-head attention for one (crop, head) on the matrix pipe: q, k, v fp16 [T][64] slices of qkv16, out fp16 [T][64] slice of
// h16 (nn.MultiheadAttention: q scaled by 1 / sqrt(64), softmax over the keys, no mask; model.py:200-206).
//   S^T = K . Q^T   per 16-query tile: MFMA A = K rows, B = Q rows -> a lane holds, for ITS query (lane & 15), the scores of the
//                   keys 16 kt + 4 (lane >> 4) + r: the softmax is an in-lane reduction plus two xor-shuffles (16, 32)
//   O^T = V^T . P^T MFMA A = V^T rows (from an LDS image transposed at load), B = the probabilities the lane already holds:
//                   the k-slot (g, j) of step ks is DEFINED as key 32 ks + 4 g + j (j < 4) / 32 ks + 16 + 4 g + j - 4 (j >= 4),
//                   i.e. exactly the accumulator rows of score tiles 2 ks and 2 ks + 1 -- a sum over keys does not care
//                   about their order, so P never moves between lanes; V^T is read with the same permutation (two 8-byte reads)
// fp32 scores / softmax / accumulation, fp16 operands (probabilities <= 1 after the max subtraction).  T <= 192.
// LDS: K and Q as [TP][72] halves (TP = T rounded up to 16; rows 144 bytes apart), V^T as [64][KP + 8] (KP = T rounded up to 32),
// pad rows / columns zeroed (a zero probability times an uninitialised value could still be NaN).
// ---------------------------------------------------------------------------
constexpr int ATT_DH = 64, ATT_LD = 72, ATT_MAX_T = 192;
__host__ __device__ inline int clip_attn_tp(int T) { return (T + 15) / 16 * 16; }
__host__ __device__ inline int clip_attn_kp(int T) { return (T + 31) / 32 * 32; }
__host__ __device__ inline int clip_attn_lds_bytes(int T) { return (2 * clip_attn_tp(T) * ATT_LD + ATT_DH * (clip_attn_kp(T) + 8)) * 2; }

__global__ void __launch_bounds__(256) k_clip_attention(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, int T, int D, int heads) {
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    const int TP = clip_attn_tp(T), KP = clip_attn_kp(T), VLD = KP + 8;
    _Float16* sQ = reinterpret_cast<_Float16*>(lds);
    _Float16* sK = sQ + TP * ATT_LD;
    _Float16* sVt = sK + TP * ATT_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l16 = lane & 15;
    const long crop = blockIdx.x / heads;
    const int head = blockIdx.x % heads;
    const _Float16* base = qkv + crop * T * (3L * D) + head * ATT_DH;
    const ch8 zero8 = ch8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < TP * 8; e += 256) {                   // 16-byte chunks: row t, halves 8 c .. 8 c + 7
        const int t = e >> 3, c = (e & 7) * 8;
        ch8 q = zero8, k = zero8, v = zero8;
        if (t < T) {
            const _Float16* row = base + (long)t * 3 * D + c;
            q = *reinterpret_cast<const ch8*>(row);
            k = *reinterpret_cast<const ch8*>(row + D);
            v = *reinterpret_cast<const ch8*>(row + 2 * D);
        }
        *reinterpret_cast<ch8*>(sQ + t * ATT_LD + c) = q;
        *reinterpret_cast<ch8*>(sK + t * ATT_LD + c) = k;
#pragma unroll
        for (int j = 0; j < 8; ++j) sVt[(c + j) * VLD + t] = v[j];
    }
    for (int e = tid; e < ATT_DH * (KP - TP); e += 256) {       // key columns TP .. KP - 1 of V^T
        const int d = e / (KP - TP), t = TP + e % (KP - TP);
        sVt[d * VLD + t] = (_Float16)0.f;
    }
    __syncthreads();
    const int nkt = TP / 16, nks = KP / 32;
    constexpr int MAX_KT = ATT_MAX_T / 16;
    for (int qt = wave; qt < nkt; qt += 4) {
        ch8 bq[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) bq[s] = *reinterpret_cast<const ch8*>(sQ + (16 * qt + l16) * ATT_LD + 32 * s + 8 * g);
        cf4 sc[MAX_KT];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < MAX_KT; ++kt) {
            sc[kt] = cf4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
            if (kt < nkt) {
                cf4 acc = cf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const ch8 a = *reinterpret_cast<const ch8*>(sK + (16 * kt + l16) * ATT_LD + 32 * s + 8 * g);
                    acc = BM_MFMA_F16_K32(a, bq[s], acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (16 * kt + 4 * g + r) < T ? acc[r] * 0.125f : -3.0e38f;      // head_dim ** -0.5; pad keys masked
                    sc[kt][r] = v;
                    mx = v > mx ? v : mx;
                }
            }
        }
        { float o = __shfl_xor(mx, 16, 64); mx = o > mx ? o : mx; o = __shfl_xor(mx, 32, 64); mx = o > mx ? o : mx; }
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < MAX_KT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = (16 * kt + 4 * g + r) < T ? BM_EXPF(sc[kt][r] - mx) : 0.f;
                    sc[kt][r] = e;
                    sum += e;
                }
            } else sc[kt] = cf4{0.f, 0.f, 0.f, 0.f};
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        cf4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = cf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MAX_KT / 2; ++ks) {
            if (ks < nks) {
                ch8 pb;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pb[j] = (_Float16)sc[2 * ks][j]; pb[4 + j] = (_Float16)sc[2 * ks + 1][j]; }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const _Float16* vr = sVt + (16 * dt + l16) * VLD + 32 * ks + 4 * g;
                    const ch4 lo = *reinterpret_cast<const ch4*>(vr), hi = *reinterpret_cast<const ch4*>(vr + 16);
                    const ch8 a = ch8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = BM_MFMA_F16_K32(a, pb, o[dt]);
                }
            }
        }
        const int q = 16 * qt + l16;
        if (q < T) {
            const float inv = 1.0f / sum;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                ch4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (_Float16)(o[dt][r] * inv);
                *reinterpret_cast<ch4*>(out + (crop * T + q) * D + head * ATT_DH + 16 * dt + 4 * g) = w;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The same attention for a COMPILE-TIME token count (ViT-B/16 on 256 x 128 crops: T = 129), three wavefronts per (crop, head):
//   * the 9 query tiles are dealt 3 per wave (k_clip_attention's four waves take 3 / 2 / 2 / 2); a wave runs them QB at a time -- one by
//     one by default: 112 registers, so three workgroups (nine waves) share a CU and one workgroup's load phase runs under the others'
//     arithmetic; QB = 3 (every K / V^T fragment read feeds three MFMAs) needs 240 registers and measured 16 % slower;
//   * Q never touches LDS: a lane's B fragments are two 16-byte global loads per tile;
//   * V stays ROW-MAJOR in LDS ([key][64] halves, rows 160 bytes apart) and the V^T fragments come from ds_read_b64_tr_b16 (BM_DS_READ_TR16_B64:
//     the 16 lanes of a lane group name the sixteen 8-byte chunks of a [4 keys][16 features] block, lane c receives column c = feature c
//     of the four keys) -- no element-wise transposed store pass (8 ds_write_b16 per 16 bytes of V in k_clip_attention);
//   * no run-time tile guards: the loops are over the compile-time tile counts, only the last key tile is masked.
// Arithmetic, operand rounding and every summation order are those of k_clip_attention: the two kernels return identical bits
// (tools/attn_prof.hip compares them; tests/test_gpu_clipreid.py runs the network through this one).
// LDS: K [TP][72] halves + V [KP][80] halves = 20.7 + 25.6 KB at T = 129: three workgroups per CU.
// ---------------------------------------------------------------------------
#ifndef BM_DS_READ_TR16_B64
typedef short cs4 __attribute__((ext_vector_type(4)));
#define BM_DS_READ_TR16_B64(lds_ptr) \
    __builtin_bit_cast(ch4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) cs4*)(lds_ptr)))
#endif
#ifndef BM_CLIP_ATTN_QB
#define BM_CLIP_ATTN_QB 1          // query tiles a wave runs together: 1 = three passes at 112 registers, three workgroups per CU (0.112 ms per layer at 512 crops);
                                   // 3 = every K / V fragment read feeds three MFMAs but 240 registers, two workgroups per CU (0.134 ms) -- profiles/r6_clip_ab.txt
#endif
constexpr int ATT_VLD = 80;             // halves per V row in LDS: 8 consecutive keys land on disjoint 8-bank windows (40 dwords apart)
template <int T>
__host__ __device__ constexpr int clip_attn_t_lds_bytes() { return (((T + 15) / 16 * 16) * ATT_LD + ((T + 31) / 32 * 32) * ATT_VLD) * 2; }

template <int T, int QB = BM_CLIP_ATTN_QB>
__global__ void __launch_bounds__(192) k_clip_attention_t(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, int D, int heads) {
    constexpr int TP = (T + 15) / 16 * 16, NKT = TP / 16, KP = (T + 31) / 32 * 32, NKS = KP / 32, NQ = NKT / 3;
    static_assert(NKT % 3 == 0 && T <= ATT_MAX_T && NQ % QB == 0, "three waves share the query tiles evenly, QB of them per pass");
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    _Float16* sK = reinterpret_cast<_Float16*>(lds);
    _Float16* sV = sK + TP * ATT_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = BM_UNIFORM_I32(tid >> 6), g = lane >> 4, l16 = lane & 15;
    const long crop = blockIdx.x / heads;
    const int head = blockIdx.x % heads;
    const _Float16* base = qkv + crop * T * (3L * D) + head * ATT_DH;
    const ch8 zero8 = ch8{0, 0, 0, 0, 0, 0, 0, 0};
    // this wave's Q fragments (rows beyond T re-read row T - 1: their results are never stored)
    ch8 bq[NQ][2];
#pragma unroll
    for (int a = 0; a < NQ; ++a) {
        int q = 16 * (wave + 3 * a) + l16;
        q = q < T ? q : T - 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) bq[a][s] = *reinterpret_cast<const ch8*>(base + (long)q * 3 * D + 32 * s + 8 * g);
    }
    for (int e = tid; e < KP * 8; e += 192) {                   // 16-byte chunks: row t, halves 8 c .. 8 c + 7; pad rows zero
        const int t = e >> 3, c = (e & 7) * 8;
        ch8 k = zero8, v = zero8;
        if (t < T) {
            const _Float16* row = base + (long)t * 3 * D + c;
            k = *reinterpret_cast<const ch8*>(row + D);
            v = *reinterpret_cast<const ch8*>(row + 2 * D);
        }
        if (t < TP) *reinterpret_cast<ch8*>(sK + t * ATT_LD + c) = k;
        *reinterpret_cast<ch8*>(sV + t * ATT_VLD + c) = v;
    }
    __syncthreads();
#pragma unroll
    for (int a0 = 0; a0 < NQ; a0 += QB) {
    // S^T = K . Q^T for the wave's three query tiles
    cf4 sc[QB][NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        ch8 ak[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) ak[s] = *reinterpret_cast<const ch8*>(sK + (16 * kt + l16) * ATT_LD + 32 * s + 8 * g);
#pragma unroll
        for (int a = 0; a < QB; ++a) {
            cf4 acc = cf4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) acc = BM_MFMA_F16_K32(ak[s], bq[a0 + a][s], acc);
            sc[a][kt] = acc;
        }
    }
    float inv[QB];
#pragma unroll
    for (int a = 0; a < QB; ++a) {
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = 16 * kt + 12 + 3 < T || (16 * kt + 4 * g + r) < T;          // (compile-time true except in the last key tile)
                const float v = valid ? sc[a][kt][r] * 0.125f : -3.0e38f;                       // head_dim ** -0.5; pad keys masked
                sc[a][kt][r] = v;
                mx = v > mx ? v : mx;
            }
        { float o = __shfl_xor(mx, 16, 64); mx = o > mx ? o : mx; o = __shfl_xor(mx, 32, 64); mx = o > mx ? o : mx; }
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = 16 * kt + 12 + 3 < T || (16 * kt + 4 * g + r) < T;
                const float e = valid ? BM_EXPF(sc[a][kt][r] - mx) : 0.f;
                sc[a][kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        inv[a] = 1.0f / sum;
    }
    // O^T = V^T . P^T: k-slot (g, j) of step ks is key 32 ks + 4 g + j (j < 4) / 32 ks + 16 + 4 g + j - 4 (j >= 4), as in k_clip_attention
    cf4 o[QB][4];
#pragma unroll
    for (int a = 0; a < QB; ++a)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[a][dt] = cf4{0.f, 0.f, 0.f, 0.f};
    // the lane's chunk of a [4 keys][16 features] block: key l16 / 4, features 4 (l16 % 4) ..
    const _Float16* vchunk = sV + (4 * g + (l16 >> 2)) * ATT_VLD + 4 * (l16 & 3);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        ch8 pb[QB];
#pragma unroll
        for (int a = 0; a < QB; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pb[a][j] = (_Float16)sc[a][2 * ks][j];
                pb[a][4 + j] = 2 * ks + 1 < NKT ? (_Float16)sc[a][(2 * ks + 1) % NKT][j] : (_Float16)0.f;
            }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const ch4 lo = BM_DS_READ_TR16_B64(vchunk + (32 * ks) * ATT_VLD + 16 * dt);
            const ch4 hi = BM_DS_READ_TR16_B64(vchunk + (32 * ks + 16) * ATT_VLD + 16 * dt);
            const ch8 av = ch8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int a = 0; a < QB; ++a) o[a][dt] = BM_MFMA_F16_K32(av, pb[a], o[a][dt]);
        }
    }
#pragma unroll
    for (int a = 0; a < QB; ++a) {
        const int q = 16 * (wave + 3 * (a0 + a)) + l16;
        if (q < T) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                ch4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (_Float16)(o[a][dt][r] * inv[a]);
                *reinterpret_cast<ch4*>(out + (crop * T + q) * D + head * ATT_DH + 16 * dt + 4 * g) = w;
            }
        }
    }
    }
}

// ---------------------------------------------------------------------------
// Head: ln_post on the class token, projection, the two BatchNorm necks (folded), concat, L2 norm
// (model.py:290-295; make_model.py:119-137; base_backend.py:206).  One workgroup per crop.
//   out[row][0 .. D) = bn(ln_post(x[crop][0]));  out[row][D .. D + E) = bn_proj(ln_post(x[crop][0]) @ proj)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_clip_head(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ proj /*[D][E]*/, const float* __restrict__ bn_scale,
                                                   const float* __restrict__ bn_shift, const float* __restrict__ bnp_scale,
                                                   const float* __restrict__ bnp_shift, float* __restrict__ out_base,
                                                   const int* __restrict__ out_rows, int T, int D, int E) {
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    float* v = reinterpret_cast<float*>(lds);            // [D] ln_post(cls)
    float* red = v + D;                                   // [8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const long crop = blockIdx.x;
    const float* src = x + crop * T * (long)D;
    float s = 0.f;
    for (int c = tid; c < D; c += blockDim.x) s += src[c];
    s = clip_wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float mean = 0.f;
    for (int w = 0; w < nw; ++w) mean += red[w];
    mean /= D;
    __syncthreads();
    float ss = 0.f;
    for (int c = tid; c < D; c += blockDim.x) ss += (src[c] - mean) * (src[c] - mean);
    ss = clip_wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    float var = 0.f;
    for (int w = 0; w < nw; ++w) var += red[w];
    const float rstd = 1.0f / sqrtf(var / D + CLIP_LN_EPS);
    for (int c = tid; c < D; c += blockDim.x) v[c] = (src[c] - mean) * rstd * gamma[c] + beta[c];
    __syncthreads();
    float* out = out_base + (out_rows ? (long)out_rows[crop] : crop) * (D + E);
    float sq = 0.f;
    for (int c = tid; c < D; c += blockDim.x) {
        const float f = v[c] * bn_scale[c] + bn_shift[c];
        out[c] = f;
        sq += f * f;
    }
    for (int e = tid; e < E; e += blockDim.x) {
        float a = 0.f;
        for (int c = 0; c < D; ++c) a += v[c] * proj[(long)c * E + e];
        const float f = a * bnp_scale[e] + bnp_shift[e];
        out[D + e] = f;
        sq += f * f;
    }
    sq = clip_wave_sum(sq);
    __syncthreads();
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < nw; ++w) tot += red[w];
    const float inv = 1.0f / sqrtf(tot);
    __syncthreads();                        // every thread's own `out` values are visible to itself; rescale them
    for (int c = tid; c < D; c += blockDim.x) out[c] *= inv;
    for (int e = tid; e < E; e += blockDim.x) out[D + e] *= inv;
}

}  // namespace bm
