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
// Attention is 0.2 % of the FLOPs (129 tokens): one workgroup per (crop, head) on the vector ALUs, fp32 softmax.
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
// Multi-head attention for one (crop, head): q, k, v fp16 [T][64] slices of qkv16, out fp16 [T][64] slice of h16
// (nn.MultiheadAttention: q scaled by 1 / sqrt(64), softmax over the keys, no mask; model.py:200-206).  K, V and Q rows in
// LDS (row stride 66 halves: a lane per key / per dimension reads conflict-free), one wavefront per query row: scores for
// the keys lane, lane + 64, ..., fp32 softmax by wave reductions, then lane d accumulates sum_k p[k] * V[k][d].
// ---------------------------------------------------------------------------
constexpr int ATT_DH = 64, ATT_LD = 66, ATT_MAX_T = 192;
__host__ __device__ inline int clip_attn_lds_bytes(int T, int nwaves) { return 3 * T * ATT_LD * 2 + nwaves * ATT_MAX_T * 4; }

__global__ void __launch_bounds__(256) k_clip_attention(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, int T, int D, int heads) {
    BM_DYNAMIC_LDS_T(unsigned char, lds);
    _Float16* sQ = reinterpret_cast<_Float16*>(lds);
    _Float16* sK = sQ + T * ATT_LD;
    _Float16* sV = sK + T * ATT_LD;
    float* sP = reinterpret_cast<float*>(sV + T * ATT_LD);              // [waves][ATT_MAX_T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const long crop = blockIdx.x / heads;
    const int head = blockIdx.x % heads;
    const _Float16* base = qkv + crop * T * (3L * D) + head * ATT_DH;
    for (int e = tid; e < T * (ATT_DH / 2); e += blockDim.x) {           // pairs of halves
        const int t = e / (ATT_DH / 2), d2 = (e % (ATT_DH / 2)) * 2;
        const _Float16* row = base + (long)t * 3 * D + d2;
        *reinterpret_cast<unsigned*>(sQ + t * ATT_LD + d2) = *reinterpret_cast<const unsigned*>(row);
        *reinterpret_cast<unsigned*>(sK + t * ATT_LD + d2) = *reinterpret_cast<const unsigned*>(row + D);
        *reinterpret_cast<unsigned*>(sV + t * ATT_LD + d2) = *reinterpret_cast<const unsigned*>(row + 2 * D);
    }
    __syncthreads();
    float* p = sP + wave * ATT_MAX_T;
    for (int q = wave; q < T; q += nw) {
        float sc[ATT_MAX_T / 64];
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < ATT_MAX_T / 64; ++j) {
            const int k = lane + 64 * j;
            float s = -3.0e38f;
            if (k < T) {
                s = 0.f;
                for (int d = 0; d < ATT_DH; d += 2) {                   // two halves per LDS dword: the Q read is a broadcast
                    typedef _Float16 ch2 __attribute__((ext_vector_type(2)));
                    const ch2 qv = *reinterpret_cast<const ch2*>(sQ + q * ATT_LD + d), kv = *reinterpret_cast<const ch2*>(sK + k * ATT_LD + d);
                    s += (float)qv[0] * (float)kv[0];
                    s += (float)qv[1] * (float)kv[1];
                }
                s *= 0.125f;                                              // head_dim ** -0.5
            }
            sc[j] = s;
            mx = s > mx ? s : mx;
        }
        mx = clip_wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < ATT_MAX_T / 64; ++j) {
            const int k = lane + 64 * j;
            if (k < T) {
                const float e = BM_EXPF(sc[j] - mx);
                p[k] = e;
                sum += e;
            }
        }
        sum = clip_wave_sum(sum);
        BM_WAVE_LDS_SYNC();                 // p[] written by the lanes of this wave is read by all of them below
        float o = 0.f;
        for (int k = 0; k < T; ++k) o += p[k] * (float)sV[k * ATT_LD + lane];
        out[(crop * T + q) * D + head * ATT_DH + lane] = (_Float16)(o / sum);
        BM_WAVE_LDS_SYNC();                 // ... before the next query row overwrites it
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
