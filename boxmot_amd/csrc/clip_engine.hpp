// Host-side CLIP-ReID (ViT-B/16) network: owns the weights of a "CLP1" blob (boxmot_amd/clip_weights.py) on the device and
// sequences clip_kernels.hpp for a batch of normalised crops.  Used by ReidEngine (reid_engine.hpp) when the blob's magic is
// CLP1, so every entry point that takes ReID weights (boxmot_hip_reid_*, the trackers' reid_model_path / set_reid_blob) serves
// both backbones.  Reference path: BaseModelBackend.get_features with a "clip" model, base_backend.py:50-54, 197-207.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "clip_kernels.hpp"
#include "reid_pack.hpp"

// A/B switches (tools/clip_bench.py with BOXMOT_HIP_LIB pointing at a variant build)
#ifndef BM_CLIP_ATTN_T
#define BM_CLIP_ATTN_T 1
#endif

namespace bm {

constexpr int CLIP_MAGIC = 0x434C5031;       // "CLP1"
constexpr int CLIP_HEADER_INTS = 16;

struct ClipLayerOff { long ln1_w, ln1_b, qkv_w, qkv_b, out_w, out_b, ln2_w, ln2_b, fc_w, fc_b, proj_w, proj_b; };

class ClipNet {
public:
    int width = 0, layers = 0, heads = 0, patch = 0, gh = 0, gw = 0, out_dim = 0, in_h = 0, in_w = 0, tokens = 0;

    // blob: header (CLIP_HEADER_INTS int32) + fp32 body in the packer's order
    ClipNet(const float* blob, long n_floats, int max_crops, std::vector<void*>& owned) : max_crops_(max_crops) {
        const int32_t* h = reinterpret_cast<const int32_t*>(blob);
        if (n_floats < CLIP_HEADER_INTS || h[0] != CLIP_MAGIC) throw std::runtime_error("CLIP-ReID blob: bad magic (expected CLP1)");
        width = h[1]; layers = h[2]; heads = h[3]; patch = h[4]; gh = h[5]; gw = h[6]; out_dim = h[7]; in_h = h[8]; in_w = h[9];
        tokens = gh * gw + 1;
        const long D = width, K0 = (long)patch * patch * 3;
        if (width % 128 != 0 || heads * 64 != width || K0 % 32 != 0 || tokens > ATT_MAX_T || layers < 1)
            throw std::runtime_error("CLIP-ReID blob: unsupported geometry (width must be a multiple of 128 with 64-wide heads, <= 192 tokens)");
        long off = 0;
        auto take = [&](long n) { const long o = off; off += n; return o; };
        o_conv_ = take(D * K0); o_cls_ = take(D); o_pos_ = take((long)tokens * D); o_lnpre_w_ = take(D); o_lnpre_b_ = take(D);
        L_.resize(layers);
        for (auto& l : L_) {
            l.ln1_w = take(D); l.ln1_b = take(D); l.qkv_w = take(3 * D * D); l.qkv_b = take(3 * D);
            l.out_w = take(D * D); l.out_b = take(D); l.ln2_w = take(D); l.ln2_b = take(D);
            l.fc_w = take(4 * D * D); l.fc_b = take(4 * D); l.proj_w = take(4 * D * D); l.proj_b = take(D);
        }
        o_lnpost_w_ = take(D); o_lnpost_b_ = take(D); o_proj_ = take(D * out_dim);
        o_bn_s_ = take(D); o_bn_b_ = take(D); o_bnp_s_ = take(out_dim); o_bnp_b_ = take(out_dim);
        if (h[10] != (int32_t)off || n_floats != CLIP_HEADER_INTS + off) throw std::runtime_error("CLIP-ReID blob: size does not match the declared geometry");
        const float* body = blob + CLIP_HEADER_INTS;
        d_w_ = alloc<float>((size_t)off, owned);
        check(hipMemcpy(d_w_, body, (size_t)off * 4, hipMemcpyHostToDevice), "upload CLIP weights");
        // fp16 copies of the GEMM operands (round to nearest even, as a tensor .half() would)
        std::vector<uint16_t> h16((size_t)off);
        for (long i = 0; i < off; ++i) h16[i] = f32_to_f16_bits(body[i]);
        d_w16_ = alloc<_Float16>((size_t)off, owned);
        check(hipMemcpy(d_w16_, h16.data(), (size_t)off * 2, hipMemcpyHostToDevice), "upload CLIP fp16 weights");
        const size_t n = (size_t)max_crops, R = n * tokens;
        patches16_ = alloc<_Float16>(n * (tokens - 1) * K0, owned);
        pe_ = alloc<float>(n * (tokens - 1) * D, owned);
        x_ = alloc<float>(R * D, owned);
        h16_ = alloc<_Float16>(R * D, owned);
        qkv16_ = alloc<_Float16>(R * 3 * D, owned);
        mlp16_ = alloc<_Float16>(R * 4 * D, owned);
        set_gemm_lds<0>(); set_gemm_lds<1>(); set_gemm_lds<2>(); set_gemm_lds<3>();
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_clip_attention), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  clip_attn_lds_bytes(tokens)), "attention LDS");
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_clip_attention_t<129>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  clip_attn_t_lds_bytes<129>()), "attention LDS");
    }
    int feature_dim() const { return width + out_dim; }

    // crops: normalised fp32 NHWC [n][in_h][in_w][3] on the device; out rows of feature_dim() floats, L2-normalised
    void forward(const float* d_crops, int n, float* d_out, const int* d_out_rows, hipStream_t st) {
        if (n > max_crops_) throw std::runtime_error("CLIP-ReID: crop batch exceeds the engine capacity");
        const int D = width, T = tokens, K0 = patch * patch * 3;
        const long R = (long)n * T, RP = (long)n * (T - 1);
        {
            const long total = RP * K0;
            const unsigned blocks = (unsigned)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
            hipLaunchKernelGGL(k_clip_patches, dim3(blocks), dim3(256), 0, st, d_crops, patches16_, n, in_h, in_w, patch, gh, gw);
        }
        gemm<3>(patches16_, d_w16_ + o_conv_, nullptr, pe_, RP, D, K0, st);
        hipLaunchKernelGGL(k_clip_tokens_lnpre, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, pe_, d_w_ + o_cls_, d_w_ + o_pos_,
                           d_w_ + o_lnpre_w_, d_w_ + o_lnpre_b_, x_, R, T, D);
        for (const ClipLayerOff& l : L_) {
            layernorm(d_w_ + l.ln1_w, d_w_ + l.ln1_b, R, st);
            gemm<0>(h16_, d_w16_ + l.qkv_w, d_w_ + l.qkv_b, qkv16_, R, 3 * D, D, st);
            if (BM_CLIP_ATTN_T && T == 129)        // ViT-B/16 on 256 x 128 crops: the compile-time-T kernel (identical bits, tools/attn_prof.hip)
                hipLaunchKernelGGL((k_clip_attention_t<129>), dim3((unsigned)(n * heads)), dim3(192), (size_t)clip_attn_t_lds_bytes<129>(), st,
                                   qkv16_, h16_, D, heads);
            else
                hipLaunchKernelGGL(k_clip_attention, dim3((unsigned)(n * heads)), dim3(256), (size_t)clip_attn_lds_bytes(T), st,
                                   qkv16_, h16_, T, D, heads);
            gemm<2>(h16_, d_w16_ + l.out_w, d_w_ + l.out_b, x_, R, D, D, st);
            layernorm(d_w_ + l.ln2_w, d_w_ + l.ln2_b, R, st);
            gemm<1>(h16_, d_w16_ + l.fc_w, d_w_ + l.fc_b, mlp16_, R, 4 * D, D, st);
            gemm<2>(mlp16_, d_w16_ + l.proj_w, d_w_ + l.proj_b, x_, R, D, 4 * D, st);
        }
        hipLaunchKernelGGL(k_clip_head, dim3(n), dim3(256), (size_t)(D + 8) * 4, st, x_, d_w_ + o_lnpost_w_, d_w_ + o_lnpost_b_,
                           d_w_ + o_proj_, d_w_ + o_bn_s_, d_w_ + o_bn_b_, d_w_ + o_bnp_s_, d_w_ + o_bnp_b_, d_out, d_out_rows, T, D, out_dim);
        check(hipGetLastError(), "CLIP-ReID launch");
    }

private:
    static void check(hipError_t e, const char* what) {
        if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
    }
    template <typename T>
    static T* alloc(size_t n, std::vector<void*>& owned) {
        void* p = nullptr;
        check(hipMalloc(&p, (n ? n : 1) * sizeof(T)), "hipMalloc");
        owned.push_back(p);
        return static_cast<T*>(p);
    }
    template <int EPI>
    static void set_gemm_lds() {
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_glds<EPI, 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  gemm_glds_lds_bytes<64>()), "GEMM LDS");
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_256<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  GEMM256_LDS_BYTES), "GEMM LDS");
    }
    template <int EPI>
    void gemm(const _Float16* X, const _Float16* W, const float* bias, void* C, long M, int N, int K, hipStream_t st) {
        if (N % GEMM_BN != 0 || K % GEMM_BK != 0) throw std::runtime_error("CLIP-ReID: GEMM shape not tileable");
        if (N % 256 == 0 && K % 64 == 0 && M >= 1024) {     // 256 x 256 tiles, phased k-loop: 739-1054 TFLOP/s on the four layer shapes against
            // 623-833 (profiles/r3_gemm_prof.txt).  (Tried in round 6 and dropped, profiles/r6_clip_ab.txt: stopping the 256-row tiles at the
            // last full round of CUs and sending the remaining rows through the 128 x 128 kernel; starting every second CU half a tile late.)
            hipLaunchKernelGGL((k_gemm_f16_256<EPI>), dim3((unsigned)(((M + 255) / 256) * (N / 256))), dim3(512), GEMM256_LDS_BYTES, st, X, W, bias, C,
                               static_cast<const _Float16*>(nullptr), (int)M, N, K, 0);
            return;
        }
        const dim3 grid((unsigned)(((M + GEMM_BM - 1) / GEMM_BM) * (N / GEMM_BN)));       // 1-D: the kernels map ids to tiles XCD-aware
        if (K % 64 == 0)     // 64-wide k-tiles: 11.7 ms per 256-crop forward vs 13.3 ms with 32-wide ones (profiles/r2_gemm_pipeline_ab.txt)
            hipLaunchKernelGGL((k_gemm_f16_glds<EPI, 64>), grid, dim3(256), gemm_glds_lds_bytes<64>(), st, X, W, bias, C,
                               static_cast<const _Float16*>(nullptr), (int)M, N, K, 0, GemmExt{});
        else
            hipLaunchKernelGGL((k_gemm_f16<EPI>), grid, dim3(256), 0, st, X, W, bias, C, static_cast<const _Float16*>(nullptr), (int)M, N, K, 0);
    }
    void layernorm(const float* g, const float* b, long R, hipStream_t st) {
        hipLaunchKernelGGL(k_clip_layernorm_f16, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x_, g, b, h16_, R, width);
    }

    int max_crops_;
    std::vector<ClipLayerOff> L_;
    long o_conv_ = 0, o_cls_ = 0, o_pos_ = 0, o_lnpre_w_ = 0, o_lnpre_b_ = 0, o_lnpost_w_ = 0, o_lnpost_b_ = 0, o_proj_ = 0;
    long o_bn_s_ = 0, o_bn_b_ = 0, o_bnp_s_ = 0, o_bnp_b_ = 0;
    float* d_w_ = nullptr;
    _Float16* d_w16_ = nullptr;
    _Float16 *patches16_ = nullptr, *h16_ = nullptr, *qkv16_ = nullptr, *mlp16_ = nullptr;
    float *pe_ = nullptr, *x_ = nullptr;
};

}  // namespace bm
