// Host side of the wide-OSNet kernel family (osnet_wide_kernels.hpp): owns the fp16 weight copies and the activation buffers
// for up to max_crops crops and sequences the launches of OSNet.forward (boxmot/reid/backbones/osnet.py:380-405).  Used by
// ReidEngine (reid_engine.hpp) in mode 1 when the OSN1 blob's widths are multiples of 32 (osnet_x1_0).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "osnet_wide_kernels.hpp"
#include "osnet_wide_pack.hpp"

namespace bm {

// ---------------------------------------------------------------------------
// Host side: owns the fp16 weight copies and the activation buffers for up to max_crops crops, sequences the launches.
// ---------------------------------------------------------------------------
class WideOsnet {
public:
    static bool supports(const OsnetLayout& L) { return wide_osnet_supports(L); }

    WideOsnet(const float* h_w, const OsnetLayout& L, const float* d_w32, int max_crops, std::vector<void*>& owned)
        : L_(L), d_w_(d_w32), max_crops_(max_crops) {
        if (!supports(L)) throw std::runtime_error("wide OSNet kernels: channel widths must be multiples of 32 (middle widths <= 128)");
        pk_ = wide_pack_w16(h_w, L);
        d_w16_ = alloc<_Float16>(pk_.data.size(), owned);
        check(hipMemcpy(d_w16_, pk_.data.data(), pk_.data.size() * 2, hipMemcpyHostToDevice), "upload fp16 weights");
        d_stem16_ = d_w16_ + pk_.stem;
        pk_.data.clear(); pk_.data.shrink_to_fit();
        const int c0 = L.c[0];
        const size_t n = (size_t)max_crops;
        const size_t crop_halves = n * WSTEM_ROWS * WSTEM_COLS * 4;
        crops16_ = alloc<_Float16>(crop_halves, owned);
        check(hipMemset(crops16_, 0, crop_halves * 2), "clear crop buffer");       // the 3-pixel border and the X channel stay zero
        size_t blk = 0, mid = 0;
        int P = 2048;
        for (int s = 0; s < 3; ++s, P /= 4) {
            const size_t a = n * P * (size_t)L.c[s + 1], m = n * P * (size_t)(L.c[s + 1] / 4);
            blk = a > blk ? a : blk; mid = m > mid ? m : mid;
        }
        const size_t first = n * 2048 * (size_t)c0;
        blk = first > blk ? first : blk;
        act_a_ = alloc<_Float16>(blk, owned); act_b_ = alloc<_Float16>(blk, owned);
        for (auto& m : mid_) m = alloc<_Float16>(mid, owned);
        gap_part_ = alloc<float>(4 * n * (64 / WIDE_BAND) * 128, owned);
        gap16_ = alloc<_Float16>(n * L.c[3], owned);
        fc32_ = alloc<float>(n * L.feat, owned);
        if (L.feat % 128 != 0 || L.c[3] % 32 != 0) throw std::runtime_error("wide OSNet: head GEMM shape");
        check(hipDeviceSynchronize(), "clear crop buffer");     // nothing asynchronous is pending if a later step of the constructor throws
        for (int b = 0; b < 6; ++b) {                            // dynamic-LDS limits for the (middle width, image width) pairs this network uses
            const int C = L.block[b].mid, W = 32 >> (b / 2);
            switch (C) {
                case 32: set_light_lds<32>(W); break;
                case 64: set_light_lds<64>(W); break;
                case 96: set_light_lds<96>(W); break;
                case 128: set_light_lds<128>(W); break;
                default: throw std::runtime_error("wide OSNet: unsupported middle width");
            }
        }
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_glds<4, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  gemm_glds_lds_bytes<32>()), "GEMM LDS");
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_glds<5, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  gemm_glds_lds_bytes<32>()), "GEMM LDS");
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_glds<6, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  gemm_glds_lds_bytes<32>()), "GEMM LDS");
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_glds<3, 32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  gemm_glds_lds_bytes<32>()), "GEMM LDS");
        // conv3 + downsample run as one GEMM: their folded-BN biases add up
        std::vector<float> bs;
        for (int b = 0; b < 6; ++b) {
            const BlockW& B = L.block[b];
            bsum_off_[b] = -1;
            if (B.down_w < 0) continue;
            bsum_off_[b] = (long)bs.size();
            for (int c = 0; c < B.cout; ++c) bs.push_back(h_w[B.conv3_b + c] + h_w[B.down_b + c]);
        }
        d_bsum_ = alloc<float>(bs.size(), owned);
        if (!bs.empty()) check(hipMemcpy(d_bsum_, bs.data(), bs.size() * 4, hipMemcpyHostToDevice), "upload summed biases");
    }

    _Float16* crops_buffer() { return crops16_; }
    int max_crops() const { return max_crops_; }

    // crops16_: normalised fp16 RGBX crops with a zero border, [n][262][136][4] (k_crop_resize_rgbx); out rows of L.feat floats
    void forward(int n, float* d_out, const int* d_out_rows, hipStream_t st) {
        if (n > max_crops_) throw std::runtime_error("wide OSNet: crop batch exceeds the engine capacity");
        if (n == 0) return;
        const int c0 = L_.c[0];
        // stem conv + ReLU + 3x3 max pool in one launch: act_a_ = [n][64 * 32][c0]
        if (c0 == 64) hipLaunchKernelGGL(k_wide_stem<64>, dim3(64 / WSTEM_PBAND, n), dim3(256), 0, st, crops16_, d_stem16_, d_w_ + L_.stem_b, act_a_);
        else if (c0 == 32) hipLaunchKernelGGL(k_wide_stem<32>, dim3(64 / WSTEM_PBAND, n), dim3(256), 0, st, crops16_, d_stem16_, d_w_ + L_.stem_b, act_a_);
        else throw std::runtime_error("wide OSNet: stem width must be 32 or 64");
        _Float16 *cur = act_a_, *other = act_b_;
        int H = 64, W = 32;
        for (int s = 0; s < 3; ++s) {
            for (int k = 0; k < 2; ++k) {
                osblock(L_.block[s * 2 + k], cur, other, n, H, W, st);
                std::swap(cur, other);
            }
            if (s < 2) {
                const int c = L_.c[s + 1];
                GemmExt pool;
                pool.pool_w = W;
                gemm(cur, d_w16_ + pk_.of(L_.trans_w[s]), d_w_ + L_.trans_b[s], other, nullptr, (long)n * H * W, c, c, 1, st, pool);
                std::swap(cur, other);
                H /= 2; W /= 2;
            }
        }
        const int c3 = L_.c[3];
        gemm(cur, d_w16_ + pk_.of(L_.conv5_w), d_w_ + L_.conv5_b, other, nullptr, (long)n * H * W, c3, c3, 1, st);
        // head: GAP -> FC (+ folded BN, ReLU) as one GEMM over all crops -> L2 + scatter
        const long g8 = (long)n * (c3 / 8);
        hipLaunchKernelGGL(k_wide_gap, dim3((unsigned)((g8 + 255) / 256)), dim3(256), 0, st, other, gap16_, H * W, c3, g8);
        {
            const long mt = ((long)n + GEMM_BM - 1) / GEMM_BM;
            hipLaunchKernelGGL((k_gemm_f16_glds<3, 32>), dim3((unsigned)(mt * (L_.feat / 128))), dim3(256), gemm_glds_lds_bytes<32>(), st, gap16_,
                               d_w16_ + pk_.of(L_.fc_w), d_w_ + L_.fc_b, static_cast<void*>(fc32_), static_cast<const _Float16*>(nullptr), n, L_.feat, c3,
                               1, GemmExt{});
        }
        hipLaunchKernelGGL(k_wide_l2, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, fc32_, d_out, d_out_rows, (long)n, L_.feat);
        check(hipGetLastError(), "wide OSNet launch");
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
    template <int C>
    static void set_light_lds(int W) {
        check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_light_fused<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  light_lds_bytes<C>(W)), "LightConv LDS");
    }
    // 1x1 convolution over n_pix pixels: out = [relu](X . W^T + bias [+ res])
    void gemm(const _Float16* X, const _Float16* W, const float* bias, _Float16* out, const _Float16* res, long M, int N, int K,
              int relu, hipStream_t st, GemmExt ext = GemmExt{}) {
        if (K % GEMM_BK != 0 || N % 32 != 0) throw std::runtime_error("wide OSNet: GEMM shape not tileable");
        const long mt = (M + GEMM_BM - 1) / GEMM_BM;          // 1-D grids: the kernels map workgroup ids to tiles XCD-aware
        void* o = static_cast<void*>(out);
        if (ext.pool_w) {       // transition: conv + ReLU + 2 x 2 average pool in one launch (M must hold whole pairs of image rows per tile)
            if (N % 128 != 0 || (ext.pool_w != 16 && ext.pool_w != 32)) throw std::runtime_error("wide OSNet: pooled GEMM shape");
            if (ext.pool_w == 32)
                hipLaunchKernelGGL((k_gemm_f16_glds<5, 32>), dim3((unsigned)(mt * (N / 128))), dim3(256), gemm_glds_lds_bytes<32>(), st, X, W, bias, o, res, (int)M, N, K, relu, ext);
            else
                hipLaunchKernelGGL((k_gemm_f16_glds<6, 32>), dim3((unsigned)(mt * (N / 128))), dim3(256), gemm_glds_lds_bytes<32>(), st, X, W, bias, o, res, (int)M, N, K, relu, ext);
        } else if (N % 128 == 0)
            hipLaunchKernelGGL((k_gemm_f16_glds<4, 32>), dim3((unsigned)(mt * (N / 128))), dim3(256), gemm_glds_lds_bytes<32>(), st, X, W, bias, o, res, (int)M, N, K, relu, ext);
        else if (ext.K2) throw std::runtime_error("wide OSNet: the two-operand GEMM needs N % 128 == 0");
        else if (N % 96 == 0) hipLaunchKernelGGL((k_gemm_f16<4, 96>), dim3((unsigned)(mt * (N / 96))), dim3(256), 0, st, X, W, bias, o, res, (int)M, N, K, relu);
        else if (N % 64 == 0) hipLaunchKernelGGL((k_gemm_f16<4, 64>), dim3((unsigned)(mt * (N / 64))), dim3(256), 0, st, X, W, bias, o, res, (int)M, N, K, relu);
        else hipLaunchKernelGGL((k_gemm_f16<4, 32>), dim3((unsigned)(mt * (N / 32))), dim3(256), 0, st, X, W, bias, o, res, (int)M, N, K, relu);
    }
    template <int C>
    void light_t(const _Float16* in, const LightW& lw, _Float16* out, float* gap, int n, int H, int W, hipStream_t st) {
        hipLaunchKernelGGL(k_light_fused<C>, dim3(H / WIDE_BAND, n), dim3(256), (size_t)light_lds_bytes<C>(W), st, in, d_w16_ + pk_.of(lw.pw),
                           d_w_ + lw.dw, d_w_ + lw.b, out, gap, H, W);
    }
    void light(int C, const _Float16* in, const LightW& lw, _Float16* out, float* gap, int n, int H, int W, hipStream_t st) {
        switch (C) {
            case 32: light_t<32>(in, lw, out, gap, n, H, W, st); break;
            case 64: light_t<64>(in, lw, out, gap, n, H, W, st); break;
            case 96: light_t<96>(in, lw, out, gap, n, H, W, st); break;
            case 128: light_t<128>(in, lw, out, gap, n, H, W, st); break;
            default: throw std::runtime_error("wide OSNet: unsupported middle width");
        }
    }
    template <int C>
    void gate_t(const BlockW& B, _Float16* const* br, _Float16* out, int n, int P, int nbands, hipStream_t st) {
        const int ppb = 128;
        hipLaunchKernelGGL(k_gate_sum4<C>, dim3(n, (P + ppb - 1) / ppb), dim3(256), 0, st, br[0], br[1], br[2], br[3], gap_part_,
                           d_w_ + B.fc1_w, d_w_ + B.fc1_b, d_w_ + B.fc2_w, d_w_ + B.fc2_b, out, P, nbands, (long)n, ppb);
    }
    void osblock(const BlockW& B, const _Float16* x, _Float16* out, int n, int H, int W, hipStream_t st) {
        const long n_pix = (long)n * H * W;
        const int nbands = H / WIDE_BAND;
        _Float16* x1 = mid_[0];
        _Float16* brs[4] = {mid_[1], mid_[2], mid_[3], mid_[4]};
        _Float16* tmp[2] = {mid_[5], mid_[6]};
        gemm(x, d_w16_ + pk_.of(B.conv1_w), d_w_ + B.conv1_b, x1, nullptr, n_pix, B.mid, B.cin, 1, st);
        int li = 0;
        for (int br = 0; br < 4; ++br) {
            // a branch is a chain of br + 1 LightConvs, one launch each
            const _Float16* cur = x1;
            const int L = br + 1;
            for (int k = 0; k < L; ++k) {
                const bool last = k + 1 == L;
                _Float16* dst = last ? brs[br] : tmp[k & 1];
                light(B.mid, cur, B.light[li + k], dst, last ? gap_part_ + (long)br * n * nbands * B.mid : nullptr, n, H, W, st);
                cur = dst;
            }
            li += L;
        }
        _Float16* x2 = mid_[7];
        switch (B.mid) {
            case 32: gate_t<32>(B, brs, x2, n, H * W, nbands, st); break;
            case 64: gate_t<64>(B, brs, x2, n, H * W, nbands, st); break;
            case 96: gate_t<96>(B, brs, x2, n, H * W, nbands, st); break;
            case 128: gate_t<128>(B, brs, x2, n, H * W, nbands, st); break;
            default: throw std::runtime_error("wide OSNet: unsupported middle width");
        }
        if (B.down_w >= 0) {        // out = relu(conv3(x2) + downsample(x)): one GEMM over the two operand pairs, the shortcut tensor never exists
            GemmExt two;
            two.X2 = x; two.W2 = d_w16_ + pk_.of(B.down_w); two.K2 = B.cin;
            gemm(x2, d_w16_ + pk_.of(B.conv3_w), d_bsum_ + bsum_off_[&B - L_.block], out, nullptr, n_pix, B.cout, B.mid, 1, st, two);
        } else
            gemm(x2, d_w16_ + pk_.of(B.conv3_w), d_w_ + B.conv3_b, out, x, n_pix, B.cout, B.mid, 1, st);
    }

    OsnetLayout L_;
    WideW16 pk_;
    const float* d_w_;                       // the engine's fp32 blob on the device (biases, depthwise taps, gate and FC weights)
    int max_crops_;
    _Float16 *d_w16_ = nullptr, *d_stem16_ = nullptr;
    _Float16 *crops16_ = nullptr, *act_a_ = nullptr, *act_b_ = nullptr;
    _Float16* mid_[8] = {};
    float* gap_part_ = nullptr;
    _Float16* gap16_ = nullptr;             // [n][c3] pooled features, the FC GEMM's activation operand
    float* fc32_ = nullptr;                 // [n][feat] FC output before the L2 norm
    float* d_bsum_ = nullptr;
    long bsum_off_[6] = {};
};

}  // namespace bm
