// Launch sequence and host side of the fp32-grade wide-OSNet kernel family (osnet_wide_hp_kernels.hpp): OSNet.forward
// (boxmot/reid/backbones/osnet.py:380-405) for widths that are multiples of 32 (osnet_x1_0), every matrix-pipe operand an fp16
// (hi, lo) pair.  Used by ReidEngine (reid_engine.hpp) in mode 2 when the blob is not OSNet-x0.25.
//
// `wide_hp_forward` is the ONE statement of the launch order: the engine passes a launcher that enqueues on a HIP stream, the test
// harness (tests/host_emu/emu_wide_hp.cpp) one that runs the same kernels on CPU threads.
#pragma once

#include <algorithm>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "osnet_wide_hp_kernels.hpp"
#include "osnet_wide_hp_pack.hpp"

#ifndef BM_WIDE_HP_SHALLOW
#define BM_WIDE_HP_SHALLOW 1
#endif

namespace bm {

// device (or emulated-device) buffers of one pass of up to `n` crops
struct WideHpBuffers {
    _Float16 *crops_h = nullptr, *crops_l = nullptr;        // [n][262][136][4] RGBX planes, zero border
    _Float16 *a_h = nullptr, *a_l = nullptr, *b_h = nullptr, *b_l = nullptr;      // block inputs / outputs, ping-pong
    _Float16 *x1_h = nullptr, *x1_l = nullptr;              // conv1 output [n][P][mid]
    _Float16 *y_h = nullptr, *y_l = nullptr;                // branch outputs [4][n][P][mid]
    _Float16 *x2_h = nullptr, *x2_l = nullptr;              // gated branch sum [n][P][mid]
    float* gap_part = nullptr;                              // [4][n][bands][mid]
    _Float16 *gap_h = nullptr, *gap_l = nullptr;            // pooled conv5 output [n][c3]
    float* fc32 = nullptr;                                  // [n][feat]
};

// halves per crop of the largest tensor of each kind
inline size_t wide_hp_act_halves(const OsnetLayout& L) {
    size_t m = (size_t)2048 * L.c[0];
    int P = 2048;
    for (int s = 0; s < 3; ++s, P /= 4) m = std::max(m, (size_t)P * L.c[s + 1]);
    return m;
}
inline size_t wide_hp_mid_halves(const OsnetLayout& L) {
    size_t m = 0;
    int P = 2048;
    for (int s = 0; s < 3; ++s, P /= 4) m = std::max(m, (size_t)P * (L.c[s + 1] / 4));
    return m;
}
constexpr int WIDE_HP_MAX_BANDS = 4;

// k_chain_hp instantiation of a stage: (image width, window rows, halo); stage 0 runs 4 bands of 16 rows with a 4-row halo, stages
// 1 and 2 hold the whole image (32 x 16, 16 x 8) in one workgroup
template <class Launch, int C>
inline void wide_hp_chain(Launch& launch, int stage, const _Float16* x1h, const _Float16* x1l, const unsigned char* w, _Float16* yh,
                          _Float16* yl, float* gap, int n) {
    if (stage == 0) {
        if constexpr (C <= 64) launch(k_chain_hp<C, 32, 24, 4>, 4, n, 512, ChainGeo<C, 32, 24, 4>::LDS_BYTES, x1h, x1l, w, yh, yl, gap, 64, (long)n);
        else throw std::runtime_error("wide OSNet (fp32-grade): middle width of stage 0 must be <= 64");
    } else if (stage == 1) {
        if constexpr (C <= 96) launch(k_chain_hp<C, 16, 32, 0>, 1, n, 512, ChainGeo<C, 16, 32, 0>::LDS_BYTES, x1h, x1l, w, yh, yl, gap, 32, (long)n);
        else throw std::runtime_error("wide OSNet (fp32-grade): middle width of stage 1 must be <= 96");
    } else
        launch(k_chain_hp<C, 8, 16, 0>, 1, n, 512, ChainGeo<C, 8, 16, 0>::LDS_BYTES, x1h, x1l, w, yh, yl, gap, 16, (long)n);
}
inline int wide_hp_bands(int stage) { return stage == 0 ? 4 : 1; }

struct WideHpNoTap { void operator()(int, const _Float16*, const _Float16*, long, int) const {} };

// `tap(block, out_h, out_l, pixels, channels)` is called after each OSBlock has been enqueued (the test harness, whose launches are
// synchronous, copies the tensor out there)
template <class Launch, class Tap = WideHpNoTap>
void wide_hp_forward(Launch& launch, const OsnetLayout& L, const WideHpPack& P, const unsigned char* wp, const float* w32,
                     const WideHpBuffers& B, int n, float* d_out, const int* d_out_rows, Tap tap = Tap()) {
    if (n == 0) return;
    auto H16 = [&](long off) { return reinterpret_cast<const _Float16*>(wp + off); };
    auto F32 = [&](long off) { return reinterpret_cast<const float*>(wp + off); };
    // 1x1 convolution over M pixel rows: out = epi(X . W^T [+ X2 . W2^T] + bias [+ res])
    auto gemm = [&](int epi, const _Float16* xh, const _Float16* xl, const GemmWHp& w, void* oh, void* ol, const _Float16* rh,
                    const _Float16* rl, long M, int relu, const _Float16* x2h = nullptr, const _Float16* x2l = nullptr,
                    const GemmWHp* w2 = nullptr) {
        GemmHpExt ext;
        if (w2) { ext.X2h = x2h; ext.X2l = x2l; ext.W2h = H16(w2->wh); ext.W2l = H16(w2->wl); ext.K2 = w2->k; }
        const int N = w.n, K = w.k;
        if (K % 32 != 0 || (w2 && w2->k % 32 != 0)) throw std::runtime_error("wide OSNet (fp32-grade): GEMM depth not a multiple of 32");
        const long mt = (M + GEMM_BM - 1) / GEMM_BM;
        const _Float16 *wh = H16(w.wh), *wl = H16(w.wl);
        const float* bias = F32(w.bias);
        // shallow products without a residual (conv1, conv3 + downsample) run the single-buffer form at four workgroups per CU
        // (k_gemm_hp; profiles/r4_c3_gemm_occupancy_ab.txt: -8 %; the residual form loses its early residual fetch there: +13 %)
        const bool shallow = BM_WIDE_HP_SHALLOW && K + (w2 ? w2->k : 0) <= 512 && epi == 0;
#define BM_HP_GEMM(EPI, BN) do { if (EPI == 0 && shallow) launch(k_gemm_hp<EPI, BN, (EPI == 0 ? 1 : 2)>, (int)(mt * (N / BN)), 1, 256, gemm_hp_lds_bytes<BN, 1>(), xh, xl, wh, wl, bias, oh, ol, rh, rl, (int)M, N, K, relu, ext); \
                                 else launch(k_gemm_hp<EPI, BN, 2>, (int)(mt * (N / BN)), 1, 256, gemm_hp_lds_bytes<BN, 2>(), xh, xl, wh, wl, bias, oh, ol, rh, rl, (int)M, N, K, relu, ext); } while (0)
        if (N % 128 == 0) {
            switch (epi) {
                case 0: BM_HP_GEMM(0, 128); break;
                case 1: BM_HP_GEMM(1, 128); break;
                case 2: BM_HP_GEMM(2, 128); break;
                case 3: BM_HP_GEMM(3, 128); break;
                default: BM_HP_GEMM(4, 128); break;
            }
        } else if (epi != 0) throw std::runtime_error("wide OSNet (fp32-grade): this epilogue needs N % 128 == 0");
        else if (N % 96 == 0) BM_HP_GEMM(0, 96);
        else if (N % 64 == 0) BM_HP_GEMM(0, 64);
        else if (N % 32 == 0) BM_HP_GEMM(0, 32);
        else throw std::runtime_error("wide OSNet (fp32-grade): GEMM width not a multiple of 32");
#undef BM_HP_GEMM
    };
    const int c0 = L.c[0];
    if (c0 == 64) launch(k_wide_stem_hp<64>, 64 / WSTEM_PBAND, n, 256, 0, (const _Float16*)B.crops_h, (const _Float16*)B.crops_l, wp + P.stem_a, F32(P.stem_b), B.a_h, B.a_l);
    else if (c0 == 32) launch(k_wide_stem_hp<32>, 64 / WSTEM_PBAND, n, 256, 0, (const _Float16*)B.crops_h, (const _Float16*)B.crops_l, wp + P.stem_a, F32(P.stem_b), B.a_h, B.a_l);
    else throw std::runtime_error("wide OSNet (fp32-grade): stem width must be 32 or 64");
    _Float16 *ch = B.a_h, *cl = B.a_l, *oh = B.b_h, *ol = B.b_l;
    int Himg = 64, Wimg = 32;
    for (int s = 0; s < 3; ++s) {
        for (int k = 0; k < 2; ++k) {
            const int b = s * 2 + k;
            const BlockW& Bw = L.block[b];
            const BlockHp& Bp = P.block[b];
            const int Pp = Himg * Wimg;
            const long n_pix = (long)n * Pp;
            gemm(0, ch, cl, Bp.conv1, B.x1_h, B.x1_l, nullptr, nullptr, n_pix, 1);
            const unsigned char* cw = wp + Bp.chain;
            switch (Bw.mid) {
                case 32: wide_hp_chain<Launch, 32>(launch, s, B.x1_h, B.x1_l, cw, B.y_h, B.y_l, B.gap_part, n); break;
                case 64: wide_hp_chain<Launch, 64>(launch, s, B.x1_h, B.x1_l, cw, B.y_h, B.y_l, B.gap_part, n); break;
                case 96: wide_hp_chain<Launch, 96>(launch, s, B.x1_h, B.x1_l, cw, B.y_h, B.y_l, B.gap_part, n); break;
                case 128: wide_hp_chain<Launch, 128>(launch, s, B.x1_h, B.x1_l, cw, B.y_h, B.y_l, B.gap_part, n); break;
                default: throw std::runtime_error("wide OSNet (fp32-grade): unsupported middle width");
            }
            const int ppb = 128, nb = wide_hp_bands(s);
            const float *f1w = w32 + Bw.fc1_w, *f1b = w32 + Bw.fc1_b, *f2w = w32 + Bw.fc2_w, *f2b = w32 + Bw.fc2_b;
#define BM_HP_GATE(CC) launch(k_gate_sum4_hp<CC>, n, (Pp + ppb - 1) / ppb, 256, 0, (const _Float16*)B.y_h, (const _Float16*)B.y_l, (const float*)B.gap_part, f1w, f1b, f2w, f2b, B.x2_h, B.x2_l, Pp, nb, (long)n, ppb)
            switch (Bw.mid) {
                case 32: BM_HP_GATE(32); break;
                case 64: BM_HP_GATE(64); break;
                case 96: BM_HP_GATE(96); break;
                default: BM_HP_GATE(128); break;
            }
#undef BM_HP_GATE
            if (Bw.down_w >= 0) gemm(0, B.x2_h, B.x2_l, Bp.conv3, oh, ol, nullptr, nullptr, n_pix, 1, ch, cl, &Bp.down);
            else gemm(1, B.x2_h, B.x2_l, Bp.conv3, oh, ol, ch, cl, n_pix, 1);
            std::swap(ch, oh); std::swap(cl, ol);
            tap(b, ch, cl, n_pix, Bw.cout);
        }
        if (s < 2) {
            gemm(Wimg == 32 ? 2 : 3, ch, cl, P.trans[s], oh, ol, nullptr, nullptr, (long)n * Himg * Wimg, 1);
            std::swap(ch, oh); std::swap(cl, ol);
            Himg /= 2; Wimg /= 2;
        }
    }
    const int c3 = L.c[3], Pp = Himg * Wimg;
    gemm(0, ch, cl, P.conv5, oh, ol, nullptr, nullptr, (long)n * Pp, 1);
    const long g8 = (long)n * (c3 / 8);
    launch(k_wide_gap_hp, (int)((g8 + 255) / 256), 1, 256, 0, (const _Float16*)oh, (const _Float16*)ol, B.gap_h, B.gap_l, Pp, c3, g8);
    gemm(4, B.gap_h, B.gap_l, P.fc, B.fc32, nullptr, nullptr, nullptr, n, 1);
    launch(k_wide_l2, (n + 3) / 4, 1, 256, 0, (const float*)B.fc32, d_out, d_out_rows, (long)n, L.feat);
}

#ifdef __HIPCC__
// ---------------------------------------------------------------------------
// Host side: owns the packed (hi, lo) weights and the activation buffers for up to max_crops crops.
// ---------------------------------------------------------------------------
class WideOsnetHP {
public:
    static bool supports(const OsnetLayout& L) { return wide_hp_supports(L); }

    WideOsnetHP(const float* h_w, const OsnetLayout& L, const float* d_w32, int max_crops, std::vector<void*>& owned)
        : L_(L), d_w_(d_w32), max_crops_(max_crops) {
        if (!supports(L)) throw std::runtime_error("wide OSNet (fp32-grade kernels): widths must be multiples of 32 / 128 (middle widths <= 64 / 96 / 128 per stage)");
        pk_ = wide_pack_hp(h_w, L);
        d_wp_ = alloc<unsigned char>(pk_.data.size(), owned);
        check(hipMemcpy(d_wp_, pk_.data.data(), pk_.data.size(), hipMemcpyHostToDevice), "upload (hi, lo) weights");
        pk_.data.clear(); pk_.data.shrink_to_fit();
        const size_t n = (size_t)max_crops;
        const size_t crop_halves = n * WSTEM_ROWS * WSTEM_COLS * 4;
        buf_.crops_h = alloc<_Float16>(crop_halves, owned); buf_.crops_l = alloc<_Float16>(crop_halves, owned);
        check(hipMemset(buf_.crops_h, 0, crop_halves * 2), "clear crop buffer");       // the 3-pixel border and the X channel stay zero
        check(hipMemset(buf_.crops_l, 0, crop_halves * 2), "clear crop buffer");
        const size_t act = n * wide_hp_act_halves(L), mid = n * wide_hp_mid_halves(L);
        buf_.a_h = alloc<_Float16>(act, owned); buf_.a_l = alloc<_Float16>(act, owned);
        buf_.b_h = alloc<_Float16>(act, owned); buf_.b_l = alloc<_Float16>(act, owned);
        buf_.x1_h = alloc<_Float16>(mid, owned); buf_.x1_l = alloc<_Float16>(mid, owned);
        buf_.y_h = alloc<_Float16>(4 * mid, owned); buf_.y_l = alloc<_Float16>(4 * mid, owned);
        buf_.x2_h = alloc<_Float16>(mid, owned); buf_.x2_l = alloc<_Float16>(mid, owned);
        buf_.gap_part = alloc<float>(4 * n * WIDE_HP_MAX_BANDS * 128, owned);
        buf_.gap_h = alloc<_Float16>(n * L.c[3], owned); buf_.gap_l = alloc<_Float16>(n * L.c[3], owned);
        buf_.fc32 = alloc<float>(n * L.feat, owned);
        check(hipDeviceSynchronize(), "clear crop buffer");
    }

    _Float16* crops_h() { return buf_.crops_h; }
    _Float16* crops_l() { return buf_.crops_l; }
    int max_crops() const { return max_crops_; }

    void forward(int n, float* d_out, const int* d_out_rows, hipStream_t st) {
        if (n > max_crops_) throw std::runtime_error("wide OSNet (fp32-grade): crop batch exceeds the engine capacity");
        HipLaunchLds launch{st};
        wide_hp_forward(launch, L_, pk_, d_wp_, d_w_, buf_, n, d_out, d_out_rows);
        check(hipGetLastError(), "wide OSNet (fp32-grade) launch");
    }

private:
    struct HipLaunchLds {
        hipStream_t stream;
        template <class K, class... A>
        void operator()(K kernel, int gx, int gy, int threads, int lds_bytes, A... args) {
            // a kernel's dynamic-LDS limit is raised once per kernel symbol (and again only if a larger request shows up)
            if (lds_bytes > 0) {
                static std::mutex mu;
                static std::unordered_map<const void*, int> allowed;
                const void* sym = reinterpret_cast<const void*>(kernel);
                std::lock_guard<std::mutex> lock(mu);
                int& have = allowed[sym];
                if (have < lds_bytes) {
                    check(hipFuncSetAttribute(sym, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes), "dynamic LDS limit");
                    have = lds_bytes;
                }
            }
            hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)gy), dim3((unsigned)threads), (size_t)lds_bytes, stream, args...);
        }
    };
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

    OsnetLayout L_;
    WideHpPack pk_;                          // offsets (the bytes live on the device)
    const float* d_w_;                       // the engine's fp32 blob on the device (gate weights)
    int max_crops_;
    unsigned char* d_wp_ = nullptr;
    WideHpBuffers buf_;
};
#endif

}  // namespace bm
