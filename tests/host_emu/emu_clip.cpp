// TEST-ONLY harness: runs the CLIP-ReID kernels (boxmot_amd/csrc/clip_kernels.hpp, the device source unchanged) on CPU threads
// with the emulated MFMA of hip_shim.hpp, in the launch order of ClipNet::forward (clip_engine.hpp).  Reduced geometries
// (width 128, a few layers, a handful of tokens) keep it to seconds; the full ViT-B/16 runs on the GPU (tests/test_gpu_clipreid.py).
#include "hip_shim.hpp"

#include <cstdlib>
#include <functional>
#include <vector>

#include "../../boxmot_amd/csrc/clip_kernels.hpp"
#include "../../boxmot_amd/csrc/reid_pack.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuDim3 gridDim;
EmuBlock* g_emu_block = nullptr;
unsigned char* g_emu_dynamic_lds = nullptr;
EmuMfmaBuf* g_emu_mfma = nullptr;

namespace {
struct TA { const std::function<void()>* fn; int tid, bx, by; };
void* tmain(void* p) {
    TA* a = static_cast<TA*>(p);
    threadIdx.x = a->tid; blockIdx.x = a->bx; blockIdx.y = a->by;
    (*a->fn)();
    return nullptr;
}
void launch(int gx, int gy, int nthr, const std::function<void()>& fn) {
    static EmuBlock block;
    static EmuMfmaBuf mf;
    static std::vector<unsigned char> lds(200000 + 64);
    g_emu_block = &block; g_emu_mfma = &mf;
    g_emu_dynamic_lds = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~uintptr_t(15));
    blockDim.x = nthr; gridDim.x = gx; gridDim.y = gy;
    block.block_barrier.init(nthr);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
    for (int by = 0; by < gy; ++by)
        for (int bx = 0; bx < gx; ++bx) {
            std::memset(lds.data(), 0xFF, lds.size());
            std::vector<TA> ta(nthr);
            for (int t = 0; t < nthr; ++t) ta[t] = TA{&fn, t, bx, by};
            emu_run_threads(nthr, tmain, ta.data(), sizeof(ta[0]), 1 << 20);
        }
}
}  // namespace

extern "C" int emu_clip_forward(const float* blob, long n_floats, const float* crops, int n, const int* rows, float* feats) {
    using namespace bm;
    const int32_t* h = reinterpret_cast<const int32_t*>(blob);
    if (h[0] != 0x434C5031) return -1;
    const int D = h[1], layers = h[2], heads = h[3], patch = h[4], gh = h[5], gw = h[6], E = h[7], H = h[8], W = h[9];
    const int T = gh * gw + 1, K0 = patch * patch * 3;
    const float* body = blob + 16;
    long off = 0;
    auto take = [&](long k) { const long o = off; off += k; return o; };
    const long o_conv = take((long)D * K0), o_cls = take(D), o_pos = take((long)T * D), o_lpw = take(D), o_lpb = take(D);
    struct LO { long ln1w, ln1b, qkvw, qkvb, outw, outb, ln2w, ln2b, fcw, fcb, pw, pb; };
    std::vector<LO> L(layers);
    for (auto& l : L) {
        l.ln1w = take(D); l.ln1b = take(D); l.qkvw = take(3L * D * D); l.qkvb = take(3 * D); l.outw = take((long)D * D); l.outb = take(D);
        l.ln2w = take(D); l.ln2b = take(D); l.fcw = take(4L * D * D); l.fcb = take(4 * D); l.pw = take(4L * D * D); l.pb = take(D);
    }
    const long o_postw = take(D), o_postb = take(D), o_proj = take((long)D * E), o_bs = take(D), o_bb = take(D), o_ps = take(E), o_pb = take(E);
    if (n_floats != 16 + off) return -2;
    std::vector<_Float16> w16((size_t)off);
    for (long i = 0; i < off; ++i) { const uint16_t b = f32_to_f16_bits(body[i]); std::memcpy(&w16[i], &b, 2); }
    const long R = (long)n * T, RP = (long)n * (T - 1);
    std::vector<_Float16> patches((size_t)RP * K0), h16((size_t)R * D), qkv((size_t)R * 3 * D), mlp((size_t)R * 4 * D);
    std::vector<float> pe((size_t)RP * D), x((size_t)R * D);
    const float* W32 = body; const _Float16* W16 = w16.data();
    { _Float16* o = patches.data(); launch(4, 1, 256, [=]() { k_clip_patches(crops, o, n, H, W, patch, gh, gw); }); }
    auto gemm = [&](int epi, const _Float16* X, const _Float16* Wt, const float* bias, void* C, long M, int N, int K) {
        const int gx = (int)((M + GEMM_BM - 1) / GEMM_BM), gy = N / GEMM_BN;
        if (K % 64 == 0) {            // the engine's choice (clip_engine.hpp): global_load_lds tiles for the large shapes
            if (epi == 0) launch(gx * gy, 1, 256, [=]() { k_gemm_f16_glds<0, 64>(X, Wt, bias, C, nullptr, (int)M, N, K, 0, GemmExt{}); });
            if (epi == 1) launch(gx * gy, 1, 256, [=]() { k_gemm_f16_glds<1, 64>(X, Wt, bias, C, nullptr, (int)M, N, K, 0, GemmExt{}); });
            if (epi == 2) launch(gx * gy, 1, 256, [=]() { k_gemm_f16_glds<2, 64>(X, Wt, bias, C, nullptr, (int)M, N, K, 0, GemmExt{}); });
            if (epi == 3) launch(gx * gy, 1, 256, [=]() { k_gemm_f16_glds<3, 64>(X, Wt, bias, C, nullptr, (int)M, N, K, 0, GemmExt{}); });
            return;
        }
        if (epi == 0) launch(gx * gy, 1, 256, [=]() { k_gemm_f16<0>(X, Wt, bias, C, nullptr, (int)M, N, K, 0); });
        if (epi == 1) launch(gx * gy, 1, 256, [=]() { k_gemm_f16<1>(X, Wt, bias, C, nullptr, (int)M, N, K, 0); });
        if (epi == 2) launch(gx * gy, 1, 256, [=]() { k_gemm_f16<2>(X, Wt, bias, C, nullptr, (int)M, N, K, 0); });
        if (epi == 3) launch(gx * gy, 1, 256, [=]() { k_gemm_f16<3>(X, Wt, bias, C, nullptr, (int)M, N, K, 0); });
    };
    gemm(3, patches.data(), W16 + o_conv, nullptr, pe.data(), RP, D, K0);
    { const float* p = pe.data(); float* xo = x.data();
      launch((int)((R + 3) / 4), 1, 256, [=]() { k_clip_tokens_lnpre(p, W32 + o_cls, W32 + o_pos, W32 + o_lpw, W32 + o_lpb, xo, R, T, D); }); }
    for (const LO& l : L) {
        { const float* xi = x.data(); _Float16* ho = h16.data();
          launch((int)((R + 3) / 4), 1, 256, [=]() { k_clip_layernorm_f16(xi, W32 + l.ln1w, W32 + l.ln1b, ho, R, D); }); }
        gemm(0, h16.data(), W16 + l.qkvw, W32 + l.qkvb, qkv.data(), R, 3 * D, D);
        { const _Float16* q = qkv.data(); _Float16* ho = h16.data();
          launch(n * heads, 1, 256, [=]() { k_clip_attention(q, ho, T, D, heads); }); }
        gemm(2, h16.data(), W16 + l.outw, W32 + l.outb, x.data(), R, D, D);
        { const float* xi = x.data(); _Float16* ho = h16.data();
          launch((int)((R + 3) / 4), 1, 256, [=]() { k_clip_layernorm_f16(xi, W32 + l.ln2w, W32 + l.ln2b, ho, R, D); }); }
        gemm(1, h16.data(), W16 + l.fcw, W32 + l.fcb, mlp.data(), R, 4 * D, D);
        gemm(2, mlp.data(), W16 + l.pw, W32 + l.pb, x.data(), R, D, 4 * D);
    }
    { const float* xi = x.data();
      launch(n, 1, 256, [=]() { k_clip_head(xi, W32 + o_postw, W32 + o_postb, W32 + o_proj, W32 + o_bs, W32 + o_bb, W32 + o_ps, W32 + o_pb, feats, rows, T, D, E); }); }
    return 0;
}

// the two attention kernels on the same q | k | v rows (T = 33 tokens: three key / query tiles, the last with ONE valid key, as 129 tokens have):
// k_clip_attention (run-time T) -> out_a, k_clip_attention_t<33> (compile-time T, ds_read_b64_tr_b16 fragments) -> out_b
extern "C" int emu_clip_attention_pair(const uint16_t* qkv_bits, uint16_t* out_a, uint16_t* out_b, int n, int heads, int D) {
    using namespace bm;
    constexpr int T = 33;
    const _Float16* q = reinterpret_cast<const _Float16*>(qkv_bits);
    _Float16* a = reinterpret_cast<_Float16*>(out_a);
    _Float16* b = reinterpret_cast<_Float16*>(out_b);
    launch(n * heads, 1, 256, [=]() { k_clip_attention(q, a, T, D, heads); });
    launch(n * heads, 1, 192, [=]() { k_clip_attention_t<T>(q, b, D, heads); });
    return 0;
}
