// TEST-ONLY harness: runs the fp32-grade wide-OSNet kernels (boxmot_amd/csrc/osnet_wide_hp_kernels.hpp, the device source unchanged, in
// the launch order of wide_hp_forward -- the same function the engine calls) on CPU threads with the emulated MFMA of hip_shim.hpp.
// A reduced architecture keeps it to seconds; osnet_x1_0 itself runs on the GPU (tests/test_gpu_long_parity.py).
#include "hip_shim.hpp"

#include <cstdlib>
#include <functional>
#include <vector>

#include "../../boxmot_amd/csrc/osnet_wide_hp.hpp"

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
struct Launcher {
    template <class K, class... A>
    void operator()(K kernel, int gx, int gy, int nthr, int lds_bytes, A... args) {
        const std::function<void()> fn = [=]() { kernel(args...); };
        static EmuBlock block;
        static EmuMfmaBuf mf;
        static std::vector<unsigned char> lds(200000 + 64);
        if (lds_bytes > 160 * 1024) std::abort();
        g_emu_block = &block; g_emu_mfma = &mf;
        g_emu_dynamic_lds = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(lds.data()) + 15) & ~uintptr_t(15));
        blockDim.x = nthr; gridDim.x = (unsigned)gx; gridDim.y = (unsigned)gy;
        block.block_barrier.init(nthr);
        for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
        for (int by = 0; by < gy; ++by)
            for (int bx = 0; bx < gx; ++bx) {
                std::memset(lds.data(), 0xFF, lds.size());          // LDS poison: nothing may rely on zeroed shared memory
                std::vector<TA> ta(nthr);
                for (int t = 0; t < nthr; ++t) ta[t] = TA{&fn, t, bx, by};
                emu_run_threads(nthr, tmain, ta.data(), sizeof(ta[0]), 1 << 20);
            }
    }
};
}  // namespace

// crops: normalised fp32 NHWC [n][256][128][3]; feats [rows][feat]; stages (optional): fp32 copies of the six block outputs
extern "C" int emu_wide_hp_forward(const float* blob, long n_floats, const float* crops, int n, const int* rows, float* feats, float** stages) {
    using namespace bm;
    const int32_t* h = reinterpret_cast<const int32_t*>(blob);
    if (h[0] != REID_MAGIC) return -1;
    const int ch[4] = {h[1], h[2], h[3], h[4]};
    const OsnetLayout L0 = make_osnet_layout(ch, h[5]);
    if (n_floats != REID_HEADER_INTS + L0.total) return -2;
    const float* W32 = blob + REID_HEADER_INTS;
    // widths the family does not take as they are run as their zero-padded copy, exactly as the engine does (reid_engine.hpp)
    OsnetLayout L = L0;
    std::vector<float> padded;
    if (!wide_hp_supports(L0)) {
        int cp[4];
        if (!osnet_padded_channels(L0, cp)) return -3;
        L = osnet_padded_layout(L0, cp);
        padded = osnet_pad_weights(W32, L0, L);
        W32 = padded.data();
    }
    if (!wide_hp_supports(L)) return -3;
    const WideHpPack pk = wide_pack_hp(W32, L);
    std::vector<unsigned char> wp(pk.data.size() + 16);
    unsigned char* wpa = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(wp.data()) + 15) & ~uintptr_t(15));
    std::memcpy(wpa, pk.data.data(), pk.data.size());
    const size_t N = (size_t)n;
    // the stem's input: (hi, lo) fp16 RGBX planes with a 3-pixel zero border (what k_crop_resize_rgbx_hl writes on the device)
    std::vector<_Float16> crops_h(N * WSTEM_ROWS * WSTEM_COLS * 4, (_Float16)0.f), crops_l(crops_h.size(), (_Float16)0.f);
    for (size_t i = 0; i < N; ++i)
        for (int y = 0; y < REID_IN_H; ++y)
            for (int x = 0; x < REID_IN_W; ++x)
                for (int c = 0; c < 3; ++c) {
                    const float v = crops[((i * REID_IN_H + y) * REID_IN_W + x) * 3 + c];
                    const _Float16 hi = (_Float16)v;
                    const size_t o = ((i * WSTEM_ROWS + y + 3) * WSTEM_COLS + x + 3) * 4 + c;
                    crops_h[o] = hi;
                    crops_l[o] = (_Float16)(v - (float)hi);
                }
    const size_t act = N * wide_hp_act_halves(L), mid = N * wide_hp_mid_halves(L);
    std::vector<_Float16> a_h(act), a_l(act), b_h(act), b_l(act), x1_h(mid), x1_l(mid), y_h(4 * mid), y_l(4 * mid), x2_h(mid), x2_l(mid);
    std::vector<_Float16> gap_h(N * L.c[3]), gap_l(N * L.c[3]);
    std::vector<float> gap_part(4 * N * WIDE_HP_MAX_BANDS * 128), fc32(N * L.feat);
    WideHpBuffers B;
    B.crops_h = crops_h.data(); B.crops_l = crops_l.data();
    B.a_h = a_h.data(); B.a_l = a_l.data(); B.b_h = b_h.data(); B.b_l = b_l.data();
    B.x1_h = x1_h.data(); B.x1_l = x1_l.data(); B.y_h = y_h.data(); B.y_l = y_l.data(); B.x2_h = x2_h.data(); B.x2_l = x2_l.data();
    B.gap_part = gap_part.data(); B.gap_h = gap_h.data(); B.gap_l = gap_l.data(); B.fc32 = fc32.data();
    Launcher launch;
    try {
        wide_hp_forward(launch, L, pk, wpa, W32, B, n, feats, rows, [&](int b, const _Float16* oh, const _Float16* ol, long n_pix, int cout) {
            if (stages && stages[b])
                for (long px = 0; px < n_pix; ++px) {          // block outputs are stored in the family's paired channel order
                    const int c_net = L0.block[b].cout;        // the network's own channels; a padded copy's extra ones must be 0
                    for (int c = 0; c < cout; ++c) {
                        const long i = px * cout + hp_paired_pos(c);
                        const float v = (float)oh[i] + (float)ol[i];
                        if (c < c_net) stages[b][px * c_net + c] = v;
                        else if (v != 0.f) throw std::runtime_error("padded channel is not zero");
                    }
                }
        });
    } catch (const std::exception&) {
        return -4;
    }
    return 0;
}
