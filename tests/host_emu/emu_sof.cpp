// TEST-ONLY harness: runs the sparse-optical-flow camera-motion kernels (boxmot_amd/csrc/cmc_sof.hpp, the device source unchanged,
// including its kernel sequence sof_frame) on CPU threads for one stream, frame after frame.
#include "hip_shim.hpp"

#include <cstdlib>
#include <functional>
#include <vector>

#include "../../boxmot_amd/csrc/cmc_sof.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuDim3 gridDim;
EmuBlock* g_emu_block = nullptr;
unsigned char* g_emu_dynamic_lds = nullptr;

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
    void operator()(K kernel, int gx, int gy, int nthr, A... args) {
        const std::function<void()> fn = [=]() { kernel(args...); };
        static EmuBlock block;
        g_emu_block = &block;
        blockDim.x = nthr; gridDim.x = gx; gridDim.y = gy;
        block.block_barrier.init(nthr);
        for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
        for (int by = 0; by < gy; ++by)
            for (int bx = 0; bx < gx; ++bx) {
                std::vector<TA> ta(nthr);
                for (int t = 0; t < nthr; ++t) ta[t] = TA{&fn, t, bx, by};
                emu_run_threads(nthr, tmain, ta.data(), sizeof(ta[0]), 1 << 19);
            }
    }
};

struct EmuSof {
    int rows, cols;
    bm::SofLevels lv;
    bm::SofParams prm;
    std::vector<uint8_t> pyr_prev, pyr_cur, mask, status;
    std::vector<short> der_prev, der_cur;
    std::vector<float> eig, prev_kps, new_kps, next_pts, valid_to;
    std::vector<int> cand;
    bm::SofState st{};
    double warp[6];
};
}  // namespace

extern "C" void* emu_sof_create(int rows, int cols, double scale, int min_inliers, double min_inlier_ratio, double thresh) {
    EmuSof* h = new EmuSof();
    h->rows = rows; h->cols = cols;
    const int w = (int)std::nearbyint(cols * scale), hh = (int)std::nearbyint(rows * scale);
    h->lv = bm::sof_levels(hh, w);
    h->prm = bm::SofParams{scale, min_inliers, min_inlier_ratio, thresh};
    const size_t T = h->lv.total, P = (size_t)hh * w, K = bm::SOF_MAX_CORNERS;
    h->pyr_prev.assign(T, 0); h->pyr_cur.assign(T, 0); h->der_prev.assign(2 * T, 0); h->der_cur.assign(2 * T, 0);
    h->eig.assign(P, 0); h->mask.assign(P, 0); h->cand.assign(P, 0);
    h->prev_kps.assign(2 * K, 0); h->new_kps.assign(2 * K, 0); h->next_pts.assign(2 * K, 0); h->valid_to.assign(2 * K, 0); h->status.assign(K, 0);
    return h;
}
extern "C" void emu_sof_destroy(void* p) { delete static_cast<EmuSof*>(p); }

// frame: BGR uint8 [rows][cols][3]; dets: [n_dets][det_stride] fp32 or null; out_state: the 12 ints of SofState after the frame
extern "C" int emu_sof_apply(void* p, const uint8_t* frame, const float* dets, int n_dets, int det_stride, double* out_warp6, int* out_state12) {
    EmuSof* h = static_cast<EmuSof*>(p);
    bm::SofBuffers B{h->pyr_prev.data(), h->pyr_cur.data(), h->der_prev.data(), h->der_cur.data(), h->eig.data(), h->mask.data(), h->cand.data(),
                     h->prev_kps.data(), h->new_kps.data(), h->next_pts.data(), h->valid_to.data(), h->status.data(), &h->st, h->warp};
    const uint8_t* fr[1] = {frame};
    Launcher L;
    bm::sof_frame(L, B, h->lv, 0, 1, fr, h->rows, h->cols, dets, dets ? &n_dets : nullptr, n_dets > 0 ? n_dets : 1, det_stride, h->prm);
    for (int i = 0; i < 6; ++i) out_warp6[i] = h->warp[i];
    if (out_state12) std::memcpy(out_state12, &h->st, sizeof(bm::SofState));
    return 0;
}

// which: 0 keypoints the next frame tracks (n = n_prev), 1 LK results of the last frame (x, y per previous keypoint), 2 status bytes as floats
extern "C" int emu_sof_points(void* p, int which, float* out, int cap) {
    EmuSof* h = static_cast<EmuSof*>(p);
    const int n = h->st.n_prev < cap ? h->st.n_prev : cap;
    if (which == 0) std::memcpy(out, h->prev_kps.data(), (size_t)n * 8);
    else if (which == 1) std::memcpy(out, h->next_pts.data(), (size_t)cap * 8);
    else for (int i = 0; i < cap; ++i) out[i] = h->status[i];
    return h->st.n_prev;
}
extern "C" int emu_sof_small(void* p, uint8_t* out) {       // level 0 of the previous (= last committed) frame
    EmuSof* h = static_cast<EmuSof*>(p);
    std::memcpy(out, h->pyr_prev.data(), (size_t)h->lv.h[0] * h->lv.w[0]);
    return h->lv.n;
}
