// TEST-ONLY harness: runs bm::botsort_step_stream (the device source, unchanged)
// on CPU threads through hip_shim.hpp.  See hip_shim.hpp for scope and limits.
#include "hip_shim.hpp"

#include <cstdlib>
#include <vector>

#include "../../boxmot_amd/csrc/botsort_alloc.hpp"
#include "../../boxmot_amd/csrc/botsort_step.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuBlock* g_emu_block = nullptr;

namespace {

// -DEMU_OBB=1: the oriented-box copy of the step (bm::obb), 7-column detections, 9-column rows, 110-double filter records
#ifndef EMU_OBB
#define EMU_OBB 0
#endif
#if EMU_OBB
namespace geo = bm::obb;
#else
namespace geo = bm;
#endif

#ifndef EMU_NTHR
#define EMU_NTHR 64   // one emulated wavefront per workgroup keeps barrier cost low
#endif
constexpr int NTHR = EMU_NTHR;

struct HostAlloc {
    std::vector<void*> owned;
    template <typename T> T* get(size_t n) {
        void* p = std::calloc(n ? n : 1, sizeof(T));
        owned.push_back(p);
        return static_cast<T*>(p);
    }
};

struct Emu {
    bm::BotSortStepArgs args{};
    HostAlloc alloc;
    int cap, nd, dim;
    float* dets; int* n_dets; float* embs; float* out; int* out_n;
    double* warp; int* warp_flag;
    int* list_sel; int* fc_set;
    EmuBlock block;
};

struct ThreadArg { Emu* e; int tid; };

int* g_s_int; double* g_s_dbl; unsigned char* g_dyn;
float (*g_sA)[bm::COST_KC + 1];      // (COST_KC / COST_TILE / lap_lds_bytes are the same in both copies)
float (*g_sB)[bm::COST_KC + 1];

void* thread_main(void* p) {
    ThreadArg* ta = static_cast<ThreadArg*>(p);
    threadIdx.x = ta->tid;
    blockIdx.x = 0;
    geo::botsort_step_stream<NTHR, true>(ta->e->args, 0, g_s_int, g_s_dbl, g_sA, g_sB, g_dyn);
    return nullptr;
}

}  // namespace

extern "C" {

// n_lists: per-class active lists (per_class=True: basetracker.py:223-263; 1 otherwise)
void* emu_create_lists(const double* cd, const int* ci, int cap, int nd, int dim, int n_lists);
void* emu_create(const double* cd, const int* ci, int cap, int nd, int dim) { return emu_create_lists(cd, ci, cap, nd, dim, 1); }
void* emu_create_lists(const double* cd, const int* ci, int cap, int nd, int dim, int n_lists) {
    Emu* e = new Emu();
    e->cap = cap; e->nd = nd; e->dim = dim;
    e->args.cfg = bm::make_config_dev(cd[0], cd[1], cd[2], cd[3], cd[4], cd[5], cd[6], cd[7], cd[8], ci[0], ci[1], ci[2],
                                      ci[3], ci[4], ci[5]);
    bm::BotSortSizes z{1, cap, nd, dim, n_lists, ci[4] > 0 ? ci[4] : 1, EMU_OBB};
    bm::botsort_allocate(e->args, z, e->alloc);
    e->dets = e->alloc.get<float>((size_t)nd * geo::DET_COLS);
    e->n_dets = e->alloc.get<int>(1);
    e->embs = e->alloc.get<float>((size_t)nd * dim);
    e->out = e->alloc.get<float>((size_t)nd * geo::OUT_COLS);
    e->out_n = e->alloc.get<int>(1);
    e->args.dets = e->dets; e->args.n_dets = e->n_dets; e->args.embs = e->embs;
    e->list_sel = e->alloc.get<int>(1); e->fc_set = e->alloc.get<int>(1);
    e->args.list_sel = nullptr; e->args.frame_count_set = nullptr;
    e->args.out = e->out; e->args.out_n = e->out_n; e->args.stream_base = 0; e->args.phase_clock = nullptr;
    e->warp = e->alloc.get<double>(6);
    e->warp_flag = e->alloc.get<int>(1);
    e->args.warp = e->warp; e->args.warp_flag = e->warp_flag;
    e->block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) e->block.wave_barrier[w].init(EMU_WAVE);
    return e;
}

// parity debugging: BotSortStepArgs::dbg_cost / dbg_shape (what boxmot_hip_botsort_debug_costs_enable + _debug_costs do in the library)
void emu_debug_costs_enable(void* h) {
    Emu* e = static_cast<Emu*>(h);
    if (e->args.dbg_cost) return;
    e->args.dbg_cost = e->alloc.get<double>((size_t)bm::DBG_STAGES * bm::DBG_PLANES * e->nd * e->cap);
    e->args.dbg_shape = e->alloc.get<int>(bm::DBG_STAGES * 2);
}
// out (rows, cols) row-major = (tracks, detections); returns rows * cols
int emu_debug_costs(void* h, int stage, int plane, double* out, int* rows, int* cols) {
    Emu* e = static_cast<Emu*>(h);
    if (!e->args.dbg_cost) return -1;
    const int R = e->args.dbg_shape[stage * 2], C = e->args.dbg_shape[stage * 2 + 1];
    const double* m = e->args.dbg_cost + ((size_t)stage * bm::DBG_PLANES + plane) * e->nd * e->cap;
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) out[(size_t)r * C + c] = m[(size_t)c * e->cap + r];
    *rows = R; *cols = C;
    return R * C;
}

void emu_destroy(void* h) {
    Emu* e = static_cast<Emu*>(h);
    for (void* p : e->alloc.owned) std::free(p);
    delete e;
}

// warp (6 doubles, 2x3 row-major) applied by the next emu_update only
void emu_set_warp(void* h, const double* w) {
    Emu* e = static_cast<Emu*>(h);
    for (int k = 0; k < 6; ++k) e->warp[k] = w[k];
    e->warp_flag[0] = 1;
}

// returns the status word; out rows (n_out, 8).  class_list: the active list of this call; frame_count >= 0 presets the frame counter
// (the per-class fan-out rewinds it for every class), -1 leaves it alone
int emu_update_list(void* h, const float* dets, int n, const float* embs, float* out, int* out_n, int class_list, int frame_count);
int emu_update(void* h, const float* dets, int n, const float* embs, float* out, int* out_n) {
    return emu_update_list(h, dets, n, embs, out, out_n, 0, -1);
}
int emu_update_list(void* h, const float* dets, int n, const float* embs, float* out, int* out_n, int class_list, int frame_count) {
    Emu* e = static_cast<Emu*>(h);
    if (n > e->nd) return -1;
    e->list_sel[0] = class_list; e->fc_set[0] = frame_count;
    e->args.list_sel = e->list_sel;
    e->args.frame_count_set = frame_count >= 0 ? e->fc_set : nullptr;
    std::memcpy(e->dets, dets, (size_t)n * geo::DET_COLS * 4);
    if (embs) std::memcpy(e->embs, embs, (size_t)n * e->dim * 4);
    e->n_dets[0] = n;
    static int s_int[bm::MAX_WAVES + 1];
    static double s_dbl[bm::MAX_WAVES];
    static float sA[bm::COST_TILE][bm::COST_KC + 1];
    static float sB[bm::COST_TILE][bm::COST_KC + 1];
    static std::vector<double> dyn;
    dyn.assign((size_t)bm::lap_lds_bytes(e->cap, e->nd) / 8 + 2, 0.0);
    g_s_int = s_int; g_s_dbl = s_dbl; g_sA = sA; g_sB = sB; g_dyn = reinterpret_cast<unsigned char*>(dyn.data());
    g_emu_block = &e->block;
    blockDim.x = NTHR;
    std::vector<ThreadArg> ta(NTHR);
    for (int t = 0; t < NTHR; ++t) ta[t] = ThreadArg{e, t};
    emu_run_threads(NTHR, thread_main, ta.data(), sizeof(ta[0]), 1 << 20);
    e->warp_flag[0] = 0;
    *out_n = e->out_n[0];
    std::memcpy(out, e->out, (size_t)e->out_n[0] * geo::OUT_COLS * 4);
    return e->args.st.status[0];
}

// block_argmin (block_prims.hpp) over NTHR emulated threads: thread t offers (v[t], idx[t]); idx < 0 = no candidate
namespace {
struct ArgminArg { const double* v; const int* idx; double* out_v; int* out_i; int tid; };
void* argmin_main(void* p) {
    ArgminArg* a = static_cast<ArgminArg*>(p);
    threadIdx.x = a->tid; blockIdx.x = 0;
    const bm::Ctx c = bm::make_ctx(g_s_int, g_s_dbl);
    double ov; int oi;
    bm::block_argmin(c, a->v[a->tid], a->idx[a->tid], ov, oi);
    if (a->tid == 0) { *a->out_v = ov; *a->out_i = oi; }
    return nullptr;
}
}  // namespace
void emu_block_argmin(const double* v, const int* idx, double* out_v, int* out_i) {
    static int s_int[bm::MAX_WAVES + 1];
    static double s_dbl[bm::MAX_WAVES];
    static EmuBlock block;
    g_s_int = s_int; g_s_dbl = s_dbl; g_emu_block = &block;
    blockDim.x = NTHR;
    block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
    std::vector<ArgminArg> ta(NTHR);
    for (int t = 0; t < NTHR; ++t) ta[t] = ArgminArg{v, idx, out_v, out_i, t};
    emu_run_threads(NTHR, argmin_main, ta.data(), sizeof(ta[0]), 1 << 20);
}
int emu_nthr() { return NTHR; }

// which: 0 active, 1 lost.  ints (rows,6), kf (rows,72), smooth (rows,dim), misc (rows,3)
int emu_dump(void* h, int which, int* ints, double* kf, float* smooth, float* misc, int* counters) {
    Emu* e = static_cast<Emu*>(h);
    const bm::BotSortState& st = e->args.st;
    const int n = which == 0 ? st.n_active[0] : st.n_lost[0];
    const int* list = which == 0 ? st.active_list : st.lost_list;
    for (int r = 0; r < n; ++r) {
        const int sl = list[r];
        int* o = ints + r * 6;
        o[0] = st.id[sl]; o[1] = st.state[sl]; o[2] = st.is_activated[sl]; o[3] = st.frame_id[sl];
        o[4] = st.start_frame[sl]; o[5] = st.tracklet_len[sl];
        std::memcpy(kf + (size_t)r * geo::KF_STRIDE, st.kf + (size_t)sl * geo::KF_STRIDE, geo::KF_STRIDE * 8);
        std::memcpy(smooth + (size_t)r * e->dim, st.smooth + (size_t)sl * e->dim, e->dim * 4);
        misc[r * 3] = st.conf[sl]; misc[r * 3 + 1] = st.cls[sl]; misc[r * 3 + 2] = st.det_ind[sl];
    }
    counters[0] = st.frame_count[0]; counters[1] = st.id_count[0]; counters[2] = st.rm_size[0];
    return n;
}

}  // extern "C"
