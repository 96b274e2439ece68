// TEST-ONLY harness: runs bm::docs_step_stream (the DeepOCSORT device source, unchanged) on CPU threads through
// hip_shim.hpp.  See hip_shim.hpp for scope and limits.
#include "hip_shim.hpp"

#include <cstdlib>
#include <vector>

#include "../../boxmot_amd/csrc/deepocsort_step.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuBlock* g_emu_block = nullptr;

namespace {

#ifndef EMU_NTHR
#define EMU_NTHR 64
#endif
constexpr int NTHR = EMU_NTHR;
#ifndef EMU_OBB
#define EMU_OBB 0
#endif
#if EMU_OBB          // the oriented copy of the step (bm::obb): 7-column detections, 9-column rows, 90 doubles of filter state
namespace geo = bm::obb;
#else
namespace geo = bm;
#endif

struct HostAlloc {
    std::vector<void*> owned;
    template <typename T> T* get(size_t n) {
        void* p = std::calloc(n ? n : 1, sizeof(T));
        owned.push_back(p);
        return static_cast<T*>(p);
    }
};

struct Emu {
    bm::DocsStepArgs args{};
    HostAlloc alloc;
    int cap, nd, dim;
    float* dets; int* n_dets; float* embs; float* out; int* out_n; double* warp; int* warp_flag;
    EmuBlock block;
};

struct ThreadArg { Emu* e; int tid; };
int* g_s_int; double* g_s_dbl; unsigned char* g_dyn;

void* thread_main(void* p) {
    ThreadArg* ta = static_cast<ThreadArg*>(p);
    threadIdx.x = ta->tid;
    blockIdx.x = 0;
    geo::docs_step_stream<NTHR>(ta->e->args, 0, g_s_int, g_s_dbl, g_dyn);
    return nullptr;
}

}  // namespace

extern "C" {

// cd: det_thresh, iou_threshold, inertia, w_emb, alpha_fixed, aw_param, q_xy, q_s, min_conf, asso_diag; ci: max_age, min_hits, delta_t, embedding_off, aw_off, use_byte, asso_mode
void* emu_docs_create(const double* cd, const int* ci, int cap, int nd, int dim) {
    Emu* e = new Emu();
    e->cap = cap; e->nd = nd; e->dim = dim;
    bm::DocsConfigDev& c = e->args.cfg;
    c.det_thresh = cd[0]; c.det_thresh_f32 = (float)cd[0]; c.iou_threshold = cd[1]; c.inertia = cd[2]; c.w_emb = cd[3];
    c.alpha_fixed = cd[4]; c.aw_param = cd[5]; c.q_xy = cd[6]; c.q_s = cd[7];
    c.max_age = ci[0]; c.min_hits = ci[1]; c.delta_t = ci[2]; c.embedding_off = ci[3]; c.aw_off = ci[4];
    c.use_byte = ci[5]; c.min_conf_f32 = (float)cd[8];
    c.asso_mode = ci[6]; c.asso_diag = cd[9];
    bm::DocsSizes z{1, cap, nd, dim, EMU_OBB};
    bm::docs_allocate(e->args, z, e->alloc);
    e->dets = e->alloc.get<float>((size_t)nd * geo::DOCS_DET_COLS);
    e->n_dets = e->alloc.get<int>(1);
    e->embs = e->alloc.get<float>((size_t)nd * dim);
    e->out = e->alloc.get<float>((size_t)cap * geo::DOCS_OUT_COLS);
    e->out_n = e->alloc.get<int>(1);
    e->args.dets = e->dets; e->args.n_dets = e->n_dets; e->args.embs = e->embs;
    e->warp = e->alloc.get<double>(6); e->warp_flag = e->alloc.get<int>(1);
    e->args.warp = e->warp; e->args.warp_flag = e->warp_flag;
    e->args.out = e->out; e->args.out_n = e->out_n; e->args.stream_base = 0;
    e->block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) e->block.wave_barrier[w].init(EMU_WAVE);
    return e;
}

// parity debugging: DocsStepArgs::dbg_cost / dbg_shape (what boxmot_hip_deepocsort_debug_costs_enable + _debug_costs do in the library)
void emu_docs_debug_costs_enable(void* h) {
    Emu* e = static_cast<Emu*>(h);
    if (e->args.dbg_cost) return;
    e->args.dbg_cost = e->alloc.get<double>((size_t)bm::DOCS_DBG_PLANES * e->nd * e->cap);
    e->args.dbg_shape = e->alloc.get<int>(4);
}
// out (detections, tracks) row-major; returns the branch (0 no matrix, 1 permutation early-out, 2 solver), -1 when not enabled
int emu_docs_debug_costs(void* h, int plane, double* out, int* rows, int* cols) {
    Emu* e = static_cast<Emu*>(h);
    if (!e->args.dbg_cost) return -1;
    const int R = e->args.dbg_shape[0], C = e->args.dbg_shape[1];
    const double* m = e->args.dbg_cost + (size_t)plane * e->nd * e->cap;
    for (int r = 0; r < R; ++r) std::memcpy(out + (size_t)r * C, m + (size_t)r * e->cap, (size_t)C * 8);
    *rows = R; *cols = C;
    return e->args.dbg_shape[2];
}

void emu_docs_destroy(void* h) {
    Emu* e = static_cast<Emu*>(h);
    for (void* p : e->alloc.owned) std::free(p);
    delete e;
}

int emu_docs_update(void* h, const float* dets, int n, const float* embs, const double* warp, float* out, int* out_n) {
    Emu* e = static_cast<Emu*>(h);
    e->warp_flag[0] = warp != nullptr;
    if (warp) std::memcpy(e->warp, warp, 48);
    if (n > e->nd) return -1;
    std::memcpy(e->dets, dets, (size_t)n * geo::DOCS_DET_COLS * 4);
    if (embs) std::memcpy(e->embs, embs, (size_t)n * e->dim * 4);
    e->n_dets[0] = n;
    static int s_int[bm::MAX_WAVES + 1];
    static double s_dbl[bm::MAX_WAVES];
    static std::vector<double> dyn;
    dyn.assign((size_t)bm::docs_lap_lds_bytes(e->cap, e->nd) / 8 + 2, 0.0);
    g_s_int = s_int; g_s_dbl = s_dbl; g_dyn = reinterpret_cast<unsigned char*>(dyn.data());
    g_emu_block = &e->block;
    blockDim.x = NTHR;
    std::vector<ThreadArg> ta(NTHR);
    for (int t = 0; t < NTHR; ++t) ta[t] = ThreadArg{e, t};
    emu_run_threads(NTHR, thread_main, ta.data(), sizeof(ta[0]), 1 << 20);
    *out_n = e->out_n[0];
    std::memcpy(out, e->out, (size_t)e->out_n[0] * geo::DOCS_OUT_COLS * 4);
    return e->args.st.status[0];
}

// tracks in list order: ints (rows,5) = id, age, time_since_update, hit_streak, observed; kf (rows,72); emb (rows,dim)
int emu_docs_dump(void* h, int* ints, double* kf, double* emb, int* counters) {
    Emu* e = static_cast<Emu*>(h);
    const bm::DocsState& st = e->args.st;
    const int n = st.n_tracks[0];
    for (int r = 0; r < n; ++r) {
        const int sl = st.list[r];
        int* o = ints + r * 5;
        o[0] = st.id[sl]; o[1] = st.age[sl]; o[2] = st.tsu[sl]; o[3] = st.hit_streak[sl]; o[4] = st.observed[sl];
        std::memcpy(kf + (size_t)r * geo::DOCS_KF_STRIDE, st.kf + (size_t)sl * geo::DOCS_KF_STRIDE, geo::DOCS_KF_STRIDE * 8);
        std::memcpy(emb + (size_t)r * e->dim, st.emb + (size_t)sl * e->dim, (size_t)e->dim * 8);
    }
    counters[0] = st.frame_count[0]; counters[1] = st.id_count[0];
    return n;
}


// The Jonker-Volgenant solver of lap_jv.hpp alone on a given n_rows x n_cols matrix (row-major), as one workgroup of NTHR threads:
// x[n_rows] = column of each row, y[n_cols] = row of each column (-1 = unassigned), what lap.lapjv(extend_cost=True[, cost_limit]) returns.
struct JvArg { int nr, nc; const double* cost; int use_limit; double limit; int* x; int* y; int tid; int* ok; };
static void* jv_main(void* p) {
    JvArg* a = static_cast<JvArg*>(p);
    threadIdx.x = a->tid;
    blockIdx.x = 0;
    const bm::Ctx c = bm::make_ctx(g_s_int, g_s_dbl);
    const bm::JvLds L = bm::jv_carve(g_dyn, a->nr + a->nc);
    const double* cm = a->cost;
    const int nc = a->nc;
    const bool ok = bm::lap_jv_extended(c, L, a->nr, a->nc, [&](int i, int j) { return cm[(long)i * nc + j]; }, a->use_limit != 0, a->limit, a->x, a->y);
    if (a->tid == 0) *a->ok = ok ? 1 : 0;
    return nullptr;
}
int emu_lap_jv(int nr, int nc, const double* cost, int use_limit, double limit, int* x, int* y) {
    static int s_int[bm::MAX_WAVES + 1];
    static double s_dbl[bm::MAX_WAVES];
    std::vector<double> dyn((size_t)bm::jv_lds_bytes(nr + nc) / 8 + 2, 0.0);
    g_s_int = s_int; g_s_dbl = s_dbl; g_dyn = reinterpret_cast<unsigned char*>(dyn.data());
    static EmuBlock block;
    g_emu_block = &block;
    blockDim.x = NTHR;
    block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
    int ok = 0;
    std::vector<JvArg> ta(NTHR);
    for (int t = 0; t < NTHR; ++t) ta[t] = JvArg{nr, nc, cost, use_limit, limit, x, y, t, &ok};
    emu_run_threads(NTHR, jv_main, ta.data(), sizeof(ta[0]), 1 << 20);
    return ok;
}

// jv_move alone (wave 0): the TODO-list permutation of one chunk segment against the sequential swap loop it replaces
struct MvArg { int n; int* cols; int base, l0; unsigned long long q; int hi; int tid; int* out_hi; };
static void* mv_main(void* p) {
    MvArg* a = static_cast<MvArg*>(p);
    threadIdx.x = a->tid;
    blockIdx.x = 0;
    if (a->tid >= bm::WAVE) return nullptr;
    bm::JvLds L{};
    L.cols = a->cols;
    const int lane = a->tid;
    int hi = a->hi;
    const int j = a->base + lane < a->n ? a->cols[a->base + lane] : 0;
    g_emu_block->wave_barrier[0].wait();
    bm::jv_move(L, a->base, a->l0, a->q, hi, j, lane);
    if (lane == 0) *a->out_hi = hi;
    return nullptr;
}
int emu_jv_move(int n, int* cols, int base, int l0, unsigned long long q, int hi) {
    static EmuBlock block;
    g_emu_block = &block;
    blockDim.x = NTHR;
    block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) block.wave_barrier[w].init(EMU_WAVE);
    int out = -1;
    std::vector<MvArg> ta(NTHR);
    for (int t = 0; t < NTHR; ++t) ta[t] = MvArg{n, cols, base, l0, q, hi, t, &out};
    emu_run_threads(NTHR, mv_main, ta.data(), sizeof(ta[0]), 1 << 20);
    return out;
}

}  // extern "C"
