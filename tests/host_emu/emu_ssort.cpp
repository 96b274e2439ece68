// TEST-ONLY harness: runs the StrongSORT device source (bm::ss_bank_distance_block + bm::ss_step_stream, unchanged)
// on CPU threads through hip_shim.hpp.  See hip_shim.hpp for scope and limits.
#include "hip_shim.hpp"

#include <cstdlib>
#include <vector>

#include "../../boxmot_amd/csrc/strongsort_step.hpp"

thread_local EmuDim3 threadIdx;
thread_local EmuDim3 blockIdx;
EmuDim3 blockDim;
EmuBlock* g_emu_block = nullptr;
EmuMfmaBuf* g_emu_mfma = nullptr;

namespace {

#ifndef EMU_NTHR
#define EMU_NTHR 64
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
    bm::SsStepArgs args{};
    HostAlloc alloc;
    int cap, nd, dim;
    float* dets; int* n_dets; float* embs; float* out; int* out_n; double* warp;
    EmuBlock block;
};

struct ThreadArg { Emu* e; int tid; int mode; int t; };
int* g_s_int; double* g_s_dbl; unsigned char* g_dyn;
float (*g_sA)[bm::SS_TILE + 1]; float (*g_sB)[bm::SS_TILE + 1]; float (*g_sMin)[bm::SS_TILE];
float* g_mfma_lds;

const double* g_lsa_cost; int g_lsa_nr, g_lsa_nc; int* g_lsa_out;
bm::SSV* g_set_view; int g_set_na, g_set_nb, g_set_nt, g_set_big, g_set_n;

void* thread_main(void* p) {
    ThreadArg* ta = static_cast<ThreadArg*>(p);
    threadIdx.x = ta->tid;
    blockIdx.x = 0;
    if (ta->mode == 2) {
        const bm::Ctx c = bm::make_ctx(g_s_int, g_s_dbl);
        const int n = g_lsa_nr > g_lsa_nc ? g_lsa_nr : g_lsa_nc;
        const bm::LsaLds l = bm::ss_carve_lsa(g_dyn, n);
        const double* cm = g_lsa_cost; const int nc = g_lsa_nc;
        bm::lsa_scipy(c, l, g_lsa_nr, g_lsa_nc, [&](int r, int q) { return cm[r * nc + q]; }, g_lsa_out);
        return nullptr;
    }
    if (ta->mode == 5) {
        const bm::Ctx c = bm::make_ctx(g_s_int, g_s_dbl);
        const int n = bm::ss_unmatched_in_set_order(c, *g_set_view, g_set_na, g_set_nb, g_set_nt, g_set_big, g_dyn);
        if (ta->tid == 0) g_set_n = n;
        return nullptr;
    }
    if (ta->mode == 3) { bm::ss_det_norm_block<NTHR>(ta->e->args, 0); return nullptr; }
    if (ta->mode == 4) { bm::ss_bank_distance_block_mfma<NTHR>(ta->e->args, 0, ta->t, g_mfma_lds); return nullptr; }
    if (ta->mode == 0) bm::ss_bank_distance_block<NTHR>(ta->e->args, 0, ta->t, g_sA, g_sB, g_sMin);
    else bm::ss_step_stream<NTHR>(ta->e->args, 0, g_s_int, g_s_dbl, g_dyn);
    return nullptr;
}

void run_block(Emu* e, int mode, int t) {
    std::vector<ThreadArg> ta(NTHR);
    for (int k = 0; k < NTHR; ++k) ta[k] = ThreadArg{e, k, mode, t};
    emu_run_threads(NTHR, thread_main, ta.data(), sizeof(ta[0]), 1 << 20);
}

}  // namespace

extern "C" {

// cd: min_conf, max_cos_dist, max_iou_dist, mc_lambda, ema_alpha; ci: max_age, n_init, budget
void* emu_ss_create(const double* cd, const int* ci, int cap, int nd, int dim) {
    Emu* e = new Emu();
    e->cap = cap; e->nd = nd; e->dim = dim;
    bm::SsConfigDev& c = e->args.cfg;
    c.min_conf = cd[0]; c.max_cos_dist = cd[1]; c.max_iou_dist = cd[2]; c.mc_lambda = cd[3];
    c.ema_alpha_f32 = (float)cd[4]; c.one_minus_alpha_f32 = (float)(1 - cd[4]);
    c.max_age = ci[0]; c.n_init = ci[1]; c.budget = ci[2];
    bm::SsSizes z{1, cap, nd, dim, ci[2]};
    bm::ss_allocate(e->args, z, e->alloc);
    e->args.st.next_id[0] = 1;
    e->dets = e->alloc.get<float>((size_t)nd * bm::DET_COLS);
    e->n_dets = e->alloc.get<int>(1);
    e->embs = e->alloc.get<float>((size_t)nd * dim);
    e->out = e->alloc.get<float>((size_t)cap * bm::OUT_COLS);
    e->out_n = e->alloc.get<int>(1);
    e->warp = e->alloc.get<double>(6);
    e->args.dets = e->dets; e->args.n_dets = e->n_dets; e->args.embs = e->embs; e->args.warp = nullptr;
    e->args.out = e->out; e->args.out_n = e->out_n; e->args.stream_base = 0;
    e->block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) e->block.wave_barrier[w].init(EMU_WAVE);
    return e;
}

// parity debugging: SsStepArgs::dbg_cost / dbg_shape (what boxmot_hip_strongsort_debug_costs_enable + _debug_costs do in the library)
void emu_ss_debug_costs_enable(void* h) {
    Emu* e = static_cast<Emu*>(h);
    if (e->args.dbg_cost) return;
    const size_t big = e->cap > e->nd ? e->cap : e->nd;
    e->args.dbg_cost = e->alloc.get<double>(4 * big * big);
    e->args.dbg_shape = e->alloc.get<int>(4);
}
// out (tracks, detections) row-major; returns rows * cols, -1 when not enabled
int emu_ss_debug_costs(void* h, int stage, int plane, double* out, int* rows, int* cols) {
    Emu* e = static_cast<Emu*>(h);
    if (!e->args.dbg_cost) return -1;
    const size_t big = e->cap > e->nd ? e->cap : e->nd;
    const int R = e->args.dbg_shape[stage * 2], C = e->args.dbg_shape[stage * 2 + 1];
    const double* m = e->args.dbg_cost + (size_t)(stage * 2 + plane) * big * big;
    for (int r = 0; r < R; ++r) std::memcpy(out + (size_t)r * C, m + (size_t)r * big, (size_t)C * 8);
    *rows = R; *cols = C;
    return R * C;
}

void emu_ss_destroy(void* h) {
    Emu* e = static_cast<Emu*>(h);
    for (void* p : e->alloc.owned) std::free(p);
    delete e;
}

int emu_ss_update(void* h, const float* dets, int n, const float* embs, const double* warp, float* out, int* out_n) {
    Emu* e = static_cast<Emu*>(h);
    if (n > e->nd) return -1;
    std::memcpy(e->dets, dets, (size_t)n * bm::DET_COLS * 4);
    if (embs) std::memcpy(e->embs, embs, (size_t)n * e->dim * 4);
    e->n_dets[0] = n;
    if (warp) { std::memcpy(e->warp, warp, 48); e->args.warp = e->warp; } else e->args.warp = nullptr;
    static int s_int[bm::MAX_WAVES + 1];
    static double s_dbl[bm::MAX_WAVES];
    static std::vector<double> dyn;
    static float sA[bm::SS_KC][bm::SS_TILE + 1], sB[bm::SS_KC][bm::SS_TILE + 1], sMin[16][bm::SS_TILE];
    const int big = e->cap > e->nd ? e->cap : e->nd;
    dyn.assign((size_t)bm::ss_lsa_lds_bytes(big) / 8 + 2, 0.0);
    g_s_int = s_int; g_s_dbl = s_dbl; g_dyn = reinterpret_cast<unsigned char*>(dyn.data());
    g_sA = sA; g_sB = sB; g_sMin = sMin;
    g_emu_block = &e->block;
    blockDim.x = NTHR;
    const int nt = e->args.st.n_tracks[0];
    run_block(e, 3, 0);
    // both bank-distance kernels on every confirmed track: the fp32-MFMA one (the shipped kernel; here the k-ordered fmaf
    // chain the instruction is) must reproduce the scalar-FMA one bit for bit
    static EmuMfmaBuf mf;
    static std::vector<float> mlds;
    mlds.assign((size_t)bm::SS_MFMA_LDS_FLOATS(NTHR) + 4, -12345.f);
    g_emu_mfma = &mf; g_mfma_lds = mlds.data();
    std::vector<float> row(e->nd);
    for (int t = 0; t < nt; ++t)
        if (e->args.st.state[e->args.st.list[t]] == bm::SS_CONFIRMED) {
            float* app = e->args.sc.app + (size_t)t * e->nd;
            run_block(e, 0, t);
            std::memcpy(row.data(), app, (size_t)n * 4);
            if (e->args.cfg.budget <= bm::SS_MT * 16) {
                for (int q = 0; q < n; ++q) app[q] = -7.f;
                run_block(e, 4, t);
                if (std::memcmp(row.data(), app, (size_t)n * 4) != 0) return -77;
            }
        }
    run_block(e, 1, 0);
    *out_n = e->out_n[0];
    std::memcpy(out, e->out, (size_t)e->out_n[0] * bm::OUT_COLS * 4);
    return e->args.st.status[0];
}

// debugging aid: scratch lists of the last step
void emu_ss_debug(void* h, int* rows_b, int* cols_b, int* m_trk, int* m_det, int* un_d, double* cost, int n) {
    Emu* e = static_cast<Emu*>(h);
    const bm::SsScratch& sc = e->args.sc;
    for (int k = 0; k < n; ++k) { rows_b[k] = sc.rows_b[k]; cols_b[k] = sc.cols_b[k]; m_trk[k] = sc.m_trk[k]; m_det[k] = sc.m_det[k]; un_d[k] = sc.un_d[k]; }
    const int big = e->cap > e->nd ? e->cap : e->nd;
    for (int r = 0; r < n; ++r) for (int q = 0; q < n; ++q) cost[r * n + q] = sc.cost[(size_t)r * big + q];
}

void emu_ss_app(void* h, float* app, int rows, int cols) {
    Emu* e = static_cast<Emu*>(h);
    for (int r = 0; r < rows; ++r) for (int q = 0; q < cols; ++q) app[r * cols + q] = e->args.sc.app[(size_t)r * e->nd + q];
}

void emu_ss_bank(void* h, int pos, int k, float* out) {
    Emu* e = static_cast<Emu*>(h);
    const int sl = e->args.st.list[pos];
    std::memcpy(out, e->args.st.bank + ((size_t)sl * e->args.st.budget + k) * e->dim, (size_t)e->dim * 4);
}

// the device assignment solver alone (nr <= nc), for comparison with scipy.optimize.linear_sum_assignment
void emu_lsa(const double* cost, int nr, int nc, int* col_of) {
    static Emu e;
    static int s_int[bm::MAX_WAVES + 1];
    static double s_dbl[bm::MAX_WAVES];
    static std::vector<double> dyn;
    dyn.assign((size_t)bm::ss_lsa_lds_bytes(nr > nc ? nr : nc) / 8 + 2, 0.0);
    g_s_int = s_int; g_s_dbl = s_dbl; g_dyn = reinterpret_cast<unsigned char*>(dyn.data());
    g_lsa_cost = cost; g_lsa_nr = nr; g_lsa_nc = nc; g_lsa_out = col_of;
    e.block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) e.block.wave_barrier[w].init(EMU_WAVE);
    g_emu_block = &e.block;
    blockDim.x = NTHR;
    run_block(&e, 2, 0);
}

// list(set(a) - set(b)) in CPython's iteration order as the frame step computes it (a ascending track positions < nt, b a subset);
// lds_big sizes the LDS area like the kernel does (max(cap, max_dets)); returns the length
int emu_set_order(const int* a, int na, const int* b, int nb, int nt, int cap, int lds_big, int* out) {
    static Emu e;
    static int s_int[bm::MAX_WAVES + 1];
    static double s_dbl[bm::MAX_WAVES];
    static std::vector<double> dyn;
    dyn.assign((size_t)bm::ss_lsa_lds_bytes(lds_big) / 8 + 2, 0.0);
    g_s_int = s_int; g_s_dbl = s_dbl; g_dyn = reinterpret_cast<unsigned char*>(dyn.data());
    std::vector<int> rows_a(a, a + na), m_trk(b, b + nb), flag(cap + 1), tmp_a(cap + 1), tmp_b(cap + 1), pyset(3 * (size_t)bm::pyset_capacity(cap));
    int status = 0;
    bm::SSV v{};
    v.cap = cap; v.rows_a = rows_a.data(); v.m_trk = m_trk.data(); v.flag_t = flag.data(); v.tmp_a = tmp_a.data(); v.tmp_b = tmp_b.data();
    v.pyset = pyset.data(); v.status = &status;
    g_set_view = &v; g_set_na = na; g_set_nb = nb; g_set_nt = nt; g_set_big = lds_big; g_set_n = -1;
    e.block.block_barrier.init(NTHR);
    for (int w = 0; w < EMU_MAX_WAVES; ++w) e.block.wave_barrier[w].init(EMU_WAVE);
    g_emu_block = &e.block;
    blockDim.x = NTHR;
    run_block(&e, 5, 0);
    if (status != 0) return -1;
    std::memcpy(out, tmp_a.data(), (size_t)g_set_n * 4);
    return g_set_n;
}

// tracks in list order: ints (rows,6) = id, state, hits, age, time_since_update, bank size; kf (rows,72); feat (rows,dim)
int emu_ss_dump(void* h, int* ints, double* kf, float* feat, int* counters) {
    Emu* e = static_cast<Emu*>(h);
    const bm::SsState& st = e->args.st;
    const int n = st.n_tracks[0];
    for (int r = 0; r < n; ++r) {
        const int sl = st.list[r];
        int* o = ints + r * 6;
        o[0] = st.id[sl]; o[1] = st.state[sl]; o[2] = st.hits[sl]; o[3] = st.age[sl]; o[4] = st.tsu[sl];
        o[5] = st.bank_n[sl] < st.budget ? st.bank_n[sl] : st.budget;
        std::memcpy(kf + (size_t)r * bm::KF_STRIDE, st.kf + (size_t)sl * bm::KF_STRIDE, bm::KF_STRIDE * 8);
        std::memcpy(feat + (size_t)r * e->dim, st.feat + (size_t)sl * e->dim, (size_t)e->dim * 4);
    }
    counters[0] = st.frame_count[0]; counters[1] = st.next_id[0];
    return n;
}

}  // extern "C"
