// StrongSORT frame step for one stream (one workgroup) + the appearance-bank distance kernel, gfx950 / wave64.
//
// Reference path (Python, boxmot 21.0.0):
//   StrongSort._update_impl                         boxmot/trackers/bbox/strongsort/strongsort.py:69-123
//   Tracker.predict / update / _match / _initiate_track   strongsort/sort/tracker.py:63-169
//   Track (camera_update, predict, update, mark_missed)   strongsort/sort/track.py:25-208
//   min_cost_matching / matching_cascade / gate_cost_matrix / NearestNeighborDistanceMetric
//                                                   strongsort/sort/linear_assignment.py:14-353
//   iou / iou_cost                                  strongsort/sort/iou_matching.py:10-87
//   KalmanFilterXYAH                                motion/kalman_filters/xyah.py:8-172 over base.py:234-355, :523-551
//   scipy.optimize.linear_sum_assignment            (SciPy 1.15 rectangular_lsap.cpp: shortest augmenting paths,
//                                                    Crouse's variant) -- restated here incl. its tie rule
//
// Layout: as the other trackers, stream-major arrays, a track lives in a slot, `list` = the order of the
// reference's `tracker.tracks` Python list.  Filter state mean[8] ++ cov[8][8] fp64 per slot (lane l of a
// wavefront <-> cov element (l>>3, l&7)).  Appearance: the track's EMA feature fp32[D] and the per-target
// sample bank of the nearest-neighbour metric, a ring of `budget` fp32[D] vectors per slot.
// fp64 filter / gating / costs with contraction off; the appearance distances are fp32 like the reference's
// float32 matrix product.
#pragma once

#include "block_prims.hpp"
#include "botsort_types.hpp"
#include "kernel_macros.hpp"

namespace bm {

constexpr int SS_TENTATIVE = 1, SS_CONFIRMED = 2, SS_DELETED = 3;      // sort/track.py:20-22
constexpr double SS_INFTY_COST = 1e5;                                    // linear_assignment.py:11
constexpr double SS_CHI2_4 = 9.4877;                                     // matching.py:14-24
constexpr double SS_STD_POS = 1.0 / 20, SS_STD_VEL = 1.0 / 160;          // base.py:60-65
constexpr double SS_INF = 1e300;

struct SsConfigDev {
    double min_conf, max_cos_dist, max_iou_dist, mc_lambda;
    float ema_alpha_f32, one_minus_alpha_f32;       // python floats meet float32 arrays: rounded to fp32 (NEP 50)
    int max_age, n_init, budget;
};

struct SsState {
    int cap, dim, budget;
    int* frame_count; int* next_id; int* n_tracks; int* status;    // [S]
    int* list; int* slot_used;                                      // [S][cap]
    double* kf;           // [S][cap][72]
    float* feat;          // [S][cap][dim]      features[-1]
    float* bank;          // [S][cap][budget][dim]
    float* bank_norm;     // [S][cap][budget]  |sample|, computed when the sample is appended
    int* bank_n;          // [S][cap]   samples ever appended (ring position = bank_n % budget)
    int* id; int* state; int* hits; int* age; int* tsu;
    float* conf; float* cls; float* det_ind;
};

struct SsScratch {
    int max_dets;
    float* app;           // [S][cap][nd]  min cosine distance (list position, detection), from ss_bank_distance
    float* det_norm;      // [S][nd]
    int* keep;            // [S][nd]
    double* det_tlwh;     // [S][nd][4]   per kept detection
    double* det_xyah;     // [S][nd][4]
    double* cost;         // [S][max(cap,nd)][max(cap,nd)]
    double* cost_t;       // same size: the transposed copy the assignment solver scans when there are more tracks than detections
    int* rows_a; int* rows_b; int* cols_b;       // track positions / detection indices of the two stages
    int* un_d; int* tmp_a; int* tmp_b;
    int* m_trk; int* m_det;
    int* row_of; int* col_of;                    // assignment result
    int* flag_t;
    int* pyset;           // [S][3][PYSET_FACTOR * cap]  hash tables of the CPython set emulation (unmatched-track order)
};

struct SsStepArgs {
    SsConfigDev cfg;
    SsState st;
    SsScratch sc;
    const float* dets;        // [S][nd][6]
    const int* n_dets;        // [S]
    const float* embs;        // [S][nd][dim]
    const double* warp;       // [S][6] camera-motion warp (2x3 row-major) or nullptr = identity
    float* out;               // [S][cap][8]
    int* out_n;               // [S]
    int stream_base;
    // Parity debugging (boxmot_hip_strongsort_debug_costs; nullptr = off, the default): the cost matrices of the two
    // min_cost_matching calls of this step (linear_assignment.py:14-79), [S][2 stages][2 planes][big][big] fp64 row-major
    // (tracks x detections, leading dimension big = max(cap, nd)) and their shapes [S][2][2] = (rows, cols).
    // stage 0: confirmed tracks x detections, the gated appearance metric (tracker.py:108-122, linear_assignment.py:145-198);
    // stage 1: the IoU stage (iou_matching.py:49-87).  plane 0: the metric's matrix as returned; plane 1: after the
    // `cost > max_distance -> max_distance + 1e-5` clamp, what linear_sum_assignment is given.
    double* dbg_cost;
    int* dbg_shape;
};

struct SsSizes { int S, cap, nd, dim, budget; };

// Optional phase clocks (tools only; never defined in the product build): thread 0 of workgroup 0 accumulates shader-clock
// cycles per phase of the frame step, read back through boxmot_hip_debug_ss_prof.
#ifdef BM_SS_PROF
__device__ unsigned long long g_ss_prof[16];
#define SS_PROF_DECL() do { if (threadIdx.x == 0 && blockIdx.x == 0) g_ss_prof[15] = clock64(); } while (0)
#define SS_PROF(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned long long t_ = clock64(); g_ss_prof[k] += t_ - g_ss_prof[15]; g_ss_prof[15] = t_; } } while (0)
#define SS_PROF_COUNT(k, n) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_ss_prof[k] += (n); } while (0)
#else
#define SS_PROF_DECL() ((void)0)
#define SS_PROF(k) ((void)0)
#define SS_PROF_COUNT(k, n) ((void)0)
#endif

// ---------------------------------------------------------------------------
// `unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches))` (linear_assignment.py:141): the order of
// that list is the iteration order of a CPython set of small ints, and it decides the row order of the IoU stage
// (tracker.py:142-144) and with it which of several equally good assignments SciPy returns.  Restated from
// CPython 3.10 Objects/setobject.c (open addressing, LINEAR_PROBES = 9, PERTURB_SHIFT = 5, growth x4, the
// copy-and-discard path when len(a) / 4 > len(b)); hash(i) == i for these keys.  Single-threaded, tiny.
// ---------------------------------------------------------------------------
__host__ __device__ inline int pyset_capacity(int cap) { int n = 8; while (n <= 8 * cap) n <<= 1; return n; }
constexpr int PYSET_EMPTY = -1, PYSET_DUMMY = -2;
struct PySetI { int* table; int mask, fill, used, capacity; bool overflow; };

__device__ inline void pyset_init(PySetI& s, int* storage, int capacity) {
    s.table = storage; s.mask = 7; s.fill = 0; s.used = 0; s.capacity = capacity; s.overflow = false;
    for (int i = 0; i < 8; ++i) storage[i] = PYSET_EMPTY;
}
__device__ inline void pyset_insert_clean(int* table, int mask, int key) {
    unsigned perturb = (unsigned)key;
    unsigned i = (unsigned)key & (unsigned)mask;
    while (true) {
        if (table[i] == PYSET_EMPTY) { table[i] = key; return; }
        if (i + 9 <= (unsigned)mask)
            for (int j = 1; j <= 9; ++j) if (table[i + j] == PYSET_EMPTY) { table[i + j] = key; return; }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned)mask;
    }
}
// rebuilds into the upper half of the storage, then copies down (the tables are small)
__device__ inline void pyset_resize(PySetI& s, int minused) {
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    if (newsize + (s.mask + 1) > s.capacity) { s.overflow = true; return; }
    int* fresh = s.table + (s.capacity - newsize);            // scratch area at the end of the storage
    for (int i = 0; i < newsize; ++i) fresh[i] = PYSET_EMPTY;
    for (int i = 0; i <= s.mask; ++i) { const int k = s.table[i]; if (k >= 0) pyset_insert_clean(fresh, newsize - 1, k); }
    for (int i = 0; i < newsize; ++i) s.table[i] = fresh[i];
    s.mask = newsize - 1;
    s.fill = s.used;
}
__device__ inline int pyset_find(const PySetI& s, int key) {
    unsigned perturb = (unsigned)key, i = (unsigned)key & (unsigned)s.mask;
    while (true) {
        const int probes = (i + 9 <= (unsigned)s.mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            const int v = s.table[i + j];
            if (v == PYSET_EMPTY) return -1;
            if (v == key) return (int)(i + j);
        }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned)s.mask;
    }
}
__device__ inline void pyset_add(PySetI& s, int key) {
    unsigned perturb = (unsigned)key, i = (unsigned)key & (unsigned)s.mask;
    int free_slot = -1, unused = -1;
    while (unused < 0) {
        const int probes = (i + 9 <= (unsigned)s.mask) ? 9 : 0;
        for (int j = 0; j <= probes; ++j) {
            const int v = s.table[i + j];
            if (v == PYSET_EMPTY) { unused = (int)(i + j); break; }
            if (v == key) return;
            if (v == PYSET_DUMMY && free_slot < 0) free_slot = (int)(i + j);
        }
        if (unused >= 0) break;
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned)s.mask;
    }
    if (free_slot >= 0) { s.used++; s.table[free_slot] = key; return; }
    s.fill++; s.used++; s.table[unused] = key;
    if ((long)s.fill * 5 < (long)s.mask * 3) return;
    pyset_resize(s, s.used > 50000 ? s.used * 2 : s.used * 4);
}
// table size CPython reaches after n distinct insertions into an empty set (growth x4 whenever fill * 5 >= mask * 3)
__device__ inline int pyset_table_size(int n) {
    int size = 8;
    while (true) {
        const int thr = ((size - 1) * 3 + 4) / 5;          // smallest fill that triggers the resize
        if (n < thr) return size;
        const int minused = thr > 50000 ? thr * 2 : thr * 4;
        int ns = 8;
        while (ns <= minused) ns <<= 1;
        size = ns;
    }
}
// ints per table that pyset_difference(a of na keys, b of nb keys) needs at most: the final table plus the rebuild area of its
// last resize; the copy path sizes the result for 2 * na keys at once
__device__ inline int pyset_need(int na, int nb) {
    int f = pyset_table_size(na > nb ? na : nb), r = 8;
    while (r <= 2 * na) r <<= 1;
    return 2 * (f > r ? f : r);
}
// out[] = list(set(a) - set(b)); returns the length (or -1 when the tables would not fit)
__device__ inline int pyset_difference(int* storage, int capacity, const int* a, int na, const int* b, int nb, int* out) {
    PySetI A, B, R;
    pyset_init(A, storage, capacity);
    pyset_init(B, storage + capacity, capacity);
    pyset_init(R, storage + 2 * capacity, capacity);
    for (int i = 0; i < na && !A.overflow; ++i) pyset_add(A, a[i]);
    for (int i = 0; i < nb && !B.overflow; ++i) pyset_add(B, b[i]);
    if (A.overflow || B.overflow) return -1;
    if ((A.used >> 2) > B.used) {
        // set_copy_and_difference: copy (one big resize, slots kept when the sizes agree), then discard
        if (A.used != 0) {
            if ((long)(R.fill + A.used) * 5 >= (long)R.mask * 3) pyset_resize(R, (R.used + A.used) * 2);
            if (R.overflow) return -1;
            if (R.mask == A.mask && A.fill == A.used) {
                for (int i = 0; i <= A.mask; ++i) R.table[i] = A.table[i];
            } else {
                for (int i = 0; i <= A.mask; ++i) { const int k = A.table[i]; if (k >= 0) pyset_insert_clean(R.table, R.mask, k); }
            }
            R.fill = A.used; R.used = A.used;
        }
        for (int i = 0; i <= B.mask; ++i) {
            const int k = B.table[i];
            if (k < 0) continue;
            const int at = pyset_find(R, k);
            if (at >= 0) { R.table[at] = PYSET_DUMMY; R.used--; }
        }
    } else {
        for (int i = 0; i <= A.mask && !R.overflow; ++i) { const int k = A.table[i]; if (k >= 0 && pyset_find(B, k) < 0) pyset_add(R, k); }
        if (R.overflow) return -1;
    }
    int n = 0;
    for (int i = 0; i <= R.mask; ++i) if (R.table[i] >= 0) out[n++] = R.table[i];
    return n;
}

// ---------------------------------------------------------------------------
// The same hash tables replayed by ONE WAVEFRONT out of LDS (the single-thread version above is the fallback when the tables
// do not fit): a probe of the up to ten consecutive slots CPython examines is one LDS read across ten lanes plus a ballot, a
// rebuild walks the old table 64 slots at a time, and the tables ping-pong between two LDS regions instead of being copied
// down.  Same insertion order, same probe sequence, same growth rule -- the same layouts.  No deletions happen while a table
// is being built (the discards of the copy path only turn keys into dummies, which the final walk skips by the flag array).
// ---------------------------------------------------------------------------
struct WSet { int* table; int* other; int mask, fill; };
__device__ inline int wset_first_set(unsigned long long m) { return __popcll((m & (0ull - m)) - 1ull); }     // index of the lowest set bit
__device__ inline void wset_insert(int* table, int mask, int key, int lane) {          // set_insert_clean / set_add_entry of a new key
    unsigned perturb = (unsigned)key, i = (unsigned)key & (unsigned)mask;
    while (true) {
        const int probes = (i + 9 <= (unsigned)mask) ? 9 : 0;
        const int val = lane <= probes ? table[i + lane] : 0;
        const unsigned long long emp = __ballot(lane <= probes && val == PYSET_EMPTY);
        if (emp) { if (lane == 0) table[i + wset_first_set(emp)] = key; break; }
        perturb >>= 5;
        i = (i * 5 + 1 + perturb) & (unsigned)mask;
    }
    BM_WAVE_LDS_SYNC();
}
__device__ inline void wset_clear(int* table, int size, int lane) {
    for (int e = lane; e < size; e += WAVE) table[e] = PYSET_EMPTY;
    BM_WAVE_LDS_SYNC();
}
// set_table_resize: the keys of the old table, in slot order, go into a fresh table of `newsize` slots in the other region
__device__ inline void wset_rebuild(WSet& s, int newsize, int lane) {
    wset_clear(s.other, newsize, lane);
    for (int base = 0; base <= s.mask; base += WAVE) {
        const int key = base + lane <= s.mask ? s.table[base + lane] : PYSET_EMPTY;
        unsigned long long occ = __ballot(key >= 0);
        while (occ) {
            const int b = wset_first_set(occ);
            occ &= occ - 1ull;
            wset_insert(s.other, newsize - 1, __shfl(key, b, WAVE), lane);
        }
    }
    int* t = s.table; s.table = s.other; s.other = t;
    s.mask = newsize - 1;
}
// set(keys[0 .. n)) built key by key (set_add_entry with the growth rule); the tables alternate between regions a and b
__device__ inline void wset_build(WSet& s, int* a, int* b, const int* keys, int n, int lane) {
    s.table = a; s.other = b; s.mask = 7; s.fill = 0;
    wset_clear(s.table, 8, lane);
    for (int base = 0; base < n; base += WAVE) {
        const int mine = base + lane < n ? keys[base + lane] : 0;
        const int cnt = n - base < WAVE ? n - base : WAVE;
        for (int q = 0; q < cnt; ++q) {
            wset_insert(s.table, s.mask, __shfl(mine, q, WAVE), lane);
            s.fill++;
            if ((long)s.fill * 5 < (long)s.mask * 3) continue;
            const int minused = s.fill > 50000 ? s.fill * 2 : s.fill * 4;
            int newsize = 8;
            while (newsize <= minused) newsize <<= 1;
            wset_rebuild(s, newsize, lane);
        }
    }
}
// keys of a list, in order, those with skip[key] != 0 left out; returns the count
__device__ inline int wset_filter(const int* keys, int n_keys, const int* skip, int* out, int lane) {
    int n = 0;
    for (int base = 0; base < n_keys; base += WAVE) {
        const int key = base + lane < n_keys ? keys[base + lane] : -1;
        const bool keep = key >= 0 && !skip[key];
        const unsigned long long m = __ballot(keep);
        if (keep) out[n + __popcll(m & ((1ull << lane) - 1ull))] = key;
        n += __popcll(m);
    }
    BM_WAVE_LDS_SYNC();
    return n;
}
// the keys of the table in slot order, those with skip[key] != 0 left out (skip may be null); returns the count
__device__ inline int wset_walk(const WSet& s, const int* skip, int* out, int lane) {
    int n = 0;
    for (int base = 0; base <= s.mask; base += WAVE) {
        const int key = base + lane <= s.mask ? s.table[base + lane] : PYSET_EMPTY;
        const bool keep = key >= 0 && !(skip && skip[key]);
        const unsigned long long m = __ballot(keep);
        if (keep) out[n + __popcll(m & ((1ull << lane) - 1ull))] = key;
        n += __popcll(m);
    }
    BM_WAVE_LDS_SYNC();
    return n;
}
// ints of LDS the wave version needs: two table regions of `fmax` slots and two key lists of `nkeys`
__host__ __device__ inline long wset_lds_ints(int fmax, int nkeys) { return 2L * fmax + 2L * nkeys + 256; }

template <class A>
void ss_allocate(SsStepArgs& args, const SsSizes& z, A& a) {
    const size_t S = z.S, cap = z.cap, nd = z.nd, dim = z.dim, big = cap > nd ? cap : nd;
    SsState& st = args.st;
    st.cap = z.cap; st.dim = z.dim; st.budget = z.budget;
    st.frame_count = a.template get<int>(S); st.next_id = a.template get<int>(S);
    st.n_tracks = a.template get<int>(S); st.status = a.template get<int>(S);
    st.list = a.template get<int>(S * cap); st.slot_used = a.template get<int>(S * cap);
    st.kf = a.template get<double>(S * cap * KF_STRIDE);
    st.feat = a.template get<float>(S * cap * dim);
    st.bank = a.template get<float>(S * cap * (size_t)z.budget * dim);
    st.bank_norm = a.template get<float>(S * cap * (size_t)z.budget);
    st.bank_n = a.template get<int>(S * cap);
    st.id = a.template get<int>(S * cap); st.state = a.template get<int>(S * cap); st.hits = a.template get<int>(S * cap);
    st.age = a.template get<int>(S * cap); st.tsu = a.template get<int>(S * cap);
    st.conf = a.template get<float>(S * cap); st.cls = a.template get<float>(S * cap); st.det_ind = a.template get<float>(S * cap);
    SsScratch& sc = args.sc;
    sc.max_dets = z.nd;
    sc.app = a.template get<float>(S * cap * nd);
    sc.det_norm = a.template get<float>(S * nd);
    sc.keep = a.template get<int>(S * nd);
    sc.det_tlwh = a.template get<double>(S * nd * 4); sc.det_xyah = a.template get<double>(S * nd * 4);
    sc.cost = a.template get<double>(S * big * big);
    sc.cost_t = a.template get<double>(S * big * big);
    sc.rows_a = a.template get<int>(S * cap); sc.rows_b = a.template get<int>(S * cap); sc.cols_b = a.template get<int>(S * nd);
    sc.un_d = a.template get<int>(S * nd); sc.tmp_a = a.template get<int>(S * big); sc.tmp_b = a.template get<int>(S * big);
    sc.m_trk = a.template get<int>(S * big); sc.m_det = a.template get<int>(S * big);
    sc.row_of = a.template get<int>(S * big); sc.col_of = a.template get<int>(S * big);
    sc.flag_t = a.template get<int>(S * cap);
    sc.pyset = a.template get<int>(S * 3 * (size_t)pyset_capacity(z.cap));
}

// ---------------------------------------------------------------------------
// NearestNeighborDistanceMetric.distance (linear_assignment.py:336-353, _nn_cosine_distance :266-284):
// app[t][j] = min over the bank of track t of 1 - <a/|a|, b/|b|>, all fp32 -- the reference's float32 matrix
// product followed by a min.  A batched GEMM: one workgroup per (track position, stream), confirmed tracks only,
// C[m][n] = <bank row m, detection n> over 64x64 tiles staged through LDS in k-chunks of 16, TMR x TNR outputs per
// thread in registers; the norms (bank rows: stored when the sample is appended; detections: ss_det_norm_block) scale
// the product afterwards.  fp32 FMAs: the fp32 matrix pipe has the same peak as the vector pipe on gfx950.
// ---------------------------------------------------------------------------
constexpr int SS_TILE = 64, SS_KC = 16;

// |b| of every detection of stream s: one wavefront per detection; `part` of `nparts` workgroups share a stream's detections
template <int NTHR>
__device__ inline void ss_det_norm_block(const SsStepArgs& a, int s, int part = 0, int nparts = 1) {
    const long dim = a.st.dim, nd = a.sc.max_dets;
    const int n_d = a.n_dets[s], lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const float* embs = a.embs + (long)s * nd * dim;
    for (int j = part * (NTHR / WAVE) + wave; j < n_d; j += nparts * (NTHR / WAVE)) {
        float ss = 0.f;
        const float* row = embs + j * dim;
        for (int k0 = lane; k0 < dim; k0 += 4 * WAVE) {        // four loads in flight; the per-lane fmaf order is the plain loop's
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = k0 + u * WAVE < dim ? row[k0 + u * WAVE] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (k0 + u * WAVE < dim) ss = fmaf(x[u], x[u], ss);
        }
        ss = wave_sum(ss);
        if (lane == 0) a.sc.det_norm[s * nd + j] = sqrtf(ss);
    }
}

template <int NTHR>
__device__ inline void ss_bank_distance_block(const SsStepArgs& a, int s, int t, float (*sA)[SS_TILE + 1], float (*sB)[SS_TILE + 1],
                                              float (*sMin)[SS_TILE]) {
    constexpr int TX = NTHR == 256 ? 16 : 8, TY = NTHR / TX, TMR = SS_TILE / TY, TNR = SS_TILE / TX;
    const SsState& st = a.st;
    const long cap = st.cap, dim = st.dim, nd = a.sc.max_dets;
    const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
    if (a.n_dets[s] < 0 || t >= st.n_tracks[s]) return;
    const int slot = st.list[s * cap + t];
    if (st.state[s * cap + slot] != SS_CONFIRMED) return;
    const int n_d = a.n_dets[s];
    const float* embs = a.embs + (long)s * nd * dim;
    const int M = st.bank_n[s * cap + slot] < st.budget ? st.bank_n[s * cap + slot] : st.budget;
    const float* bank = st.bank + ((long)(s * cap + slot)) * st.budget * dim;
    const float* bnorm = st.bank_norm + (long)(s * cap + slot) * st.budget;
    const float* dnorm = a.sc.det_norm + s * nd;
    float* out = a.sc.app + ((long)s * cap + t) * nd;
    for (int n0 = 0; n0 < n_d; n0 += SS_TILE) {
        float best[TNR];
        for (int q = 0; q < TNR; ++q) best[q] = 3.0e38f;
        for (int m0 = 0; m0 < M; m0 += SS_TILE) {
            float acc[TMR][TNR];
            for (int p = 0; p < TMR; ++p) for (int q = 0; q < TNR; ++q) acc[p][q] = 0.f;
            for (int k0 = 0; k0 < dim; k0 += SS_KC) {
                // stage A[m][k] and B[n][k] as [k][m], [k][n]: consecutive threads read consecutive k (coalesced 64-byte rows)
                for (int e = tid; e < SS_TILE * SS_KC; e += NTHR) {
                    const int r = e / SS_KC, k = e % SS_KC;
                    sA[k][r] = (m0 + r < M && k0 + k < dim) ? bank[(long)(m0 + r) * dim + k0 + k] : 0.f;
                    sB[k][r] = (n0 + r < n_d && k0 + k < dim) ? embs[(long)(n0 + r) * dim + k0 + k] : 0.f;
                }
                __syncthreads();
                for (int k = 0; k < SS_KC; ++k) {
                    float av[TMR], bv[TNR];
                    for (int p = 0; p < TMR; ++p) av[p] = sA[k][ty * TMR + p];
                    for (int q = 0; q < TNR; ++q) bv[q] = sB[k][tx * TNR + q];
                    for (int p = 0; p < TMR; ++p) for (int q = 0; q < TNR; ++q) acc[p][q] = fmaf(av[p], bv[q], acc[p][q]);
                }
                __syncthreads();
            }
            for (int p = 0; p < TMR; ++p) {
                const int m = m0 + ty * TMR + p;
                if (m >= M) continue;
                const float an = bnorm[m];
                for (int q = 0; q < TNR; ++q) {
                    const int n = n0 + tx * TNR + q;
                    if (n >= n_d) continue;
                    const float dist = 1.0f - acc[p][q] / (an * dnorm[n]);
                    best[q] = dist < best[q] ? dist : best[q];
                }
            }
        }
        for (int q = 0; q < TNR; ++q) sMin[ty][tx * TNR + q] = best[q];
        __syncthreads();
        for (int n = tid; n < SS_TILE && n0 + n < n_d; n += NTHR) {
            float m = sMin[0][n];
            for (int y = 1; y < TY; ++y) m = sMin[y][n] < m ? sMin[y][n] : m;
            out[n0 + n] = m;
        }
        __syncthreads();
    }
}

// The same distances on the fp32 matrix pipe (the shipped kernel; the scalar-FMA version above is kept as its
// on-device cross-check).  One workgroup of NTHR / 64 waves per (track position, stream):
//   C[m][n] = <bank row m, detection n>,  m < M <= budget (up to SS_MT = 7 row tiles of 16: budgets <= 112),
//   n in passes of 16 * (NTHR / 64) detections: wave w owns one 16-detection column tile and all row tiles (7 accumulators).
// K runs in chunks of SS_MKC = 32 through LDS (row stride 36 floats: 16-byte rows for the staging writes, 2-way at most for
// the one-dword fragment reads); the next chunk's global loads are issued before the current chunk's MFMAs and written to
// LDS after them, so the only exposed round trip is the first.  v_mfma_f32_16x16x4_f32 is the exact k-ordered fmaf chain:
// the sums are bit-identical to the scalar version's (k ascending, one fmaf per k) -- same distances, same argmin.
// Rate: 64 FLOP/clk/SIMD like the vector pipe, but issued as 1 instruction per 2 KiFLOP with the operands read once per
// 16x16 tile, and with the vector ALUs free for the staging (MI355X_MICROARCH.md: 122 TF untuned vs 52 TF for a VALU GEMM).
// ---------------------------------------------------------------------------
constexpr int SS_MT = 7, SS_MKC = 32, SS_MLD = 36;
constexpr int SS_MFMA_LDS_FLOATS(int nthr) { return (SS_MT * 16 + 16 * (nthr / 64)) * SS_MLD; }

template <int NTHR>
__device__ inline void ss_bank_distance_block_mfma(const SsStepArgs& a, int s, int t, float* lds) {
    constexpr int NW = NTHR / WAVE, NB = 16 * NW, ROWS_A = SS_MT * 16;
    constexpr int QA = (ROWS_A * SS_MKC / 4 + NTHR - 1) / NTHR, QB = (NB * SS_MKC / 4 + NTHR - 1) / NTHR;     // float4 slots per thread
    const SsState& st = a.st;
    const long cap = st.cap, dim = st.dim, nd = a.sc.max_dets;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE, g = lane >> 4, l16 = lane & 15;
    if (a.n_dets[s] < 0 || t >= st.n_tracks[s]) return;
    const int slot = st.list[s * cap + t];
    if (st.state[s * cap + slot] != SS_CONFIRMED) return;
    const int n_d = a.n_dets[s];
    const float* embs = a.embs + (long)s * nd * dim;
    const int M = st.bank_n[s * cap + slot] < st.budget ? st.bank_n[s * cap + slot] : st.budget;
    const float* bank = st.bank + ((long)(s * cap + slot)) * st.budget * dim;
    const float* bnorm = st.bank_norm + (long)(s * cap + slot) * st.budget;
    const float* dnorm = a.sc.det_norm + s * nd;
    float* out = a.sc.app + ((long)s * cap + t) * nd;
    float* sA = lds;                        // [ROWS_A][SS_MLD]
    float* sB = lds + ROWS_A * SS_MLD;      // [NB][SS_MLD]
    const int mt_used = (M + 15) / 16;      // row tiles that hold samples (wave-uniform)
    for (int n0 = 0; n0 < n_d; n0 += NB) {
        bm_f4 acc[SS_MT];
#pragma unroll
        for (int mt = 0; mt < SS_MT; ++mt) acc[mt] = bm_f4{0.f, 0.f, 0.f, 0.f};
        bm_f4 ra[QA], rb[QB];
        // chunk loader: slot e of the A part = (row e / 8, four consecutive k's e % 8); rows / k's beyond the data read as 0
        auto load_chunk = [&](int k0) {
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const int e = tid + q * NTHR, r = e / (SS_MKC / 4), c4 = (e % (SS_MKC / 4)) * 4;
                bm_f4 v = bm_f4{0.f, 0.f, 0.f, 0.f};
                if (e < ROWS_A * SS_MKC / 4 && r < M) {
                    const float* src = bank + (long)r * dim + k0 + c4;
                    if (k0 + c4 + 3 < dim) v = *reinterpret_cast<const bm_f4*>(src);
                    else for (int j = 0; j < 4; ++j) if (k0 + c4 + j < dim) v[j] = src[j];
                }
                ra[q] = v;
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int e = tid + q * NTHR, r = e / (SS_MKC / 4), c4 = (e % (SS_MKC / 4)) * 4;
                bm_f4 v = bm_f4{0.f, 0.f, 0.f, 0.f};
                if (e < NB * SS_MKC / 4 && n0 + r < n_d) {
                    const float* src = embs + (long)(n0 + r) * dim + k0 + c4;
                    if (k0 + c4 + 3 < dim) v = *reinterpret_cast<const bm_f4*>(src);
                    else for (int j = 0; j < 4; ++j) if (k0 + c4 + j < dim) v[j] = src[j];
                }
                rb[q] = v;
            }
        };
        auto store_chunk = [&]() {
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const int e = tid + q * NTHR, r = e / (SS_MKC / 4), c4 = (e % (SS_MKC / 4)) * 4;
                if (e < ROWS_A * SS_MKC / 4) *reinterpret_cast<bm_f4*>(sA + r * SS_MLD + c4) = ra[q];
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int e = tid + q * NTHR, r = e / (SS_MKC / 4), c4 = (e % (SS_MKC / 4)) * 4;
                if (e < NB * SS_MKC / 4) *reinterpret_cast<bm_f4*>(sB + r * SS_MLD + c4) = rb[q];
            }
        };
        load_chunk(0);
        for (int k0 = 0; k0 < dim; k0 += SS_MKC) {
            __syncthreads();                // every wave is done reading the previous chunk
            store_chunk();
            __syncthreads();
            if (k0 + SS_MKC < dim) load_chunk(k0 + SS_MKC);          // in flight while this chunk multiplies
#pragma unroll
            for (int kk = 0; kk < SS_MKC / 4; ++kk) {
                const float b = sB[(wave * 16 + l16) * SS_MLD + 4 * kk + g];
#pragma unroll
                for (int mt = 0; mt < SS_MT; ++mt)
                    if (mt < mt_used) acc[mt] = BM_MFMA_F32_K4(sA[(mt * 16 + l16) * SS_MLD + 4 * kk + g], b, acc[mt]);
            }
        }
        // D layout: column (detection) l16, rows (samples) 16 mt + 4 g + r: distance, minimum over the lane's samples, then over g
        const int n = n0 + wave * 16 + l16;
        const float dn = n < n_d ? dnorm[n] : 1.f;
        float best = 3.0e38f;
#pragma unroll
        for (int mt = 0; mt < SS_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + 4 * g + r;
                if (m < M) {
                    const float dist = 1.0f - acc[mt][r] / (bnorm[m] * dn);
                    best = dist < best ? dist : best;
                }
            }
        const float o1 = __shfl_xor(best, 16, WAVE);
        best = o1 < best ? o1 : best;
        const float o2 = __shfl_xor(best, 32, WAVE);
        best = o2 < best ? o2 : best;
        if (g == 0 && n < n_d) out[n] = best;
    }
}

// ---------------------------------------------------------------------------
// KalmanFilterXYAH, one wavefront per track (lane l <-> cov element (l>>3, l&7))
// ---------------------------------------------------------------------------
// predict (base.py:259-275 + xyah.py:105-108): mean . F^T; F (cov F^T) + Q with Q from the PRE-motion height
__device__ inline void ss_kf_predict_wave(double* kf, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const double mj = kf[j];
    const double h = __shfl(mj, 3, WAVE);
    const double p = kf[KF_DIM + lane];
    const double mhi = __shfl(mj, (j + 4) & 7, WAVE);
    double mnew = (j < 4) ? (mj + mhi) : mj;
    if (j == 2 || j == 3) mnew = mnew > 1e-4 ? mnew : 1e-4;
    // T = cov F^T : T[i][j] = P[i][j] + P[i][j+4];  U = F T : U[i][j] = T[i][j] + T[i+4][j]   (multi_dot picks F (cov F^T))
    const double p_rt = __shfl(p, (lane & ~7) | ((j + 4) & 7), WAVE);
    const double t = (j < 4) ? (p + p_rt) : p;
    const double t_dn = __shfl(t, (lane + 32) & 63, WAVE);
    double c = (i < 4) ? (t + t_dn) : t;
    if (i == j) {
        double sd;
        if (i == 2) sd = 1e-2; else if (i == 6) sd = 1e-5;
        else sd = ((i < 4) ? SS_STD_POS : SS_STD_VEL) * h;
        c = c + sd * sd;
    } else {
        c = c + 0.0;
    }
    kf[KF_DIM + lane] = c;
    if (i == 0) kf[j] = mnew;
}

// update with measurement z (xyah) and detection confidence (NSA: std scaled by 1 - conf), base.py:286-355 + xyah.py:126-150
__device__ inline bool ss_kf_update_wave(double* kf, const double* z, double confidence, int lane) {
    const int i = lane >> 3, j = lane & 7;
    double m[8];
    for (int k = 0; k < 8; ++k) m[k] = kf[k];
    const double* P = kf + KF_DIM;
    double S[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) S[a][b] = P[a * 8 + b];
    for (int a = 0; a < 4; ++a) {
        const double base = (a == 2) ? 1e-1 : SS_STD_POS * m[3];
        const double sd = (1 - confidence) * base;
        S[a][a] = S[a][a] + sd * sd;
    }
    double L[4][4];
    bool ok = true;
    for (int c = 0; c < 4; ++c) {
        double d = S[c][c];
        for (int k = 0; k < c; ++k) d -= L[c][k] * L[c][k];
        if (!(d > 0.0)) ok = false;
        d = sqrt(d);
        L[c][c] = d;
        for (int r = c + 1; r < 4; ++r) {
            double t = S[r][c];
            for (int k = 0; k < c; ++k) t -= L[r][k] * L[c][k];
            L[r][c] = t / d;
        }
    }
    double Ki[4], Kj[4];
    for (int which = 0; which < 2; ++which) {
        const int r = which ? j : i;
        double y[4];
        for (int k = 0; k < 4; ++k) {
            double t = P[r * 8 + k];
            for (int q = 0; q < k; ++q) t -= L[k][q] * y[q];
            y[k] = t / L[k][k];
        }
        double* K = which ? Kj : Ki;
        for (int k = 3; k >= 0; --k) {
            double t = y[k];
            for (int q = k + 1; q < 4; ++q) t -= L[q][k] * K[q];
            K[k] = t / L[k][k];
        }
    }
    double acc = 0.0;
    for (int a = 0; a < 4; ++a) acc += (z[a] - m[a]) * Ki[a];
    double mnew = m[i] + acc;
    if (i == 2 || i == 3) mnew = mnew > 1e-4 ? mnew : 1e-4;
    double ksk = 0.0;
    for (int a = 0; a < 4; ++a) {
        double mj = 0.0;
        for (int b = 0; b < 4; ++b) mj += S[a][b] * Kj[b];
        ksk += Ki[a] * mj;
    }
    const double pnew = P[lane] - ksk;
    const double pn = __shfl(pnew, lane, WAVE);       // wave-wide dependency: every load above precedes the stores
    kf[KF_DIM + lane] = pn;
    if (j == 0) kf[i] = mnew;
    return ok;
}

// squared Mahalanobis distance of measurement z to the projected state (confidence 0), base.py:523-551
// split in two so that the factor, which depends on the track only, is computed once per track: the same operations on the same
// values as the one-piece form
struct SsGate { double L[4][4]; double m[4]; };
__device__ inline void ss_gating_factor(const double* kf, SsGate& g) {
    const double* P = kf + KF_DIM;
    double S[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) S[a][b] = P[a * 8 + b];
    for (int a = 0; a < 4; ++a) { const double sd = (a == 2) ? 1e-1 : SS_STD_POS * kf[3]; S[a][a] = S[a][a] + sd * sd; }
    for (int c = 0; c < 4; ++c) {
        double d = S[c][c];
        for (int k = 0; k < c; ++k) d -= g.L[c][k] * g.L[c][k];
        d = sqrt(d);
        g.L[c][c] = d;
        for (int r = c + 1; r < 4; ++r) {
            double t = S[r][c];
            for (int k = 0; k < c; ++k) t -= g.L[r][k] * g.L[c][k];
            g.L[r][c] = t / d;
        }
    }
    for (int k = 0; k < 4; ++k) g.m[k] = kf[k];
}
__device__ inline double ss_gating_apply(const SsGate& g, const double* z) {
    double y[4], s = 0.0;
    for (int k = 0; k < 4; ++k) {
        double t = z[k] - g.m[k];
        for (int q = 0; q < k; ++q) t -= g.L[k][q] * y[q];
        y[k] = t / g.L[k][k];
    }
    for (int k = 0; k < 4; ++k) s += y[k] * y[k];
    return s;
}
__device__ inline double ss_gating_distance(const double* kf, const double* z) {
    SsGate g;
    ss_gating_factor(kf, g);
    return ss_gating_apply(g, z);
}

// ---------------------------------------------------------------------------
// scipy.optimize.linear_sum_assignment (rectangular_lsap.cpp): rows of the smaller side are inserted in order with
// a shortest augmenting path over the `remaining` columns (filled in reverse, removed by swapping in the last one);
// among equal path costs the scan keeps the first column unless a later one is unassigned (then the last such).
// cost_of(r, c) is the LOGICAL matrix (already transposed by the caller when it has more rows than columns).
// Solver state in dynamic LDS.  Output: col_of[r] for r < nr (every row is assigned).
// ---------------------------------------------------------------------------
struct LsaLds { double* u; double* v; double* spc; int* path; int* col4row; int* row4col; int* sr; int* sc; int* remaining; int cap_n; };
__host__ __device__ inline long ss_lsa_lds_bytes(int n) { return (long)n * (8 * 3 + 4 * 6) + 64 + 2 * MAX_WAVES * 16; }
__device__ inline LsaLds ss_carve_lsa(unsigned char* base, int n) {
    LsaLds l;
    l.cap_n = n;
    l.u = reinterpret_cast<double*>(base); l.v = l.u + n; l.spc = l.v + n;
    l.path = reinterpret_cast<int*>(l.spc + n); l.col4row = l.path + n; l.row4col = l.col4row + n;
    l.sr = l.row4col + n; l.sc = l.sr + n; l.remaining = l.sc + n;
    return l;
}

// One barrier per scan.  Every wavefront reduces (lowest, first position among the minima, last unassigned position among the
// minima) over its columns and leaves the result in a 16-byte slot of the current parity; after the barrier every thread
// combines the slots itself (one ds_read_b128 each; the same comparisons a single combiner would make).  Positions travel
// packed with their column (position << 16 | column: positions are distinct, so min / max on the packed word is min / max on
// the position), so nobody reads `remaining` after the barrier; the swap that removes the chosen column from `remaining` is
// deferred to the owner of that position at the start of the next scan, and the slots alternate between two parities, which
// together make the single barrier sufficient.  sr / sc marks are only read by the dual update, behind its own barrier.
// Measured on configuration 5 (256 rows x 1024 columns, 263 scans per frame): six barriers per scan with a serial combiner
// 9.5 k cycles per scan; a single-wavefront solver without any barrier 14.7 k.
// Wavefronts beyond the first BM_LSA_SCAN_THREADS / 64 do not scan (they only follow the control flow): a scan is a thousand
// columns at most, and with sixteen scanning waves the barrier and the combine cost more than the second column per thread saves.
#ifndef BM_LSA_SCAN_THREADS
#define BM_LSA_SCAN_THREADS 512
#endif
struct alignas(16) LsaSlot { double lowest; int first; int last_un; };       // packed: position << 16 | column, -1 = none
constexpr int LSA_SLOT_BYTES = 2 * MAX_WAVES * 16;
__device__ inline LsaSlot* ss_lsa_slots(unsigned char* base, int n) { return reinterpret_cast<LsaSlot*>(base + (((long)n * (8 * 3 + 4 * 6) + 63) & ~63L)); }
template <class CostFn>
__device__ inline bool lsa_scipy(const Ctx& c, const LsaLds& L, int nr, int nc, CostFn cost_of, int* col_of) {
    SS_PROF_COUNT(11, 1);
    LsaSlot* const slot_base = ss_lsa_slots(reinterpret_cast<unsigned char*>(L.u), L.cap_n);
    for (int r = c.tid; r < nr; r += c.nthr) { L.u[r] = 0.0; L.col4row[r] = -1; }
    for (int j = c.tid; j < nc; j += c.nthr) { L.v[j] = 0.0; L.row4col[j] = -1; L.path[j] = -1; }
    __syncthreads();
    bool feasible = true;
    int parity = 0;
    const int nscan = c.nthr < BM_LSA_SCAN_THREADS ? c.nthr : BM_LSA_SCAN_THREADS, nscan_waves = nscan / WAVE;
    const bool scans = c.tid < nscan;
    for (int cur = 0; cur < nr && feasible; ++cur) {
        double min_val = 0.0;
        int i = cur, num_remaining = nc, sink = -1;
        int pend_pos = -1, pend_from = -1;           // remaining[pend_pos] = remaining[pend_from], owed by the owner of pend_pos
        for (int it = c.tid; it < nc; it += c.nthr) { L.remaining[it] = nc - it - 1; L.sc[it] = 0; L.spc[it] = SS_INF; }
        for (int r = c.tid; r < nr; r += c.nthr) L.sr[r] = 0;
        __syncthreads();
        while (sink == -1) {
            SS_PROF_COUNT(10, 1);
            if (c.tid == 0) L.sr[i] = 1;
            const double ui = L.u[i];
            double lowest = SS_INF;
            int first = -1, last_un = -1;
            if (scans)
            for (int it = c.tid; it < num_remaining; it += nscan) {
                if (it == pend_pos) L.remaining[it] = L.remaining[pend_from];
                const int j = L.remaining[it];
                const double cij = cost_of(i, j), vj = L.v[j];
                double sp = L.spc[j];
                const bool un = L.row4col[j] == -1;
                const double r = min_val + cij - ui - vj;
                if (r < sp) { L.path[j] = i; L.spc[j] = r; sp = r; }
                const int packed = (it << 16) | j;
                if (sp < lowest) { lowest = sp; first = packed; last_un = un ? packed : -1; }
                else if (sp == lowest && un) last_un = packed;
            }
            LsaSlot* slots = slot_base + parity * MAX_WAVES;
            if (scans) {           // wave-uniform
                for (int off = WAVE / 2; off > 0; off >>= 1) {
                    const double ov = __shfl_xor(lowest, off, WAVE);
                    const int of = __shfl_xor(first, off, WAVE), ol = __shfl_xor(last_un, off, WAVE);
                    if (of >= 0 && (first < 0 || ov < lowest)) { lowest = ov; first = of; last_un = ol; }
                    else if (of >= 0 && ov == lowest) { first = of < first ? of : first; last_un = ol > last_un ? ol : last_un; }
                }
                if (c.lane == 0) { LsaSlot o; o.lowest = lowest; o.first = first; o.last_un = last_un; slots[c.wave] = o; }
            }
            __syncthreads();
            double gl = SS_INF;
            int gf = -1, gu = -1;
            for (int w = 0; w < nscan_waves; ++w) {
                const LsaSlot o = slots[w];
                if (o.first < 0) continue;
                if (gf < 0 || o.lowest < gl) { gl = o.lowest; gf = o.first; gu = o.last_un; }
                else if (o.lowest == gl) { gf = o.first < gf ? o.first : gf; gu = o.last_un > gu ? o.last_un : gu; }
            }
            parity ^= 1;
            if (gf < 0 || !(gl < SS_INF)) { feasible = false; break; }
            min_val = gl;
            const int chosen = gu >= 0 ? gu : gf;
            const int index = chosen >> 16, j = chosen & 0xffff;
            const int owner = L.row4col[j];
            if (owner == -1) sink = j; else i = owner;
            if (c.tid == 0) L.sc[j] = 1;
            --num_remaining;
            pend_pos = index; pend_from = num_remaining;         // the last live position moves into the hole
            if (pend_pos == pend_from) pend_pos = -1;
        }
        __syncthreads();           // sr / sc marks and the last scan's spc are visible to the dual update
        if (!feasible) break;
        // dual variables (rectangular_lsap.cpp: u[curRow] += minVal; u[i] += minVal - spc[col4row[i]]; v[j] -= minVal - spc[j])
        for (int r = c.tid; r < nr; r += c.nthr) {
            if (r == cur) L.u[r] += min_val;
            else if (L.sr[r]) L.u[r] += min_val - L.spc[L.col4row[r]];
        }
        for (int j = c.tid; j < nc; j += c.nthr) if (L.sc[j]) L.v[j] -= min_val - L.spc[j];
        __syncthreads();
        if (c.tid == 0) {
            int j = sink, guard = 0;
            while (guard++ <= nr) {
                const int r = L.path[j];
                L.row4col[j] = r;
                const int prev = L.col4row[r];
                L.col4row[r] = j;
                j = prev;
                if (r == cur) break;
            }
        }
        __syncthreads();
    }
    for (int r = c.tid; r < nr; r += c.nthr) col_of[r] = feasible ? L.col4row[r] : -1;
    __syncthreads();
    return feasible;
}

// Per-stream view
struct SSV {
    SsConfigDev cfg;
    int cap, dim, nd, budget;
    int* frame_count; int* next_id; int* n_tracks; int* status; int* list; int* slot_used;
    double* kf; float* feat; float* bank; float* bank_norm; int* bank_n; int* id; int* state; int* hits; int* age; int* tsu;
    float* conf; float* cls; float* det_ind;
    float* app; int* keep; double* det_tlwh; double* det_xyah; double* cost; double* cost_t;
    int* rows_a; int* rows_b; int* cols_b; int* un_d; int* tmp_a; int* tmp_b; int* m_trk; int* m_det; int* row_of; int* col_of; int* flag_t; int* pyset;
    const float* dets; int n_dets; const float* embs; const double* warp; float* out; int* out_n;
};

__device__ inline SSV ss_view(const SsStepArgs& a, int s) {
    SSV v;
    const SsState& st = a.st; const SsScratch& sc = a.sc;
    const long cap = st.cap, dim = st.dim, nd = sc.max_dets, big = cap > nd ? cap : nd;
    v.cfg = a.cfg; v.cap = st.cap; v.dim = st.dim; v.nd = sc.max_dets; v.budget = st.budget;
    v.frame_count = st.frame_count + s; v.next_id = st.next_id + s; v.n_tracks = st.n_tracks + s; v.status = st.status + s;
    v.list = st.list + s * cap; v.slot_used = st.slot_used + s * cap;
    v.kf = st.kf + s * cap * KF_STRIDE; v.feat = st.feat + s * cap * dim;
    v.bank = st.bank + s * cap * (long)st.budget * dim; v.bank_norm = st.bank_norm + s * cap * (long)st.budget; v.bank_n = st.bank_n + s * cap;
    v.id = st.id + s * cap; v.state = st.state + s * cap; v.hits = st.hits + s * cap; v.age = st.age + s * cap; v.tsu = st.tsu + s * cap;
    v.conf = st.conf + s * cap; v.cls = st.cls + s * cap; v.det_ind = st.det_ind + s * cap;
    v.app = sc.app + s * cap * nd; v.keep = sc.keep + s * nd; v.det_tlwh = sc.det_tlwh + s * nd * 4; v.det_xyah = sc.det_xyah + s * nd * 4;
    v.cost = sc.cost + s * big * big;
    v.cost_t = sc.cost_t + s * big * big;
    v.rows_a = sc.rows_a + s * cap; v.rows_b = sc.rows_b + s * cap; v.cols_b = sc.cols_b + s * nd; v.un_d = sc.un_d + s * nd;
    v.tmp_a = sc.tmp_a + s * big; v.tmp_b = sc.tmp_b + s * big; v.m_trk = sc.m_trk + s * big; v.m_det = sc.m_det + s * big;
    v.row_of = sc.row_of + s * big; v.col_of = sc.col_of + s * big; v.flag_t = sc.flag_t + s * cap;
    v.pyset = sc.pyset + (long)s * 3 * pyset_capacity(st.cap);
    v.dets = a.dets + s * nd * DET_COLS; v.n_dets = a.n_dets[s]; v.embs = a.embs + s * nd * dim;
    v.warp = a.warp ? a.warp + s * 6 : nullptr;
    v.out = a.out + s * cap * OUT_COLS; v.out_n = a.out_n + s;
    return v;
}

// `unmatched_tracks = list(set(rows_a) - set(m_trk))` in CPython's iteration order -> v.tmp_a, returns the length.
// rows_a: the confirmed track positions (ascending), m_trk[0 .. n_match): the matched ones; nt = live tracks.
__device__ inline int ss_unmatched_in_set_order(const Ctx& c, SSV& v, int n_conf, int n_match, int nt, int big, unsigned char* dyn_lds) {
    int* s_int = c.s_int;
    auto ident = [](int i) { return i; };
    // The order of that list is the slot order of CPython's hash table (hash(i) == i).  B only matters as a set (a flag per
    // track position).  A's slot order: track positions are small ascending ints, and whenever every key is below the table
    // size in force when it is inserted (checked in parallel) no probe ever collides, each key sits in the slot of its own
    // value and the order is ascending; otherwise wave 0 replays the table out of LDS (wset_*).  The result either shares A's
    // layout (copy path with equal table sizes: the unmatched keys in A's order) or is rebuilt by the same wave.
    const int na = n_conf, nb = n_match;
    {
        for (int t = c.tid; t < nt; t += c.nthr) v.flag_t[t] = 0;
        __syncthreads();
        for (int q = c.tid; q < nb; q += c.nthr) v.flag_t[v.m_trk[q]] = 1;
        __syncthreads();
        const int bad_a = block_append_if(c, na, [&](int i) { return v.rows_a[i] >= pyset_table_size(i); }, ident, v.tmp_b, 0);
        // unmatched keys in ascending order: the answer whenever every table involved has the identity layout
        const int n_asc = block_append_if(c, na, [&](int i) { return !v.flag_t[v.rows_a[i]]; }, [&](int i) { return v.rows_a[i]; }, v.tmp_a, 0);
        const bool copy_path = (na >> 2) > nb;                 // set_difference -> set_copy_and_difference
        const int ta = pyset_table_size(na);
        int tr = 8;                                            // the copy is sized for 2 na keys at once (set_merge)
        if (copy_path && (long)na * 5 >= 7 * 3) while (tr <= 2 * na) tr <<= 1;
        int bad_r = 0;
        if (!bad_a && na > 0) {
            if (copy_path) bad_r = (tr != ta && v.rows_a[na - 1] >= tr) ? 1 : 0;
            else bad_r = block_append_if(c, n_asc, [&](int i) { return v.tmp_a[i] >= pyset_table_size(i); }, ident, v.tmp_b, 0);
        }
        int n = n_asc;
        if (bad_a || bad_r) {
            int fmax = ta;
            if (copy_path && tr > fmax) fmax = tr;
            n = -2;
            if (wset_lds_ints(fmax, na) * (long)sizeof(int) <= ss_lsa_lds_bytes(big)) {
                if (c.wave == 0) {
                    int* reg_a = reinterpret_cast<int*>(dyn_lds);
                    int* reg_b = reg_a + fmax + 64;
                    int* list_a = reg_b + fmax + 64;           // A's keys in slot order
                    int* list_b = list_a + na + 64;            // the candidates of the growing result
                    WSet A, R;
                    const int* order = v.rows_a;
                    int n_order = na;
                    if (bad_a) {
                        wset_build(A, reg_a, reg_b, v.rows_a, na, c.lane);
                        n_order = wset_walk(A, nullptr, list_a, c.lane);
                        order = list_a;
                    }
                    if (copy_path) {
                        if (tr == ta) n = wset_filter(order, n_order, v.flag_t, v.tmp_a, c.lane);        // slots copied one to one
                        else {             // set_insert_clean of A's keys, in slot order, into tr empty slots; the discards leave dummies
                            wset_clear(reg_a, tr, c.lane);
                            for (int base = 0; base < n_order; base += WAVE) {
                                const int mine = base + c.lane < n_order ? order[base + c.lane] : 0;
                                const int cnt = n_order - base < WAVE ? n_order - base : WAVE;
                                for (int q = 0; q < cnt; ++q) wset_insert(reg_a, tr - 1, __shfl(mine, q, WAVE), c.lane);
                            }
                            R.table = reg_a; R.other = reg_b; R.mask = tr - 1; R.fill = n_order;
                            n = wset_walk(R, v.flag_t, v.tmp_a, c.lane);
                        }
                    } else {               // the unmatched keys, in A's slot order, enter an empty set one by one
                        const int m = wset_filter(order, n_order, v.flag_t, list_b, c.lane);
                        wset_build(R, reg_a, reg_b, list_b, m, c.lane);
                        n = wset_walk(R, nullptr, v.tmp_a, c.lane);
                    }
                    if (c.lane == 0) s_int[0] = n;
                }
                __syncthreads();
                n = s_int[0];
                __syncthreads();
            }
            if (n == -2) {                                     // tables larger than the LDS area: one thread, global memory
                if (c.tid == 0) {
                    int r = pyset_difference(v.pyset, pyset_capacity(v.cap), v.rows_a, na, v.m_trk, nb, v.tmp_a);
                    if (r < 0) { *v.status = STATUS_TRACK_CAPACITY; r = 0; }
                    s_int[0] = r;
                }
                __syncthreads();
                n = s_int[0];
                __syncthreads();
            }
        }
        if (c.tid == 0) s_int[0] = n;
    }
    __syncthreads();
    const int n_out = s_int[0];
    __syncthreads();
    return n_out;
}

// min_cost_matching (linear_assignment.py:14-79) on `cost` (nr x nc row-major, leading dimension ld), rows = track
// positions `rows`, columns = kept-detection indices `cols`.  Appends the valid matches to (m_trk, m_det) and
// writes the unmatched rows to un_rows (never assigned ascending, then assigned-but-too-costly in row order) and the
// unmatched columns to un_cols likewise.
struct MatchOut { int n_match, n_un_rows, n_un_cols; };
__device__ inline MatchOut ss_min_cost_matching(const Ctx& c, SSV& v, const LsaLds& lsa, const int* rows, int nr, const int* cols, int nc,
                                                long ld, double max_distance, int n_match0, int* un_rows, int* un_cols,
                                                double* dbg = nullptr, int* dbg_shape = nullptr) {
    MatchOut o{n_match0, 0, 0};
    if (dbg_shape && c.tid == 0) { dbg_shape[0] = nr; dbg_shape[1] = nc; }
    auto ident = [](int i) { return i; };
    if (nr == 0 || nc == 0) {
        for (int r = c.tid; r < nr; r += c.nthr) un_rows[r] = rows[r];
        for (int q = c.tid; q < nc; q += c.nthr) un_cols[q] = cols[q];
        __syncthreads();
        o.n_un_rows = nr; o.n_un_cols = nc;
        return o;
    }
    const double clamp = max_distance + 1e-5;
    // the solver works on the matrix with no more rows than columns; it scans one solver row per step, so the transposed case
    // gets a transposed copy (rows of length nr) and every scan reads consecutive addresses
    const bool transposed = nr > nc;
    double* cmT = v.cost_t;
    // 8 x 8 micro-tiles per wave instruction: the read touches 8 full 64-byte lines of `cost`, the transposed write 8 full lines of
    // the copy
    {
        const int rr = c.lane >> 3, qq = c.lane & 7;
        const int tiles_q = (nc + 7) >> 3, tiles = ((nr + 7) >> 3) * tiles_q;
        for (int tile = c.wave; tile < tiles; tile += c.nwaves) {
            const int r = (tile / tiles_q) * 8 + rr, q = (tile % tiles_q) * 8 + qq;
            if (r < nr && q < nc) {
                double x = v.cost[r * ld + q];
                if (dbg) dbg[r * ld + q] = x;
                if (x > max_distance) { x = clamp; v.cost[r * ld + q] = x; }
                if (dbg) dbg[ld * ld + r * ld + q] = x;
                if (transposed) cmT[(long)q * ld + r] = x;
            }
        }
    }
    __syncthreads();
    SS_PROF(12);
    const double* cm = v.cost;
    bool ok;
    if (!transposed) {
        ok = lsa_scipy(c, lsa, nr, nc, [&](int r, int q) { return cm[r * ld + q]; }, v.row_of);       // row_of[r] = column
        for (int q = c.tid; q < nc; q += c.nthr) v.col_of[q] = -1;
        __syncthreads();
        for (int r = c.tid; r < nr; r += c.nthr) if (v.row_of[r] >= 0) v.col_of[v.row_of[r]] = r;
    } else {
        ok = lsa_scipy(c, lsa, nc, nr, [&](int q, int r) { return cmT[(long)q * ld + r]; }, v.col_of);       // col_of[q] = row
        for (int r = c.tid; r < nr; r += c.nthr) v.row_of[r] = -1;
        __syncthreads();
        for (int q = c.tid; q < nc; q += c.nthr) if (v.col_of[q] >= 0) v.row_of[v.col_of[q]] = q;
    }
    __syncthreads();
    SS_PROF(13);
    if (!ok && c.tid == 0) *v.status = STATUS_LAP_STALL;
    auto valid = [&](int r) { return v.row_of[r] >= 0 && !(cm[r * ld + v.row_of[r]] > max_distance); };
    o.n_un_cols = block_append_if(c, nc, [&](int q) { return v.col_of[q] < 0; }, [&](int q) { return cols[q]; }, un_cols, 0);
    o.n_un_rows = block_append_if(c, nr, [&](int r) { return v.row_of[r] < 0; }, [&](int r) { return rows[r]; }, un_rows, 0);
    o.n_un_rows = block_append_if(c, nr, [&](int r) { return v.row_of[r] >= 0 && !valid(r); }, [&](int r) { return rows[r]; }, un_rows, o.n_un_rows);
    o.n_un_cols = block_append_if(c, nr, [&](int r) { return v.row_of[r] >= 0 && !valid(r); }, [&](int r) { return cols[v.row_of[r]]; }, un_cols, o.n_un_cols);
    const int n_new = block_append_if(c, nr, valid, ident, v.tmp_b, 0);
    for (int k = c.tid; k < n_new; k += c.nthr) {
        const int r = v.tmp_b[k];
        v.m_trk[n_match0 + k] = rows[r];
        v.m_det[n_match0 + k] = cols[v.row_of[r]];
    }
    __syncthreads();
    o.n_match = n_match0 + n_new;
    return o;
}

// Per-wavefront passes over a dim-long fp32 vector (lane l owns elements l, l + 64, ...).  Four loads per lane are issued before
// the first use (a plain loop over global memory exposes one round trip per element); the per-lane order of the operations, and
// with it every rounding, is the order of the plain loop.
template <class Load, class Use>
__device__ inline void ss_wave_pass(int lane, int dim, Load load, Use use) {
    for (int e0 = lane; e0 < dim; e0 += 4 * WAVE) {
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = e0 + u * WAVE; x[u] = e < dim ? load(e) : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = e0 + u * WAVE; if (e < dim) use(e, x[u]); }
    }
}
template <class Load1, class Load2, class Use>
__device__ inline void ss_wave_pass2(int lane, int dim, Load1 load1, Load2 load2, Use use) {
    for (int e0 = lane; e0 < dim; e0 += 4 * WAVE) {
        float x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = e0 + u * WAVE; x[u] = e < dim ? load1(e) : 0.f; y[u] = e < dim ? load2(e) : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int e = e0 + u * WAVE; if (e < dim) use(e, x[u], y[u]); }
    }
}

template <int NTHR>
__device__ inline void ss_step_stream(const SsStepArgs& args, int s, int* s_int, double* s_dbl, unsigned char* dyn_lds) {
    if (args.n_dets[s] < 0) {                 // stream not stepped in this call
        if (threadIdx.x == 0) args.out_n[s] = 0;
        return;
    }
    const Ctx c = make_ctx(s_int, s_dbl);
    SS_PROF_DECL();
    SSV v = ss_view(args, s);
    const SsConfigDev& cfg = v.cfg;
    const int big = v.cap > v.nd ? v.cap : v.nd;
    const LsaLds lsa = ss_carve_lsa(dyn_lds, big);
    const long ld = big;
    const int dim = v.dim;
    auto ident = [](int i) { return i; };

    if (c.tid == 0) *v.frame_count += 1;
    // ---- detections with conf >= min_conf (strongsort.py:74-76), tlwh and xyah in fp64 of the fp32 inputs ----
    const int nk = block_append_if(c, v.n_dets, [&](int j) { return (double)v.dets[j * DET_COLS + 4] >= cfg.min_conf; }, ident, v.keep, 0);
    for (int k = c.tid; k < nk; k += c.nthr) {
        const float* d = v.dets + v.keep[k] * DET_COLS;
        const double x1 = d[0], y1 = d[1], w = (double)d[2] - (double)d[0], h = (double)d[3] - (double)d[1];
        double* t = v.det_tlwh + k * 4;
        t[0] = x1; t[1] = y1; t[2] = w; t[3] = h;
        double* z = v.det_xyah + k * 4;                 // Detection.to_xyah, detection.py:35-42
        z[0] = x1 + w / 2; z[1] = y1 + h / 2; z[2] = w / h; z[3] = h;
    }
    int nt = *v.n_tracks;
    __syncthreads();

    // ---- camera_update for every track (strongsort.py:83-86, track.py:139-148); identity unless a warp was supplied ----
    if (nt >= 1) {
        double W[6] = {1, 0, 0, 0, 1, 0};
        if (v.warp) for (int q = 0; q < 6; ++q) W[q] = v.warp[q];
        for (int t = c.tid; t < nt; t += c.nthr) {
            double* m = v.kf + (long)v.list[t] * KF_STRIDE;
            const double ww = m[2] * m[3];
            const double x1 = m[0] - ww / 2, y1 = m[1] - m[3] / 2;
            const double x2 = x1 + ww, y2 = y1 + m[3];
            const double x1_ = (W[0] * x1 + W[1] * y1) + W[2] * 1.0, y1_ = (W[3] * x1 + W[4] * y1) + W[5] * 1.0;
            const double x2_ = (W[0] * x2 + W[1] * y2) + W[2] * 1.0, y2_ = (W[3] * x2 + W[4] * y2) + W[5] * 1.0;
            const double w = x2_ - x1_, h = y2_ - y1_;
            m[0] = x1_ + w / 2; m[1] = y1_ + h / 2; m[2] = w / h; m[3] = h;
        }
        __syncthreads();
    }

    // ---- predict (tracker.py:63-69, track.py:154-160), wave per track ----
    for (int base = 0; base < nt; base += c.nwaves) {
        const int t = base + c.wave;
        if (t < nt) {
            const int slot = v.list[t];
            ss_kf_predict_wave(v.kf + (long)slot * KF_STRIDE, c.lane);
            if (c.lane == 0) { v.age[slot] += 1; v.tsu[slot] += 1; }
        }
    }
    __syncthreads();

    SS_PROF(0);
    // ---- stage A: confirmed tracks vs all detections by gated appearance (tracker.py:107-139) ----
    const int n_conf = block_append_if(c, nt, [&](int t) { return v.state[v.list[t]] == SS_CONFIRMED; }, ident, v.rows_a, 0);
    for (int k = c.tid; k < nk; k += c.nthr) v.tmp_a[k] = k;            // detection_indices = range(len(detections))
    __syncthreads();
    if (n_conf > 0 && nk > 0) {
        for (int r = c.wave; r < n_conf; r += c.nwaves) {          // a wavefront per track: the gate's factor once, lanes over detections
            const int t = v.rows_a[r];
            SsGate g;
            ss_gating_factor(v.kf + (long)v.list[t] * KF_STRIDE, g);
            for (int k = c.lane; k < nk; k += WAVE) {
                double cst = (double)v.app[(long)t * v.nd + v.keep[k]];
                const double gd = ss_gating_apply(g, v.det_xyah + k * 4);
                if (gd > SS_CHI2_4) cst = SS_INFTY_COST;
                v.cost[r * ld + k] = cfg.mc_lambda * cst + (1 - cfg.mc_lambda) * gd;
            }
        }
        __syncthreads();
    }
    SS_PROF(1);
    // un_rows of stage A -> tmp list (flag_t reused as storage for unmatched confirmed rows)
    double* dbg = args.dbg_cost ? args.dbg_cost + (long)s * 4 * ld * ld : nullptr;          // parity debugging only
    int* dbg_shape = args.dbg_cost ? args.dbg_shape + s * 4 : nullptr;
    MatchOut a = ss_min_cost_matching(c, v, lsa, v.rows_a, n_conf, v.tmp_a, nk, ld, cfg.max_cos_dist, 0, v.flag_t, v.un_d, dbg, dbg_shape);
    SS_PROF(3);
    // unmatched confirmed tracks = list(set(confirmed) - set(matched)) in CPython's set order (ss_unmatched_in_set_order)
    s_int[0] = ss_unmatched_in_set_order(c, v, n_conf, a.n_match, nt, big, dyn_lds);
    __syncthreads();
    const int n_un_a = s_int[0];
    __syncthreads();
    SS_PROF(4);
    // ---- stage B: unconfirmed + just-missed confirmed tracks vs the remaining detections by IoU (tracker.py:141-158) ----
    int n_b = block_append_if(c, nt, [&](int t) { return v.state[v.list[t]] != SS_CONFIRMED; }, ident, v.rows_b, 0);
    n_b = block_append_if(c, n_un_a, [&](int q) { return v.tsu[v.list[v.tmp_a[q]]] == 1; }, [&](int q) { return v.tmp_a[q]; }, v.rows_b, n_b);
    const int n_stale = block_append_if(c, n_un_a, [&](int q) { return v.tsu[v.list[v.tmp_a[q]]] != 1; }, [&](int q) { return v.tmp_a[q]; },
                                        v.tmp_b, 0);                       // unmatched_tracks_a after the split
    for (int q = c.tid; q < n_stale; q += c.nthr) v.flag_t[q] = v.tmp_b[q];
    __syncthreads();
    for (int q = c.tid; q < n_stale; q += c.nthr) v.tmp_a[q] = v.flag_t[q];
    for (int q = c.tid; q < a.n_un_cols; q += c.nthr) v.cols_b[q] = v.un_d[q];
    __syncthreads();
    const int n_cb = a.n_un_cols;
    if (n_b > 0 && n_cb > 0) {
        for (int r = c.wave; r < n_b; r += c.nwaves)
        for (int q = c.lane; q < n_cb; q += WAVE) {
            const int slot = v.list[v.rows_b[r]];
            double cst;
            if (v.tsu[slot] > 1) cst = SS_INFTY_COST;
            else {
                const double* m = v.kf + (long)slot * KF_STRIDE;
                const double bw = m[2] * m[3], bx = m[0] - bw / 2, by = m[1] - m[3] / 2, bh = m[3];      // to_tlwh
                const double* d = v.det_tlwh + v.cols_b[q] * 4;
                const double tlx = bx > d[0] ? bx : d[0], tly = by > d[1] ? by : d[1];
                const double brx = (bx + bw) < (d[0] + d[2]) ? (bx + bw) : (d[0] + d[2]);
                const double bry = (by + bh) < (d[1] + d[3]) ? (by + bh) : (d[1] + d[3]);
                double iw = brx - tlx, ih = bry - tly;
                iw = iw > 0.0 ? iw : 0.0; ih = ih > 0.0 ? ih : 0.0;
                const double inter = iw * ih;
                cst = 1.0 - inter / (bw * bh + d[2] * d[3] - inter);
            }
            v.cost[r * ld + q] = cst;
        }
        __syncthreads();
    }
    MatchOut b = ss_min_cost_matching(c, v, lsa, v.rows_b, n_b, v.cols_b, n_cb, ld, cfg.max_iou_dist, a.n_match, v.flag_t, v.un_d,
                                      dbg ? dbg + 2 * ld * ld : nullptr, dbg ? dbg_shape + 2 : nullptr);
    const int n_match = b.n_match;

    SS_PROF(5);
    // ---- Track.update for the matches (track.py:162-189), wave per match ----
    for (int base = 0; base < n_match; base += c.nwaves) {
        const int q = base + c.wave;
        if (q < n_match) {
            const int slot = v.list[v.m_trk[q]], k = v.m_det[q], j = v.keep[k];
            const float* d = v.dets + j * DET_COLS;
            const bool ok = ss_kf_update_wave(v.kf + (long)slot * KF_STRIDE, v.det_xyah + k * 4, (double)d[4], c.lane);
            // feature = det / |det|; smooth = alpha * features[-1] + (1 - alpha) * feature; smooth /= |smooth|   (all fp32)
            const float* de = v.embs + (long)j * dim;
            float* tf = v.feat + (long)slot * dim;
            float ss = 0.f;
            ss_wave_pass(c.lane, dim, [&](int e) { return de[e]; }, [&](int, float x) { ss = fmaf(x, x, ss); });
            const float dn = sqrtf(wave_sum(ss));
            float s2 = 0.f;
            ss_wave_pass2(c.lane, dim, [&](int e) { return tf[e]; }, [&](int e) { return de[e]; }, [&](int e, float t0, float x) {
                const float sm = cfg.ema_alpha_f32 * t0 + cfg.one_minus_alpha_f32 * (x / dn);
                tf[e] = sm;
                s2 = fmaf(sm, sm, s2);
            });
            const float sn = sqrtf(wave_sum(s2));
            ss_wave_pass(c.lane, dim, [&](int e) { return tf[e]; }, [&](int e, float x) { tf[e] = x / sn; });
            if (c.lane == 0) {
                if (!ok) *v.status = STATUS_LAP_STALL + 1;
                v.conf[slot] = d[4]; v.cls[slot] = d[5]; v.det_ind[slot] = (float)j;
                v.hits[slot] += 1; v.tsu[slot] = 0;
                if (v.state[slot] == SS_TENTATIVE && v.hits[slot] >= cfg.n_init) v.state[slot] = SS_CONFIRMED;
            }
        }
    }
    SS_PROF(6);
    // ---- mark_missed (track.py:191-196): stale confirmed, unmatched of stage B ----
    auto missed = [&](int t) {
        const int slot = v.list[t];
        if (v.state[slot] == SS_TENTATIVE) v.state[slot] = SS_DELETED;
        else if (v.tsu[slot] > cfg.max_age) v.state[slot] = SS_DELETED;
    };
    for (int q = c.tid; q < n_stale; q += c.nthr) missed(v.tmp_a[q]);
    for (int q = c.tid; q < b.n_un_rows; q += c.nthr) missed(v.flag_t[q]);
    __syncthreads();

    // ---- births for the unmatched detections, in their order (tracker.py:92-93, :159-169; track.py:72-109) ----
    int n_new = b.n_un_cols;
    if (n_new > 0) {
        const int n_free = block_append_if(c, v.cap, [&](int sl) { return v.slot_used[sl] == 0; }, ident, v.tmp_b, 0);
        if (n_free < n_new) { if (c.tid == 0) *v.status = STATUS_TRACK_CAPACITY; n_new = n_free; }
        const int id0 = *v.next_id;
        __syncthreads();
        for (int base = 0; base < n_new; base += c.nwaves) {
            const int q = base + c.wave;
            if (q < n_new) {
                const int slot = v.tmp_b[q], k = v.un_d[q], j = v.keep[k];
                const float* d = v.dets + j * DET_COLS;
                const double* z = v.det_xyah + k * 4;
                const int i = c.lane >> 3, jj = c.lane & 7;
                double p = 0.0;
                if (i == jj) {
                    double sd;
                    if (i == 2) sd = 1e-2; else if (i == 6) sd = 1e-5;
                    else sd = (i < 4) ? 2 * SS_STD_POS * z[3] : 10 * SS_STD_VEL * z[3];
                    p = sd * sd;
                }
                v.kf[(long)slot * KF_STRIDE + KF_DIM + c.lane] = p;
                if (c.lane < 8) {
                    double mv = c.lane < 4 ? z[c.lane] : 0.0;
                    if (c.lane == 2 || c.lane == 3) mv = mv > 1e-4 ? mv : 1e-4;
                    v.kf[(long)slot * KF_STRIDE + c.lane] = mv;
                }
                const float* de = v.embs + (long)j * dim;
                float ss = 0.f;
                ss_wave_pass(c.lane, dim, [&](int e) { return de[e]; }, [&](int, float x) { ss = fmaf(x, x, ss); });
                const float dn = sqrtf(wave_sum(ss));
                float* nf = v.feat + (long)slot * dim;
                ss_wave_pass(c.lane, dim, [&](int e) { return de[e]; }, [&](int e, float x) { nf[e] = x / dn; });
                if (c.lane == 0) {
                    v.slot_used[slot] = 1;
                    v.id[slot] = id0 + q;
                    v.state[slot] = SS_TENTATIVE; v.hits[slot] = 1; v.age[slot] = 1; v.tsu[slot] = 0; v.bank_n[slot] = 0;
                    v.conf[slot] = d[4]; v.cls[slot] = d[5]; v.det_ind[slot] = (float)j;
                    v.list[nt + q] = slot;
                }
            }
        }
        __syncthreads();
        nt += n_new;
        if (c.tid == 0) *v.next_id = id0 + n_new;
        __syncthreads();
    }

    SS_PROF(7);
    // ---- drop deleted tracks (tracker.py:94), then feed the sample bank of every confirmed track (:96-106) ----
    const int n_live = block_append_if(c, nt, [&](int t) { return v.state[v.list[t]] != SS_DELETED; }, [&](int t) { return v.list[t]; }, v.tmp_a, 0);
    if (n_live != nt) {
        for (int t = c.tid; t < nt; t += c.nthr) if (v.state[v.list[t]] == SS_DELETED) v.slot_used[v.list[t]] = 0;
        __syncthreads();
        for (int t = c.tid; t < n_live; t += c.nthr) v.list[t] = v.tmp_a[t];
        __syncthreads();
    }
    nt = n_live;
    if (c.tid == 0) *v.n_tracks = nt;
    for (int base = 0; base < nt; base += c.nwaves) {
        const int t = base + c.wave;
        if (t < nt) {
            const int slot = v.list[t];
            if (v.state[slot] == SS_CONFIRMED) {
                const int bn = __shfl(v.bank_n[slot], 0, WAVE);       // every lane has read the count before lane 0 bumps it
                float* dst = v.bank + ((long)slot * v.budget + bn % v.budget) * dim;
                float ss = 0.f;
                const float* sf = v.feat + (long)slot * dim;
                ss_wave_pass(c.lane, dim, [&](int e) { return sf[e]; }, [&](int e, float f) { dst[e] = f; ss = fmaf(f, f, ss); });
                ss = wave_sum(ss);
                if (c.lane == 0) { v.bank_norm[(long)slot * v.budget + bn % v.budget] = sqrtf(ss); v.bank_n[slot] = bn + 1; }
            }
        }
    }
    __syncthreads();

    SS_PROF(8);
    // ---- output rows (strongsort.py:103-123): confirmed and updated this frame, list order ----
    const int n_out = block_append_if(c, nt, [&](int t) { return v.state[v.list[t]] == SS_CONFIRMED && v.tsu[v.list[t]] < 1; },
                                      [&](int t) { return v.list[t]; }, v.tmp_a, 0);
    for (int q = c.tid; q < n_out; q += c.nthr) {
        const int slot = v.tmp_a[q];
        const double* m = v.kf + (long)slot * KF_STRIDE;
        const double w = m[2] * m[3], x1 = m[0] - w / 2, y1 = m[1] - m[3] / 2;
        float* o = v.out + q * OUT_COLS;
        o[0] = (float)x1; o[1] = (float)y1; o[2] = (float)(x1 + w); o[3] = (float)(y1 + m[3]);
        o[4] = (float)v.id[slot]; o[5] = v.conf[slot]; o[6] = v.cls[slot]; o[7] = v.det_ind[slot];
    }
    if (c.tid == 0) *v.out_n = n_out;
    __syncthreads();
    SS_PROF(9);
}

}  // namespace bm
