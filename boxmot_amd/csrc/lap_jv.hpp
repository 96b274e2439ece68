// Dense Jonker-Volgenant assignment on the device, tie-for-tie.
//
// `lap.lapjv(cost, extend_cost=True[, cost_limit=L])` (boxmot/trackers/association/association.py:20-24, matching.py:28-43)
// is the third-party lapx solver: column reduction + reduction transfer, two rounds of augmenting row reduction, then
// shortest-augmenting-path augmentation on the (n_rows + n_cols)^2 extended matrix (R. Jonker, A. Volgenant, Computing 38,
// 1987).  Every exact solver returns the same assignment when the optimum is unique; when it is not -- many zero-IoU pairs at
// cost 0, clamped costs -- WHICH optimal assignment comes back decides, through the order of the unmatched lists, which
// detection gets which new id.  So the device does not just solve the problem: it executes the same algorithm with the same
// scan orders, comparisons and list manipulations, and returns the assignment the sequential code returns.
//
// Parallel where the sequential order cannot be observed, sequential where it can:
//   * column minima (n^2 evaluations): one thread per column, rows in ascending order (first minimum wins, as the scan does);
//   * the column -> row claim ("last column claims the row"): an atomic max per row;
//   * every O(n) scan of a row (reduction transfer, the two-smallest search of the row reduction, the relaxation of the
//     augmentation) runs on the 64 lanes of the first wavefront with order-exact reductions: lexicographic (value, index)
//     minima reproduce "first index wins"; the TODO-list permutations of the augmentation (`cols`) are replayed in index order
//     from a ballot of the lanes whose element the sequential loop would have moved;
//   * the control flow between scans (free-row list, the `rr_cnt` guard, path update) is scalar and uniform.
// The extended matrix is never materialised: e(i, j) = cost (i < n_rows, j < n_cols), 0 (both beyond), fill otherwise, with
// fill = L / 2 or max(cost) + 1.
#pragma once

#include "block_prims.hpp"
#include "kernel_macros.hpp"

#ifndef BM_JV_PROF
#define BM_JV_PROF(k)            // tools/jv_prof.hip defines it to record a wall clock per phase
#endif
#ifndef BM_JV_T
#define BM_JV_T_DECL
#define BM_JV_T(k)               // tools/jv_prof.hip -DJV_FINE: shader clocks per segment of a row-reduction iteration / a fast augmentation
#endif

namespace bm {

constexpr double JV_LARGE = 1.7976931348623157e308;          // DBL_MAX, the sequential code's LARGE

struct JvLds { double* v; double* d; int* x; int* y; int* pred; int* cols; int* free_rows; };
__host__ __device__ inline long jv_lds_bytes(int n) { return (long)n * (8 + 8 + 5 * 4) + 16; }
__device__ inline JvLds jv_carve(unsigned char* base, int n) {
    JvLds l;
    l.v = reinterpret_cast<double*>(base); l.d = l.v + n;
    l.x = reinterpret_cast<int*>(l.d + n); l.y = l.x + n; l.pred = l.y + n; l.cols = l.pred + n; l.free_rows = l.cols + n;
    return l;
}

// Wave reductions on the DPP path (row rotations inside the 16-lane rows, then the four row results as scalars): the sequential
// part of the solver is a chain of a few thousand of these, and the LDS crossbar (__shfl) costs several times as much.
template <int CTRL> __device__ inline double jv_dpp(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = BM_DPP_U32(0u, (unsigned)b, CTRL, false), hi = BM_DPP_U32(0u, (unsigned)(b >> 32), CTRL, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
template <int CTRL> __device__ inline int jv_dpp(int v) { return (int)BM_DPP_U32(0u, (unsigned)v, CTRL, false); }
template <int LANE> __device__ inline double jv_lane(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = BM_READLANE_U32((unsigned)b, LANE), hi = BM_READLANE_U32((unsigned)(b >> 32), LANE);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
template <int LANE> __device__ inline int jv_lane(int v) { return (int)BM_READLANE_U32((unsigned)v, LANE); }

__device__ inline double jv_min2(double a, double b) { return b < a ? b : a; }
__device__ inline double jv_wave_min(double v) {
    v = jv_min2(v, jv_dpp<0x128>(v)); v = jv_min2(v, jv_dpp<0x124>(v)); v = jv_min2(v, jv_dpp<0x122>(v)); v = jv_min2(v, jv_dpp<0x121>(v));
    return jv_min2(jv_min2(jv_lane<0>(v), jv_lane<16>(v)), jv_min2(jv_lane<32>(v), jv_lane<48>(v)));
}
__device__ inline int jv_wave_min(int v) {
    int o;
    o = jv_dpp<0x128>(v); v = o < v ? o : v; o = jv_dpp<0x124>(v); v = o < v ? o : v;
    o = jv_dpp<0x122>(v); v = o < v ? o : v; o = jv_dpp<0x121>(v); v = o < v ? o : v;
    const int a = jv_lane<0>(v), b = jv_lane<16>(v), c = jv_lane<32>(v), d = jv_lane<48>(v);
    const int ab = b < a ? b : a, cd = d < c ? d : c;
    return cd < ab ? cd : ab;
}
// The two smallest (value, column) pairs of a row scan, lexicographic -- "first column wins" among equal values, as the sequential
// scan keeps them.  A lane visits its columns in ascending order, so inside a lane strict comparisons keep the earlier column
// (the form of the sequential loop); across the lanes: the minimum value, the lowest column holding it, then the same again
// with the winning lane offering its runner-up.  (A NaN cost never enters: every comparison with it is false.)
struct JvTwo { double v1, v2; int j1, j2; };
constexpr int JV_NO_COL = 0x7fffffff;
__device__ inline void jv_two_insert(JvTwo& t, double r, int j) {         // (selects: no divergent branches in the scans)
    const bool lt1 = r < t.v1, lt2 = r < t.v2;
    t.v2 = lt1 ? t.v1 : (lt2 ? r : t.v2);
    t.j2 = lt1 ? t.j1 : (lt2 ? j : t.j2);
    t.v1 = lt1 ? r : t.v1;
    t.j1 = lt1 ? j : t.j1;
}
__device__ inline JvTwo jv_wave_two(const JvTwo& t) {
    JvTwo o;
    o.v1 = jv_wave_min(t.v1);
    o.j1 = jv_wave_min(t.v1 == o.v1 ? t.j1 : JV_NO_COL);
    const bool own = t.j1 == o.j1;
    const double cv = own ? t.v2 : t.v1;
    const int cj = own ? t.j2 : t.j1;
    o.v2 = jv_wave_min(cv);
    o.j2 = jv_wave_min(cv == o.v2 ? cj : JV_NO_COL);
    return o;
}

// The same for lanes that each scanned a contiguous, ascending run of columns (lane l's columns all precede lane l + 1's): the
// lowest lane holding the minimum holds its first column.
__device__ inline JvTwo jv_wave_two_contig(const JvTwo& t, int lane) {
    JvTwo o;
    o.v1 = jv_wave_min(t.v1);
    const unsigned long long b1 = __ballot(t.j1 != JV_NO_COL && t.v1 == o.v1);
    const int l1 = b1 ? __builtin_ctzll(b1) : 0;
    o.j1 = b1 ? (int)BM_READLANE_U32((unsigned)t.j1, l1) : JV_NO_COL;
    const bool own = b1 && lane == l1;
    const double cv = own ? t.v2 : t.v1;
    const int cj = own ? t.j2 : t.j1;
    o.v2 = jv_wave_min(cv);
    const unsigned long long b2 = __ballot(cj != JV_NO_COL && cv == o.v2);
    o.j2 = b2 ? (int)BM_READLANE_U32((unsigned)cj, __builtin_ctzll(b2)) : JV_NO_COL;
    return o;
}

// The TODO-list permutation of _find_dense / _scan_dense for the lanes [l0, seg_end) of one 64-column chunk at list position
// `base`, in one step.  The sequential loop visits the positions in order; a position whose column joins the ready run (bit set
// in `q`) is swapped with position `hi`, which then advances.  Seen from the list, the positions [hi, k) are a queue of passed-over
// columns: a passed-over column is appended, a joining column sends the queue's head to its tail.  Every visited position is
// therefore one push; the t-th joining column takes list position hi + t and pushes what the queue's t-th pop returns -- the
// t-th push overall (one of the g0 columns queued before this chunk, or the push of lane l0 + t - g0, itself possibly a
// re-push: resolved by pointer jumping).  The queue ends as pushes r.. at positions hi + r.., so a passed-over column keeps its
// place unless a joining column lands on it.  A leading run of joining columns on an empty queue swaps with itself.
__device__ inline void jv_move(const JvLds& L, int base, int l0, unsigned long long q, int& hi, int j, int lane) {
    int g0 = base + l0 - hi;
    if (g0 == 0) {
        const unsigned long long rest = ~(q >> l0);
        const int run = rest ? __builtin_ctzll(rest) : WAVE;
        hi += run; l0 += run;
        q = l0 >= WAVE ? 0ull : (q >> l0) << l0;
    }
    const int r = __builtin_popcountll(q);
    if (r == 0) return;
    const int H = hi;
    const bool isq = (q >> lane) & 1ull;
    const int t = __builtin_popcountll(q & ((1ull << lane) - 1ull));
    bool resolved = true;
    int val = j, ptr = lane;
    if (isq) {
        if (t < g0) val = L.cols[H + t];
        else { resolved = false; ptr = l0 + t - g0; }
    }
    while (__ballot(!resolved)) {
        const int pv = __shfl(val, ptr, WAVE), pr = __shfl((int)resolved, ptr, WAVE), pp = __shfl(ptr, ptr, WAVE);
        if (!resolved) { if (pr) { val = pv; resolved = true; } else ptr = pp; }
    }
    BM_WAVE_LDS_SYNC();                              // every old list entry is read before the first is overwritten
    if (isq) {
        L.cols[H + t] = j;
        if (g0 + lane - l0 >= r) L.cols[base + lane] = val;
    }
    BM_WAVE_LDS_SYNC();
    hi = H + r;
}

// n_rows x n_cols problem, cost_of(i, j).  out_col_of_row[i] (x) and out_row_of_col[j] (y), -1 = unassigned, as lapx returns
// them.  Whole workgroup calls; returns false (uniform) if an iteration guard tripped.
template <class CostFn>
__device__ inline bool lap_jv_extended(const Ctx& c, const JvLds& L, int n_rows, int n_cols, CostFn cost_of, bool use_limit, double limit,
                                       int* out_col_of_row, int* out_row_of_col) {
    const int n = n_rows + n_cols;
    if (n_rows == 0 || n_cols == 0) {
        for (int i = c.tid; i < n_rows; i += c.nthr) out_col_of_row[i] = -1;
        for (int j = c.tid; j < n_cols; j += c.nthr) out_row_of_col[j] = -1;
        __syncthreads();
        return true;
    }
    // ---- fill value of the extension ----
    double fill;
    if (use_limit) fill = limit / 2.0;
    else {
        double mx = -JV_LARGE;
        for (int k = c.tid; k < n_rows * n_cols; k += c.nthr) { const double e = cost_of(k / n_cols, k % n_cols); mx = e > mx ? e : mx; }
        for (int off = WAVE / 2; off > 0; off >>= 1) { const double o = __shfl_xor(mx, off, WAVE); mx = o > mx ? o : mx; }
        if (c.lane == 0) c.s_dbl[c.wave] = mx;
        __syncthreads();
        mx = c.s_dbl[0];
        for (int w = 1; w < c.nwaves; ++w) mx = c.s_dbl[w] > mx ? c.s_dbl[w] : mx;
        __syncthreads();
        fill = mx + 1.0;
    }
    auto e_of = [&](int i, int j) -> double {
        if (i < n_rows) return j < n_cols ? cost_of(i, j) : fill;
        return j < n_cols ? fill : 0.0;
    };
    BM_JV_PROF(0);
    // ---- column reduction (_ccrrt_dense): column minima, first row wins ----
    for (int i = c.tid; i < n; i += c.nthr) { L.x[i] = -1; L.pred[i] = 0; }
    for (int j = c.tid; j < n; j += c.nthr) {
        double vj = JV_LARGE;
        int yj = 0;
        for (int i0 = 0; i0 < n_rows; i0 += 4) {                  // four independent loads in flight
            double e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e[u] = i0 + u < n_rows ? e_of(i0 + u, j) : JV_LARGE;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + u < n_rows && e[u] < vj) { vj = e[u]; yj = i0 + u; }
        }
        const double ed = j < n_cols ? fill : 0.0;               // rows n_rows.. all hold this value: the first of them wins, if any
        if (ed < vj) { vj = ed; yj = n_rows; }
        L.v[j] = vj; L.y[j] = yj;
    }
    __syncthreads();
    // the descending column loop: the LAST column of a row's claimants keeps it, the others lose their row; a row claimed once is "unique"
    for (int j = c.tid; j < n; j += c.nthr) { const int i = L.y[j]; atomicMax(&L.x[i], j); atomicAdd(&L.pred[i], 1); }
    __syncthreads();
    for (int j = c.tid; j < n; j += c.nthr) if (L.x[L.y[j]] != j) L.y[j] = -1;
    __syncthreads();
    bool ok = true;
    BM_JV_T_DECL
    BM_JV_PROF(1);
    if (c.wave == 0) {
        const int lane = c.lane;
        // contiguous column run of a lane (row reduction, augmentation start); an odd run length keeps the 8-byte LDS accesses of the
        // lanes (stride = run length) on distinct banks
        const int cpl = ((n + WAVE - 1) / WAVE) | 1;
        const int jbeg = lane * cpl < n ? lane * cpl : n, jend = jbeg + cpl < n ? jbeg + cpl : n;
        // r(j) = e(i, j) - v[j] over the lane's run, eight columns' loads in flight at a time (the scans are latency chains otherwise)
        // Branch-free (clamped addresses, selects): a column outside the run visits as r = LARGE with a row, which no comparison below
        // lets win; the row kind (cost row | extension row: fill | 0) is a uniform branch.
        auto scan_run = [&](int i, auto&& visit) {
            const bool ext = i >= n_rows;
            for (int jb = jbeg; jb < jend; jb += 8) {
                double e[8], vv[8];
                int yy[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = jb + u, jc = j < jend ? j : jend - 1;
                    vv[u] = L.v[jc]; yy[u] = L.y[jc];
                    if (ext) e[u] = jc < n_cols ? fill : 0.0;
                    else { const double ce = cost_of(i, jc < n_cols ? jc : n_cols - 1); e[u] = jc < n_cols ? ce : fill; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const bool in = jb + u < jend; visit(jb + u, in ? e[u] - vv[u] : JV_LARGE, in ? yy[u] : 0); }
            }
        };
        // ---- free rows (ascending) and reduction transfer, row by row: a transfer changes v for the rows after it ----
        int n_free = 0;
        for (int i = 0; i < n; ++i) {
            const int xi = L.x[i];
            if (xi < 0) { if (lane == 0) L.free_rows[n_free] = i; ++n_free; }
            else if (L.pred[i] == 1) {
                double mn = JV_LARGE;
                for (int jb = lane; jb < n; jb += 4 * WAVE) {
                    double e[4], vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int j = jb + u * WAVE; const bool in = j < n; e[u] = in ? e_of(i, j) : 0.0; vv[u] = in ? L.v[j] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int j = jb + u * WAVE; if (j < n && j != xi) { const double r = e[u] - vv[u]; mn = r < mn ? r : mn; } }
                }
                mn = jv_wave_min(mn);
                if (lane == 0) L.v[xi] -= mn;
                BM_WAVE_LDS_SYNC();
            }
        }
        BM_WAVE_LDS_SYNC();
        BM_JV_PROF(2);
        // ---- augmenting row reduction (_carr_dense), twice ----
        for (int round = 0; round < 2 && n_free > 0; ++round) {
            int current = 0, new_free = 0;
            long rr_cnt = 0;
            const long guard_max = 4L * n * n + 64;
            long guard = 0;
            while (current < n_free) {
                if (++guard > guard_max) { ok = false; break; }
                ++rr_cnt;
                BM_JV_T(0);
                const int fi = L.free_rows[current++];
                // each lane scans a CONTIGUOUS run of columns, ascending: "first column wins" across the lanes is then "lowest lane
                // wins" -- one value reduction + a ballot per rank instead of a (value, column) reduction pair
                JvTwo t{JV_LARGE, JV_LARGE, JV_NO_COL, JV_NO_COL};
                scan_run(fi, [&](int j, double r, int) { jv_two_insert(t, r, j); });
                BM_JV_T(1);
                t = jv_wave_two_contig(t, lane);
                BM_JV_T(2);
                int j1 = t.j1, j2 = n > 1 ? t.j2 : -1;
                const double v1 = t.v1, v2 = n > 1 ? t.v2 : JV_LARGE;
                int i0 = L.y[j1];
                const int i0_second = j2 >= 0 ? L.y[j2] : -1;     // read before lane 0 writes y (lanes run in lockstep on the device, not in the CPU harness)
                const double vj1 = L.v[j1];
                const double v1_new = vj1 - (v2 - v1);
                const bool lowers = v1_new < vj1;
                BM_WAVE_LDS_SYNC();
                if (rr_cnt < (long)current * n) {
                    if (lowers) { if (lane == 0) L.v[j1] = v1_new; }
                    else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = i0_second; }
                    if (i0 >= 0) {
                        if (lowers) { --current; if (lane == 0) L.free_rows[current] = i0; }
                        else { if (lane == 0) L.free_rows[new_free] = i0; ++new_free; }
                    }
                } else if (i0 >= 0) {
                    if (lane == 0) L.free_rows[new_free] = i0;
                    ++new_free;
                }
                if (lane == 0) { L.x[fi] = j1; L.y[j1] = fi; }
                BM_WAVE_LDS_SYNC();
                BM_JV_T(3);
            }
            n_free = new_free;
        }
        BM_JV_PROF(3);
        // ---- augmentation (_ca_dense): shortest augmenting path per remaining free row ----
        for (int f = 0; f < n_free && ok; ++f) {
            const int fi = L.free_rows[f];
            {
                // The first _find_dense of a search starts from the identity list: its ready run is every column at the minimum of
                // d = e(fi, .) - v in ascending order (a restart discards what was gathered before the first occurrence of the final
                // minimum), and the search ends there if one of them has no row -- the LAST such column (the loop over the run keeps
                // overwriting).  That is the whole augmentation then (no dual update: n_ready = 0; the path is the single edge):
                // two reductions instead of the list machinery.  DeepOCSORT's recovery rounds (a few rows against hundreds of
                // columns, nearly all pairs tied at 0) are ~400 such searches per call.
                double dm = JV_LARGE;
                int last_free = -1;              // the lane's last column without a row among those at its own minimum
                scan_run(fi, [&](int j, double r, int yj) {
                    const bool lt = r < dm, fr = yj < 0;
                    last_free = lt ? (fr ? j : -1) : ((r == dm && fr) ? j : last_free);
                    dm = lt ? r : dm;
                });
                const double m = jv_wave_min(dm);
                if (dm != m) last_free = -1;
                const unsigned long long fb = __ballot(last_free >= 0);
                if (fb) {
                    const int fj = (int)BM_READLANE_U32((unsigned)last_free, 63 - __builtin_clzll(fb));
                    BM_WAVE_LDS_SYNC();
                    if (lane == 0) { L.x[fi] = fj; L.y[fj] = fi; }
                    BM_WAVE_LDS_SYNC();
                    continue;
                }
            }
            for (int jb = lane; jb < n; jb += 4 * WAVE) {
                double e[4], vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int j = jb + u * WAVE; const bool in = j < n; e[u] = in ? e_of(fi, j) : 0.0; vv[u] = in ? L.v[j] : 0.0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int j = jb + u * WAVE; if (j < n) { L.cols[j] = j; L.pred[j] = fi; L.d[j] = e[u] - vv[u]; } }
            }
            BM_WAVE_LDS_SYNC();
            int lo = 0, hi = 0, final_j = -1, n_ready = 0;
            long guard = 0;
            // A pop whose row is an EXTENSION row (all of them are the same row: fill | 0) relaxes with cred(j) = e_ext(j) - v[j] - h;
            // if such a pop changed nothing (no d[j] lowered), any later one with the same h changes nothing either while d is
            // unchanged (v is fixed during a search and the TODO list only shrinks): it is `lo++` and nothing else.  With thousands
            // of tied ready columns these are 95 - 98 % of all pops (profiles/r4_jv_prof.txt); they are recognised 64 at a time.
            bool memo_ok = false;
            double memo_h = 0.0;
            while (final_j == -1) {
                if (++guard > 4L * n + 64) { ok = false; break; }
                if (lo == hi) {
                    // _find_dense: move the columns at the minimum of d over cols[lo..n) to the front, in scan order.  A column
                    // below the running minimum restarts the ready run at `lo` (the earlier run stays where it was put).
                    n_ready = lo;
                    hi = lo + 1;
                    double mind = L.d[L.cols[lo]];
                    for (int base0 = hi; base0 < n; base0 += 4 * WAVE) {
                        int jj[4];
                        double ss[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int k = base0 + u * WAVE + lane; jj[u] = k < n ? L.cols[k] : 0; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) ss[u] = base0 + u * WAVE + lane < n ? L.d[jj[u]] : JV_LARGE;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int base = base0 + u * WAVE;
                            if (base >= n) break;
                            const bool in = base + lane < n;
                            const double s = ss[u];
                            int l0 = 0;
                            while (true) {
                                const unsigned long long from = l0 >= WAVE ? 0ull : ~0ull << l0;
                                if (!(__ballot(in && s <= mind) & from)) break;
                                const unsigned long long lt = __ballot(in && s < mind) & from;
                                const int seg_end = lt ? __builtin_ctzll(lt) : WAVE;
                                const unsigned long long upto = seg_end >= WAVE ? ~0ull : (1ull << seg_end) - 1ull;
                                jv_move(L, base, l0, __ballot(in && s == mind) & from & upto, hi, jj[u], lane);
                                if (!lt) break;
                                mind = __shfl(s, seg_end, WAVE);
                                hi = lo;
                                l0 = seg_end;
                            }
                        }
                    }
                    // the last ready column without a row ends the search
                    int last = -1;
                    for (int kb = lo; kb < hi; kb += WAVE) {
                        const int k = kb + lane;
                        const unsigned long long fr = __ballot(k < hi && L.y[L.cols[k]] < 0);
                        if (fr) last = kb + 63 - __builtin_clzll(fr);
                    }
                    if (last >= 0) final_j = L.cols[last];
                }
                if (final_j == -1) {
                    // _scan_dense: relax from the ready columns; columns that reach the minimum join the ready set
                    int found = -1;
                    while (lo != hi && found < 0) {
                        if (memo_ok) {
                            const int kq = lo + lane;
                            bool hit = false;
                            if (kq < hi) {
                                const int jq = L.cols[kq], iq = L.y[jq];
                                if (iq >= n_rows) hit = e_of(iq, jq) - L.v[jq] - L.d[jq] == memo_h;
                            }
                            const unsigned long long miss = ~__ballot(hit);
                            const int run = miss ? __builtin_ctzll(miss) : WAVE;
                            if (run > 0) { lo += run; continue; }
                        }
                        bool any_lower = false;
                        const int j0 = L.cols[lo++];
                        const int i = L.y[j0];
                        const double mind = L.d[j0];
                        const double h = e_of(i, j0) - L.v[j0] - mind;
                        const int hi0 = hi;
                        for (int base0 = hi0; base0 < n && found < 0; base0 += 4 * WAVE) {
                            // four chunks of the TODO list in flight: the swaps of a chunk touch positions at or before it only
                            int jj[4], yy[4];
                            double e[4], vv[4], dd[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) { const int k = base0 + u * WAVE + lane; jj[u] = k < n ? L.cols[k] : 0; }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const bool in = base0 + u * WAVE + lane < n;
                                e[u] = in ? e_of(i, jj[u]) : 0.0; vv[u] = in ? L.v[jj[u]] : 0.0; dd[u] = in ? L.d[jj[u]] : 0.0; yy[u] = in ? L.y[jj[u]] : 0;
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int base = base0 + u * WAVE;
                                if (base >= n || found >= 0) break;
                                const bool in = base + lane < n;
                                const int j = jj[u];
                                bool flag = false;
                                if (in) {
                                    const double cred = e[u] - vv[u] - h;
                                    if (cred < dd[u]) { L.d[j] = cred; L.pred[j] = i; flag = cred == mind; }
                                }
                                any_lower = any_lower || __ballot(in && e[u] - vv[u] - h < dd[u]) != 0ull;
                                const unsigned long long fl = __ballot(flag);
                                if (fl) {
                                    const unsigned long long un = __ballot(flag && yy[u] < 0);      // the first of these ends the search
                                    const int seg_end = un ? __builtin_ctzll(un) : WAVE;
                                    jv_move(L, base, 0, fl & (seg_end >= WAVE ? ~0ull : (1ull << seg_end) - 1ull), hi, j, lane);
                                    if (un) found = __shfl(j, seg_end, WAVE);
                                }
                            }
                        }
                        BM_WAVE_LDS_SYNC();
                        if (any_lower) memo_ok = false;
                        else if (i >= n_rows && found < 0) { memo_ok = true; memo_h = h; }
                        if (found >= 0) --lo;      // the sequential code returns before writing `lo` back
                    }
                    final_j = found;
                }
            }
            if (!ok) break;
            const double mind = L.d[L.cols[lo]];
            for (int k = lane; k < n_ready; k += WAVE) { const int j = L.cols[k]; L.v[j] += L.d[j] - mind; }
            BM_WAVE_LDS_SYNC();
            if (lane == 0) {
                int i = -1, j = final_j, steps = 0;
                while (i != fi && steps++ <= n) {
                    i = L.pred[j];
                    L.y[j] = i;
                    const int t = j; j = L.x[i]; L.x[i] = t;
                }
                if (i != fi) L.free_rows[0] = -2;        // path did not close: report through the guard below
            }
            BM_WAVE_LDS_SYNC();
            if (L.free_rows[0] == -2) ok = false;
        }
        BM_JV_PROF(4);
        if (lane == 0) c.s_int[0] = ok ? 1 : 0;
    }
    __syncthreads();
    ok = c.s_int[0] != 0;
    __syncthreads();
    for (int i = c.tid; i < n_rows; i += c.nthr) out_col_of_row[i] = (!ok || L.x[i] >= n_cols) ? -1 : L.x[i];
    for (int j = c.tid; j < n_cols; j += c.nthr) out_row_of_col[j] = (!ok || L.y[j] >= n_rows) ? -1 : L.y[j];
    __syncthreads();
    return ok;
}

}  // namespace bm
