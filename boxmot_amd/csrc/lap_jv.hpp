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

__device__ inline double jv_wave_min(double v) {
    for (int off = WAVE / 2; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, WAVE); v = o < v ? o : v; }
    return v;
}
__device__ inline int jv_wave_max_int(int v) {
    for (int off = WAVE / 2; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, WAVE); v = o > v ? o : v; }
    return v;
}
// the two lexicographically smallest (value, index) pairs of the wave's candidates (each lane brings its own two)
struct JvTwo { double v1, v2; int j1, j2; };
__device__ inline bool jv_less(double va, int ja, double vb, int jb) { return va < vb || (va == vb && ja < jb); }
__device__ inline void jv_two_insert(JvTwo& t, double r, int j) {
    if (jv_less(r, j, t.v1, t.j1)) { t.v2 = t.v1; t.j2 = t.j1; t.v1 = r; t.j1 = j; }
    else if (jv_less(r, j, t.v2, t.j2)) { t.v2 = r; t.j2 = j; }
}
__device__ inline JvTwo jv_wave_two(JvTwo t) {
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        const double ov1 = __shfl_xor(t.v1, off, WAVE), ov2 = __shfl_xor(t.v2, off, WAVE);
        const int oj1 = __shfl_xor(t.j1, off, WAVE), oj2 = __shfl_xor(t.j2, off, WAVE);
        jv_two_insert(t, ov1, oj1);
        jv_two_insert(t, ov2, oj2);
    }
    return t;
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
    // ---- column reduction (_ccrrt_dense): column minima, first row wins ----
    for (int i = c.tid; i < n; i += c.nthr) { L.x[i] = -1; L.pred[i] = 0; }
    for (int j = c.tid; j < n; j += c.nthr) {
        double vj = JV_LARGE;
        int yj = 0;
        for (int i = 0; i < n; ++i) { const double e = e_of(i, j); if (e < vj) { vj = e; yj = i; } }
        L.v[j] = vj; L.y[j] = yj;
    }
    __syncthreads();
    // the descending column loop: the LAST column of a row's claimants keeps it, the others lose their row; a row claimed once is "unique"
    for (int j = c.tid; j < n; j += c.nthr) { const int i = L.y[j]; atomicMax(&L.x[i], j); atomicAdd(&L.pred[i], 1); }
    __syncthreads();
    for (int j = c.tid; j < n; j += c.nthr) if (L.x[L.y[j]] != j) L.y[j] = -1;
    __syncthreads();
    bool ok = true;
    if (c.wave == 0) {
        const int lane = c.lane;
        // ---- free rows (ascending) and reduction transfer, row by row: a transfer changes v for the rows after it ----
        int n_free = 0;
        for (int i = 0; i < n; ++i) {
            const int xi = L.x[i];
            if (xi < 0) { if (lane == 0) L.free_rows[n_free] = i; ++n_free; }
            else if (L.pred[i] == 1) {
                double mn = JV_LARGE;
                for (int j = lane; j < n; j += WAVE) if (j != xi) { const double r = e_of(i, j) - L.v[j]; mn = r < mn ? r : mn; }
                mn = jv_wave_min(mn);
                if (lane == 0) L.v[xi] -= mn;
                BM_WAVE_LDS_SYNC();
            }
        }
        BM_WAVE_LDS_SYNC();
        // ---- augmenting row reduction (_carr_dense), twice ----
        for (int round = 0; round < 2 && n_free > 0; ++round) {
            int current = 0, new_free = 0;
            long rr_cnt = 0;
            const long guard_max = 4L * n * n + 64;
            long guard = 0;
            while (current < n_free) {
                if (++guard > guard_max) { ok = false; break; }
                ++rr_cnt;
                const int fi = L.free_rows[current++];
                JvTwo t{JV_LARGE, JV_LARGE, 0x7fffffff, 0x7fffffff};
                for (int j = lane; j < n; j += WAVE) jv_two_insert(t, e_of(fi, j) - L.v[j], j);
                t = jv_wave_two(t);
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
            }
            n_free = new_free;
        }
        // ---- augmentation (_ca_dense): shortest augmenting path per remaining free row ----
        for (int f = 0; f < n_free && ok; ++f) {
            const int fi = L.free_rows[f];
            for (int j = lane; j < n; j += WAVE) { L.cols[j] = j; L.pred[j] = fi; L.d[j] = e_of(fi, j) - L.v[j]; }
            BM_WAVE_LDS_SYNC();
            int lo = 0, hi = 0, final_j = -1, n_ready = 0;
            long guard = 0;
            while (final_j == -1) {
                if (++guard > 4L * n + 64) { ok = false; break; }
                if (lo == hi) {
                    // _find_dense: move the columns at the minimum of d over cols[lo..n) to the front, in scan order
                    n_ready = lo;
                    hi = lo + 1;
                    double mind = L.d[L.cols[lo]];
                    for (int base = hi; base < n; base += WAVE) {
                        const int k = base + lane;
                        const bool in = k < n;
                        const int j = in ? L.cols[k] : 0;
                        const double s = in ? L.d[j] : JV_LARGE;
                        // exclusive prefix minimum over the lanes, seeded with the running minimum
                        double pm = s;
                        for (int off = 1; off < WAVE; off <<= 1) { const double o = __shfl(pm, lane >= off ? lane - off : lane, WAVE); if (lane >= off && o < pm) pm = o; }
                        double ex = __shfl(pm, lane > 0 ? lane - 1 : 0, WAVE);
                        if (lane == 0 || mind < ex) ex = mind;
                        unsigned long long ev = __ballot(in && s <= ex);
                        while (ev) {
                            const int el = __builtin_ctzll(ev);
                            ev &= ev - 1;
                            const double se = __shfl(s, el, WAVE);
                            const int je = __shfl(j, el, WAVE);
                            if (se < mind) { hi = lo; mind = se; }
                            if (lane == 0) { L.cols[base + el] = L.cols[hi]; L.cols[hi] = je; }
                            ++hi;
                            BM_WAVE_LDS_SYNC();
                        }
                    }
                    // the last ready column without a row ends the search
                    int last = -1;
                    for (int k = lo + lane; k < hi; k += WAVE) if (L.y[L.cols[k]] < 0) last = k;
                    last = jv_wave_max_int(last);
                    if (last >= 0) final_j = L.cols[last];
                }
                if (final_j == -1) {
                    // _scan_dense: relax from the ready columns; columns that reach the minimum join the ready set
                    int found = -1;
                    while (lo != hi && found < 0) {
                        const int j0 = L.cols[lo++];
                        const int i = L.y[j0];
                        const double mind = L.d[j0];
                        const double h = e_of(i, j0) - L.v[j0] - mind;
                        const int hi0 = hi;
                        for (int base = hi0; base < n && found < 0; base += WAVE) {
                            const int k = base + lane;
                            const bool in = k < n;
                            const int j = in ? L.cols[k] : 0;
                            bool flag = false;
                            if (in) {
                                const double cred = e_of(i, j) - L.v[j] - h;
                                if (cred < L.d[j]) { L.d[j] = cred; L.pred[j] = i; flag = cred == mind; }
                            }
                            const bool unassigned = in && L.y[j] < 0;
                            unsigned long long ev = __ballot(flag);
                            while (ev) {
                                const int el = __builtin_ctzll(ev);
                                ev &= ev - 1;
                                const int je = __shfl(j, el, WAVE);
                                if (__shfl((int)unassigned, el, WAVE)) { found = je; break; }
                                if (lane == 0) { L.cols[base + el] = L.cols[hi]; L.cols[hi] = je; }
                                ++hi;
                                BM_WAVE_LDS_SYNC();
                            }
                        }
                        BM_WAVE_LDS_SYNC();
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
