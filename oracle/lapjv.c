/*
 * oracle/lapjv.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into
 * the product library; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * Restatement of the dense Jonker-Volgenant LAP solver behind
 * `lap.lapjv(cost, extend_cost=True, cost_limit=L)`:
 *
 *   - third-party dependency `lapx==0.9.4` (reference uv.lock:2522-2523), NOT
 *     vendored under /root/reference and not installable offline.  Call sites
 *     this oracle anchors on: boxmot/trackers/association/matching.py:28-43
 *     (BoT-SORT / ByteTrack, `extend_cost=True, cost_limit=thresh`) and
 *     boxmot/trackers/association/association.py:20-24 (`extend_cost=True`).
 *   - published algorithm: R. Jonker, A. Volgenant, "A shortest augmenting path
 *     algorithm for dense and sparse linear assignment problems", Computing 38
 *     (1987): column reduction + reduction transfer, two rounds of augmenting
 *     row reduction, then shortest-augmenting-path augmentation (the structure
 *     the lap/lapx C++ core follows).
 *   - the `_lapjv` wrapper semantics restated here (from knowledge of the
 *     gatagat/lap sources; PARITY UNPINNED -- the package is unavailable, so
 *     this cannot be diffed against the real library in this container):
 *       extend_cost or finite cost_limit L  ->  n = n_rows + n_cols square,
 *       filled with L/2 (finite L) or max(cost)+1 (no limit), bottom-right
 *       n_cols x n_rows block = 0, top-left = cost;  after solving,
 *       x[x >= n_cols] = -1, y[y >= n_rows] = -1, truncate.
 *
 * Exact optimality is what the trackers consume; when the optimum is unique
 * any exact solver returns the same (x, y).  tests/test_oracle_lap.py checks
 * this file against scipy.optimize.linear_sum_assignment on the same extended
 * matrix.
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define LARGE DBL_MAX

static int ccrrt_dense(int n, const double *c, int *free_rows, int *x, int *y, double *v)
{
    int i, j;
    for (i = 0; i < n; i++) { x[i] = -1; v[i] = LARGE; y[i] = 0; }
    for (i = 0; i < n; i++) {
        const double *ci = c + (size_t)i * n;
        for (j = 0; j < n; j++) {
            if (ci[j] < v[j]) { v[j] = ci[j]; y[j] = i; }
        }
    }
    char *unique = (char *)malloc((size_t)n);
    memset(unique, 1, (size_t)n);
    j = n;
    do {
        j--;
        i = y[j];
        if (x[i] < 0) x[i] = j;
        else { unique[i] = 0; y[j] = -1; }
    } while (j > 0);
    int n_free = 0;
    for (i = 0; i < n; i++) {
        if (x[i] < 0) free_rows[n_free++] = i;
        else if (unique[i]) {
            const double *ci = c + (size_t)i * n;
            const int j1 = x[i];
            double mn = LARGE;
            for (j = 0; j < n; j++) {
                if (j == j1) continue;
                const double r = ci[j] - v[j];
                if (r < mn) mn = r;
            }
            v[j1] -= mn;
        }
    }
    free(unique);
    return n_free;
}

static int carr_dense(int n, const double *c, int n_free, int *free_rows, int *x, int *y, double *v)
{
    int current = 0, new_free = 0;
    long rr_cnt = 0;
    while (current < n_free) {
        rr_cnt++;
        const int fi = free_rows[current++];
        const double *ci = c + (size_t)fi * n;
        int j1 = 0, j2 = -1;
        double v1 = ci[0] - v[0], v2 = LARGE;
        for (int j = 1; j < n; j++) {
            const double r = ci[j] - v[j];
            if (r < v2) {
                if (r >= v1) { v2 = r; j2 = j; }
                else { v2 = v1; v1 = r; j2 = j1; j1 = j; }
            }
        }
        int i0 = y[j1];
        const double v1_new = v[j1] - (v2 - v1);
        const int lowers = v1_new < v[j1];
        if (rr_cnt < (long)current * n) {
            if (lowers) v[j1] = v1_new;
            else if (i0 >= 0 && j2 >= 0) { j1 = j2; i0 = y[j2]; }
            if (i0 >= 0) {
                if (lowers) free_rows[--current] = i0;
                else free_rows[new_free++] = i0;
            }
        } else if (i0 >= 0) {
            free_rows[new_free++] = i0;
        }
        x[fi] = j1;
        y[j1] = fi;
    }
    return new_free;
}

static int find_dense(int n, int lo, const double *d, int *cols)
{
    int hi = lo + 1;
    double mind = d[cols[lo]];
    for (int k = hi; k < n; k++) {
        const int j = cols[k];
        if (d[j] <= mind) {
            if (d[j] < mind) { hi = lo; mind = d[j]; }
            cols[k] = cols[hi];
            cols[hi++] = j;
        }
    }
    return hi;
}

static int scan_dense(int n, const double *c, int *plo, int *phi, double *d, int *cols,
                      int *pred, const int *y, const double *v)
{
    int lo = *plo, hi = *phi;
    while (lo != hi) {
        int j = cols[lo++];
        const int i = y[j];
        const double mind = d[j];
        const double *ci = c + (size_t)i * n;
        const double h = ci[j] - v[j] - mind;
        for (int k = hi; k < n; k++) {
            j = cols[k];
            const double cred = ci[j] - v[j] - h;
            if (cred < d[j]) {
                d[j] = cred;
                pred[j] = i;
                if (cred == mind) {
                    if (y[j] < 0) return j;
                    cols[k] = cols[hi];
                    cols[hi++] = j;
                }
            }
        }
    }
    *plo = lo; *phi = hi;
    return -1;
}

static int find_path_dense(int n, const double *c, int start_i, const int *y, double *v,
                           int *pred, int *cols, double *d)
{
    int lo = 0, hi = 0, final_j = -1, n_ready = 0;
    const double *cs = c + (size_t)start_i * n;
    for (int j = 0; j < n; j++) { cols[j] = j; pred[j] = start_i; d[j] = cs[j] - v[j]; }
    while (final_j == -1) {
        if (lo == hi) {
            n_ready = lo;
            hi = find_dense(n, lo, d, cols);
            for (int k = lo; k < hi; k++) {
                const int j = cols[k];
                if (y[j] < 0) final_j = j;
            }
        }
        if (final_j == -1) final_j = scan_dense(n, c, &lo, &hi, d, cols, pred, y, v);
    }
    const double mind = d[cols[lo]];
    for (int k = 0; k < n_ready; k++) {
        const int j = cols[k];
        v[j] += d[j] - mind;
    }
    return final_j;
}

/* Solve the square n x n problem; x[i] = column of row i, y[j] = row of column j. */
int oracle_lapjv_square(int n, const double *c, int *x, int *y)
{
    if (n <= 0) return 0;
    int *free_rows = (int *)malloc(sizeof(int) * (size_t)n);
    double *v = (double *)malloc(sizeof(double) * (size_t)n);
    int ret = ccrrt_dense(n, c, free_rows, x, y, v);
    for (int i = 0; ret > 0 && i < 2; i++) ret = carr_dense(n, c, ret, free_rows, x, y, v);
    if (ret > 0) {
        int *pred = (int *)malloc(sizeof(int) * (size_t)n);
        int *cols = (int *)malloc(sizeof(int) * (size_t)n);
        double *d = (double *)malloc(sizeof(double) * (size_t)n);
        for (int f = 0; f < ret; f++) {
            const int fi = free_rows[f];
            int i = -1, j = find_path_dense(n, c, fi, y, v, pred, cols, d);
            while (i != fi) {
                i = pred[j];
                y[j] = i;
                const int t = j; j = x[i]; x[i] = t;
            }
        }
        free(pred); free(cols); free(d);
    }
    free(free_rows); free(v);
    return 0;
}

/*
 * lap.lapjv(cost, extend_cost=True, cost_limit=limit) semantics on an
 * n_rows x n_cols row-major matrix.  use_limit == 0 -> "no limit" padding
 * (max(cost)+1).  Outputs x[n_rows], y[n_cols] with -1 for unassigned.
 */
int oracle_lapjv_extended(int n_rows, int n_cols, const double *cost, int use_limit,
                          double limit, int *x, int *y)
{
    const int n = n_rows + n_cols;
    if (n_rows == 0 || n_cols == 0) {
        for (int i = 0; i < n_rows; i++) x[i] = -1;
        for (int j = 0; j < n_cols; j++) y[j] = -1;
        return 0;
    }
    double fill;
    if (use_limit) fill = limit / 2.0;
    else {
        double mx = cost[0];
        for (size_t k = 1; k < (size_t)n_rows * n_cols; k++) if (cost[k] > mx) mx = cost[k];
        fill = mx + 1.0;
    }
    double *e = (double *)malloc(sizeof(double) * (size_t)n * n);
    for (size_t k = 0; k < (size_t)n * n; k++) e[k] = fill;
    for (int i = n_rows; i < n; i++)
        for (int j = n_cols; j < n; j++) e[(size_t)i * n + j] = 0.0;
    for (int i = 0; i < n_rows; i++)
        memcpy(e + (size_t)i * n, cost + (size_t)i * n_cols, sizeof(double) * (size_t)n_cols);
    int *xe = (int *)malloc(sizeof(int) * (size_t)n);
    int *ye = (int *)malloc(sizeof(int) * (size_t)n);
    oracle_lapjv_square(n, e, xe, ye);
    for (int i = 0; i < n_rows; i++) x[i] = xe[i] >= n_cols ? -1 : xe[i];
    for (int j = 0; j < n_cols; j++) y[j] = ye[j] >= n_rows ? -1 : ye[j];
    free(e); free(xe); free(ye);
    return 0;
}

/*
 * The OTHER reading of lapx's `extend_cost=True` without a cost limit (SURVEY.md section 8(c) "alternative reading"; round-4 review,
 * Weak 3): the rectangular matrix zero-padded to max(n_rows, n_cols) square, solved, assignments into the padding dropped.
 * Same optimum as the (n_rows + n_cols) square form above; possibly another choice among exactly tied optima.  Kept ONLY so that
 * tests/test_lap_forms.py can show every golden row of the DeepOCSORT / OC-SORT call site (association.py:20-24) is invariant
 * under the choice of form -- the form above stays the oracle's (and the device solver's) default.
 */
int oracle_lapjv_zero_padded(int n_rows, int n_cols, const double *cost, int *x, int *y)
{
    const int n = n_rows > n_cols ? n_rows : n_cols;
    if (n_rows == 0 || n_cols == 0) {
        for (int i = 0; i < n_rows; i++) x[i] = -1;
        for (int j = 0; j < n_cols; j++) y[j] = -1;
        return 0;
    }
    double *e = (double *)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n_rows; i++)
        memcpy(e + (size_t)i * n, cost + (size_t)i * n_cols, sizeof(double) * (size_t)n_cols);
    int *xe = (int *)malloc(sizeof(int) * (size_t)n);
    int *ye = (int *)malloc(sizeof(int) * (size_t)n);
    oracle_lapjv_square(n, e, xe, ye);
    for (int i = 0; i < n_rows; i++) x[i] = xe[i] >= n_cols ? -1 : xe[i];
    for (int j = 0; j < n_cols; j++) y[j] = ye[j] >= n_rows ? -1 : ye[j];
    free(e); free(xe); free(ye);
    return 0;
}
