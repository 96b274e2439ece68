// Body of the BoT-SORT / ByteTrack frame step (botsort_step.hpp includes it twice: once in namespace bm with the axis-aligned layout,
// once in namespace bm::obb with the oriented one -- BM_OBB, KF_DIM / KF_STRIDE / DET_COLS / OUT_COLS / BOX_W / CONF_COL are the
// enclosing namespace's).  No include guard, no includes.

// the detection box the IoU functions take: fp32 xyxy (axis-aligned) | fp32 (cx, cy, w, h, theta) (oriented)
#undef DET_BOX
#if BM_OBB
#define DET_BOX(v, d) ((v).det_xywh + (d) * BOX_W)
#else
#define DET_BOX(v, d) ((v).det_xyxy + (d) * 4)
#endif
constexpr double LAP_INF = 1.0e300;
constexpr int COST_TILE = 64;     // rows x cols of one cosine tile
constexpr int COST_KC = 32;       // k-chunk staged in LDS
constexpr int SPARSE_MAX_ALLOC = 4096;   // capacity of the per-stream ungated-pair list

// Per-stream views (pointers already offset to this stream).
struct SV {
    BotSortConfigDev cfg;
    int cap, dim, nd;
    // state
    int* frame_count; int* id_count; int* n_active; int* n_lost; int* rm_head; int* rm_size;
    int* stamp; int* status;
    int* active_list; int* lost_list; int* removed_ring; int removed_alloc;
    double* kf; float* smooth;
    int* id; int* state; int* is_activated; int* frame_id; int* start_frame; int* tracklet_len;
    int* slot_used; int* mark;
    float* conf; float* cls; float* det_ind;
    int* hist_n; float* hist_cls; float* hist_w;
    // scratch
    float* det_xywh; float* det_xyxy; float* det_area; float* det_feat;
    double* det_norm; double* trk_norm;
    int* first_idx; int* second_idx; int* left_idx;
    int* pool; int* unconf; int* remain; int* list_a; int* list_b;
    int* activated; int* refound; int* newly_lost; int* newly_removed;
    int* match_slot; int* match_det; int* match_flag; int* drop_a; int* drop_b;
    double* cost; int* lap_x; int* lap_y; double* lap_u; double* lap_v; double* lap_minv;
    int* lap_way; int* lap_used; double* box_a; int* pair_list;
    // io
    const float* dets; int n_dets; const float* embs; float* out; int* out_n;
};

__device__ inline SV make_view(const BotSortStepArgs& a, int s, int sel) {
    SV v;
    const BotSortState& st = a.st;
    const BotSortScratch& sc = a.sc;
    const long cap = st.cap, dim = st.dim, nd = sc.max_dets;
    v.cfg = a.cfg;
    v.cap = st.cap; v.dim = st.dim; v.nd = sc.max_dets;
    v.frame_count = st.frame_count + s; v.id_count = st.id_count + s;
    v.n_active = st.n_active + (long)s * st.n_lists + sel; v.n_lost = st.n_lost + s;
    v.rm_head = st.rm_head + s; v.rm_size = st.rm_size + s; v.stamp = st.stamp + s;
    v.status = st.status + s;
    v.active_list = st.active_list + ((long)s * st.n_lists + sel) * cap;
    v.lost_list = st.lost_list + s * cap;
    v.removed_alloc = st.removed_alloc;
    v.removed_ring = st.removed_ring + (long)s * st.removed_alloc;
    v.kf = st.kf + s * cap * KF_STRIDE;
    v.smooth = st.smooth + s * cap * dim;
    v.id = st.id + s * cap; v.state = st.state + s * cap; v.is_activated = st.is_activated + s * cap;
    v.frame_id = st.frame_id + s * cap; v.start_frame = st.start_frame + s * cap;
    v.tracklet_len = st.tracklet_len + s * cap; v.slot_used = st.slot_used + s * cap;
    v.mark = st.mark + s * cap;
    v.conf = st.conf + s * cap; v.cls = st.cls + s * cap; v.det_ind = st.det_ind + s * cap;
    v.hist_n = st.hist_n + s * cap; v.hist_cls = st.hist_cls + s * cap * KCLS;
    v.hist_w = st.hist_w + s * cap * KCLS;
    v.det_xywh = sc.det_xywh + s * nd * BOX_W; v.det_xyxy = sc.det_xyxy + s * nd * 4;
    v.det_area = sc.det_area + s * nd; v.det_feat = sc.det_feat + s * nd * dim;
    v.det_norm = sc.det_norm + s * nd; v.trk_norm = sc.trk_norm + s * cap;
    v.first_idx = sc.first_idx + s * nd; v.second_idx = sc.second_idx + s * nd;
    v.left_idx = sc.left_idx + s * nd;
    v.pool = sc.pool + s * cap; v.unconf = sc.unconf + s * cap; v.remain = sc.remain + s * cap;
    v.list_a = sc.list_a + s * cap; v.list_b = sc.list_b + s * cap;
    v.activated = sc.activated + s * cap; v.refound = sc.refound + s * cap;
    v.newly_lost = sc.newly_lost + s * cap; v.newly_removed = sc.newly_removed + s * cap;
    v.match_slot = sc.match_slot + s * nd; v.match_det = sc.match_det + s * nd;
    v.match_flag = sc.match_flag + s * nd;
    v.drop_a = sc.drop_a + s * cap; v.drop_b = sc.drop_b + s * cap;
    v.cost = sc.cost + s * cap * nd;
    v.lap_x = sc.lap_x + s * cap; v.lap_y = sc.lap_y + s * nd; v.lap_u = sc.lap_u + s * nd;
    v.lap_v = sc.lap_v + s * cap; v.lap_minv = sc.lap_minv + s * cap;
    v.lap_way = sc.lap_way + s * cap; v.lap_used = sc.lap_used + s * cap;
    v.box_a = sc.box_a + s * cap * BOX_W;
    v.pair_list = sc.pair_list + (long)s * SPARSE_MAX_ALLOC;
    v.dets = a.dets + s * nd * DET_COLS;
    v.n_dets = a.n_dets[s];
    v.embs = a.embs ? a.embs + s * nd * dim : nullptr;
    v.out = a.out + s * nd * OUT_COLS;
    v.out_n = a.out_n + s;
    return v;
}

#if !BM_OBB
// ---------------------------------------------------------------------------
// Kalman filter, one wavefront per track; lane l owns cov element (l>>3, l&7).
// ---------------------------------------------------------------------------
constexpr double STD_POS = 1.0 / 20;
constexpr double STD_VEL = 1.0 / 160;
constexpr double MIN_SIZE = 1e-4;

// multi_predict for one track (base.py:311-327, xywh.py:149-160) incl. the
// "zero (vw, vh) unless Tracked" rule of STrack.multi_predict (botsort_track.py:104-109).
// `xyah`: KalmanFilterXYAH noise model (xyah.py:70-89: every std from the height, constants for the aspect ratio)
// and ByteTrack's rule "zero vh unless Tracked" (bytetrack.py:63-74).
// (`mj_in`, `p_in`: the lane's mean[lane & 7] and covariance element, loaded by the caller -- the predict loop requests the next
// track's state before it computes this one)
BM_STEP_FN void kf_predict_wave(double* kf, double mj_in, double p_in, bool zero_size_vel, int lane, bool xyah = false) {
    const int i = lane >> 3, j = lane & 7;
    double mj = mj_in;                       // mean[j]
    if (zero_size_vel && (xyah ? j == 7 : j >= 6)) mj = 0.0;
    const double w = __shfl(mj, 2, WAVE), h = __shfl(mj, 3, WAVE);   // pre-motion w, h
    const double p = p_in;
    // mean' = mean . F^T  (one rounding: m_k + m_{k+4})
    const double mhi = __shfl(mj, (j + 4) & 7, WAVE);
    double mnew = (j < 4) ? (mj + mhi) : mj;
    if (j == 2 || j == 3) mnew = mnew > MIN_SIZE ? mnew : MIN_SIZE;
    // left = F P ; cov' = left F^T + Q   (same two-step rounding as np.dot twice)
    const double p_dn = __shfl(p, (lane + 32) & 63, WAVE);        // P[i+4][j]
    const double fp = (i < 4) ? (p + p_dn) : p;
    const double fp_rt = __shfl(fp, (lane + 4) & 63, WAVE);       // (FP)[i][j+4]
    double c = (j < 4) ? (fp + fp_rt) : fp;
    if (i == j) {
        const double dim_v = xyah ? h : ((i & 1) ? h : w);
        double sd = ((i < 4) ? STD_POS : STD_VEL) * dim_v;
        if (xyah && i == 2) sd = 1e-2;
        if (xyah && i == 6) sd = 1e-5;
        c = c + sd * sd;
    } else {
        c = c + 0.0;
    }
    kf[KF_DIM + lane] = c;
    if (i == 0) kf[j] = mnew;
}

// STrack.multi_gmc for one track (botsort_track.py:117-132): mean <- kron(I4, R) mean (+ t on x, y),
// cov <- R8 cov R8^T with R8 = kron(I4, R); W = [r00 r01 tx; r10 r11 ty].  Every element is a sum of two
// products; numpy evaluates the matrix-vector product as mul, mul, add and the two 8x8 matrix products as
// fma(b_hi, a_hi, b_lo * a_lo) (k ascending, OpenBLAS dgemm micro-kernel) -- reproduced here so the fp64 state
// follows the NumPy reference to the last bit on such hosts (the tests accept 1e-9 relative).
BM_STEP_FN void kf_warp_wave(double* kf, const double* W, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const double m = kf[j];
    const double p = kf[KF_DIM + lane];
    const double mp = __shfl(m, (lane & ~7) | (j ^ 1), WAVE);
    const double m0 = (j & 1) ? mp : m, m1 = (j & 1) ? m : mp;
    double mn = W[(j & 1) * 3 + 0] * m0 + W[(j & 1) * 3 + 1] * m1;
    if (j < 2) mn = mn + W[j * 3 + 2];
    const double pp = __shfl(p, lane ^ 8, WAVE);                      // P[i ^ 1][j]
    const double p0 = (i & 1) ? pp : p, p1 = (i & 1) ? p : pp;
    const double a = fma(W[(i & 1) * 3 + 1], p1, W[(i & 1) * 3 + 0] * p0);   // (R8 P)[i][j]
    const double ap = __shfl(a, lane ^ 1, WAVE);                      // (R8 P)[i][j ^ 1]
    const double a0 = (j & 1) ? ap : a, a1 = (j & 1) ? a : ap;
    const double cnew = fma(a1, W[(j & 1) * 3 + 1], a0 * W[(j & 1) * 3 + 0]);
    const double cn = __shfl(cnew, lane, WAVE);                       // wave-wide dependency: all loads precede the stores
    kf[KF_DIM + lane] = cn;
    if (i == 0) kf[j] = mn;
}

// KalmanFilterXYWH.update for one track with measurement z (fp32 xywh);
// base.py:286-355 (confidence = 0: BoT-SORT never passes it, botsort_track.py:269-271).
// `xyah`: z is (x, y, aspect, height) and the measurement noise follows xyah.py:57-68.
BM_STEP_FN void kf_update_wave(double* kf, const float* z, int lane, bool xyah = false) {
    const int i = lane >> 3, j = lane & 7;
    double m[8];
    for (int k = 0; k < 8; ++k) m[k] = kf[k];
    const double* P = kf + KF_DIM;
    // S = H P H^T + diag(std^2)
    double S[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) S[a][b] = P[a * 8 + b];
    for (int a = 0; a < 4; ++a) {
        double sd = STD_POS * (xyah ? m[3] : m[2 + (a & 1)]);
        if (xyah && a == 2) sd = 1e-1;
        S[a][a] = S[a][a] + sd * sd;
    }
    // lower Cholesky factor (dpotrf, lower triangle of S only)
    double L[4][4];
    for (int c = 0; c < 4; ++c) {
        double d = S[c][c];
        for (int k = 0; k < c; ++k) d -= L[c][k] * L[c][k];
        d = sqrt(d);
        L[c][c] = d;
        for (int r = c + 1; r < 4; ++r) {
            double t = S[r][c];
            for (int k = 0; k < c; ++k) t -= L[r][k] * L[c][k];
            L[r][c] = t / d;
        }
    }
    // gain rows i and j: K[r][:] = S^-1 P[r][0:4]^T  (cho_solve: L y = b, L^T x = y)
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
    // mean' = mean + innovation . K^T
    double acc = 0.0;
    for (int a = 0; a < 4; ++a) acc += ((double)z[a] - m[a]) * Ki[a];
    double mnew = m[i] + acc;
    if (i == 2 || i == 3) mnew = mnew > MIN_SIZE ? mnew : MIN_SIZE;
    // cov' = cov - K (S K^T)
    double ksk = 0.0;
    for (int a = 0; a < 4; ++a) {
        double mj = 0.0;
        for (int b = 0; b < 4; ++b) mj += S[a][b] * Kj[b];
        ksk += Ki[a] * mj;
    }
    const double pnew = P[lane] - ksk;
    // all lanes have finished reading kf through the loads above (register
    // values); stores happen after a wave-wide data dependency on `pnew`.
    const double pn = __shfl(pnew, lane, WAVE);
    kf[KF_DIM + lane] = pn;
    if (j == 0) kf[i] = mnew;
}

// KalmanFilterXYWH.initiate (xywh.py:136-142, base.py:234-244, std xywh.py:22-36)
// `xyah`: xyah.py:22-37, 99-105.
BM_STEP_FN void kf_initiate_wave(double* kf, const float* z, int lane, bool xyah = false) {
    const int i = lane >> 3, j = lane & 7;
    const double w = (double)z[2], h = (double)z[3];
    double c = 0.0;
    if (i == j) {
        const double dim_v = xyah ? h : ((i & 1) ? h : w);
        double sd = (i < 4) ? (2 * STD_POS) * dim_v : (10 * STD_VEL) * dim_v;
        if (xyah && i == 2) sd = 1e-2;
        if (xyah && i == 6) sd = 1e-5;
        c = sd * sd;
    }
    kf[KF_DIM + lane] = c;
    if (i == 0) {
        double mv = (j < 4) ? (double)z[j] : 0.0;
        if (j == 2 || j == 3) mv = mv > MIN_SIZE ? mv : MIN_SIZE;
        kf[j] = mv;
    }
}

#else
// ---------------------------------------------------------------------------
// Kalman filter for ORIENTED boxes: KalmanFilterXYWH(ndim=5), state (cx, cy, w, h, theta) + velocities, one wavefront per track.
// kf[0..9] = mean, kf[10..109] = the 10 x 10 covariance, row-major; lane l owns covariance elements l and l + 64.
// Follows boxmot/motion/kalman_filters/xywh.py:16-206 over base.py:116-355 (the checker's restatement of the same lines is
// pinned bit for bit on the reference filter, tests/).
// ---------------------------------------------------------------------------
constexpr double STD_POS = 1.0 / 20;
constexpr double STD_VEL = 1.0 / 160;
constexpr double MIN_SIZE = 1e-4;
// (OBB_PI, obb_wrap_angle: obb_geometry.hpp)
// per-state process / initial standard deviation's size factor: w for x and w, h for y and h (xywh.py:22-83)
__device__ inline double obb_dim_of(int i5, double w, double h) { return (i5 & 1) ? h : w; }

// multi_predict for one track incl. "zero (vw, vh, vtheta) unless Tracked" (botsort_track.py:104-109, bytetrack.py:55-58)
BM_STEP_FN void kf_predict_obb_wave(double* kf, bool zero_vel, int lane) {
    double m[10];
    for (int k = 0; k < 10; ++k) m[k] = kf[k];
    if (zero_vel) { m[7] = 0.0; m[8] = 0.0; m[9] = 0.0; }
    const double w = m[2], h = m[3];                    // pre-motion sizes drive the process noise
    const double* P = kf + KF_DIM;
    double cnew[2];
    for (int q = 0; q < 2; ++q) {
        const int e = lane + q * WAVE;
        cnew[q] = 0.0;
        if (e >= 100) continue;
        const int i = e / 10, j = e % 10;
        // left = F P ; cov' = left F^T + Q   (the two-step rounding of np.dot twice)
        const double fp = (i < 5) ? (P[i * 10 + j] + P[(i + 5) * 10 + j]) : P[i * 10 + j];
        double c = fp;
        if (j < 5) {
            const double fp_rt = (i < 5) ? (P[i * 10 + j + 5] + P[(i + 5) * 10 + j + 5]) : P[i * 10 + j + 5];
            c = fp + fp_rt;
        }
        if (i == j) {
            const int i5 = i < 5 ? i : i - 5;
            double sd = (i5 == 4) ? (i < 5 ? 1e-2 : 1e-5) : ((i < 5 ? STD_POS : STD_VEL) * obb_dim_of(i5, w, h));
            c = c + sd * sd;
        } else {
            c = c + 0.0;
        }
        cnew[q] = c;
    }
    double mnew = 0.0;
    if (lane < 10) {
        mnew = (lane < 5) ? (m[lane] + m[lane + 5]) : m[lane];
        if (lane == 2 || lane == 3) mnew = mnew > MIN_SIZE ? mnew : MIN_SIZE;
        if (lane == 4) mnew = obb_wrap_angle(mnew);
    }
    const double c0 = __shfl(cnew[0], lane, WAVE), c1 = __shfl(cnew[1], lane, WAVE);      // wave-wide dependency: every load precedes the stores
    kf[KF_DIM + lane] = c0;
    if (lane + WAVE < 100) kf[KF_DIM + lane + WAVE] = c1;
    if (lane < 10) kf[lane] = mnew;
}

// KalmanFilterXYWH._align_obb_measurement (xywh.py:85-123, base.py:122-157): of (w, h, t), (w, h, t + pi), (h, w, t + pi / 2),
// (h, w, t - pi / 2) the parameterisation closest to the state
__device__ inline void obb_align_measurement(double* z, const double* ref) {
    const double ref_w = ref[2] > 1e-6 ? ref[2] : 1e-6, ref_h = ref[3] > 1e-6 ? ref[3] : 1e-6, ref_t = ref[4];
    const double w = z[2] > 1e-6 ? z[2] : 1e-6, h = z[3] > 1e-6 ? z[3] : 1e-6, t = z[4];
    double best_cost = 1.0 / 0.0, bs0 = w, bs1 = h, bt = t;
    for (int k = 0; k < 4; ++k) {
        double s0 = (k < 2) ? w : h, s1 = (k < 2) ? h : w;
        const double th = k == 0 ? t : (k == 1 ? t + OBB_PI : (k == 2 ? t + (OBB_PI / 2.0) : t - (OBB_PI / 2.0)));
        s0 = s0 > 1e-6 ? s0 : 1e-6; s1 = s1 > 1e-6 ? s1 : 1e-6;
        const double ta = ref_t + obb_wrap_angle(th - ref_t);
        const double cost = fabs(ta - ref_t) + (0.05 * (fabs(log(s0 / ref_w)) + fabs(log(s1 / ref_h))));
        if (cost < best_cost) { best_cost = cost; bs0 = s0; bs1 = s1; bt = ta; }
    }
    z[2] = bs0; z[3] = bs1; z[4] = bt;
}

// KalmanFilterXYWH.update for one track with measurement z (fp32 cx, cy, w, h, theta); base.py:286-355, xywh.py:162-185
BM_STEP_FN void kf_update_wave(double* kf, const float* z32, int lane, bool /*xyah*/ = false) {
    double m[10];
    for (int k = 0; k < 10; ++k) m[k] = kf[k];
    const double* P = kf + KF_DIM;
    double z[5];
    for (int k = 0; k < 5; ++k) z[k] = (double)z32[k];
    obb_align_measurement(z, m);
    double S[5][5];
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) S[a][b] = P[a * 10 + b];
    for (int a = 0; a < 5; ++a) {
        const double sd = a == 4 ? 1e-1 : STD_POS * m[2 + (a & 1)];
        S[a][a] = S[a][a] + sd * sd;
    }
    double L[5][5];
    for (int c = 0; c < 5; ++c) {
        double d = S[c][c];
        for (int k = 0; k < c; ++k) d -= L[c][k] * L[c][k];
        d = sqrt(d);
        L[c][c] = d;
        for (int r = c + 1; r < 5; ++r) {
            double t = S[r][c];
            for (int k = 0; k < c; ++k) t -= L[r][k] * L[c][k];
            L[r][c] = t / d;
        }
    }
    auto gain_row = [&](int r, double (&K)[5]) {        // K[r][:] = S^-1 P[r][0:5]^T (cho_solve)
        double y[5];
        for (int k = 0; k < 5; ++k) {
            double t = P[r * 10 + k];
            for (int q = 0; q < k; ++q) t -= L[k][q] * y[q];
            y[k] = t / L[k][k];
        }
        for (int k = 4; k >= 0; --k) {
            double t = y[k];
            for (int q = k + 1; q < 5; ++q) t -= L[q][k] * K[q];
            K[k] = t / L[k][k];
        }
    };
    double pnew[2];
    for (int q = 0; q < 2; ++q) {
        const int e = lane + q * WAVE;
        pnew[q] = 0.0;
        if (e >= 100) continue;
        const int i = e / 10, j = e % 10;
        double Ki[5], Kj[5];
        gain_row(i, Ki);
        gain_row(j, Kj);
        double ksk = 0.0;
        for (int a = 0; a < 5; ++a) {
            double mj = 0.0;
            for (int b = 0; b < 5; ++b) mj += S[a][b] * Kj[b];
            ksk += Ki[a] * mj;
        }
        pnew[q] = P[e] - ksk;
    }
    double mnew = 0.0;
    if (lane < 10) {
        double Ki[5];
        gain_row(lane, Ki);
        double acc = 0.0;
        for (int a = 0; a < 5; ++a) acc += (z[a] - m[a]) * Ki[a];
        mnew = m[lane] + acc;
        if (lane == 9) mnew *= 0.8;                      // _damp_theta_velocity
        if (lane == 2 || lane == 3) mnew = mnew > MIN_SIZE ? mnew : MIN_SIZE;
        if (lane == 4) mnew = obb_wrap_angle(mnew);
    }
    const double p0 = __shfl(pnew[0], lane, WAVE), p1 = __shfl(pnew[1], lane, WAVE);
    kf[KF_DIM + lane] = p0;
    if (lane + WAVE < 100) kf[KF_DIM + lane + WAVE] = p1;
    if (lane < 10) kf[lane] = mnew;
}

// KalmanFilterXYWH.initiate (xywh.py:133-140, base.py:234-244, std xywh.py:22-36)
BM_STEP_FN void kf_initiate_wave(double* kf, const float* z32, int lane, bool /*xyah*/ = false) {
    const double w = (double)z32[2], h = (double)z32[3];
    for (int q = 0; q < 2; ++q) {
        const int e = lane + q * WAVE;
        if (e >= 100) continue;
        const int i = e / 10, j = e % 10;
        double c = 0.0;
        if (i == j) {
            const int i5 = i < 5 ? i : i - 5;
            const double sd = (i5 == 4) ? (i < 5 ? 1e-2 : 1e-5) : ((i < 5) ? (2 * STD_POS) * obb_dim_of(i5, w, h) : (10 * STD_VEL) * obb_dim_of(i5, w, h));
            c = sd * sd;
        }
        kf[KF_DIM + e] = c;
    }
    if (lane < 10) {
        double mv = lane < 5 ? (double)z32[lane] : 0.0;
        if (lane == 4) mv = obb_wrap_angle(obb_wrap_angle(mv));     // wrapped on the way in and by _enforce_xywh_constraints
        if (lane == 2 || lane == 3) mv = mv > MIN_SIZE ? mv : MIN_SIZE;
        kf[lane] = mv;
    }
}

// STrack.multi_gmc_obb for one track (botsort_track.py:134-230): the box is warped as its four corners (fp32, cv2.boxPoints /
// cv2.transform), refitted by the minimum-area enclosing rectangle (cv2.minAreaRect: every edge of the warped quadrilateral is tried,
// fp64 on the fp32 points, result in fp32) and re-aligned to the box it came from; (vx, vy) <- L (vx, vy), vw *= sx, vh *= sy and
// cov <- T cov T^T with T = diag-blocks (L, sx, sy, 1, L, sx, sy, 1), all of them the fp32 values of the warp.  W = [l00 l01 tx; l10 l11 ty].
// The two OpenCV calls are restated (OpenCV is absent offline: parity unpinned for them).
BM_STEP_FN void kf_warp_wave(double* kf, const double* W, int lane) {
    float w32[6];
    for (int k = 0; k < 6; ++k) w32[k] = (float)W[k];
    const float sxf = sqrtf(w32[0] * w32[0] + w32[3] * w32[3]), syf = sqrtf(w32[1] * w32[1] + w32[4] * w32[4]);     // column norms, fp32
    const double sx = (double)sxf > 1e-6 ? (double)sxf : 1e-6, sy = (double)syf > 1e-6 ? (double)syf : 1e-6;
    double m[10];
    for (int k = 0; k < 10; ++k) m[k] = kf[k];
    // the box as fp32, its corners, the warped corners
    double ref[5];
    for (int k = 0; k < 5; ++k) ref[k] = (double)(float)m[k];
    double rect[5] = {ref[0], ref[1], ref[2] > 1e-4 ? ref[2] : 1e-4, ref[3] > 1e-4 ? ref[3] : 1e-4, ref[4]};
    double p[4][2];
    obb_corners_deg(rect, (double)((float)m[4] * (180.0f / 3.14159274f)), p);      // np.degrees of an fp32 angle: x * (180.0f / NPY_PIf), fp32
    double q[4][2];
    for (int k = 0; k < 4; ++k) {
        const float x = (float)p[k][0], y = (float)p[k][1];
        q[k][0] = (double)(w32[0] * x + w32[1] * y + w32[2]);
        q[k][1] = (double)(w32[3] * x + w32[4] * y + w32[5]);
    }
    // minimum-area enclosing rectangle: a side lies on an edge
    double best_area = 0.0, bcx = q[0][0], bcy = q[0][1], bw = 0.0, bh = 0.0, bang = 0.0;
    bool have = false;
    for (int e = 0; e < 4; ++e) {
        const int n = (e + 1) & 3;
        const double ex = q[n][0] - q[e][0], ey = q[n][1] - q[e][1];
        const double len = hypot(ex, ey);
        if (len == 0.0) continue;
        const double ux = ex / len, uy = ey / len, vx = -uy, vy = ux;
        double amin = 0, amax = 0, bmin = 0, bmax = 0;
        for (int k = 0; k < 4; ++k) {
            const double a = q[k][0] * ux + q[k][1] * uy, b = q[k][0] * vx + q[k][1] * vy;
            if (k == 0 || a < amin) amin = a;
            if (k == 0 || a > amax) amax = a;
            if (k == 0 || b < bmin) bmin = b;
            if (k == 0 || b > bmax) bmax = b;
        }
        const double ww = amax - amin, hh = bmax - bmin;
        if (!have || ww * hh < best_area) {
            have = true; best_area = ww * hh;
            const double ca = (amax + amin) / 2, cb = (bmax + bmin) / 2;
            bcx = ux * ca + vx * cb; bcy = uy * ca + vy * cb; bw = ww; bh = hh;
            bang = atan2(uy, ux) * (180.0 / OBB_PI);
        }
    }
    double z[5];
    z[0] = (double)(float)bcx; z[1] = (double)(float)bcy;
    const double fw = (double)(float)bw, fh = (double)(float)bh, fang = (double)(float)bang;
    z[2] = (double)(float)(fw > 1e-4 ? fw : 1e-4); z[3] = (double)(float)(fh > 1e-4 ? fh : 1e-4);
    z[4] = (double)(float)(fang * (OBB_PI / 180.0));                                       // np.deg2rad into the fp32 box
    obb_align_measurement(z, ref);
    double mn = 0.0;
    if (lane < 10) {
        if (lane < 5) mn = (double)(float)z[lane];
        else if (lane == 5) mn = (double)w32[0] * m[5] + (double)w32[1] * m[6];
        else if (lane == 6) mn = (double)w32[3] * m[5] + (double)w32[4] * m[6];
        else if (lane == 7) mn = m[7] * sx;
        else if (lane == 8) mn = m[8] * sy;
        else mn = m[9];
    }
    // cov <- (T P) T^T: a row of T has the entries of L in rows 0, 1 (columns 0, 1) and 5, 6 (columns 5, 6), one diagonal entry elsewhere
    const double* P = kf + KF_DIM;
    auto trow = [&](int i, int (&idx)[2], double (&cf)[2]) {
        if (i == 0 || i == 1 || i == 5 || i == 6) {
            const int b0 = i < 2 ? 0 : 5, r = i - b0;
            idx[0] = b0; idx[1] = b0 + 1; cf[0] = (double)w32[r * 3 + 0]; cf[1] = (double)w32[r * 3 + 1];
            return 2;
        }
        idx[0] = i; idx[1] = i;
        cf[0] = (i == 2 || i == 7) ? (double)(float)sx : ((i == 3 || i == 8) ? (double)(float)sy : 1.0); cf[1] = 0.0;
        return 1;
    };
    double cnew[2];
    for (int qq = 0; qq < 2; ++qq) {
        const int e = lane + qq * WAVE;
        cnew[qq] = 0.0;
        if (e >= 100) continue;
        const int i = e / 10, j = e % 10;
        int ia[2], jb[2];
        double ca[2], cb[2];
        const int na = trow(i, ia, ca), nb = trow(j, jb, cb);
        double acc = 0.0;
        for (int b = 0; b < nb; ++b) {
            double tp = 0.0;                                   // (T P)[i][jb[b]]
            for (int a = 0; a < na; ++a) tp = a == 0 ? ca[a] * P[ia[a] * 10 + jb[b]] : fma(ca[a], P[ia[a] * 10 + jb[b]], tp);
            acc = b == 0 ? tp * cb[b] : fma(tp, cb[b], acc);
        }
        cnew[qq] = acc;
    }
    const double c0 = __shfl(cnew[0], lane, WAVE), c1 = __shfl(cnew[1], lane, WAVE);      // wave-wide dependency: every load precedes the stores
    kf[KF_DIM + lane] = c0;
    if (lane + WAVE < 100) kf[KF_DIM + lane + WAVE] = c1;
    if (lane < 10) kf[lane] = mn;
}

#endif

// STrack.update_cls (botsort_track.py:69-82); single lane.
__device__ inline void vote_cls(SV& v, int slot, float cls, float conf) {
    float best = 0.0f;
    bool seen = false;
    const int n = v.hist_n[slot];
    float* hc = v.hist_cls + slot * KCLS;
    float* hw = v.hist_w + slot * KCLS;
    for (int k = 0; k < n; ++k) {
        if (cls == hc[k]) { hw[k] = hw[k] + conf; seen = true; }
        if (hw[k] > best) { best = hw[k]; v.cls[slot] = hc[k]; }
    }
    if (!seen) {
        if (n < KCLS) { hc[n] = cls; hw[n] = conf; v.hist_n[slot] = n + 1; }
        else *v.status = STATUS_CLASS_CAPACITY;
        v.cls[slot] = cls;
    }
}

// fp32 L2 norm of a vector by one wavefront (np.linalg.norm, fp32 accumulate).
__device__ inline float wave_norm_f32(const float* x, int dim, int lane) {
    float s = 0.0f;
    for (int k = lane; k < dim; k += WAVE) s += x[k] * x[k];
    return sqrtf(wave_sum(s));
}

constexpr int VEC_REGS_PER_LANE = 8;    // appearance vectors up to 64 * 8 = 512 floats stay in registers; longer ones are streamed

// STrack.update_features (botsort_track.py:58-67) for a live track: the matched
// detection's vector is normalised once more, blended, renormalised.  Up to 512 floats: one read of each vector, one write,
// everything else in registers.  Longer vectors (CLIP-ReID's 1280) are streamed in three passes with the SAME arithmetic per
// element and the same per-lane summation order (k = lane, lane + 64, ...), so both forms return the same bits -- a register
// form sized for 2048 floats kept 64 registers live across the frame step's largest phase and was most of its spill traffic.
BM_STEP_FN void blend_feature_wave(float* smooth, const float* feat, int dim, int lane) {
    if (dim <= VEC_REGS_PER_LANE * WAVE) {
        float f[VEC_REGS_PER_LANE], sm[VEC_REGS_PER_LANE];
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < VEC_REGS_PER_LANE; ++q) {
            const int k = lane + q * WAVE;
            f[q] = k < dim ? feat[k] : 0.0f;
            sm[q] = k < dim ? smooth[k] : 0.0f;
            s += f[q] * f[q];
        }
        const float nf = sqrtf(wave_sum(s));
        s = 0.0f;
#pragma unroll
        for (int q = 0; q < VEC_REGS_PER_LANE; ++q) {
            const float b = 0.9f * sm[q] + 0.1f * (f[q] / nf);
            sm[q] = b;
            s += b * b;
        }
        const float ns = sqrtf(wave_sum(s));
#pragma unroll
        for (int q = 0; q < VEC_REGS_PER_LANE; ++q) {
            const int k = lane + q * WAVE;
            if (k < dim) smooth[k] = sm[q] / ns;
        }
        return;
    }
    float s = 0.0f;
    for (int k = lane; k < dim; k += WAVE) { const float f = feat[k]; s += f * f; }
    const float nf = sqrtf(wave_sum(s));
    s = 0.0f;
    for (int k = lane; k < dim; k += WAVE) { const float b = 0.9f * smooth[k] + 0.1f * (feat[k] / nf); s += b * b; }
    const float ns = sqrtf(wave_sum(s));
    for (int k = lane; k < dim; k += WAVE) { const float b = 0.9f * smooth[k] + 0.1f * (feat[k] / nf); smooth[k] = b / ns; }
}

// STrack constructor's update_features on a fresh detection (botsort_track.py:58-66):
// feat /= |feat|; smooth = feat; smooth /= |smooth|  -> the vector is normalised twice.
BM_STEP_FN void normalize_twice_wave(const float* src, float* dst, int dim, int lane) {
    if (dim <= VEC_REGS_PER_LANE * WAVE) {
        float x[VEC_REGS_PER_LANE];
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < VEC_REGS_PER_LANE; ++q) {
            const int k = lane + q * WAVE;
            x[q] = k < dim ? src[k] : 0.0f;
            s += x[q] * x[q];
        }
        const float n1 = sqrtf(wave_sum(s));
        s = 0.0f;
#pragma unroll
        for (int q = 0; q < VEC_REGS_PER_LANE; ++q) { x[q] = x[q] / n1; s += x[q] * x[q]; }
        const float n2 = sqrtf(wave_sum(s));
#pragma unroll
        for (int q = 0; q < VEC_REGS_PER_LANE; ++q) {
            const int k = lane + q * WAVE;
            if (k < dim) dst[k] = x[q] / n2;
        }
        return;
    }
    float s = 0.0f;
    for (int k = lane; k < dim; k += WAVE) { const float x = src[k]; s += x * x; }
    const float n1 = sqrtf(wave_sum(s));
    s = 0.0f;
    for (int k = lane; k < dim; k += WAVE) { const float x = src[k] / n1; s += x * x; }
    const float n2 = sqrtf(wave_sum(s));
    for (int k = lane; k < dim; k += WAVE) dst[k] = (src[k] / n1) / n2;
}

// STrack.update / re_activate (botsort_track.py:244-282) for a list of
// (slot, det) pairs, one wavefront per pair.
BM_STEP_FN void apply_matches(const Ctx& c, SV& v, int n_match, int frame, bool with_feat) {
    for (int base = 0; base < n_match; base += c.nwaves) {
        const int k = base + c.wave;
        if (k < n_match) {
            const int slot = v.match_slot[k], d = v.match_det[k];
            const bool was_tracked = v.match_flag[k] != 0;
            kf_update_wave(v.kf + (long)slot * KF_STRIDE, v.det_xywh + d * BOX_W, c.lane, v.cfg.kind == 1);
            if (with_feat) blend_feature_wave(v.smooth + (long)slot * v.dim, v.det_feat + (long)d * v.dim, v.dim, c.lane);
            if (c.lane == 0) {
                v.tracklet_len[slot] = was_tracked ? v.tracklet_len[slot] + 1 : 0;
                v.frame_id[slot] = frame;
                v.state[slot] = ST_TRACKED;
                v.is_activated[slot] = 1;
                const float conf = v.dets[d * DET_COLS + CONF_COL], cls = v.dets[d * DET_COLS + CONF_COL + 1];
                v.conf[slot] = conf;
                v.cls[slot] = cls;
                v.det_ind[slot] = (float)d;
                if (v.cfg.kind == 0) vote_cls(v, slot, cls, conf);      // ByteTrack takes the detection's class (bytetrack.py:139)
            }
        }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// Cost matrices
// ---------------------------------------------------------------------------
#if !BM_OBB
// fp64 xyxy of the listed tracks from their Kalman mean (STrack.xyxy, botsort_track.py:310-316)
__device__ inline void track_boxes(const Ctx& c, const SV& v, const int* rows, int n, double* box) {
    for (int r = c.tid; r < n; r += c.nthr) {
        const double* m = v.kf + (long)rows[r] * KF_STRIDE;
        const double hw = (v.cfg.kind == 1 ? m[2] * m[3] : m[2]) / 2, hh = m[3] / 2;     // xyah: w = a * h (bytetrack.py:185-186)
        box[r * 4 + 0] = m[0] - hw;
        box[r * 4 + 1] = m[1] - hh;
        box[r * 4 + 2] = m[0] + hw;
        box[r * 4 + 3] = m[1] + hh;
    }
    __syncthreads();
}

// 1 - IoU between an fp64 box and an fp32 detection box whose area was rounded to
// fp32 first (iou.py:133-150 with mixed dtypes, matching.py:46-80).
__device__ inline double iou_dist_td(const double* a, const float* b, float area_b) {
    const double bx1 = b[0], by1 = b[1], bx2 = b[2], by2 = b[3];
    const double xx1 = a[0] > bx1 ? a[0] : bx1, yy1 = a[1] > by1 ? a[1] : by1;
    const double xx2 = a[2] < bx2 ? a[2] : bx2, yy2 = a[3] < by2 ? a[3] : by2;
    double w = xx2 - xx1, h = yy2 - yy1;
    w = w > 0.0 ? w : 0.0;
    h = h > 0.0 ? h : 0.0;
    const double wh = w * h;
    const double o = wh / ((a[2] - a[0]) * (a[3] - a[1]) + (double)area_b - wh);
    return 1 - o;
}

__device__ inline double iou_dist_tt(const double* a, const double* b) {
    const double xx1 = a[0] > b[0] ? a[0] : b[0], yy1 = a[1] > b[1] ? a[1] : b[1];
    const double xx2 = a[2] < b[2] ? a[2] : b[2], yy2 = a[3] < b[3] ? a[3] : b[3];
    double w = xx2 - xx1, h = yy2 - yy1;
    w = w > 0.0 ? w : 0.0;
    h = h > 0.0 ? h : 0.0;
    const double wh = w * h;
    const double o = wh / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - wh);
    return 1 - o;
}

#else
// STrack.xywha of the listed tracks: the fp32 of the filter mean's (cx, cy, w, h, theta) (botsort_track.py:319-327, bytetrack.py:191-198)
__device__ inline void track_boxes(const Ctx& c, const SV& v, const int* rows, int n, double* box) {
    for (int r = c.tid; r < n; r += c.nthr) {
        const double* m = v.kf + (long)rows[r] * KF_STRIDE;
        for (int k = 0; k < 5; ++k) box[r * 5 + k] = (double)(float)m[k];
    }
    __syncthreads();
}

// (obb_corners, obb_iou -- the rotated IoU of two (cx, cy, w, h, theta) boxes: obb_geometry.hpp)
// track (fp64 of its fp32 xywha) against a detection's fp32 xywha; `area_b` is not used in this layout
__device__ inline double iou_dist_td(const double* a, const float* b, float /*area_b*/) {
    double bd[5];
    for (int k = 0; k < 5; ++k) bd[k] = (double)b[k];
    return 1 - obb_iou(a, bd);
}
__device__ inline double iou_dist_tt(const double* a, const double* b) { return 1 - obb_iou(a, b); }

#endif
// Association cost for rows (track slots) x cols (detection indices), stored DETECTION-MAJOR
// (cost[c * cap + r]) because the assignment solver scans one detection column over all tracks:
//   iou_d, gate = iou_d > proximity, optional fuse_score, cosine distance (optionally / emb_scale),
//   appearance + proximity gating, element-wise min.
// botsort.py:306-317 (first association) and :396-413 (unconfirmed tracks).
// Gated pairs get emb = 1.0 whatever their cosine is (botsort.py:314), so the cosine is evaluated
// only for the pairs that pass the IoU gate (typically ~1 per detection): one wavefront per pair,
// fp32 x fp32 products accumulated in fp64.  Scenes with more than SPARSE_MAX ungated pairs fall
// back to the dense LDS-tiled contraction (64x64 pair tiles, k staged through LDS in chunks of 32,
// every pair accumulated in ascending k like scipy's cdist loop).
#ifndef BM_SPARSE_MAX
#define BM_SPARSE_MAX 4096
#endif
constexpr int SPARSE_MAX = BM_SPARSE_MAX;     // tests build a variant with 0 to force the dense path

// embedding_distance of one pair (matching.py:85-107): np.maximum(0.0, cdist(..., "cosine"))
__device__ inline double cosine_dist(double dot, double nu, double nv) {
    double cosv = dot / (nu * nv);
    if (fabs(cosv) > 1.0) cosv = cosv > 0 ? 1.0 : -1.0;       // scipy clips rounding overshoot
    const double e = 1.0 - cosv;
    return e > 0.0 ? e : (e != e ? e : 0.0);                    // np.maximum(0.0, e)
}
__device__ inline double cosine_gate(double dot, double nu, double nv, double emb_scale, double app, bool gate) {
    double e = cosine_dist(dot, nu, nv);
    if (emb_scale > 0.0) e = e / emb_scale;
    if (e > app) e = 1.0;
    if (gate) e = 1.0;
    return e;
}
__device__ inline double np_minimum(double a, double b) { return (a != a || b != b) ? (a != a ? a : b) : (a < b ? a : b); }

template <int NTHR>
BM_STEP_BIG_FN void assoc_cost(const Ctx& c, SV& v, const int* rows, int n_rows, const int* cols,
                                  int n_cols, bool use_emb, double emb_scale, bool fuse,
                                  float (*sA)[COST_KC + 1], float (*sB)[COST_KC + 1], int* s_count, double* dbg = nullptr) {
    if (n_rows == 0 || n_cols == 0) return;
    track_boxes(c, v, rows, n_rows, v.box_a);
    const long ld = v.cap;
    // parity debugging only (BotSortStepArgs::dbg_cost): planes 1 and 2 of this stage
    double* dbg_iou = dbg ? dbg + (long)v.nd * ld : nullptr;
    double* dbg_emb = dbg ? dbg + 2 * (long)v.nd * ld : nullptr;
    if (c.tid == 0) *s_count = 0;
    __syncthreads();
    // pass 1: IoU part for every pair; collect the pairs whose appearance term matters
    for (int o = c.tid; o < n_rows * n_cols; o += c.nthr) {
        const int cc = o / n_rows, r = o % n_rows;
        const int d = cols[cc];
        double iou_d = iou_dist_td(v.box_a + r * BOX_W, DET_BOX(v, d), v.det_area[d]);
        const bool gate = iou_d > v.cfg.proximity_thresh;
        if (dbg) { dbg_iou[cc * ld + r] = iou_d; dbg_emb[cc * ld + r] = __builtin_nan(""); }
        if (fuse) {
            const double sim = 1 - iou_d;
            iou_d = 1 - sim * (double)v.dets[d * DET_COLS + CONF_COL];
        }
        double out = iou_d;
        if (use_emb) {
            if (gate) out = np_minimum(iou_d, 1.0);
            else {
                const int k = atomicAdd(s_count, 1);
                if (k < SPARSE_MAX) v.pair_list[k] = o;
            }
        }
        v.cost[cc * ld + r] = out;
    }
    __syncthreads();
    if (!use_emb) return;
    const int n_pairs = *s_count;
    __syncthreads();
    if (n_pairs <= SPARSE_MAX || n_pairs == 0) {
        for (int base = 0; base < n_pairs; base += c.nwaves) {
            const int k = base + c.wave;
            if (k < n_pairs) {
                const int o = v.pair_list[k];
                const int cc = o / n_rows, r = o % n_rows;
                const float* a = v.smooth + (long)rows[r] * v.dim;
                const float* b = v.det_feat + (long)cols[cc] * v.dim;
                double dot = 0.0, na = 0.0, nb = 0.0;
                for (int q = c.lane; q < v.dim; q += WAVE) {
                    const double x = (double)a[q], y = (double)b[q];
                    dot += x * y; na += x * x; nb += y * y;
                }
                dot = wave_sum(dot); na = wave_sum(na); nb = wave_sum(nb);
                if (c.lane == 0) {
                    const double e = cosine_gate(dot, sqrt(na), sqrt(nb), emb_scale, v.cfg.appearance_thresh, false);
                    double* dst = v.cost + cc * ld + r;
                    *dst = np_minimum(*dst, e);
                    if (dbg) dbg_emb[cc * ld + r] = cosine_dist(dot, sqrt(na), sqrt(nb));
                }
            }
        }
        __syncthreads();
        return;
    }
    // dense path: norms (scipy _row_norms: sqrt(sum x^2) in fp64), then the LDS-tiled contraction
    for (int base = 0; base < n_rows + n_cols; base += c.nwaves) {
        const int q = base + c.wave;
        if (q < n_rows + n_cols) {
            const float* x = (q < n_rows) ? v.smooth + (long)rows[q] * v.dim
                                          : v.det_feat + (long)cols[q - n_rows] * v.dim;
            double s = 0.0;
            for (int k = c.lane; k < v.dim; k += WAVE) s += (double)x[k] * (double)x[k];
            s = wave_sum(s);
            if (c.lane == 0) {
                if (q < n_rows) v.trk_norm[q] = sqrt(s);
                else v.det_norm[q - n_rows] = sqrt(s);
            }
        }
    }
    __syncthreads();
    constexpr int per_thread = (COST_TILE * COST_TILE + NTHR - 1) / NTHR;   // fp64 accumulators per thread
    for (int r0 = 0; r0 < n_rows; r0 += COST_TILE) {
        for (int c0 = 0; c0 < n_cols; c0 += COST_TILE) {
            double acc[per_thread];
#pragma unroll
            for (int m = 0; m < per_thread; ++m) acc[m] = 0.0;
            for (int k0 = 0; k0 < v.dim; k0 += COST_KC) {
                for (int e = c.tid; e < COST_TILE * COST_KC; e += c.nthr) {
                    const int rr = e / COST_KC, kk = e % COST_KC;
                    const int k = k0 + kk;
                    float a = 0.0f, b = 0.0f;
                    if (k < v.dim) {
                        if (r0 + rr < n_rows) a = v.smooth[(long)rows[r0 + rr] * v.dim + k];
                        if (c0 + rr < n_cols) b = v.det_feat[(long)cols[c0 + rr] * v.dim + k];
                    }
                    sA[rr][kk] = a;
                    sB[rr][kk] = b;
                }
                __syncthreads();
#pragma unroll
                for (int m = 0; m < per_thread; ++m) {
                    const int o = c.tid + m * NTHR;
                    if (o < COST_TILE * COST_TILE) {
                        const int rr = o / COST_TILE, cc = o % COST_TILE;
                        double s = acc[m];
                        for (int kk = 0; kk < COST_KC; ++kk) s += (double)sA[rr][kk] * (double)sB[cc][kk];
                        acc[m] = s;
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int m = 0; m < per_thread; ++m) {
                const int o = c.tid + m * NTHR;
                if (o >= COST_TILE * COST_TILE) continue;
                const int r = r0 + o / COST_TILE, cc = c0 + o % COST_TILE;
                if (r >= n_rows || cc >= n_cols) continue;
                const int d = cols[cc];
                // the gate is re-derived from the un-fused IoU distance
                const double raw = iou_dist_td(v.box_a + r * BOX_W, DET_BOX(v, d), v.det_area[d]);
                const bool gate = raw > v.cfg.proximity_thresh;
                if (dbg) dbg_emb[cc * ld + r] = cosine_dist(acc[m], v.trk_norm[r], v.det_norm[cc]);
                if (gate) continue;                                   // pass 1 already stored min(iou_d, 1)
                const double e = cosine_gate(acc[m], v.trk_norm[r], v.det_norm[cc], emb_scale, v.cfg.appearance_thresh, false);
                double* dst = v.cost + cc * ld + r;
                *dst = np_minimum(*dst, e);
            }
        }
    }
    __syncthreads();
}

// Parity debugging (BotSortStepArgs::dbg_cost): the matrix the solver is about to be given -> plane 0 of `stage`, and its shape.
__device__ inline double* dbg_stage(const BotSortStepArgs& args, const SV& v, int s, int stage) {
    return args.dbg_cost ? args.dbg_cost + ((long)s * DBG_STAGES + stage) * DBG_PLANES * (long)v.nd * v.cap : nullptr;
}
__device__ inline void dbg_copy_cost(const Ctx& c, const BotSortStepArgs& args, const SV& v, int s, int stage, int n_rows, int n_cols,
                                     bool iou_only = false) {
    double* dst = dbg_stage(args, v, s, stage);
    if (!dst) return;
    const long ld = v.cap;
    for (int o = c.tid; o < n_rows * n_cols; o += c.nthr) {
        const int cc = o / n_rows, r = o % n_rows;
        dst[cc * ld + r] = v.cost[cc * ld + r];
        if (iou_only) dst[(long)v.nd * ld + cc * ld + r] = v.cost[cc * ld + r];     // the IoU-only stage: plane 1 is the same matrix
    }
    if (c.tid == 0) { args.dbg_shape[(s * DBG_STAGES + stage) * 2] = n_rows; args.dbg_shape[(s * DBG_STAGES + stage) * 2 + 1] = n_cols; }
    __syncthreads();
}

// IoU-only cost (second association, botsort.py:356), detection-major.
BM_STEP_BIG_FN void iou_cost(const Ctx& c, SV& v, const int* rows, int n_rows, const int* cols, int n_cols) {
    if (n_rows == 0 || n_cols == 0) return;
    track_boxes(c, v, rows, n_rows, v.box_a);
    const long ld = v.cap;
    for (int o = c.tid; o < n_rows * n_cols; o += c.nthr) {
        const int cc = o / n_rows, r = o % n_rows;
        const int d = cols[cc];
        v.cost[cc * ld + r] = iou_dist_td(v.box_a + r * BOX_W, DET_BOX(v, d), v.det_area[d]);
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// Linear assignment with cost limit:  lap.lapjv(cost, extend_cost=True,
// cost_limit=L) (matching.py:28-43).  The extended (R+C)^2 problem lapx builds
// is equivalent to  min sum_{matched}(c_ij - L)  over partial matchings, so it
// is solved directly on the R x C matrix: detections are inserted one by one
// with a shortest-augmenting-path search (Dijkstra over the tracks, potentials
// u/v), each detection also owning a private zero-cost "stay unmatched" sink.
// One workgroup; solver state (potentials, distances, predecessor links,
// matching) lives in LDS, the cost column of the detection being relaxed is one
// coalesced read; the relaxation is parallel over tracks, the arg-min is a
// wave-shuffle + LDS reduction.  Exact; ties broken towards the lower index.
// ---------------------------------------------------------------------------
struct LapLds {     // carved from dynamic LDS: R = cap entries for tracks, C = max_dets for detections
    double* v; double* minv; double* u; int* x; int* way; int* used; int* y;
};
__device__ inline LapLds carve_lap(unsigned char* base, int cap, int nd) {
    LapLds l;
    l.v = reinterpret_cast<double*>(base);
    l.minv = l.v + cap;
    l.u = l.minv + cap;
    l.x = reinterpret_cast<int*>(l.u + nd);
    l.way = l.x + cap;
    l.used = l.way + cap;
    l.y = l.used + cap;
    return l;
}
__host__ __device__ inline long lap_lds_bytes(int cap, int nd) { return (long)cap * (8 + 8 + 4 + 4 + 4) + (long)nd * (8 + 4) + 16; }

BM_STEP_BIG_FN void lap_solve(const Ctx& c, SV& v, const LapLds& L, int R, int C, double limit) {
    const long ld = v.cap;
    for (int t = c.tid; t < R; t += c.nthr) { L.x[t] = -1; L.v[t] = 0.0; }
    for (int d = c.tid; d < C; d += c.nthr) { L.y[d] = -1; L.u[d] = 0.0; }
    __syncthreads();
    bool stalled = false;
    if (R > 0 && C > 0) {
        // the first cost column of detection s + 1 is requested while detection s is inserted (one element per thread when the tracks
        // fit the workgroup): the first relaxation of an insertion -- most insertions have no second -- does not wait for L2
        const bool pre = R <= c.nthr;
        double nxt = (pre && c.tid < R) ? v.cost[c.tid] : 0.0;
        for (int s = 0; s < C && !stalled; ++s) {
            const double col0 = nxt;
            if (pre && s + 1 < C && c.tid < R) nxt = v.cost[(long)(s + 1) * ld + c.tid];
            for (int t = c.tid; t < R; t += c.nthr) { L.minv[t] = LAP_INF; L.used[t] = 0; L.way[t] = -1; }
            int cur = s;          // detection being relaxed
            int via = -1;         // track through which `cur` was reached (-1 = root)
            double sink_best = LAP_INF;
            int sink_via = -1;
            int end_track = -1;
            bool end_sink = false;
            const int max_iter = R + 2;
            int iter = 0;
            for (; iter < max_iter; ++iter) {
                const double ucur = L.u[cur];
                const double* col = v.cost + cur * ld;
                double best = LAP_INF;
                int best_t = -1;
                for (int t = c.tid; t < R; t += c.nthr) {
                    if (L.used[t]) continue;
                    const double cst = (pre && iter == 0) ? col0 : col[t];
                    double mv = L.minv[t];
                    if (cst < limit) {
                        const double cand = (cst - limit) - ucur - L.v[t];
                        if (cand < mv) { mv = cand; L.minv[t] = cand; L.way[t] = via; }
                    }
                    if (mv < best || (mv == best && best_t < 0)) { best = mv; best_t = t; }
                }
                if (best >= LAP_INF) best_t = -1;
                const double sink_cand = 0.0 - ucur;
                if (sink_cand < sink_best) { sink_best = sink_cand; sink_via = via; }
                double gmin;
                int gt;
                block_argmin(c, best, best_t, gmin, gt);
                end_sink = (gt < 0) || (sink_best <= gmin);
                const double delta = end_sink ? sink_best : gmin;
                // potentials
                for (int t = c.tid; t < R; t += c.nthr) {
                    if (L.used[t]) {
                        L.u[L.x[t]] += delta;
                        L.v[t] -= delta;
                    } else if (L.minv[t] < LAP_INF) {
                        L.minv[t] -= delta;
                    }
                }
                if (c.tid == 0) L.u[s] += delta;
                sink_best -= delta;
                if (end_sink) break;
                if (c.tid == (gt % c.nthr)) L.used[gt] = 1;
                __syncthreads();
                if (L.x[gt] < 0) { end_track = gt; break; }
                via = gt;
                cur = L.x[gt];
            }
            __syncthreads();
            if (iter >= max_iter) { stalled = true; break; }
            // augment along the way[] chain (short; one thread)
            if (c.tid == 0) {
                int t;
                if (end_sink) {
                    t = sink_via;            // detection reached through `t` (or the root) stays unmatched
                    if (t >= 0) L.y[L.x[t]] = -1;
                } else {
                    t = end_track;
                }
                int guard = 0;
                while (t >= 0 && guard++ <= R) {
                    const int prev = L.way[t];
                    const int det = (prev >= 0) ? L.x[prev] : s;
                    L.x[t] = det;
                    L.y[det] = t;
                    t = prev;
                }
            }
            __syncthreads();
        }
    }
    if (stalled && c.tid == 0) *v.status = STATUS_LAP_STALL;
    // publish the matching for the bookkeeping phases
    for (int t = c.tid; t < R; t += c.nthr) v.lap_x[t] = stalled ? -1 : L.x[t];
    for (int d = c.tid; d < C; d += c.nthr) v.lap_y[d] = stalled ? -1 : L.y[d];
    __syncthreads();
}

// ---------------------------------------------------------------------------
// The frame step
// ---------------------------------------------------------------------------
// DBG: the parity-debugging instantiation (writes BotSortStepArgs::dbg_cost when it is set).  The product kernel is compiled with
// DBG = false: every debug branch folds away and its register allocation is what it was without the feature.
template <int NTHR, bool DBG = false>
__device__ inline void botsort_step_stream(const BotSortStepArgs& args, int s, int* s_int, double* s_dbl,
                                           float (*sA)[COST_KC + 1], float (*sB)[COST_KC + 1], unsigned char* dyn_lds) {
    if (args.n_dets[s] < 0) {                 // "no update for this stream in this call" (frames without detections are
        if (threadIdx.x == 0) args.out_n[s] = 0;      // not passed to the tracker by the reference's replay loop, replay.py:318-341)
        return;
    }
    const Ctx c = make_ctx(s_int, s_dbl);
    const LapLds lap = carve_lap(dyn_lds, args.st.cap, args.sc.max_dets);
    int* s_count = s_int + MAX_WAVES;      // one spare LDS word (s_int has MAX_WAVES + 1 entries)
    const int sel = args.list_sel ? args.list_sel[s] : 0;
    SV v = (make_view)(args, s, sel);      // (parenthesised: no argument-dependent lookup -- the body exists in two namespaces)
    const BotSortConfigDev& cfg = v.cfg;
    const bool reid = cfg.with_reid != 0;
    const int cap = v.cap, dim = v.dim;

    if (c.tid == 0) {
        if (args.frame_count_set) *v.frame_count = args.frame_count_set[s];
        *v.frame_count += 1;                       // botsort.py:183
        *v.stamp += 4;
    }
    __syncthreads();
    const int frame = *v.frame_count;
    const int stamp = *v.stamp;                    // marks: stamp+0..3 are fresh this step
    long long* pclk = (args.phase_clock && s == args.stream_base && c.tid == 0) ? args.phase_clock : nullptr;
    int pidx = 0;
    auto tick = [&]() { if (pclk) pclk[pidx++] = BM_CLOCK(); };
    tick();
    const int n = v.n_dets;

    // ---- detections: fp32 xywh / xyxy / area, confidence split (botsort.py:251-261,
    //      botsort_track.py:46-50, geometry.py:10-42) ----
    for (int j = c.tid; j < n; j += c.nthr) {
        const float* d = v.dets + j * DET_COLS;
#if BM_OBB
        float* q5 = v.det_xywh + j * BOX_W;             // the detection's (cx, cy, w, h, theta) as it is (botsort_track.py:52-56)
        for (int k = 0; k < 5; ++k) q5[k] = d[k];
        continue;
#endif
        const float cx = (d[0] + d[2]) * 0.5f, cy = (d[1] + d[3]) * 0.5f;
        const float w = d[2] - d[0], h = d[3] - d[1];
        float* q = v.det_xywh + j * 4;
        q[0] = cx; q[1] = cy; q[2] = w; q[3] = h;
        if (cfg.kind == 1) {        // the filter's measurement is tlwh2xyah(xywh2tlwh(xywh)) in fp32 (bytetrack.py:33-35, geometry.py:56-99)
            const float tl = cx - w / 2.0f, tt = cy - h / 2.0f;
            q[0] = tl + (w / 2); q[1] = tt + (h / 2); q[2] = w / h;
        }
        float* b = v.det_xyxy + j * 4;
        b[0] = cx - w * 0.5f; b[1] = cy - h * 0.5f; b[2] = cx + w * 0.5f; b[3] = cy + h * 0.5f;
        v.det_area[j] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    __syncthreads();
    auto conf_of = [&](int j) { return (double)v.dets[j * DET_COLS + CONF_COL]; };
    auto ident = [](int i) { return i; };
    const int n_first = block_append_if(c, n, [&](int j) { return conf_of(j) > cfg.track_high_thresh; }, ident, v.first_idx, 0);
    const int n_second = block_append_if(c, n, [&](int j) {
        const double cf = conf_of(j);
        return cf > cfg.track_low_thresh && cf < cfg.track_high_thresh; }, ident, v.second_idx, 0);

    tick();
    // ---- detection appearance vectors: L2-normalised twice, as the STrack
    //      constructor does (botsort_track.py:58-66: feat /= |feat|; smooth = feat; smooth /= |smooth|) ----
    if (reid) {
        for (int base = 0; base < n_first; base += c.nwaves) {
            const int k = base + c.wave;
            if (k < n_first) {
                const int j = v.first_idx[k];
                normalize_twice_wave(v.embs + (long)j * dim, v.det_feat + (long)j * dim, dim, c.lane);
            }
        }
        __syncthreads();
    }

    tick();
    // ---- unconfirmed / confirmed split and the association pool (botsort.py:276-283, :202) ----
    const int n_act0 = *v.n_active, n_lost0 = *v.n_lost;
    const int* al = v.active_list;
    auto act_slot = [&](int i) { return al[i]; };
    const int n_unconf = block_append_if(c, n_act0, [&](int i) { return v.is_activated[al[i]] == 0; }, act_slot, v.unconf, 0);
    int n_pool = block_append_if(c, n_act0, [&](int i) { return v.is_activated[al[i]] != 0; }, act_slot, v.pool, 0);
    for (int i = c.tid; i < n_pool; i += c.nthr) v.mark[v.pool[i]] = stamp;
    __syncthreads();
    n_pool = block_append_if(c, n_lost0, [&](int i) { return v.mark[v.lost_list[i]] != stamp; },
                             [&](int i) { return v.lost_list[i]; }, v.pool, n_pool);

    tick();
    // ---- Kalman prediction of the pool, one wavefront per track (botsort_track.py:96-115) ----
#if !BM_OBB
    // (a wave's tracks are independent: the state of track k + nwaves is in flight while track k is computed -- the loop is a chain
    // of global round trips otherwise)
    {
        int k = c.wave;
        int slot = k < n_pool ? v.pool[k] : 0;
        double mj = 0.0, p = 0.0;
        int st = 0;
        if (k < n_pool) { const double* kf = v.kf + (long)slot * KF_STRIDE; mj = kf[c.lane & 7]; p = kf[KF_DIM + c.lane]; st = v.state[slot]; }
        while (k < n_pool) {
            const int kn = k + c.nwaves;
            const int slot_n = kn < n_pool ? v.pool[kn] : 0;
            double mj_n = 0.0, p_n = 0.0;
            int st_n = 0;
            if (kn < n_pool) { const double* kf = v.kf + (long)slot_n * KF_STRIDE; mj_n = kf[c.lane & 7]; p_n = kf[KF_DIM + c.lane]; st_n = v.state[slot_n]; }
            kf_predict_wave(v.kf + (long)slot * KF_STRIDE, mj, p, st != ST_TRACKED, c.lane, cfg.kind == 1);
            k = kn; slot = slot_n; mj = mj_n; p = p_n; st = st_n;
        }
    }
    __syncthreads();
#else
    for (int base = 0; base < n_pool; base += c.nwaves) {
        const int k = base + c.wave;
        if (k < n_pool) {
            const int slot = v.pool[k];
            kf_predict_obb_wave(v.kf + (long)slot * KF_STRIDE, v.state[slot] != ST_TRACKED, c.lane);
        }
    }
    __syncthreads();
#endif
    // ---- camera-motion warp of the pool and the unconfirmed tracks (botsort.py:134-145, :300-303) ----
    if (args.warp_flag && args.warp_flag[s]) {          // (oriented boxes: STrack.multi_gmc_obb, kf_warp_wave of the oriented layout)
        const double* W = args.warp + (long)s * 6;
        for (int base = 0; base < n_pool + n_unconf; base += c.nwaves) {
            const int k = base + c.wave;
            if (k < n_pool + n_unconf) {
                const int slot = k < n_pool ? v.pool[k] : v.unconf[k - n_pool];
                kf_warp_wave(v.kf + (long)slot * KF_STRIDE, W, c.lane);
            }
        }
        __syncthreads();
    }

    tick();
    // ---- first association (botsort.py:285-333) ----
    assoc_cost<NTHR>(c, v, v.pool, n_pool, v.first_idx, n_first, reid, 0.0, cfg.fuse_first_associate != 0, sA, sB, s_count, DBG ? dbg_stage(args, v, s, 0) : nullptr);
    if (DBG) dbg_copy_cost(c, args, v, s, 0, n_pool, n_first);
    tick();
    lap_solve(c, v, lap, n_pool, n_first, cfg.match_thresh);
    tick();
    int n_match = block_append_if(c, n_pool, [&](int r) { return v.lap_x[r] >= 0; }, [&](int r) { return r; }, v.match_slot, 0);
    for (int k = c.tid; k < n_match; k += c.nthr) {
        const int r = v.match_slot[k];
        const int slot = v.pool[r];
        v.match_det[k] = v.first_idx[v.lap_x[r]];
        v.match_slot[k] = slot;
        v.match_flag[k] = v.state[slot] == ST_TRACKED;
    }
    __syncthreads();
    int n_activated = block_append_if(c, n_match, [&](int k) { return v.match_flag[k] != 0; },
                                      [&](int k) { return v.match_slot[k]; }, v.activated, 0);
    int n_refound = block_append_if(c, n_match, [&](int k) { return v.match_flag[k] == 0; },
                                    [&](int k) { return v.match_slot[k]; }, v.refound, 0);
    // unmatched first-stage detections (ascending), kept for the unconfirmed stage
    const int n_left = block_append_if(c, n_first, [&](int k) { return v.lap_y[k] < 0; },
                                       [&](int k) { return v.first_idx[k]; }, v.left_idx, 0);
    // remaining Tracked pool tracks (botsort.py:350-354) -- state read BEFORE the updates
    // (matched tracks are excluded by lap_x anyway)
    const int n_remain = block_append_if(c, n_pool, [&](int r) { return v.lap_x[r] < 0 && v.state[v.pool[r]] == ST_TRACKED; },
                                         [&](int r) { return v.pool[r]; }, v.remain, 0);
    apply_matches(c, v, n_match, frame, reid);
    tick();

    // ---- second association: IoU only, low-confidence detections (botsort.py:335-378) ----
    iou_cost(c, v, v.remain, n_remain, v.second_idx, n_second);
    if (DBG) dbg_copy_cost(c, args, v, s, 1, n_remain, n_second, true);
    lap_solve(c, v, lap, n_remain, n_second, cfg.second_match_thresh);
    n_match = block_append_if(c, n_remain, [&](int r) { return v.lap_x[r] >= 0; }, [&](int r) { return r; }, v.match_slot, 0);
    for (int k = c.tid; k < n_match; k += c.nthr) {
        const int r = v.match_slot[k];
        v.match_det[k] = v.second_idx[v.lap_x[r]];
        v.match_slot[k] = v.remain[r];
        v.match_flag[k] = 1;
    }
    __syncthreads();
    n_activated = block_append_if(c, n_match, [&](int) { return true; }, [&](int k) { return v.match_slot[k]; }, v.activated, n_activated);
    const int n_newly_lost = block_append_if(c, n_remain, [&](int r) { return v.lap_x[r] < 0; },
                                             [&](int r) { return v.remain[r]; }, v.newly_lost, 0);
    apply_matches(c, v, n_match, frame, false);
    for (int k = c.tid; k < n_newly_lost; k += c.nthr) v.state[v.newly_lost[k]] = ST_LOST;   // mark_lost
    __syncthreads();

    tick();
    // ---- unconfirmed tracks vs the left-over high-confidence detections (botsort.py:380-431) ----
    assoc_cost<NTHR>(c, v, v.unconf, n_unconf, v.left_idx, n_left, reid, cfg.unconfirmed_emb_scale, true, sA, sB, s_count, DBG ? dbg_stage(args, v, s, 2) : nullptr);
    if (DBG) dbg_copy_cost(c, args, v, s, 2, n_unconf, n_left);
    lap_solve(c, v, lap, n_unconf, n_left, cfg.unconfirmed_match_thresh);
    n_match = block_append_if(c, n_unconf, [&](int r) { return v.lap_x[r] >= 0; }, [&](int r) { return r; }, v.match_slot, 0);
    for (int k = c.tid; k < n_match; k += c.nthr) {
        const int r = v.match_slot[k];
        v.match_det[k] = v.left_idx[v.lap_x[r]];
        v.match_slot[k] = v.unconf[r];
        v.match_flag[k] = 1;
    }
    __syncthreads();
    n_activated = block_append_if(c, n_match, [&](int) { return true; }, [&](int k) { return v.match_slot[k]; }, v.activated, n_activated);
    int n_newly_removed = block_append_if(c, n_unconf, [&](int r) { return v.lap_x[r] < 0; },
                                          [&](int r) { return v.unconf[r]; }, v.newly_removed, 0);
    apply_matches(c, v, n_match, frame, reid);
    for (int k = c.tid; k < n_newly_removed; k += c.nthr) v.state[v.newly_removed[k]] = ST_REMOVED;
    __syncthreads();

    tick();
    // ---- births (botsort.py:433-440, botsort_track.py:232-242): unmatched left-over
    //      detections with conf >= new_track_thresh, ids in ascending detection order ----
    // list_a <- detection indices to initialise; list_b <- free slots
    const int n_birth = block_append_if(c, n_left, [&](int k) {
        return v.lap_y[k] < 0 && !(v.dets[v.left_idx[k] * DET_COLS + CONF_COL] < cfg.new_track_thresh_f32); },
        [&](int k) { return v.left_idx[k]; }, v.list_a, 0);
    const int n_free = block_append_if(c, cap, [&](int sl) { return v.slot_used[sl] == 0; }, ident, v.list_b, 0);
    int n_born = n_birth;
    if (n_birth > n_free) {
        n_born = n_free;
        if (c.tid == 0) *v.status = STATUS_TRACK_CAPACITY;
    }
    const int id0 = *v.id_count;
    __syncthreads();
    for (int base = 0; base < n_born; base += c.nwaves) {
        const int k = base + c.wave;
        if (k < n_born) {
            const int d = v.list_a[k], slot = v.list_b[k];
            kf_initiate_wave(v.kf + (long)slot * KF_STRIDE, v.det_xywh + d * BOX_W, c.lane, cfg.kind == 1);
            if (reid) for (int q = c.lane; q < dim; q += WAVE) v.smooth[(long)slot * dim + q] = v.det_feat[(long)d * dim + q];
            if (c.lane == 0) {
                const float conf = v.dets[d * DET_COLS + CONF_COL], cls = v.dets[d * DET_COLS + CONF_COL + 1];
                v.slot_used[slot] = 1;
                v.id[slot] = id0 + k + 1;                    // BaseTrack.next_id, basetrack.py:71-80
                v.state[slot] = ST_TRACKED;
                v.is_activated[slot] = (frame == 1);         // botsort_track.py:239-240
                v.frame_id[slot] = frame;
                v.start_frame[slot] = frame;
                v.tracklet_len[slot] = 0;
                v.conf[slot] = conf; v.cls[slot] = cls; v.det_ind[slot] = (float)d;
                v.hist_n[slot] = cfg.kind == 1 ? 0 : 1;     // ByteTrack: "has been marked removed" flag, see below
                v.hist_cls[slot * KCLS] = cls;
                v.hist_w[slot * KCLS] = conf;
            }
        }
    }
    if (c.tid == 0) *v.id_count = id0 + n_born;
    __syncthreads();
    n_activated = block_append_if(c, n_born, [&](int) { return true; }, [&](int k) { return v.list_b[k]; }, v.activated, n_activated);

    tick();
    // ---- lost tracks past max_time_lost -> Removed (botsort.py:472-476) ----
    n_newly_removed = block_append_if(c, n_lost0, [&](int i) { return frame - v.frame_id[v.lost_list[i]] > cfg.max_time_lost; },
                                      [&](int i) { return v.lost_list[i]; }, v.newly_removed, n_newly_removed);
    for (int k = c.tid; k < n_newly_removed; k += c.nthr) v.state[v.newly_removed[k]] = ST_REMOVED;
    __syncthreads();

    // ---- list bookkeeping (botsort.py:478-492, botsort_utils.py:10-52) ----
    // A = [t in active if Tracked] + (activated \ A) + (refound \ A)
    const int mk_a = stamp + 1;
    int n_a = block_append_if(c, n_act0, [&](int i) { return v.state[al[i]] == ST_TRACKED; }, act_slot, v.list_a, 0);
    for (int i = c.tid; i < n_a; i += c.nthr) v.mark[v.list_a[i]] = mk_a;
    __syncthreads();
    const int n_a1 = block_append_if(c, n_activated, [&](int k) { return v.mark[v.activated[k]] != mk_a; },
                                     [&](int k) { return v.activated[k]; }, v.list_a, n_a);
    for (int i = n_a + c.tid; i < n_a1; i += c.nthr) v.mark[v.list_a[i]] = mk_a;
    __syncthreads();
    const int n_a2 = block_append_if(c, n_refound, [&](int k) { return v.mark[v.refound[k]] != mk_a; },
                                     [&](int k) { return v.refound[k]; }, v.list_a, n_a1);
    for (int i = n_a1 + c.tid; i < n_a2; i += c.nthr) v.mark[v.list_a[i]] = mk_a;
    __syncthreads();
    n_a = n_a2;
    // L = ((lost \ A) + newly_lost) \ removed_ids
    const int rm_size = *v.rm_size, rm_head = *v.rm_head, rm_cap = cfg.removed_cap;
    auto in_removed = [&](int slot) {
        // ByteTrack's removed list is unbounded (bytetrack.py:393): a track keeps its slot while it is in either list,
        // so "its id is in the list" is a per-slot flag, set below once this frame's subtraction is done
        if (cfg.kind == 1) return v.hist_n[slot] != 0;
        const int tid_ = v.id[slot];
        for (int q = 0; q < rm_size; ++q)
            if (v.removed_ring[(rm_head + q) % v.removed_alloc] == tid_) return true;
        return false;
    };
    int n_l = block_append_if(c, n_lost0, [&](int i) { const int sl = v.lost_list[i]; return v.mark[sl] != mk_a && !in_removed(sl); },
                              [&](int i) { return v.lost_list[i]; }, v.list_b, 0);
    n_l = block_append_if(c, n_newly_lost, [&](int k) { return !in_removed(v.newly_lost[k]); },
                          [&](int k) { return v.newly_lost[k]; }, v.list_b, n_l);
    // removed deque .extend (maxlen semantics), after the subtraction above
    if (c.tid == 0 && rm_cap > 0) {
        int head = rm_head, size = rm_size;
        for (int k = 0; k < n_newly_removed; ++k) {
            const int tid_ = v.id[v.newly_removed[k]];
            if (size == rm_cap) { head = (head + 1) % v.removed_alloc; --size; }
            v.removed_ring[(head + size) % v.removed_alloc] = tid_;
            ++size;
        }
        *v.rm_head = head;
        *v.rm_size = size;
    }
    __syncthreads();
    if (cfg.kind == 1) for (int k = c.tid; k < n_newly_removed; k += c.nthr) v.hist_n[v.newly_removed[k]] = 1;
    // remove_duplicate_stracks (botsort_utils.py:55-82)
    for (int i = c.tid; i < n_a; i += c.nthr) v.drop_a[i] = 0;
    for (int i = c.tid; i < n_l; i += c.nthr) v.drop_b[i] = 0;
    track_boxes(c, v, v.list_a, n_a, v.box_a);
    {
        double* box_b = v.cost;   // reuse: n_l * BOX_W doubles
        track_boxes(c, v, v.list_b, n_l, box_b);
        for (int o = c.tid; o < n_a * n_l; o += c.nthr) {
            const int p = o / n_l, q = o % n_l;
            const double pd = iou_dist_tt(v.box_a + p * BOX_W, box_b + q * BOX_W);
            if (pd < 0.15) {
                const int sa = v.list_a[p], sb = v.list_b[q];
                const int tp = v.frame_id[sa] - v.start_frame[sa];
                const int tq = v.frame_id[sb] - v.start_frame[sb];
                if (tp > tq) v.drop_b[q] = 1; else v.drop_a[p] = 1;
            }
        }
        __syncthreads();
    }
    const int mk_live = stamp + 2;
    const int n_act1 = block_append_if(c, n_a, [&](int i) { return v.drop_a[i] == 0; }, [&](int i) { return v.list_a[i]; }, v.pool, 0);
    const int n_lost1 = block_append_if(c, n_l, [&](int i) { return v.drop_b[i] == 0; }, [&](int i) { return v.list_b[i]; }, v.remain, 0);
    for (int i = c.tid; i < n_act1; i += c.nthr) v.mark[v.pool[i]] = mk_live;
    for (int i = c.tid; i < n_lost1; i += c.nthr) v.mark[v.remain[i]] = mk_live;
    __syncthreads();
    // release the slots of tracks that left both lists (old active, old lost, births)
    for (int i = c.tid; i < n_act0; i += c.nthr) { const int sl = al[i]; if (v.mark[sl] != mk_live) v.slot_used[sl] = 0; }
    for (int i = c.tid; i < n_lost0; i += c.nthr) { const int sl = v.lost_list[i]; if (v.mark[sl] != mk_live) v.slot_used[sl] = 0; }
    for (int i = c.tid; i < n_activated; i += c.nthr) { const int sl = v.activated[i]; if (v.mark[sl] != mk_live) v.slot_used[sl] = 0; }
    __syncthreads();
    for (int i = c.tid; i < n_act1; i += c.nthr) v.active_list[i] = v.pool[i];
    for (int i = c.tid; i < n_lost1; i += c.nthr) v.lost_list[i] = v.remain[i];
    if (c.tid == 0) { *v.n_active = n_act1; *v.n_lost = n_lost1; }
    __syncthreads();

    tick();
    // ---- output rows (botsort.py:494-500): activated tracks in active-list order ----
    const int n_out = block_append_if(c, n_act1, [&](int i) { return v.is_activated[v.active_list[i]] != 0; },
                                      [&](int i) { return v.active_list[i]; }, v.list_a, 0);
    for (int k = c.tid; k < n_out; k += c.nthr) {
        const int sl = v.list_a[k];
        const double* m = v.kf + (long)sl * KF_STRIDE;
        float* o = v.out + k * OUT_COLS;
#if BM_OBB
        for (int q = 0; q < 5; ++q) o[q] = (float)m[q];         // t.xywha (botsort.py:495, bytetrack.py:397)
        o[5] = (float)v.id[sl]; o[6] = v.conf[sl]; o[7] = v.cls[sl]; o[8] = v.det_ind[sl];
#else
        const double hw = (cfg.kind == 1 ? m[2] * m[3] : m[2]) / 2, hh = m[3] / 2;
        o[0] = (float)(m[0] - hw); o[1] = (float)(m[1] - hh);
        o[2] = (float)(m[0] + hw); o[3] = (float)(m[1] + hh);
        o[4] = (float)v.id[sl]; o[5] = v.conf[sl]; o[6] = v.cls[sl]; o[7] = v.det_ind[sl];
#endif
    }
    if (c.tid == 0) *v.out_n = n_out;
    __syncthreads();
    tick();
}

