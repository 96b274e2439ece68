// Body of the DeepOCSORT / OC-SORT frame step: included twice by deepocsort_step.hpp -- in namespace bm with the axis-aligned layout
// (BM_OBB 0) and in namespace bm::obb with the oriented one (BM_OBB 1; DOCS_KF_STRIDE / DOCS_BOX / DOCS_OBS / DOCS_DET_COLS /
// DOCS_OUT_COLS are the enclosing namespace's).  No include guard on purpose.

#if !BM_OBB
// ---------------------------------------------------------------------------
// 7-state filter in the registers of one wavefront: lane l holds p = P[l>>3][l&7] and xv = x[l&7].
// ---------------------------------------------------------------------------
struct Kf7 { double p, xv; };

__device__ inline Kf7 kf7_load(const double* kf, int lane) { return Kf7{kf[KF_DIM + lane], kf[lane & 7]}; }
__device__ inline void kf7_store(double* kf, const Kf7& k, int lane) {
    kf[KF_DIM + lane] = k.p;
    if (lane < 8) kf[lane] = k.xv;
}
__device__ inline double kf7_x(const Kf7& k, int a) { return __shfl(k.xv, a, WAVE); }
__device__ inline double kf7_p(const Kf7& k, int i, int j) { return __shfl(k.p, i * 8 + j, WAVE); }

// _enforce_state_constraints, xysr.py:154-161
__device__ inline void kf7_constrain(Kf7& k, int lane) {
    const int i = lane >> 3, j = lane & 7;
    if (j == 2 || j == 3) k.xv = k.xv > 1e-6 ? k.xv : 1e-6;     // np.maximum(x, 1e-6)
    const double pt = __shfl(k.p, j * 8 + i, WAVE);
    k.p = 0.5 * (k.p + pt);
}

// KalmanFilterXYSR.predict, xysr.py:368-377 over base.py:366-391: x <- F x, P <- F P F^T + Q (F = I + shift by 4 on
// x, y, s), then the constraints.  Same roundings as np.dot twice (the products with the 0/1 entries of F are exact).
__device__ inline void kf7_predict(Kf7& k, double q_xy, double q_s, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const double xhi = __shfl(k.xv, (j + 4) & 7, WAVE);
    if (j < 3) k.xv = k.xv + xhi;
    const double p_dn = __shfl(k.p, (lane + 32) & 63, WAVE);          // P[i+4][j]
    const double fp = (i < 3) ? (k.p + p_dn) : k.p;
    const double fp_rt = __shfl(fp, (lane & ~7) | ((j + 4) & 7), WAVE);   // (FP)[i][j+4]
    double c = (j < 3) ? (fp + fp_rt) : fp;
    double q = 0.0;
    if (i == j && i < 7) q = (i < 4) ? 1.0 : (i < 6 ? q_xy : q_s);    // Q = diag(1,1,1,1,Q_xy,Q_xy,Q_s), deepocsort.py:108-109
    c = 1.0 * c + q;
    k.p = (i < 7 && j < 7) ? c : 0.0;
    kf7_constrain(k, lane);
}

// KalmanFilterXYSR.update with a measurement (xysr.py:456-476 -> base.py:414-459): S = 0.5 (S + S^T) with
// S = H P H^T + R, Cholesky (with the jitter ladder of _safe_cho_factor), K = (S^-1 (P H^T)^T)^T, x += K y,
// P = (I-KH) (P (I-KH)^T) + K (R K^T), symmetrised, then the constraints.  Returns false when no factorisation exists.
__device__ inline bool kf7_update(Kf7& k, const double* m, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const double Rd[4] = {1.0, 1.0, 10.0, 10.0};                       // deepocsort.py:103
    double S[4][4];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) S[a][b] = kf7_p(k, a, b) + (a == b ? Rd[a] : 0.0);
    for (int a = 0; a < 4; ++a)
        for (int b = a + 1; b < 4; ++b) { const double sy = 0.5 * (S[a][b] + S[b][a]); S[a][b] = sy; S[b][a] = sy; }
    for (int a = 0; a < 4; ++a) S[a][a] = 0.5 * (S[a][a] + S[a][a]);
    double L[4][4];
    double scale = 0.0;
    for (int a = 0; a < 4; ++a) { const double d = S[a][a] < 0 ? -S[a][a] : S[a][a]; scale = d > scale ? d : scale; }
    if (!(scale > 0.0) || !(scale < 1.7e308)) scale = 1.0;
    bool ok = false;
    for (int attempt = -13; attempt < 4 && !ok; ++attempt) {           // -13: no jitter; -12..3: scale * 10^e
        const double P10[16] = {1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3};
        const double jit = attempt >= -12 ? scale * P10[attempt + 12] : 0.0;
        ok = true;
        for (int c = 0; c < 4 && ok; ++c) {
            double d = S[c][c] + jit;
            for (int q = 0; q < c; ++q) d -= L[c][q] * L[c][q];
            if (!(d > 0.0)) { ok = false; break; }
            d = sqrt(d);
            L[c][c] = d;
            for (int r = c + 1; r < 4; ++r) {
                double t = S[r][c];
                for (int q = 0; q < c; ++q) t -= L[r][q] * L[c][q];
                L[r][c] = t / d;
            }
        }
    }
    if (!ok) return false;
    // gain rows i and j
    double Ki[4], Kj[4];
    for (int which = 0; which < 2; ++which) {
        const int r = which ? j : i;
        double y[4];
        for (int q = 0; q < 4; ++q) {
            double t = kf7_p(k, r & 7, q);
            for (int w = 0; w < q; ++w) t -= L[q][w] * y[w];
            y[q] = t / L[q][q];
        }
        double* K = which ? Kj : Ki;
        for (int q = 3; q >= 0; --q) {
            double t = y[q];
            for (int w = q + 1; w < 4; ++w) t -= L[w][q] * K[w];
            K[q] = t / L[q][q];
        }
    }
    // x += K y   (lane uses row j of K for its x[j])
    double yv[4];
    for (int a = 0; a < 4; ++a) yv[a] = m[a] - kf7_x(k, a);
    double acc = Kj[0] * yv[0];
    for (int a = 1; a < 4; ++a) acc = fma(Kj[a], yv[a], acc);
    const double xnew = k.xv + acc;
    // T = P (I-KH)^T : T[i][j] = sum_q P[i][q] * IKH[j][q],  IKH[j][q] = delta_jq - (q < 4 ? K[j][q] : 0)
    double t_ij = 0.0;
    for (int q = 0; q < 7; ++q) {
        const double piq = kf7_p(k, i & 7, q);
        const double ikh = (q == j ? 1.0 : 0.0) - (q < 4 ? Kj[q] : 0.0);
        t_ij = q == 0 ? piq * ikh : fma(piq, ikh, t_ij);
    }
    // U = (I-KH) T : U[i][j] = sum_q IKH[i][q] T[q][j]
    double u_ij = 0.0;
    for (int q = 0; q < 7; ++q) {
        const double tqj = __shfl(t_ij, q * 8 + j, WAVE);
        const double ikh = (q == i ? 1.0 : 0.0) - (q < 4 ? Ki[q] : 0.0);
        u_ij = q == 0 ? ikh * tqj : fma(ikh, tqj, u_ij);
    }
    // V = K (R K^T) : V[i][j] = sum_a K[i][a] * (R_aa * K[j][a])
    double v_ij = 0.0;
    for (int a = 0; a < 4; ++a) {
        const double rk = Rd[a] * Kj[a];
        v_ij = a == 0 ? Ki[a] * rk : fma(Ki[a], rk, v_ij);
    }
    double pn = u_ij + v_ij;
    pn = (i < 7 && j < 7) ? pn : 0.0;
    const double pt = __shfl(pn, j * 8 + i, WAVE);
    k.p = 0.5 * (pn + pt);
    k.xv = (j < 7) ? xnew : 0.0;
    kf7_constrain(k, lane);
    return true;
}

// KalmanFilterXYSR.apply_affine_correction, axis-aligned branch (xysr.py:311-366): x[:2] <- m x[:2] + t, x[4:6] <- m x[4:6],
// P[:2,:2] <- m P[:2,:2] m^T, P[4:6,4:6] <- m P[4:6,4:6] m^T (diagonal blocks only), then the constraints.
// W = [m00 m01 tx; m10 m11 ty].
__device__ inline void kf7_affine(Kf7& k, const double* W, int lane) {
    const int i = lane >> 3, j = lane & 7;
    // state: lanes j in {0,1,4,5}
    const int jb = j & ~1;                                   // 0 or 4 for the affected pairs
    const double xa = __shfl(k.xv, (lane & ~7) | jb, WAVE), xb = __shfl(k.xv, (lane & ~7) | (jb + 1), WAVE);
    if (j < 2 || (j >= 4 && j < 6)) {
        double nv = W[(j & 1) * 3 + 0] * xa + W[(j & 1) * 3 + 1] * xb;
        if (j < 2) nv = nv + W[(j & 1) * 3 + 2];
        k.xv = nv;
    }
    // covariance blocks (rows/cols {0,1} and {4,5})
    const int r0 = i & ~1;
    const bool in_block = (r0 == 0 || r0 == 4) && (j & ~1) == r0;
    const double p0 = __shfl(k.p, r0 * 8 + j, WAVE), p1 = __shfl(k.p, (r0 + 1) * 8 + j, WAVE);
    const double a_ij = fma(W[(i & 1) * 3 + 1], p1, W[(i & 1) * 3 + 0] * p0);          // (m P)[i][j]
    const double a0 = __shfl(a_ij, i * 8 + (j & ~1), WAVE), a1 = __shfl(a_ij, i * 8 + (j & ~1) + 1, WAVE);
    const double c_ij = fma(a1, W[(j & 1) * 3 + 1], a0 * W[(j & 1) * 3 + 0]);           // ((m P) m^T)[i][j]
    if (in_block) k.p = c_ij;
    kf7_constrain(k, lane);
}

// KalmanBoxTracker.apply_affine_correction (deepocsort.py:190-209) for one track: the observation boxes, the filter and --
// for an unobserved track -- its frozen copy.  `last_observation` and the newest entry of `observations` are ONE array in
// the reference (deepocsort.py:168-170), so that box is transformed twice when it is recent enough; reproduced.
__device__ inline void docs_affine_wave(double* kf, double* kf_saved, bool fix_saved, double* last_obs, double* obs_box, const int* obs_age,
                                        int n_obs, int age, int delta_t, const double* W, int lane) {
    if (lane == 0) {
        auto T = [&](double* b) {
            const double x1 = fma(W[1], b[1], W[0] * b[0]) + W[2], y1 = fma(W[4], b[1], W[3] * b[0]) + W[5];
            const double x2 = fma(W[1], b[3], W[0] * b[2]) + W[2], y2 = fma(W[4], b[3], W[3] * b[2]) + W[5];
            b[0] = x1; b[1] = y1; b[2] = x2; b[3] = y2;
        };
        if (last_obs[0] + last_obs[1] + last_obs[2] + last_obs[3] + last_obs[4] > 0) T(last_obs);
        const int n_have = n_obs < 3 ? n_obs : 3;
        for (int q = 3 - n_have; q < 3; ++q) {
            const int a = obs_age[q];
            if (a < age - delta_t || a > age) continue;
            if (q == 2) T(last_obs);                       // shared storage with last_observation
            else T(obs_box + q * 5);
        }
        if (n_have > 0) for (int e = 0; e < 4; ++e) obs_box[2 * 5 + e] = last_obs[e];
    }
    Kf7 k = kf7_load(kf, lane);
    kf7_affine(k, W, lane);
    kf7_store(kf, k, lane);
    if (fix_saved) {
        Kf7 ks = kf7_load(kf_saved, lane);
        kf7_affine(ks, W, lane);
        kf7_store(kf_saved, ks, lane);
    }
}

// xyxy2xysr + _prepare_measurement (geometry.py:103-124, xysr.py:139-152)
__device__ inline void box_to_z(const double* b, double* z) {
    const double w = b[2] - b[0], h = b[3] - b[1];
    z[0] = b[0] + w / 2.0;
    z[1] = b[1] + h / 2.0;
    z[2] = w * h;
    z[3] = w / (h + 1e-6);
    z[2] = z[2] > 1e-6 ? z[2] : 1e-6;
    z[3] = z[3] > 1e-6 ? z[3] : 1e-6;
}

// convert_x_to_bbox, deepocsort.py:29-40
__device__ inline void x_to_box(double x0, double x1, double x2, double x3, double* b) {
    const double w = sqrt(x2 * x3);
    const double h = x2 / w;
    b[0] = x0 - w / 2.0; b[1] = x1 - h / 2.0; b[2] = x0 + w / 2.0; b[3] = x1 + h / 2.0;
}

#else
// ---------------------------------------------------------------------------
// 9-state filter of an ORIENTED track: KalmanFilterXYSR(dim_x=9, dim_z=5) as KalmanBoxTracker configures it (ocsort.py:121-154),
// state (x, y, s, r, theta, vx, vy, vs, vtheta).  One THREAD per track: the state lives in the thread's private memory for the whole
// predict / update / observation-centric re-update chain (the 81-element covariance does not map onto 64 lanes, and this path is not a
// benchmark line); fp64, contraction off, the operation order of the NumPy calls.  The three large functions are calls (one copy each,
// the state addressed in scratch) instead of being unrolled into the frame step's register allocation.
// ---------------------------------------------------------------------------
struct Kf9 { double x[9]; double P[81]; };

__device__ inline void kf9_load(Kf9& k, const double* kf) {
    for (int e = 0; e < 9; ++e) k.x[e] = kf[e];
    for (int e = 0; e < 81; ++e) k.P[e] = kf[9 + e];
}
__device__ inline void kf9_store(double* kf, const Kf9& k) {
    for (int e = 0; e < 9; ++e) kf[e] = k.x[e];
    for (int e = 0; e < 81; ++e) kf[9 + e] = k.P[e];
}
// the state a velocity feeds (xysr.py:54-66: x += vx, y += vy, s += vs, theta += vtheta; r static), -1: none
__device__ inline int kf9_vel_of(int i) { return i == 0 ? 5 : (i == 1 ? 6 : (i == 2 ? 7 : (i == 4 ? 8 : -1))); }

// _enforce_state_constraints, xysr.py:154-161 over base.py:160-180
__device__ inline void kf9_constrain(Kf9& k) {
    k.x[2] = k.x[2] > 1e-6 ? k.x[2] : 1e-6;
    k.x[3] = k.x[3] > 1e-6 ? k.x[3] : 1e-6;
    k.x[4] = obb_wrap_angle(k.x[4]);
    for (int i = 0; i < 9; ++i)
        for (int j = i; j < 9; ++j) {
            const double sy = 0.5 * (k.P[i * 9 + j] + k.P[j * 9 + i]);
            k.P[i * 9 + j] = sy; k.P[j * 9 + i] = sy;
        }
}

// KalmanFilterXYSR.predict, xysr.py:368-377 over base.py:366-391: x <- F x, P <- 1.0 * (F P) F^T + Q, then the constraints
__device__ __noinline__ inline void kf9_predict(Kf9& k, double q_xy, double q_s, double q_a) {
    for (int i = 0; i < 5; ++i) { const int s = kf9_vel_of(i); if (s >= 0) k.x[i] = k.x[i] + k.x[s]; }
    double fp[81];
    for (int i = 0; i < 9; ++i) {
        const int s = kf9_vel_of(i);
        for (int j = 0; j < 9; ++j) fp[i * 9 + j] = s >= 0 ? (k.P[i * 9 + j] + k.P[s * 9 + j]) : k.P[i * 9 + j];
    }
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            const int s = kf9_vel_of(j);
            double c = s >= 0 ? (fp[i * 9 + j] + fp[i * 9 + s]) : fp[i * 9 + j];
            double q = 0.0;
            if (i == j) q = i < 5 ? 1.0 : (i < 7 ? q_xy : (i == 7 ? q_s : q_a));      // ocsort.py:151-153
            k.P[i * 9 + j] = 1.0 * c + q;
        }
    kf9_constrain(k);
}

// KalmanFilterXYSR._prepare_measurement for an oriented box (xysr.py:96-152 over base.py:122-157): floors on s and r, the angle
// wrapped, then of (r, t), (r, t + pi), (1 / r, t + pi / 2), (1 / r, t - pi / 2) the form closest to the state `ref`
__device__ __noinline__ inline void kf9_prepare(double* m, const double* ref) {
    m[2] = m[2] > 1e-6 ? m[2] : 1e-6;
    m[3] = m[3] > 1e-6 ? m[3] : 1e-6;
    m[4] = obb_wrap_angle(m[4]);
    const double ref_r = ref[3] > 1e-6 ? ref[3] : 1e-6, ref_t = ref[4];
    const double s = m[2] > 1e-6 ? m[2] : 1e-6, r = m[3] > 1e-6 ? m[3] : 1e-6, t = m[4];
    double best_cost = 1.0 / 0.0, br = r, bt = t;
    for (int c = 0; c < 4; ++c) {
        double s1 = c < 2 ? r : 1.0 / r;
        const double th = c == 0 ? t : (c == 1 ? t + OBB_PI : (c == 2 ? t + (OBB_PI / 2.0) : t - (OBB_PI / 2.0)));
        s1 = s1 > 1e-6 ? s1 : 1e-6;
        const double ta = ref_t + obb_wrap_angle(th - ref_t);
        const double cost = fabs(ta - ref_t) + (0.05 * (fabs(log(1.0 / 1.0)) + fabs(log(s1 / ref_r))));
        if (cost < best_cost) { best_cost = cost; br = s1; bt = ta; }
    }
    m[2] = s > 1e-6 ? s : 1e-6;
    m[3] = br > 1e-6 ? br : 1e-6;
    m[4] = bt;
}

// KalmanFilterXYSR.update with a prepared measurement (xysr.py:466-473 -> base.py:414-459): S = 0.5 (S + S^T) with S = H P H^T + R,
// Cholesky (with the jitter ladder of _safe_cho_factor), K = (S^-1 (P H^T)^T)^T, x += K y, P = (I-KH) (P (I-KH)^T) + K (R K^T),
// symmetrised; the theta velocity damped by 0.8 (base.py:222-232), then the constraints.  False when no factorisation exists.
__device__ __noinline__ inline bool kf9_update(Kf9& k, const double* m) {
    const double Rd[5] = {1.0, 1.0, 10.0, 10.0, 10.0};                 // ocsort.py:145
    double S[5][5];
    for (int a = 0; a < 5; ++a)
        for (int c = 0; c < 5; ++c) S[a][c] = k.P[a * 9 + c] + (a == c ? Rd[a] : 0.0);
    for (int a = 0; a < 5; ++a)
        for (int c = a; c < 5; ++c) { const double sy = 0.5 * (S[a][c] + S[c][a]); S[a][c] = sy; S[c][a] = sy; }
    double L[5][5];
    double scale = 0.0;
    for (int a = 0; a < 5; ++a) { const double d = S[a][a] < 0 ? -S[a][a] : S[a][a]; scale = d > scale ? d : scale; }
    if (!(scale > 0.0) || !(scale < 1.7e308)) scale = 1.0;
    bool ok = false;
    for (int attempt = -13; attempt < 4 && !ok; ++attempt) {           // -13: no jitter; -12..3: scale * 10^e
        const double P10[16] = {1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3};
        const double jit = attempt >= -12 ? scale * P10[attempt + 12] : 0.0;
        ok = true;
        for (int c = 0; c < 5 && ok; ++c) {
            double d = S[c][c] + jit;
            for (int q = 0; q < c; ++q) d -= L[c][q] * L[c][q];
            if (!(d > 0.0)) { ok = false; break; }
            d = sqrt(d);
            L[c][c] = d;
            for (int r = c + 1; r < 5; ++r) {
                double t = S[r][c];
                for (int q = 0; q < c; ++q) t -= L[r][q] * L[c][q];
                L[r][c] = t / d;
            }
        }
    }
    if (!ok) return false;
    double K[9][5];
    for (int r = 0; r < 9; ++r) {
        double y[5];
        for (int q = 0; q < 5; ++q) {
            double t = k.P[r * 9 + q];
            for (int w = 0; w < q; ++w) t -= L[q][w] * y[w];
            y[q] = t / L[q][q];
        }
        for (int q = 4; q >= 0; --q) {
            double t = y[q];
            for (int w = q + 1; w < 5; ++w) t -= L[w][q] * K[r][w];
            K[r][q] = t / L[q][q];
        }
    }
    double yv[5];
    for (int a = 0; a < 5; ++a) yv[a] = m[a] - k.x[a];
    for (int r = 0; r < 9; ++r) {
        double acc = K[r][0] * yv[0];
        for (int a = 1; a < 5; ++a) acc = fma(K[r][a], yv[a], acc);
        k.x[r] = k.x[r] + acc;
    }
    // T = P (I-KH)^T, U = (I-KH) T, V = K (R K^T)
    double T[81];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            double t = 0.0;
            for (int q = 0; q < 9; ++q) {
                const double ikh = (q == j ? 1.0 : 0.0) - (q < 5 ? K[j][q] : 0.0);
                t = q == 0 ? k.P[i * 9 + q] * ikh : fma(k.P[i * 9 + q], ikh, t);
            }
            T[i * 9 + j] = t;
        }
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) {
            double u = 0.0;
            for (int q = 0; q < 9; ++q) {
                const double ikh = (q == i ? 1.0 : 0.0) - (q < 5 ? K[i][q] : 0.0);
                u = q == 0 ? ikh * T[q * 9 + j] : fma(ikh, T[q * 9 + j], u);
            }
            double vv = 0.0;
            for (int a = 0; a < 5; ++a) {
                const double rk = Rd[a] * K[j][a];
                vv = a == 0 ? K[i][a] * rk : fma(K[i][a], rk, vv);
            }
            k.P[i * 9 + j] = u + vv;
        }
    for (int i = 0; i < 9; ++i)
        for (int j = i; j < 9; ++j) {
            const double sy = 0.5 * (k.P[i * 9 + j] + k.P[j * 9 + i]);
            k.P[i * 9 + j] = sy; k.P[j * 9 + i] = sy;
        }
    k.x[8] = k.x[8] * 0.8;
    kf9_constrain(k);
    return true;
}

// convert_obb_to_z, ocsort.py:49-59
__device__ inline void obb_to_z(const double* b, double* z) {
    const double w = b[2] > 1e-6 ? b[2] : 1e-6, h = b[3] > 1e-6 ? b[3] : 1e-6;
    z[0] = b[0]; z[1] = b[1]; z[2] = w * h; z[3] = w / h; z[4] = b[4];
}
// convert_x_to_obb, ocsort.py:62-72
__device__ inline void x_to_obb(const double* x, double* b) {
    const double sr = x[2] * x[3];
    const double w = sqrt(sr > 1e-12 ? sr : 1e-12);
    const double h = x[2] / (w > 1e-6 ? w : 1e-6);
    b[0] = x[0]; b[1] = x[1]; b[2] = w; b[3] = h; b[4] = x[4];
}

#endif

// Per-stream view
struct DV {
    DocsConfigDev cfg;
    int cap, dim, nd;
    int* frame_count; int* id_count; int* n_tracks; int* status;
    int* list; int* slot_used;
    double* kf; double* kf_saved; double* last_obs; double* last_z; double* obs_box; int* obs_age; int* n_obs;
    double* velocity; int* has_vel; double* emb;
    int* id; int* age; int* tsu; int* hits; int* hit_streak; int* observed; int* has_saved; int* n_miss;
    float* conf; float* cls; float* det_ind;
    int* keep; double* alpha; double* trk_box; double* kobs; double* iou; double* cost; double* embc;
    double* row_w; double* col_w; int* row_cnt; int* col_cnt; int* m_det; int* m_trk; int* un_d; int* un_t;
    int* tmp_a; int* tmp_b; int* lap_x; int* lap_y; int* flag_t; int* flag_d;
    const float* dets; int n_dets; const float* embs; float* out; int* out_n;
};

__device__ inline DV docs_view(const DocsStepArgs& a, int s) {
    DV v;
    const DocsState& st = a.st; const DocsScratch& sc = a.sc;
    const long cap = st.cap, dim = st.dim, nd = sc.max_dets, big = cap > nd ? cap : nd;
    v.cfg = a.cfg; v.cap = st.cap; v.dim = st.dim; v.nd = sc.max_dets;
    v.frame_count = st.frame_count + s; v.id_count = st.id_count + s; v.n_tracks = st.n_tracks + s; v.status = st.status + s;
    v.list = st.list + s * cap; v.slot_used = st.slot_used + s * cap;
    v.kf = st.kf + s * cap * DOCS_KF_STRIDE; v.kf_saved = st.kf_saved + s * cap * DOCS_KF_STRIDE;
    v.last_obs = st.last_obs + s * cap * DOCS_OBS; v.last_z = st.last_z + s * cap * DOCS_BOX; v.obs_box = st.obs_box + s * cap * (3 * DOCS_OBS); v.obs_age = st.obs_age + s * cap * 3;
    v.n_obs = st.n_obs + s * cap; v.velocity = st.velocity + s * cap * 2; v.has_vel = st.has_vel + s * cap;
    v.emb = st.emb + s * cap * dim;
    v.id = st.id + s * cap; v.age = st.age + s * cap; v.tsu = st.tsu + s * cap; v.hits = st.hits + s * cap;
    v.hit_streak = st.hit_streak + s * cap; v.observed = st.observed + s * cap; v.has_saved = st.has_saved + s * cap;
    v.n_miss = st.n_miss + s * cap;
    v.conf = st.conf + s * cap; v.cls = st.cls + s * cap; v.det_ind = st.det_ind + s * cap;
    v.keep = sc.keep + s * nd; v.alpha = sc.alpha + s * nd; v.trk_box = sc.trk_box + s * cap * DOCS_BOX; v.kobs = sc.kobs + s * cap * DOCS_OBS;
    v.iou = sc.iou + s * nd * cap; v.cost = sc.cost + s * nd * cap; v.embc = sc.embc + s * nd * cap;
    v.row_w = sc.row_w + s * nd; v.col_w = sc.col_w + s * cap; v.row_cnt = sc.row_cnt + s * nd; v.col_cnt = sc.col_cnt + s * cap;
    v.m_det = sc.m_det + s * nd; v.m_trk = sc.m_trk + s * nd; v.un_d = sc.un_d + s * nd; v.un_t = sc.un_t + s * cap;
    v.tmp_a = sc.tmp_a + s * big; v.tmp_b = sc.tmp_b + s * big;
    v.lap_x = sc.lap_x + s * (cap + nd); v.lap_y = sc.lap_y + s * nd;
    v.flag_t = sc.flag_t + s * cap; v.flag_d = sc.flag_d + s * nd;
    v.dets = a.dets + s * nd * DOCS_DET_COLS; v.n_dets = a.n_dets[s];
    v.embs = a.embs ? a.embs + s * nd * dim : nullptr;
    v.out = a.out + s * cap * DOCS_OUT_COLS; v.out_n = a.out_n + s;
    return v;
}

// ---------------------------------------------------------------------------
// linear_assignment(cost) of association.py:20-24 -- lap.lapjv(cost, extend_cost=True) -- on a det-major matrix
// cost[d * ld + t]: R columns (tracks), C rows (detections).  The solver is the Jonker-Volgenant code of lap_jv.hpp, which
// returns, tie for tie, the assignment the sequential algorithm returns; its state lives in dynamic LDS.
// out_x[t] = row of column t, out_y[d] = column of row d, -1 = unassigned.
// ---------------------------------------------------------------------------
// One out-of-line copy for the three call sites of a frame step (first association, BYTE, recovery round): the solver gets its
// own register allocation instead of sharing the frame step's.
__device__ __noinline__ inline bool docs_assign(const Ctx& c, unsigned char* dyn_lds, int R, int C, const double* cost, long ld, int* out_x, int* out_y) {
    return lap_jv_extended(c, jv_carve(dyn_lds, R + C), C, R, [=](int d, int t) { return cost[d * ld + t]; }, false, 0.0, out_y, out_x);
}

#if !BM_OBB
__device__ inline double iou_pair(const double* a, const double* b) {      // iou.py:134-150
    const double xx1 = a[0] > b[0] ? a[0] : b[0], yy1 = a[1] > b[1] ? a[1] : b[1];
    const double xx2 = a[2] < b[2] ? a[2] : b[2], yy2 = a[3] < b[3] ? a[3] : b[3];
    double w = xx2 - xx1, h = yy2 - yy1;
    w = w > 0.0 ? w : 0.0; h = h > 0.0 ? h : 0.0;
    const double wh = w * h;
    return wh / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - wh);
}

// The association function the constructor named (AssociationFunction._get_asso_func, iou.py:408-423), a = detection, b = track
// box: the elementwise expressions of the batch functions in their order of evaluation (fp64; +, -, *, /, sqrt round as NumPy's
// do -- `ciou` alone calls a libm function, arctan).  np.maximum / np.minimum propagate a NaN operand; the comparisons below
// return the second operand then, which differs only for boxes that are NaN already (no detection or live track is).
enum { ASSO_IOU = 0, ASSO_GIOU = 1, ASSO_DIOU = 2, ASSO_CIOU = 3, ASSO_HMIOU = 4, ASSO_CENTROID = 5 };
__device__ inline double asso_pair(int mode, double diag, const double* a, const double* b) {
    if (mode == ASSO_IOU) return iou_pair(a, b);
    if (mode == ASSO_CENTROID) {                                    // iou.py:253-268
        const double dx = (a[0] + a[2]) / 2 - (b[0] + b[2]) / 2, dy = (a[1] + a[3]) / 2 - (b[1] + b[3]) / 2;
        return 1 - sqrt(dx * dx + dy * dy) / diag;
    }
    const double xx1 = a[0] > b[0] ? a[0] : b[0], yy1 = a[1] > b[1] ? a[1] : b[1];
    const double xx2 = a[2] < b[2] ? a[2] : b[2], yy2 = a[3] < b[3] ? a[3] : b[3];
    double w = xx2 - xx1, h = yy2 - yy1;
    w = w > 0.0 ? w : 0.0; h = h > 0.0 ? h : 0.0;
    const double wh = w * h;
    const double area1 = (a[2] - a[0]) * (a[3] - a[1]), area2 = (b[2] - b[0]) * (b[3] - b[1]);
    const double xxc1 = a[0] < b[0] ? a[0] : b[0], yyc1 = a[1] < b[1] ? a[1] : b[1];
    const double xxc2 = a[2] > b[2] ? a[2] : b[2], yyc2 = a[3] > b[3] ? a[3] : b[3];
    if (mode == ASSO_HMIOU) {                                       // iou.py:153-203
        double uh = yyc2 - yyc1;
        uh = uh > 1e-10 ? uh : 1e-10;
        const double o = h / uh;
        return wh / (area1 + area2 - wh + 1e-10) * o;
    }
    if (mode == ASSO_GIOU) {                                        // iou.py:205-244 (its assert on the enclosing box is not restated)
        const double uni = area1 + area2 - wh, iou = wh / uni;
        const double enc = (xxc2 - xxc1) * (yyc2 - yyc1);
        return (iou - (enc - uni) / enc + 1.0) / 2.0;
    }
    const double cdx = (a[0] + a[2]) / 2.0 - (b[0] + b[2]) / 2.0, cdy = (a[1] + a[3]) / 2.0 - (b[1] + b[3]) / 2.0;
    const double inner = cdx * cdx + cdy * cdy;
    const double ox = xxc2 - xxc1, oy = yyc2 - yyc1;
    if (mode == ASSO_DIOU) {                                        // iou.py:346-391
        const double iou = wh / (area1 + area2 - wh);
        return (iou - inner / (ox * ox + oy * oy) + 1) / 2.0;
    }
    // ciou, iou.py:283-344
    const double eps = 1e-7;
    const double iou = wh / (area1 + area2 - wh + eps);
    const double outer = ox * ox + oy * oy + eps;
    const double w1 = a[2] - a[0], h1 = a[3] - a[1] + eps, w2 = b[2] - b[0], h2 = b[3] - b[1] + eps;
    const double ad = atan(w2 / h2) - atan(w1 / h1);
    const double vv = (4 / (3.141592653589793 * 3.141592653589793)) * (ad * ad);
    const double alpha = vv / ((1 - iou) + vv + eps);
    return (iou - inner / outer + alpha * vv + 1) / 2.0;
}
#else
// the association function of an oriented tracker (AssociationFunction._get_asso_func, iou.py:408-417: "iou_obb" and "centroid_obb" are
// the oriented names): iou_batch_obb (iou.py:38-115, :152-154) or centroid_batch_obb (iou.py:263-274); a = detection, b = track
__device__ inline double asso_pair(int mode, double diag, const double* a, const double* b) {
    if (mode == bm::ASSO_CENTROID) {
        const double dx = a[0] - b[0], dy = a[1] - b[1];
        return 1 - sqrt(dx * dx + dy * dy) / diag;
    }
    return obb_iou(a, b);
}
#endif

#if !BM_OBB
// KalmanBoxTracker.update(det) + update_emb for one matched (track slot, kept detection) pair: one wavefront.
// deepocsort.py:143-185, xysr.py:383-476.
__device__ inline void docs_apply_match(DV& v, int slot, int kd, int lane) {
    const int j = v.keep[kd];
    const float* d = v.dets + j * DET_COLS;
    double box[5];
    for (int q = 0; q < 5; ++q) box[q] = (double)d[q];
    double* lo = v.last_obs + slot * 5;
    const int age = v.age[slot];
    // velocity from the observation delta_t steps back (or the last one)
    if (lo[0] + lo[1] + lo[2] + lo[3] + lo[4] >= 0) {
        const double* prev = nullptr;
        const int n_have = v.n_obs[slot] < 3 ? v.n_obs[slot] : 3;
        for (int dt = v.cfg.delta_t; dt > 0 && !prev; --dt)
            for (int q = 3 - n_have; q < 3; ++q)
                if (v.obs_age[slot * 3 + q] == age - dt) { prev = v.obs_box + (slot * 3 + q) * 5; break; }
        if (!prev) prev = lo;
        const double cx1 = (prev[0] + prev[2]) / 2.0, cy1 = (prev[1] + prev[3]) / 2.0;
        const double cx2 = (box[0] + box[2]) / 2.0, cy2 = (box[1] + box[3]) / 2.0;
        const double sy = cy2 - cy1, sx = cx2 - cx1;
        const double norm = sqrt(sy * sy + sx * sx) + 1e-6;
        if (lane == 0) { v.velocity[slot * 2] = sy / norm; v.velocity[slot * 2 + 1] = sx / norm; v.has_vel[slot] = 1; }
    }
    // the filter: observation-centric re-update first when the track was unobserved (unfreeze), then the update
    double z[4];
    box_to_z(box, z);
    Kf7 k = kf7_load(v.kf + (long)slot * KF_STRIDE, lane);
    bool ok = true;
    if (!v.observed[slot] && v.has_saved[slot]) {
        k = kf7_load(v.kf_saved + (long)slot * KF_STRIDE, lane);
        double zl[4];
        for (int q = 0; q < 4; ++q) zl[q] = v.last_z[slot * 4 + q];      // last observed measurement (history_obs[index1])
        const int gap = v.n_miss[slot] + 1;
        const double w1 = sqrt(zl[2] * zl[3]), h1 = sqrt(zl[2] / zl[3]);
        const double w2 = sqrt(z[2] * z[3]), h2 = sqrt(z[2] / z[3]);
        const double dx = (z[0] - zl[0]) / gap, dy = (z[1] - zl[1]) / gap, dw = (w2 - w1) / gap, dh = (h2 - h1) / gap;
        for (int i = 0; i < gap; ++i) {
            const double xx = zl[0] + (i + 1) * dx, yy = zl[1] + (i + 1) * dy;
            const double ww = w1 + (i + 1) * dw, hh = h1 + (i + 1) * dh;
            double zv[4] = {xx, yy, ww * hh, ww / hh};
            zv[2] = zv[2] > 1e-6 ? zv[2] : 1e-6; zv[3] = zv[3] > 1e-6 ? zv[3] : 1e-6;
            ok = kf7_update(k, zv, lane) && ok;
            if (i != gap - 1) kf7_predict(k, v.cfg.q_xy, v.cfg.q_s, lane);
        }
    }
    ok = kf7_update(k, z, lane) && ok;
    kf7_store(v.kf + (long)slot * KF_STRIDE, k, lane);
    if (lane == 0) {
        if (!ok) *v.status = STATUS_LAP_STALL + 1;
        for (int q = 0; q < 5; ++q) lo[q] = box[q];
        for (int q = 0; q < 4; ++q) v.last_z[slot * 4 + q] = z[q];
        // observations[age] = box (keep the three most recent)
        for (int q = 0; q < 2; ++q) {
            v.obs_age[slot * 3 + q] = v.obs_age[slot * 3 + q + 1];
            for (int e = 0; e < 5; ++e) v.obs_box[(slot * 3 + q) * 5 + e] = v.obs_box[(slot * 3 + q + 1) * 5 + e];
        }
        v.obs_age[slot * 3 + 2] = age;
        for (int e = 0; e < 5; ++e) v.obs_box[(slot * 3 + 2) * 5 + e] = box[e];
        v.n_obs[slot] += 1;
        v.tsu[slot] = 0; v.hits[slot] += 1; v.hit_streak[slot] += 1;
        v.observed[slot] = 1; v.n_miss[slot] = 0;
        v.conf[slot] = d[4]; v.cls[slot] = d[5]; v.det_ind[slot] = (float)j;
    }
    // update_emb: emb = alpha * emb + (1 - alpha) * det_emb; emb /= |emb|
    if (!v.cfg.embedding_off && v.embs) {
        const double al = v.alpha[kd];
        const float* de = v.embs + (long)j * v.dim;
        double* te = v.emb + (long)slot * v.dim;
        double ss = 0.0;
        for (int e = lane; e < v.dim; e += WAVE) {
            const double nv = al * te[e] + (1 - al) * (double)de[e];
            te[e] = nv;
            ss += nv * nv;
        }
        ss = wave_sum(ss);
        const double nrm = sqrt(ss);
        for (int e = lane; e < v.dim; e += WAVE) te[e] = te[e] / nrm;
    }
}
// the matches of a round, one wavefront each
__device__ inline void docs_apply_matches(const Ctx& c, DV& v, int n) {
    for (int base = 0; base < n; base += c.nwaves) {
        const int q = base + c.wave;
        if (q < n) docs_apply_match(v, v.list[v.m_trk[q]], v.m_det[q], c.lane);
    }
}

#else
// KalmanBoxTracker.update(det) of an oriented track for one matched (track slot, kept detection) pair: one THREAD.
// ocsort.py:241-278, xysr.py:383-476.
__device__ inline void docs_apply_match(DV& v, int slot, int kd) {
    const int j = v.keep[kd];
    const float* d = v.dets + j * DOCS_DET_COLS;
    double box[6];
    for (int q = 0; q < 6; ++q) box[q] = (double)d[q];                  // (cx, cy, w, h, theta, conf)
    double* lo = v.last_obs + slot * 6;
    const int age = v.age[slot];
    // velocity from the observation delta_t steps back (or the last one): speed_direction_obb, ocsort.py:82-87
    if (lo[0] + lo[1] + lo[2] + lo[3] + lo[4] + lo[5] >= 0) {
        const double* prev = nullptr;
        const int n_have = v.n_obs[slot] < 3 ? v.n_obs[slot] : 3;
        for (int dt = v.cfg.delta_t; dt > 0 && !prev; --dt)
            for (int q = 3 - n_have; q < 3; ++q)
                if (v.obs_age[slot * 3 + q] == age - dt) { prev = v.obs_box + (slot * 3 + q) * 6; break; }
        if (!prev) prev = lo;
        const double sy = box[1] - prev[1], sx = box[0] - prev[0];
        const double norm = sqrt(sy * sy + sx * sx) + 1e-6;
        v.velocity[slot * 2] = sy / norm; v.velocity[slot * 2 + 1] = sx / norm; v.has_vel[slot] = 1;
    }
    // the filter.  The measurement is prepared (aligned) against the state as it is NOW; an unobserved track first replays
    // interpolated boxes from its frozen state (unfreeze), each prepared against the replayed state; then the update itself
    Kf9 k;
    kf9_load(k, v.kf + (long)slot * DOCS_KF_STRIDE);
    double z[5];
    obb_to_z(box, z);
    kf9_prepare(z, k.x);
    bool ok = true;
    if (!v.observed[slot] && v.has_saved[slot]) {
        kf9_load(k, v.kf_saved + (long)slot * DOCS_KF_STRIDE);
        double zl[5];
        for (int q = 0; q < 5; ++q) zl[q] = v.last_z[slot * 5 + q];      // last observed measurement (history_obs[index1])
        const int gap = v.n_miss[slot] + 1;
        const double w1 = sqrt(zl[2] * zl[3]), h1 = sqrt(zl[2] / zl[3]);
        const double w2 = sqrt(z[2] * z[3]), h2 = sqrt(z[2] / z[3]);
        const double dx = (z[0] - zl[0]) / gap, dy = (z[1] - zl[1]) / gap, dw = (w2 - w1) / gap, dh = (h2 - h1) / gap;
        const double dth = obb_wrap_angle(z[4] - zl[4]) / gap;
        for (int i = 0; i < gap; ++i) {
            const double ww = w1 + (i + 1) * dw, hh = h1 + (i + 1) * dh;
            double zv[5] = {zl[0] + (i + 1) * dx, zl[1] + (i + 1) * dy, ww * hh, ww / hh, obb_wrap_angle(zl[4] + (i + 1) * dth)};
            kf9_prepare(zv, k.x);
            ok = kf9_update(k, zv) && ok;
            if (i != gap - 1) kf9_predict(k, v.cfg.q_xy, v.cfg.q_s, v.cfg.q_s);
        }
    }
    ok = kf9_update(k, z) && ok;
    kf9_store(v.kf + (long)slot * DOCS_KF_STRIDE, k);
    if (!ok) *v.status = STATUS_LAP_STALL + 1;
    for (int q = 0; q < 6; ++q) lo[q] = box[q];
    for (int q = 0; q < 5; ++q) v.last_z[slot * 5 + q] = z[q];
    // observations[age] = box (keep the three most recent)
    for (int q = 0; q < 2; ++q) {
        v.obs_age[slot * 3 + q] = v.obs_age[slot * 3 + q + 1];
        for (int e = 0; e < 6; ++e) v.obs_box[(slot * 3 + q) * 6 + e] = v.obs_box[(slot * 3 + q + 1) * 6 + e];
    }
    v.obs_age[slot * 3 + 2] = age;
    for (int e = 0; e < 6; ++e) v.obs_box[(slot * 3 + 2) * 6 + e] = box[e];
    v.n_obs[slot] += 1;
    v.tsu[slot] = 0; v.hits[slot] += 1; v.hit_streak[slot] += 1;
    v.observed[slot] = 1; v.n_miss[slot] = 0;
    v.conf[slot] = d[5]; v.cls[slot] = d[6]; v.det_ind[slot] = (float)j;
}
// the matches of a round, one thread each
__device__ inline void docs_apply_matches(const Ctx& c, DV& v, int n) {
    for (int q = c.tid; q < n; q += c.nthr) docs_apply_match(v, v.list[v.m_trk[q]], v.m_det[q]);
}

#endif

// ---------------------------------------------------------------------------
// The frame step
// ---------------------------------------------------------------------------
template <int NTHR>
__device__ inline void docs_step_stream(const DocsStepArgs& args, int s, int* s_int, double* s_dbl, unsigned char* dyn_lds) {
    if (args.n_dets[s] < 0) {                 // stream not stepped in this call
        if (threadIdx.x == 0) args.out_n[s] = 0;
        return;
    }
    const Ctx c = make_ctx(s_int, s_dbl);
    DV v = (docs_view)(args, s);        // (parenthesised: no argument-dependent lookup into the enclosing namespace's copy)
    const DocsConfigDev& cfg = v.cfg;
    const long ld = v.cap;
    const int dim = v.dim;
    const bool use_emb = !cfg.embedding_off && v.embs != nullptr;
    auto ident = [](int i) { return i; };

    if (c.tid == 0) *v.frame_count += 1;
    __syncthreads();
    const int frame = *v.frame_count;

    // ---- detections above det_thresh, in order (deepocsort.py:331-335); trust -> dets_alpha (:353-356) ----
    const int nk = block_append_if(c, v.n_dets, [&](int j) { return v.dets[j * DOCS_DET_COLS + DOCS_BOX] > cfg.det_thresh_f32; }, ident, v.keep, 0);
    // BYTE candidates are appended behind the kept detections: keep[nk .. nk + n_byte)
    int n_byte = 0;
    if (cfg.use_byte)
        n_byte = block_append_if(c, v.n_dets, [&](int j) {
            const float s = v.dets[j * DOCS_DET_COLS + DOCS_BOX];
            return s > cfg.min_conf_f32 && s < cfg.det_thresh_f32; }, ident, v.keep, nk) - nk;
    for (int k = c.tid; k < nk; k += c.nthr) {
        const double conf = (double)v.dets[v.keep[k] * DOCS_DET_COLS + DOCS_BOX];
        const double trust = (conf - cfg.det_thresh) / (1 - cfg.det_thresh);
        v.alpha[k] = cfg.alpha_fixed + (1 - cfg.alpha_fixed) * (1 - trust);
    }

    int nt = *v.n_tracks;
#if !BM_OBB
    // ---- camera-motion correction of every track (deepocsort.py:347-351), wave per track ----
    if (args.warp_flag && args.warp_flag[s]) {
        const double* W = args.warp + (long)s * 6;
        for (int base = 0; base < nt; base += c.nwaves) {
            const int t = base + c.wave;
            if (t < nt) {
                const int slot = v.list[t];
                docs_affine_wave(v.kf + (long)slot * KF_STRIDE, v.kf_saved + (long)slot * KF_STRIDE, !v.observed[slot] && v.has_saved[slot],
                                 v.last_obs + slot * 5, v.obs_box + slot * 15, v.obs_age + slot * 3, v.n_obs[slot], v.age[slot],
                                 cfg.delta_t, W, c.lane);
            }
        }
        __syncthreads();
    }
    // ---- predict every track (deepocsort.py:358-368, :211-225), wave per track ----
    for (int base = 0; base < nt; base += c.nwaves) {
        const int t = base + c.wave;
        if (t < nt) {
            const int slot = v.list[t];
            Kf7 k = kf7_load(v.kf + (long)slot * KF_STRIDE, c.lane);
            const double x6 = kf7_x(k, 6), x2 = kf7_x(k, 2);
            if ((x6 + x2) <= 0 && (c.lane & 7) == 6) k.xv = k.xv * 0.0;
            kf7_predict(k, cfg.q_xy, cfg.q_s, c.lane);
            kf7_store(v.kf + (long)slot * KF_STRIDE, k, c.lane);
            double b[4];
            x_to_box(kf7_x(k, 0), kf7_x(k, 1), kf7_x(k, 2), kf7_x(k, 3), b);
            if (c.lane == 0) {
                v.age[slot] += 1;
                if (v.tsu[slot] > 0) v.hit_streak[slot] = 0;
                v.tsu[slot] += 1;
                bool bad = false;
                for (int q = 0; q < 4; ++q) { v.trk_box[t * 4 + q] = b[q]; bad = bad || (b[q] != b[q]); }
                v.flag_t[t] = bad ? 1 : 0;
            }
        }
    }
#else       // (camera-motion compensation of oriented tracks is not built: the host refuses a warp for such a handle)
    // ---- predict every track (ocsort.py:408-415, :280-299), a thread per track ----
    for (int t = c.tid; t < nt; t += c.nthr) {
        const int slot = v.list[t];
        Kf9 k;
        kf9_load(k, v.kf + (long)slot * DOCS_KF_STRIDE);
        if ((k.x[7] + k.x[2]) <= 0) k.x[7] = k.x[7] * 0.0;
        kf9_predict(k, cfg.q_xy, cfg.q_s, cfg.q_s);                      // Q_a_scaling = Q_s_scaling (ocsort.py:530)
        kf9_store(v.kf + (long)slot * DOCS_KF_STRIDE, k);
        double b[5];
        x_to_obb(k.x, b);
        v.age[slot] += 1;
        if (v.tsu[slot] > 0) v.hit_streak[slot] = 0;
        v.tsu[slot] += 1;
        bool bad = false;
        for (int q = 0; q < 5; ++q) { v.trk_box[t * 5 + q] = b[q]; bad = bad || (b[q] != b[q]); }
        v.flag_t[t] = bad ? 1 : 0;
    }
#endif
    __syncthreads();
    // tracks whose prediction is NaN are dropped (deepocsort.py:364-375)
    {
        const int n_ok = block_append_if(c, nt, [&](int t) { return v.flag_t[t] == 0; }, ident, v.tmp_a, 0);
        if (n_ok != nt) {
            for (int t = c.tid; t < nt; t += c.nthr) if (v.flag_t[t]) v.slot_used[v.list[t]] = 0;
            __syncthreads();
            for (int q = c.tid; q < n_ok; q += c.nthr) { v.tmp_b[q] = v.list[v.tmp_a[q]]; }
            __syncthreads();
            for (int q = c.tid; q < n_ok; q += c.nthr) {
                const int src = v.tmp_a[q];
                v.list[q] = v.tmp_b[q];
                if (src != q) for (int e = 0; e < DOCS_BOX; ++e) v.cost[q * DOCS_BOX + e] = v.trk_box[src * DOCS_BOX + e];   // staged below
            }
            __syncthreads();
            for (int q = c.tid; q < n_ok; q += c.nthr) if (v.tmp_a[q] != q) for (int e = 0; e < DOCS_BOX; ++e) v.trk_box[q * DOCS_BOX + e] = v.cost[q * DOCS_BOX + e];
            nt = n_ok;
            if (c.tid == 0) *v.n_tracks = nt;
            __syncthreads();
        }
    }

    // ---- k_previous_obs per track (deepocsort.py:17-26, :381-382) ----
    for (int t = c.tid; t < nt; t += c.nthr) {
        const int slot = v.list[t];
        const int n_have = v.n_obs[slot] < 3 ? v.n_obs[slot] : 3;
        const double* src = nullptr;
        if (n_have > 0) {
            const int age = v.age[slot];
            for (int dt = cfg.delta_t; dt > 0 && !src; --dt)
                for (int q = 3 - n_have; q < 3; ++q)
                    if (v.obs_age[slot * 3 + q] == age - dt) { src = v.obs_box + (slot * 3 + q) * DOCS_OBS; break; }
            if (!src) src = v.obs_box + (slot * 3 + 2) * DOCS_OBS;           // the observation with the largest age
        }
        for (int e = 0; e < DOCS_OBS; ++e) v.kobs[t * DOCS_OBS + e] = src ? src[e] : -1.0;
    }
    __syncthreads();

    // ---- first association (association.py:61-152) ----
    int n_match = 0, n_ud = 0, n_ut = 0;
    double* dbg = args.dbg_cost ? args.dbg_cost + (long)s * DOCS_DBG_PLANES * v.nd * ld : nullptr;     // parity debugging only
    if (dbg && c.tid == 0) { args.dbg_shape[s * 4] = nk; args.dbg_shape[s * 4 + 1] = nt; args.dbg_shape[s * 4 + 2] = 0; args.dbg_shape[s * 4 + 3] = 0; }
    if (nt == 0) {
        for (int k = c.tid; k < nk; k += c.nthr) v.un_d[k] = k;
        n_ud = nk;
        __syncthreads();
    } else {
        // IoU, velocity-direction consistency, optional appearance similarity; det-major matrices
        for (int k = c.tid; k < nk; k += c.nthr) v.row_cnt[k] = 0;
        for (int t = c.tid; t < nt; t += c.nthr) v.col_cnt[t] = 0;
        __syncthreads();
        const long total = (long)nk * nt;
        for (int k = c.wave; k < nk; k += c.nwaves)            // a wavefront per detection row, lanes over tracks (no per-element division)
        for (int t = c.lane; t < nt; t += WAVE) {
            const float* df = v.dets + v.keep[k] * DOCS_DET_COLS;
            double db[DOCS_BOX];
            for (int e = 0; e < DOCS_BOX; ++e) db[e] = (double)df[e];
            const double score = (double)df[DOCS_BOX];
            const double io = asso_pair(cfg.asso_mode, cfg.asso_diag, db, v.trk_box + t * DOCS_BOX);
            v.iou[k * ld + t] = io;
            const double* ko = v.kobs + t * DOCS_OBS;      // (oriented rows go through the SAME column arithmetic in the reference, association.py:8-17, :97-98: reproduced)
            const double cx1 = (db[0] + db[2]) / 2.0, cy1 = (db[1] + db[3]) / 2.0;
            const double cx2 = (ko[0] + ko[2]) / 2.0, cy2 = (ko[1] + ko[3]) / 2.0;
            const double dx = cx1 - cx2, dy = cy1 - cy2;
            const double norm = sqrt(dx * dx + dy * dy) + 1e-6;
            const double X = dx / norm, Y = dy / norm;
            const int slot = v.list[t];
            const double iy = v.has_vel[slot] ? v.velocity[slot * 2] : 0.0, ix = v.has_vel[slot] ? v.velocity[slot * 2 + 1] : 0.0;
            double cs = ix * X + iy * Y;
            cs = cs < -1.0 ? -1.0 : (cs > 1.0 ? 1.0 : cs);
            const double ang = acos(cs);
            const double diff = (3.141592653589793 / 2.0 - (ang < 0 ? -ang : ang)) / 3.141592653589793;
            const double valid = ko[4] < 0 ? 0.0 : 1.0;
            v.cost[k * ld + t] = ((valid * diff) * cfg.inertia) * score;      // angle_diff_cost for now
            if (io > cfg.iou_threshold) { atomicAdd(&v.row_cnt[k], 1); atomicAdd(&v.col_cnt[t], 1); }
        }
        __syncthreads();
        // "already a permutation" early-out (association.py:104-108)
        int mx = 0;
        for (int k = c.tid; k < nk; k += c.nthr) mx = v.row_cnt[k] > mx ? v.row_cnt[k] : mx;
        for (int t = c.tid; t < nt; t += c.nthr) mx = v.col_cnt[t] > mx ? v.col_cnt[t] : mx;
        int mn_needed = 0;                 // max over rows must be exactly 1 (and over columns)
        for (int k = c.tid; k < nk; k += c.nthr) mn_needed |= v.row_cnt[k] == 1;
        {
            // reduce (max count, any count == 1) over the workgroup
            for (int off = WAVE / 2; off > 0; off >>= 1) {
                const int o = __shfl_xor(mx, off, WAVE); mx = o > mx ? o : mx;
                mn_needed |= __shfl_xor(mn_needed, off, WAVE);
            }
            if (c.lane == 0) { c.s_int[c.wave] = mx; c.s_dbl[c.wave] = (double)mn_needed; }
            __syncthreads();
            int gm = 0, ga = 0;
            for (int w = 0; w < c.nwaves; ++w) { gm = c.s_int[w] > gm ? c.s_int[w] : gm; ga |= (int)c.s_dbl[w]; }
            __syncthreads();
            mx = gm; mn_needed = ga;
        }
        const bool have_matrix = nk > 0;   // min(iou.shape) != 0 (nt > 0 here)
        const bool permutation = have_matrix && mx == 1 && mn_needed;
        if (dbg) {
            for (int k = c.wave; k < nk; k += c.nwaves)
                for (int t = c.lane; t < nt; t += WAVE) dbg[(long)v.nd * ld + k * ld + t] = v.iou[k * ld + t];
            if (c.tid == 0) args.dbg_shape[s * 4 + 2] = !have_matrix ? 0 : (permutation ? 1 : 2);
        }
        if (!have_matrix) {
            for (int t = c.tid; t < nt; t += c.nthr) v.lap_x[t] = -1;
        } else if (permutation) {
            for (int t = c.tid; t < nt; t += c.nthr) v.lap_x[t] = -1;
            for (int k = c.tid; k < nk; k += c.nthr) v.lap_y[k] = -1;
            __syncthreads();
            for (int k = c.wave; k < nk; k += c.nwaves)
            for (int t = c.lane; t < nt; t += WAVE) {
                if (v.iou[k * ld + t] > cfg.iou_threshold) { v.lap_y[k] = t; v.lap_x[t] = k; }
            }
        } else {
            if (use_emb) {
                // emb_cost = dets_embs @ trk_embs.T, zeroed where IoU <= 0 (deepocsort.py:387-390, association.py:113)
                for (long e = c.wave; e < total; e += c.nwaves) {
                    const int k = (int)(e / nt), t = (int)(e % nt);
                    double acc = 0.0;
                    if (v.iou[k * ld + t] > 0) {
                        const float* de = v.embs + (long)v.keep[k] * dim;
                        const double* te = v.emb + (long)v.list[t] * dim;
                        for (int q = c.lane; q < dim; q += WAVE) acc += (double)de[q] * te[q];
                        acc = wave_sum(acc);
                    }
                    if (c.lane == 0) v.embc[k * ld + t] = (v.iou[k * ld + t] > 0) ? acc : 0.0;
                }
                __syncthreads();
                if (!cfg.aw_off) {
                    // compute_aw_max_metric (association.py:29-58): weights from the two largest entries of each row / column
                    for (int k = c.tid; k < nk; k += c.nthr) {
                        double w = 1.0;
                        if (nt >= 2) {
                            double m1 = -DOCS_INF, m2 = -DOCS_INF;
                            for (int t = 0; t < nt; ++t) { const double x = v.embc[k * ld + t]; if (x > m1) { m2 = m1; m1 = x; } else if (x > m2) m2 = x; }
                            if (m1 == 0) w = 0.0;
                            else { double r = m2 / m1 - cfg.aw_param; r = r > 0 ? r : 0; w = 1 - r / (1 - cfg.aw_param); }
                        }
                        v.row_w[k] = w;
                    }
                    for (int t = c.tid; t < nt; t += c.nthr) {
                        double w = 1.0;
                        if (nk >= 2) {
                            double m1 = -DOCS_INF, m2 = -DOCS_INF;
                            for (int k = 0; k < nk; ++k) { const double x = v.embc[k * ld + t]; if (x > m1) { m2 = m1; m1 = x; } else if (x > m2) m2 = x; }
                            if (m1 == 0) w = 0.0;
                            else { double r = m2 / m1 - cfg.aw_param; r = r > 0 ? r : 0; w = 1 - r / (1 - cfg.aw_param); }
                        }
                        v.col_w[t] = w;
                    }
                    __syncthreads();
                }
            }
            // final_cost = -(iou + angle_diff_cost + emb_cost)
            for (int k = c.wave; k < nk; k += c.nwaves)
            for (int t = c.lane; t < nt; t += WAVE) {
                double em = 0.0;
                if (use_emb) {
                    const double raw = v.embc[k * ld + t];
                    if (cfg.aw_off) em = raw * cfg.w_emb;
                    else {
                        double w = cfg.w_emb;
                        if (nt >= 2) w = w * v.row_w[k];
                        if (nk >= 2) w = w * v.col_w[t];
                        em = w * raw;
                    }
                }
                v.cost[k * ld + t] = -((v.iou[k * ld + t] + v.cost[k * ld + t]) + em);
                if (dbg) { dbg[k * ld + t] = v.cost[k * ld + t]; dbg[2 * (long)v.nd * ld + k * ld + t] = em; }
            }
            __syncthreads();
            const double* cm = v.cost;
            if (!(docs_assign)(c, dyn_lds, nt, nk, cm, ld, v.lap_x, v.lap_y) && c.tid == 0)
                *v.status = STATUS_LAP_STALL;
        }
        __syncthreads();
        // unmatched detections: never assigned (ascending), then the assigned-but-low-IoU ones in detection order;
        // same for the tracks, low-IoU ones in the order of their detections (association.py:128-150)
        auto good = [&](int k) { return v.lap_y[k] >= 0 && !(v.iou[k * ld + v.lap_y[k]] < cfg.iou_threshold); };
        n_match = block_append_if(c, nk, good, ident, v.m_det, 0);
        for (int q = c.tid; q < n_match; q += c.nthr) v.m_trk[q] = v.lap_y[v.m_det[q]];
        n_ud = block_append_if(c, nk, [&](int k) { return v.lap_y[k] < 0; }, ident, v.un_d, 0);
        n_ud = block_append_if(c, nk, [&](int k) { return v.lap_y[k] >= 0 && !good(k); }, ident, v.un_d, n_ud);
        n_ut = block_append_if(c, nt, [&](int t) { return v.lap_x[t] < 0; }, ident, v.un_t, 0);
        n_ut = block_append_if(c, nk, [&](int k) { return v.lap_y[k] >= 0 && !good(k); }, [&](int k) { return v.lap_y[k]; }, v.un_t, n_ut);
        // matched tracks take their detections (deepocsort.py:407-409)
        docs_apply_matches(c, v, n_match);
        __syncthreads();
    }

    // ---- OC-SORT only: BYTE association of the low-score detections with the predicted boxes of the unmatched tracks
    //      (ocsort.py:456-485); matched tracks take the detection, the rest stay unmatched (np.setdiff1d: ascending) ----
    if (cfg.use_byte && n_byte > 0 && n_ut > 0) {
        double mxi = -DOCS_INF;
        for (int a = c.wave; a < n_byte; a += c.nwaves)
        for (int b = c.lane; b < n_ut; b += WAVE) {
            const float* df = v.dets + v.keep[nk + a] * DOCS_DET_COLS;
            double db[DOCS_BOX];
            for (int e = 0; e < DOCS_BOX; ++e) db[e] = (double)df[e];
            const double io = asso_pair(cfg.asso_mode, cfg.asso_diag, db, v.trk_box + v.un_t[b] * DOCS_BOX);
            v.iou[a * ld + b] = io;
            v.cost[a * ld + b] = -io;
            mxi = io > mxi ? io : mxi;
        }
        {
            for (int off = WAVE / 2; off > 0; off >>= 1) { const double o = __shfl_xor(mxi, off, WAVE); mxi = o > mxi ? o : mxi; }
            if (c.lane == 0) c.s_dbl[c.wave] = mxi;
            __syncthreads();
            double g = -DOCS_INF;
            for (int w = 0; w < c.nwaves; ++w) g = c.s_dbl[w] > g ? c.s_dbl[w] : g;
            __syncthreads();
            mxi = g;
        }
        if (mxi > cfg.iou_threshold) {
            const double* cm = v.cost;
            if (!(docs_assign)(c, dyn_lds, n_ut, n_byte, cm, ld, v.lap_x, v.lap_y) && c.tid == 0)
                *v.status = STATUS_LAP_STALL;
            auto good_b = [&](int a) { return v.lap_y[a] >= 0 && !(v.iou[a * ld + v.lap_y[a]] < cfg.iou_threshold); };
            const int nb = block_append_if(c, n_byte, good_b, ident, v.tmp_a, 0);
            for (int q = c.tid; q < nb; q += c.nthr) { const int a = v.tmp_a[q]; v.m_det[q] = nk + a; v.m_trk[q] = v.un_t[v.lap_y[a]]; }
            __syncthreads();
            docs_apply_matches(c, v, nb);
            for (int t = c.tid; t < nt; t += c.nthr) v.flag_t[t] = 0;
            __syncthreads();
            for (int b = c.tid; b < n_ut; b += c.nthr) v.flag_t[v.un_t[b]] = 1;
            __syncthreads();
            for (int q = c.tid; q < nb; q += c.nthr) v.flag_t[v.m_trk[q]] = 0;
            __syncthreads();
            n_ut = block_append_if(c, nt, [&](int t) { return v.flag_t[t] != 0; }, ident, v.un_t, 0);
        }
    }

    // ---- second round: observation-centric recovery against the last observations (deepocsort.py:411-450) ----
    if (n_ud > 0 && n_ut > 0) {
        double mxi = -DOCS_INF;
        for (int a = c.wave; a < n_ud; a += c.nwaves)
        for (int b = c.lane; b < n_ut; b += WAVE) {
            const float* df = v.dets + v.keep[v.un_d[a]] * DOCS_DET_COLS;
            double db[DOCS_BOX];
            for (int e = 0; e < DOCS_BOX; ++e) db[e] = (double)df[e];
            const double io = asso_pair(cfg.asso_mode, cfg.asso_diag, db, v.last_obs + v.list[v.un_t[b]] * DOCS_OBS);
            v.iou[a * ld + b] = io;
            v.cost[a * ld + b] = -io;
            mxi = io > mxi ? io : mxi;              // NaN (degenerate placeholder boxes) never wins: comparisons are false
        }
        {
            for (int off = WAVE / 2; off > 0; off >>= 1) { const double o = __shfl_xor(mxi, off, WAVE); mxi = o > mxi ? o : mxi; }
            if (c.lane == 0) c.s_dbl[c.wave] = mxi;
            __syncthreads();
            double g = -DOCS_INF;
            for (int w = 0; w < c.nwaves; ++w) g = c.s_dbl[w] > g ? c.s_dbl[w] : g;
            __syncthreads();
            mxi = g;
        }
        if (mxi > cfg.iou_threshold) {
            const double* cm = v.cost;
            if (!(docs_assign)(c, dyn_lds, n_ut, n_ud, cm, ld, v.lap_x, v.lap_y) && c.tid == 0)
                *v.status = STATUS_LAP_STALL;
            auto good2 = [&](int a) { return v.lap_y[a] >= 0 && !(v.iou[a * ld + v.lap_y[a]] < cfg.iou_threshold); };
            const int n2 = block_append_if(c, n_ud, good2, ident, v.tmp_a, 0);
            for (int q = c.tid; q < n2; q += c.nthr) { const int a = v.tmp_a[q]; v.m_det[q] = v.un_d[a]; v.m_trk[q] = v.un_t[v.lap_y[a]]; }
            __syncthreads();
            docs_apply_matches(c, v, n2);
            // np.setdiff1d: what is left, sorted ascending and unique
            for (int k = c.tid; k < nk; k += c.nthr) v.flag_d[k] = 0;
            for (int t = c.tid; t < nt; t += c.nthr) v.flag_t[t] = 0;
            __syncthreads();
            for (int a = c.tid; a < n_ud; a += c.nthr) v.flag_d[v.un_d[a]] = 1;
            for (int b = c.tid; b < n_ut; b += c.nthr) v.flag_t[v.un_t[b]] = 1;
            __syncthreads();
            for (int q = c.tid; q < n2; q += c.nthr) { v.flag_d[v.m_det[q]] = 0; v.flag_t[v.m_trk[q]] = 0; }
            __syncthreads();
            n_ud = block_append_if(c, nk, [&](int k) { return v.flag_d[k] != 0; }, ident, v.un_d, 0);
            n_ut = block_append_if(c, nt, [&](int t) { return v.flag_t[t] != 0; }, ident, v.un_t, 0);
        }
    }

    // ---- unmatched tracks: update(None) (deepocsort.py:452-453, xysr.py:456-468) ----
    for (int b = c.tid; b < n_ut; b += c.nthr) {
        const int slot = v.list[v.un_t[b]];
        if (v.observed[slot]) {                         // freeze(): remember the filter before the blind stretch
            for (int e = 0; e < DOCS_KF_STRIDE; ++e) v.kf_saved[(long)slot * DOCS_KF_STRIDE + e] = v.kf[(long)slot * DOCS_KF_STRIDE + e];
            v.has_saved[slot] = 1;
            v.n_miss[slot] = 0;
        }
        v.observed[slot] = 0;
        v.n_miss[slot] += 1;
    }
    __syncthreads();

    // ---- births (deepocsort.py:455-466): slots in ascending order, ids in the order of the unmatched list ----
    if (n_ud > 0) {
        const int n_free = block_append_if(c, v.cap, [&](int sl) { return v.slot_used[sl] == 0; }, ident, v.tmp_a, 0);
        if (n_free < n_ud) { if (c.tid == 0) *v.status = STATUS_TRACK_CAPACITY; n_ud = n_free; }
        const int id0 = *v.id_count;
        __syncthreads();
#if !BM_OBB
        for (int base = 0; base < n_ud; base += c.nwaves) {
            const int q = base + c.wave;
            if (q < n_ud) {
                const int slot = v.tmp_a[q], kd = v.un_d[q], j = v.keep[kd];
                const float* d = v.dets + j * DET_COLS;
                const double box[4] = {(double)d[0], (double)d[1], (double)d[2], (double)d[3]};
                double z[4];
                const double w = box[2] - box[0], h = box[3] - box[1];
                z[0] = box[0] + w / 2.0; z[1] = box[1] + h / 2.0; z[2] = w * h; z[3] = w / (h + 1e-6);   // no clamp at birth (deepocsort.py:114)
                const int i = c.lane >> 3, jj = c.lane & 7;
                double p = 0.0;
                if (i == jj && i < 7) p = (i < 4) ? 10.0 : 10000.0;    // P = 10 * diag(1,1,1,1,1000,1000,1000)
                v.kf[(long)slot * KF_STRIDE + KF_DIM + c.lane] = p;
                if (c.lane < 8) v.kf[(long)slot * KF_STRIDE + c.lane] = c.lane < 4 ? z[c.lane] : 0.0;
                if (use_emb) for (int e = c.lane; e < dim; e += WAVE) v.emb[(long)slot * dim + e] = (double)v.embs[(long)j * dim + e];
                if (c.lane == 0) {
                    v.slot_used[slot] = 1;
                    v.id[slot] = id0 + 1 + q;                         // ids start at 1 (KalmanBoxTracker.count = 1, deepocsort.py:293)
                    v.age[slot] = 0; v.tsu[slot] = 0; v.hits[slot] = 0; v.hit_streak[slot] = 0;
                    v.observed[slot] = 0; v.has_saved[slot] = 0; v.n_miss[slot] = 0; v.n_obs[slot] = 0; v.has_vel[slot] = 0;
                    for (int e = 0; e < 5; ++e) v.last_obs[slot * 5 + e] = -1.0;
                    v.conf[slot] = d[4]; v.cls[slot] = d[5]; v.det_ind[slot] = (float)j;
                    v.list[nt + q] = slot;
                }
            }
        }
#else
        for (int q = c.tid; q < n_ud; q += c.nthr) {                       // KalmanBoxTracker(is_obb=True).__init__, ocsort.py:121-154, :192-215
            const int slot = v.tmp_a[q], kd = v.un_d[q], j = v.keep[kd];
            const float* d = v.dets + j * DOCS_DET_COLS;
            double box[5], z[5];
            for (int e = 0; e < 5; ++e) box[e] = (double)d[e];
            obb_to_z(box, z);
            double* kf = v.kf + (long)slot * DOCS_KF_STRIDE;
            for (int e = 0; e < 9; ++e) kf[e] = e < 5 ? z[e] : 0.0;
            for (int e = 0; e < 81; ++e) kf[9 + e] = (e / 9 == e % 9) ? (e / 9 < 5 ? 10.0 : 10000.0) : 0.0;      // P = 10 * diag(1 x5, 1000 x4)
            v.slot_used[slot] = 1;
            v.id[slot] = id0 + 1 + q;                                     // rows carry id + 1 with KalmanBoxTracker.count = 0 (ocsort.py:358, :545)
            v.age[slot] = 0; v.tsu[slot] = 0; v.hits[slot] = 0; v.hit_streak[slot] = 0;
            v.observed[slot] = 0; v.has_saved[slot] = 0; v.n_miss[slot] = 0; v.n_obs[slot] = 0; v.has_vel[slot] = 0;
            for (int e = 0; e < 6; ++e) v.last_obs[slot * 6 + e] = -1.0;
            v.conf[slot] = d[5]; v.cls[slot] = d[6]; v.det_ind[slot] = (float)j;
            v.list[nt + q] = slot;
        }
#endif
        __syncthreads();
        nt += n_ud;
        if (c.tid == 0) { *v.id_count = id0 + n_ud; *v.n_tracks = nt; }
        __syncthreads();
    }

    // ---- output rows in reversed list order, then drop the tracks that are too old (deepocsort.py:467-489) ----
    auto emits = [&](int r) {
        const int slot = v.list[nt - 1 - r];
        return v.tsu[slot] < 1 && (v.hit_streak[slot] >= cfg.min_hits || frame <= cfg.min_hits);
    };
    const int n_out = block_append_if(c, nt, emits, [&](int r) { return v.list[nt - 1 - r]; }, v.tmp_a, 0);
    for (int q = c.tid; q < n_out; q += c.nthr) {
        const int slot = v.tmp_a[q];
        const double* lo = v.last_obs + slot * DOCS_OBS;
        double b[DOCS_BOX];
#if !BM_OBB
        if (lo[0] + lo[1] + lo[2] + lo[3] + lo[4] < 0) {
            const double* x = v.kf + (long)slot * KF_STRIDE;
            x_to_box(x[0], x[1], x[2], x[3], b);
        } else {
            for (int e = 0; e < 4; ++e) b[e] = lo[e];
        }
#else       // ocsort.py:532-547: the state's box for a track without an observation, else the last observation
        if (lo[0] + lo[1] + lo[2] + lo[3] + lo[4] + lo[5] < 0) x_to_obb(v.kf + (long)slot * DOCS_KF_STRIDE, b);
        else for (int e = 0; e < 5; ++e) b[e] = lo[e];
#endif
        float* o = v.out + q * DOCS_OUT_COLS;
        for (int e = 0; e < DOCS_BOX; ++e) o[e] = (float)b[e];
        o[DOCS_BOX] = (float)v.id[slot]; o[DOCS_BOX + 1] = v.conf[slot]; o[DOCS_BOX + 2] = v.cls[slot]; o[DOCS_BOX + 3] = v.det_ind[slot];
    }
    if (c.tid == 0) *v.out_n = n_out;
    const int n_live = block_append_if(c, nt, [&](int t) { return !(v.tsu[v.list[t]] > cfg.max_age); }, [&](int t) { return v.list[t]; }, v.tmp_b, 0);
    if (n_live != nt) {
        for (int t = c.tid; t < nt; t += c.nthr) if (v.tsu[v.list[t]] > cfg.max_age) v.slot_used[v.list[t]] = 0;
        __syncthreads();
        for (int t = c.tid; t < n_live; t += c.nthr) v.list[t] = v.tmp_b[t];
        if (c.tid == 0) *v.n_tracks = n_live;
    }
    __syncthreads();
}
