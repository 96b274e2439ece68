"""CPU restatement of the reference StrongSORT update path -- TEST INFRASTRUCTURE ONLY.

Follows (statement order and NumPy / SciPy / torch calls kept, so results are bit-identical on one host):
  * boxmot/trackers/bbox/strongsort/strongsort.py:69-123 (StrongSort._update_impl);
  * strongsort/sort/tracker.py:63-169 (Tracker.predict / update / _match / _initiate_track);
  * strongsort/sort/track.py:25-208 (Track: camera_update, predict, update, mark_missed, EMA feature);
  * strongsort/sort/linear_assignment.py:14-79 (min_cost_matching, SciPy LSA), :82-142 (matching_cascade),
    :145-198 (gate_cost_matrix), :201-284 (_cosine_distance, _nn_cosine_distance), :286-353
    (NearestNeighborDistanceMetric, per-target sample bank with budget);
  * strongsort/sort/iou_matching.py:10-87 (iou, iou_cost); sort/detection.py (to_xyah);
  * motion/kalman_filters/xyah.py:8-172 over base.py:234-355, :523-551 (initiate / predict / project with the
    NSA confidence scaling / update / gating_distance).
Pinned against the reference classes themselves (tests/test_oracle_vs_reference.py, tests/golden/strongsort_golden.npz);
the assignment is the real ``scipy.optimize.linear_sum_assignment`` here, so this oracle has no unpinned part.
The reference applies its ECC camera-motion object unconditionally; the oracle takes the 2x3 warp as an argument
(identity by default), which is what the reference computes with an identity estimator injected.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg
import torch
from scipy.optimize import linear_sum_assignment

DEFAULTS = dict(
    max_age=30,                                                               # basetracker.py:19-31
    min_conf=0.1, max_cos_dist=0.2, max_iou_dist=0.7, n_init=3, nn_budget=100, mc_lambda=0.98, ema_alpha=0.9,
)
INFTY_COST = 1e5
CHI2_4 = 9.4877
TENTATIVE, CONFIRMED, DELETED = 1, 2, 3
STD_POS, STD_VEL = 1.0 / 20, 1.0 / 160

_F = np.eye(8)
for _i in range(4):
    _F[_i, 4 + _i] = 1.0
_H = np.eye(4, 8)


# ---- KalmanFilterXYAH (stateless use) ----
def kf_initiate(z):
    z = np.asarray(z, dtype=float).copy()
    mean = np.r_[z, np.zeros_like(z)]
    std = [2 * STD_POS * z[3], 2 * STD_POS * z[3], 1e-2, 2 * STD_POS * z[3],
           10 * STD_VEL * z[3], 10 * STD_VEL * z[3], 1e-5, 10 * STD_VEL * z[3]]
    cov = np.diag(np.square(std))
    mean[2] = max(float(mean[2]), 1e-4)
    mean[3] = max(float(mean[3]), 1e-4)
    return mean, cov


def kf_predict(mean, cov):
    std_pos = [STD_POS * mean[3], STD_POS * mean[3], 1e-2, STD_POS * mean[3]]
    std_vel = [STD_VEL * mean[3], STD_VEL * mean[3], 1e-5, STD_VEL * mean[3]]
    motion_cov = np.diag(np.square(np.r_[std_pos, std_vel]))
    mean = np.dot(mean, _F.T)
    cov = np.linalg.multi_dot((_F, cov, _F.T)) + motion_cov
    mean[2] = max(float(mean[2]), 1e-4)
    mean[3] = max(float(mean[3]), 1e-4)
    return mean, cov


def kf_project(mean, cov, confidence=0.0):
    std = [STD_POS * mean[3], STD_POS * mean[3], 1e-1, STD_POS * mean[3]]
    std = [(1 - confidence) * x for x in std]
    innovation_cov = np.diag(np.square(std))
    return np.dot(_H, mean), np.linalg.multi_dot((_H, cov, _H.T)) + innovation_cov


def kf_update(mean, cov, z, confidence):
    pm, pc = kf_project(mean, cov, confidence)
    cf, lower = scipy.linalg.cho_factor(pc, lower=True, check_finite=False)
    K = scipy.linalg.cho_solve((cf, lower), np.dot(cov, _H.T).T, check_finite=False).T
    new_mean = mean + np.dot(z - pm, K.T)
    new_cov = cov - np.linalg.multi_dot((K, pc, K.T))
    new_mean[2] = max(float(new_mean[2]), 1e-4)
    new_mean[3] = max(float(new_mean[3]), 1e-4)
    return new_mean, new_cov


def kf_gating_distance(mean, cov, measurements):
    pm, pc = kf_project(mean, cov)
    d = measurements - pm
    L = np.linalg.cholesky(pc)
    z = scipy.linalg.solve_triangular(L, d.T, lower=True, check_finite=False, overwrite_b=True)
    return np.sum(z * z, axis=0)


# ---- the device's fp64 operation order for the Kalman update and the gating distance ("dot_rule = device") ----
# The reference solves the 4 x 4 innovation system with LAPACK (cho_factor / cho_solve / solve_triangular) and forms
# K S K^T with BLAS; the kernels (strongsort_step.hpp: ss_kf_update_wave, ss_gating_distance) use explicit scalar loops --
# same algebra, sums in a fixed order, no fused multiply-add.  The two agree to ~1e-13 relative, which is enough to flip
# SciPy's choice among exactly tied (clamped) assignment entries; restated here operation for operation so that the
# oracle's filter state and gated costs are bit-identical to the device's.
def _chol4(S):
    L = [[0.0] * 4 for _ in range(4)]
    for c in range(4):
        d = S[c][c]
        for k in range(c):
            d -= L[c][k] * L[c][k]
        d = math.sqrt(d)
        L[c][c] = d
        for r in range(c + 1, 4):
            t = S[r][c]
            for k in range(c):
                t -= L[r][k] * L[c][k]
            L[r][c] = t / d
    return L


def _innovation(mean, cov, confidence, scaled):
    S = [[float(cov[a, b]) for b in range(4)] for a in range(4)]
    for a in range(4):
        base = 1e-1 if a == 2 else STD_POS * float(mean[3])
        sd = (1 - confidence) * base if scaled else base
        S[a][a] = S[a][a] + sd * sd
    return S


def kf_update_device_rule(mean, cov, z, confidence):
    m = [float(x) for x in mean]
    P = [[float(cov[a, b]) for b in range(8)] for a in range(8)]
    S = _innovation(mean, cov, float(confidence), True)
    L = _chol4(S)
    K = []
    for r in range(8):                              # row r of the gain: L y = P[r, :4], L^T K = y
        y = [0.0] * 4
        for k in range(4):
            t = P[r][k]
            for q in range(k):
                t -= L[k][q] * y[q]
            y[k] = t / L[k][k]
        Kr = [0.0] * 4
        for k in range(3, -1, -1):
            t = y[k]
            for q in range(k + 1, 4):
                t -= L[q][k] * Kr[q]
            Kr[k] = t / L[k][k]
        K.append(Kr)
    new_mean = np.empty(8)
    for i in range(8):
        acc = 0.0
        for a in range(4):
            acc += (float(z[a]) - m[a]) * K[i][a]
        v = m[i] + acc
        if i in (2, 3):
            v = v if v > 1e-4 else 1e-4
        new_mean[i] = v
    new_cov = np.empty((8, 8))
    for i in range(8):
        for j in range(8):
            ksk = 0.0
            for a in range(4):
                mj = 0.0
                for b in range(4):
                    mj += S[a][b] * K[j][b]
                ksk += K[i][a] * mj
            new_cov[i, j] = P[i][j] - ksk
    return new_mean, new_cov


def kf_gating_distance_device_rule(mean, cov, measurements):
    L = _chol4(_innovation(mean, cov, 0.0, False))
    out = np.empty(len(measurements))
    for n, z in enumerate(measurements):
        y = [0.0] * 4
        s = 0.0
        for k in range(4):
            t = float(z[k]) - float(mean[k])
            for q in range(k):
                t -= L[k][q] * y[q]
            y[k] = t / L[k][k]
        for k in range(4):
            s += y[k] * y[k]
        out[n] = s
    return out


class _Det:
    def __init__(self, tlwh, conf, cls, det_ind, feat):
        self.tlwh, self.conf, self.cls, self.det_ind, self.feat = tlwh, conf, cls, det_ind, feat

    def to_xyah(self):
        ret = self.tlwh.copy()
        ret[:2] += ret[2:] / 2
        ret[2] /= ret[3]
        return ret


class _Track:
    def __init__(self, det, tid, n_init, max_age, ema_alpha, norm=np.linalg.norm):
        self.id = tid
        self.norm = norm                            # np.linalg.norm (reference) or the device's summation order (dot_rule="device")
        self.device_rule = norm is not np.linalg.norm
        self.conf, self.cls, self.det_ind = det.conf, det.cls, det.det_ind
        self.hits = self.age = 1
        self.time_since_update = 0
        self.ema_alpha = ema_alpha
        self.state = TENTATIVE                      # GITHUB_ACTIONS unset (track.py:91-98)
        self.features = []
        if det.feat is not None:
            det.feat /= self.norm(det.feat)
            self.features.append(det.feat)
        self.n_init, self.max_age = n_init, max_age
        self.mean, self.cov = kf_initiate(det.to_xyah())

    def to_tlwh(self):
        ret = self.mean[:4].copy()
        ret[2] *= ret[3]
        ret[:2] -= ret[2:] / 2
        return ret

    def to_tlbr(self):
        ret = self.to_tlwh()
        ret[2:] = ret[:2] + ret[2:]
        return ret

    def camera_update(self, warp):                  # track.py:139-148
        a, b = warp
        m = np.array([a, b, [0, 0, 1]]).tolist()
        x1, y1, x2, y2 = self.to_tlbr()
        x1_, y1_, _ = m @ np.array([x1, y1, 1]).T
        x2_, y2_, _ = m @ np.array([x2, y2, 1]).T
        w, h = x2_ - x1_, y2_ - y1_
        cx, cy = x1_ + w / 2, y1_ + h / 2
        self.mean[:4] = [cx, cy, w / h, h]

    def predict(self):
        self.mean, self.cov = kf_predict(self.mean, self.cov)
        self.age += 1
        self.time_since_update += 1

    def update(self, det):
        self.conf, self.cls, self.det_ind = det.conf, det.cls, det.det_ind
        self.mean, self.cov = (kf_update_device_rule if self.device_rule else kf_update)(self.mean, self.cov, det.to_xyah(), self.conf)
        feature = det.feat / self.norm(det.feat)
        smooth = self.ema_alpha * self.features[-1] + (1 - self.ema_alpha) * feature
        smooth /= self.norm(smooth)
        self.features = [smooth]
        self.hits += 1
        self.time_since_update = 0
        if self.state == TENTATIVE and self.hits >= self.n_init:
            self.state = CONFIRMED

    def mark_missed(self):
        if self.state == TENTATIVE:
            self.state = DELETED
        elif self.time_since_update > self.max_age:
            self.state = DELETED


def _cosine_distance(a, b):                         # linear_assignment.py:201-220 (data_is_normalized=False)
    a = np.asarray(a) / np.linalg.norm(a, axis=1, keepdims=True)
    b = np.asarray(b) / np.linalg.norm(b, axis=1, keepdims=True)
    return 1.0 - np.dot(a, b.T)


def _nn_cosine_distance(x, y):                      # :266-284
    x_ = torch.from_numpy(np.asarray(x))
    y_ = torch.from_numpy(np.asarray(y))
    return _cosine_distance(x_, y_).min(axis=0)


# ---- the device's documented fp32 summation order ("dot_rule = device") ----
# The reference hands the appearance product to NumPy -> OpenBLAS sgemm, whose summation order depends on the operand shapes
# and on the host CPU's kernel: the cost is only defined to a few fp32 ulps, and SciPy's choice among exactly tied (clamped)
# rows can hang on those bits.  The HIP kernels use ONE documented order (strongsort_step.hpp):
#   |v|    : lane l of a 64-lane wavefront accumulates v[k]^2 for k = l, l + 64, ... with fmaf, the 64 partial sums are added
#            as a butterfly (offsets 32, 16, ..., 1), sqrtf
#   <a, b> : one fmaf per k, k ascending, from 0 (v_mfma_f32_16x16x4_f32 is bit-for-bit that chain)
#   dist   : 1 - <a, b> / (|a| * |b|) in fp32, min over the bank
# Under this rule the oracle's cost matrix is bit-identical to the device's, which makes the id comparison exact on every
# sequence.  fmaf is emulated in 80-bit long double
# (a 24 x 24-bit product is exact there; the sum is rounded once more, which matters with probability ~2^-40 per operation).
def _fma32(a, b, c):
    return (a.astype(np.longdouble) * b.astype(np.longdouble) + c.astype(np.longdouble)).astype(np.float32)


def _norm_device_rule(v):
    v = np.ascontiguousarray(v, dtype=np.float32)
    n, dim = v.shape
    ss = np.zeros((n, 64), dtype=np.float32)
    for k0 in range(0, dim, 64):
        blk = v[:, k0:k0 + 64]
        w = blk.shape[1]
        ss[:, :w] = _fma32(blk, blk, ss[:, :w])
    for off in (32, 16, 8, 4, 2, 1):
        ss = (ss[:, :off] + ss[:, off:2 * off]).astype(np.float32)
    return np.sqrt(ss[:, 0]).astype(np.float32)


def _nn_cosine_distance_device_rule(x, y):
    x = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
    y = np.ascontiguousarray(np.asarray(y), dtype=np.float32)
    acc = np.zeros((len(x), len(y)), dtype=np.float32)
    for k in range(x.shape[1]):
        acc = _fma32(x[:, k:k + 1], y[None, :, k], acc)
    den = (_norm_device_rule(x)[:, None] * _norm_device_rule(y)[None, :]).astype(np.float32)
    dist = (np.float32(1.0) - (acc / den).astype(np.float32)).astype(np.float32)
    return dist.min(axis=0)


def iou_tlwh(bbox, candidates):                     # iou_matching.py:10-46
    bbox_tl, bbox_br = bbox[:2], bbox[:2] + bbox[2:]
    c_tl, c_br = candidates[:, :2], candidates[:, :2] + candidates[:, 2:]
    tl = np.c_[np.maximum(bbox_tl[0], c_tl[:, 0])[:, np.newaxis], np.maximum(bbox_tl[1], c_tl[:, 1])[:, np.newaxis]]
    br = np.c_[np.minimum(bbox_br[0], c_br[:, 0])[:, np.newaxis], np.minimum(bbox_br[1], c_br[:, 1])[:, np.newaxis]]
    wh = np.maximum(0.0, br - tl)
    inter = wh.prod(axis=1)
    return inter / (bbox[2:].prod() + candidates[:, 2:].prod(axis=1) - inter)


class StrongSortOracle:
    def __init__(self, reid=None, dot_rule: str = "numpy", **kw):
        """``dot_rule``: "numpy" = the reference's calls (NumPy / OpenBLAS float32 product); "device" = the HIP kernels'
        documented fp32 summation order (see _nn_cosine_distance_device_rule) -- same algorithm, a defined rounding."""
        if dot_rule not in ("numpy", "device"):
            raise ValueError("dot_rule must be 'numpy' or 'device'")
        self.dot_rule = dot_rule
        cfg = dict(DEFAULTS)
        unknown = set(kw) - set(cfg)
        if unknown:
            raise TypeError(f"unknown StrongSORT options: {sorted(unknown)}")
        cfg.update(kw)
        self.cfg = cfg
        self.reid = reid
        self.tracks = []
        self.next_id = 1
        self.samples = {}
        self.frame_count = 0

    # ---- association ----
    def _min_cost_matching(self, metric, max_distance, tracks, dets, track_idx, det_idx):   # :14-79
        if len(det_idx) == 0 or len(track_idx) == 0:
            self.last_costs.append(None)
            return [], track_idx, det_idx
        cost = metric(tracks, dets, track_idx, det_idx)
        # (cost-value parity tests: the metric's matrix as returned -- gated appearance, or IoU -- and the clamped one the solver gets)
        self.last_costs.append({"raw": np.array(cost, dtype=np.float64)})
        cost[cost > max_distance] = max_distance + 1e-5
        self.last_costs[-1]["clamped"] = np.array(cost, dtype=np.float64)
        rows, cols = linear_sum_assignment(cost)
        matches, un_t, un_d = [], [], []
        for col, d in enumerate(det_idx):
            if col not in cols:
                un_d.append(d)
        for row, t in enumerate(track_idx):
            if row not in rows:
                un_t.append(t)
        for row, col in zip(rows, cols):
            t, d = track_idx[row], det_idx[col]
            if cost[row, col] > max_distance:
                un_t.append(t)
                un_d.append(d)
            else:
                matches.append((t, d))
        self.last_cost = cost
        return matches, un_t, un_d

    def _gated_metric(self, tracks, dets, track_idx, det_idx):       # tracker.py:110-125, linear_assignment.py:145-198
        feats = np.array([dets[i].feat for i in det_idx])
        cost = np.zeros((len(track_idx), len(feats)))
        self.last_app = {}                  # (track id, detection index) -> raw appearance distance of this frame (parity debugging)
        for i, t in enumerate(track_idx):
            cost[i, :] = (_nn_cosine_distance_device_rule if self.dot_rule == "device" else _nn_cosine_distance)(self.samples[tracks[t].id], feats)
            for j, di in enumerate(det_idx):
                self.last_app[(tracks[t].id, int(dets[di].det_ind))] = np.float32(cost[i, j])
        meas = np.asarray([dets[i].to_xyah() for i in det_idx])
        for row, t in enumerate(track_idx):
            gd = (kf_gating_distance_device_rule if self.dot_rule == "device" else kf_gating_distance)(tracks[t].mean, tracks[t].cov, meas)
            cost[row, gd > CHI2_4] = INFTY_COST
            cost[row] = self.cfg["mc_lambda"] * cost[row] + (1 - self.cfg["mc_lambda"]) * gd
        return cost

    @staticmethod
    def _iou_cost(tracks, dets, track_idx, det_idx):                  # iou_matching.py:49-87
        cost = np.zeros((len(track_idx), len(det_idx)))
        for row, t in enumerate(track_idx):
            if tracks[t].time_since_update > 1:
                cost[row, :] = INFTY_COST
                continue
            cand = np.asarray([dets[i].tlwh for i in det_idx])
            cost[row, :] = 1.0 - iou_tlwh(tracks[t].to_tlwh(), cand)
        return cost

    def update(self, dets, img=None, embs=None, warp=None):
        """dets (N,6) [x1,y1,x2,y2,conf,cls] -> (M,8) fp32 rows (what ``StrongSort.update`` hands back)."""
        c = self.cfg
        self.frame_count += 1
        self.last_app = {}
        self.last_costs = []            # one entry per min_cost_matching call of this frame: stage A (appearance), stage B (IoU)
        dets = np.asarray(dets)
        if dets.size == 0:
            dets = np.empty((0, 7), dtype=np.float32)
        else:
            dets = np.hstack([dets, np.arange(len(dets), dtype=np.int32).reshape(-1, 1)])
        keep = dets[:, 4] >= c["min_conf"]
        dets = dets[keep]
        xyxy, confs, clss, det_ind = dets[:, :4], dets[:, 4], dets[:, 5], dets[:, 6]
        if len(self.tracks) >= 1:
            w = np.eye(2, 3) if warp is None else warp
            for t in self.tracks:
                t.camera_update(w)
        feats = embs[keep] if embs is not None else self.reid.get_features(xyxy, img)
        tlwh = np.copy(xyxy)
        tlwh[..., 2] = xyxy[..., 2] - xyxy[..., 0]
        tlwh[..., 3] = xyxy[..., 3] - xyxy[..., 1]
        D = [_Det(b, cf, cl, di, f) for b, cf, cl, di, f in zip(tlwh, confs, clss, det_ind, feats)]

        for t in self.tracks:
            t.predict()
        # ---- Tracker.update ----
        tracks = self.tracks
        confirmed = [i for i, t in enumerate(tracks) if t.state == CONFIRMED]
        unconfirmed = [i for i, t in enumerate(tracks) if t.state != CONFIRMED]
        matches_a, _, un_d = self._min_cost_matching(self._gated_metric, c["max_cos_dist"], tracks, D, list(confirmed),
                                                     list(range(len(D))))
        un_t_a = list(set(confirmed) - set(k for k, _ in matches_a))
        iou_cand = unconfirmed + [k for k in un_t_a if tracks[k].time_since_update == 1]
        un_t_a = [k for k in un_t_a if tracks[k].time_since_update != 1]
        matches_b, un_t_b, un_d = self._min_cost_matching(self._iou_cost, c["max_iou_dist"], tracks, D, iou_cand, un_d)
        matches = matches_a + matches_b
        un_t = list(set(un_t_a + un_t_b))
        for ti, di in matches:
            tracks[ti].update(D[di])
        for ti in un_t:
            tracks[ti].mark_missed()
        for di in un_d:
            norm = (lambda v: _norm_device_rule(np.asarray(v, dtype=np.float32)[None, :])[0]) if self.dot_rule == "device" else np.linalg.norm
            tracks.append(_Track(D[di], self.next_id, c["n_init"], c["max_age"], c["ema_alpha"], norm))
            self.next_id += 1
        self.tracks = tracks = [t for t in tracks if t.state != DELETED]
        active = [t.id for t in tracks if t.state == CONFIRMED]
        for t in tracks:
            if t.state != CONFIRMED:
                continue
            for f in t.features:
                self.samples.setdefault(t.id, []).append(f)
                if c["nn_budget"] is not None:
                    self.samples[t.id] = self.samples[t.id][-c["nn_budget"]:]
        self.samples = {k: self.samples[k] for k in active}

        out = []
        for t in tracks:
            if t.state != CONFIRMED or t.time_since_update >= 1:
                continue
            x1, y1, x2, y2 = t.to_tlbr()
            out.append(np.concatenate(([x1, y1, x2, y2], [t.id], [t.conf], [t.cls], [t.det_ind])).reshape(1, -1))
        raw = np.concatenate(out) if out else np.empty((0, 8), dtype=float)
        return np.asarray(raw, dtype=np.float32)

    def dump(self):
        t = self.tracks
        return {
            "id": np.array([k.id for k in t], dtype=np.int64),
            "state": np.array([k.state for k in t], dtype=np.int64),
            "hits": np.array([k.hits for k in t], dtype=np.int64),
            "age": np.array([k.age for k in t], dtype=np.int64),
            "time_since_update": np.array([k.time_since_update for k in t], dtype=np.int64),
            "mean": np.array([k.mean for k in t], dtype=np.float64).reshape(len(t), 8),
            "cov": np.array([k.cov for k in t], dtype=np.float64).reshape(len(t), 8, 8),
            "feat": [np.asarray(k.features[-1]) for k in t],
            "bank": {k: len(v) for k, v in self.samples.items()},
            "next_id": self.next_id,
        }
