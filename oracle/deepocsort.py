"""CPU restatement of the reference DeepOCSORT update path -- TEST INFRASTRUCTURE ONLY.

Follows (statement order and NumPy calls kept, so results are bit-identical on one host):
  * boxmot/trackers/bbox/deepocsort/deepocsort.py:17-49 (k_previous_obs, convert_x_to_bbox, speed_direction),
    :51-233 (KalmanBoxTracker), :302-492 (DeepOcSort._update_impl);
  * boxmot/motion/kalman_filters/xysr.py:368-476 (predict / freeze / unfreeze / update of the 7-state filter)
    over base.py:366-459 (predict_state / project_state / update_state, Joseph form) and :461-500
    (_safe_cho_factor jitter ladder);
  * boxmot/trackers/association/association.py:8-152 (speed_direction_batch, compute_aw_max_metric, associate),
    iou.py:134-150 (iou_batch), common/geometry.py:103-124 (xyxy2xysr).
Pinned against the reference classes themselves (tests/test_oracle_vs_reference.py; fixtures
tests/golden/deepocsort_golden.npz).  ``lap.lapjv`` is the oracle stand-in (oracle/lap.py, parity unpinned).
Scope: axis-aligned boxes, ``cmc_off=True`` (or a warp supplied by the caller), every axis-aligned ``asso_func``
(iou.py:118-423: iou, giou, diou, ciou, hmiou, centroid).
"""
from __future__ import annotations

from collections import deque
from copy import deepcopy

import numpy as np
import scipy.linalg

from oracle import lap as oracle_lap

DEFAULTS = dict(
    det_thresh=0.3, max_age=30, max_obs=50, min_hits=3, iou_threshold=0.3,           # basetracker.py:19-31
    delta_t=3, inertia=0.2, w_association_emb=0.5, alpha_fixed_emb=0.95, aw_param=0.5,  # deepocsort.py:263-276
    embedding_off=False, aw_off=False, Q_xy_scaling=0.01, Q_s_scaling=0.0001,
    asso_func="iou",                    # basetracker.py:28; the axis-aligned names of iou.py:408-417
    frame_wh=None,                      # (w, h) for `centroid`; None: read off the first image (basetracker.py:175-180)
)

_F = np.eye(7)
_F[0, 4] = _F[1, 5] = _F[2, 6] = 1.0
_H = np.zeros((4, 7))
_H[:4, :4] = np.eye(4)


def _to_z(box):
    """geometry.py:103-124 -- [x1,y1,x2,y2] -> (4,1) [cx, cy, area, w/(h+1e-6)]."""
    y = np.copy(box[0:4])
    w = y[2] - y[0]
    h = y[3] - y[1]
    y[0] = y[0] + w / 2.0
    y[1] = y[1] + h / 2.0
    y[2] = w * h
    y[3] = w / (h + 1e-6)
    return y.reshape((4, 1))


def _to_box(x):
    """deepocsort.py:29-40 (score=None)."""
    w = np.sqrt(x[2] * x[3])
    h = x[2] / w
    return np.array([x[0] - w / 2.0, x[1] - h / 2.0, x[0] + w / 2.0, x[1] + h / 2.0]).reshape((1, 4))


def _safe_cho_factor(m):
    """base.py:461-500."""
    try:
        return scipy.linalg.cho_factor(m, lower=True, check_finite=False)
    except scipy.linalg.LinAlgError:
        pass
    diag = np.diagonal(m)
    scale = float(np.max(np.abs(diag))) if diag.size else 1.0
    if not np.isfinite(scale) or scale <= 0.0:
        scale = 1.0
    eye = np.eye(m.shape[0])
    for exponent in range(-12, 4):
        try:
            return scipy.linalg.cho_factor(m + scale * (10.0 ** exponent) * eye, lower=True, check_finite=False)
        except scipy.linalg.LinAlgError:
            continue
    sym = 0.5 * (m + m.T)
    vals, vecs = np.linalg.eigh(sym)
    vals = np.clip(vals, max(scale * 1e-6, 1e-12), None)
    rep = (vecs * vals) @ vecs.T
    return scipy.linalg.cho_factor(0.5 * (rep + rep.T), lower=True, check_finite=False)


class _FilterXYSR:
    """KalmanFilterXYSR(dim_x=7, dim_z=4) as KalmanBoxTracker configures it (deepocsort.py:82-114)."""

    def __init__(self, z0, q_xy, q_s):
        self.x = np.zeros((7, 1))
        self.P = np.eye(7)
        self.Q = np.eye(7)
        self.R = np.eye(4)
        self.R[2:, 2:] *= 10.0
        self.P[4:, 4:] *= 1000.0
        self.P *= 10.0
        self.Q[4:6, 4:6] *= q_xy
        self.Q[-1, -1] *= q_s
        self.x[:4] = z0
        self.history = deque([], maxlen=50)       # KalmanFilterXYSR default max_obs (xysr.py:18,47-48)
        self.saved = None
        self.observed = False

    def _constrain(self):                          # xysr.py:154-161
        self.x[2, :] = np.maximum(self.x[2, :], 1e-6)
        self.x[3, :] = np.maximum(self.x[3, :], 1e-6)
        self.P = 0.5 * (self.P + self.P.T)

    def predict(self):                             # xysr.py:368-377, base.py:366-391
        self.x = np.dot(_F, self.x)
        self.P = 1.0 * np.dot(np.dot(_F, self.P), _F.T) + self.Q
        self._constrain()

    @staticmethod
    def _prepare(z):                               # xysr.py:139-152
        m = np.asarray(z, dtype=float)
        if m.shape != (4, 1):
            m = m.reshape((4, 1))
        m[2, 0] = max(float(m[2, 0]), 1e-6)
        m[3, 0] = max(float(m[3, 0]), 1e-6)
        return m

    def update(self, z):                           # xysr.py:442-476
        m = None if z is None else self._prepare(z)
        self.history.append(None if m is None else m.copy())
        if m is None:
            if self.observed and len(self.history) >= 2:
                self.saved = deepcopy(self.__dict__)        # freeze()
            self.observed = False
            return
        if not self.observed:
            self._unfreeze()
        self.observed = True
        self._update_state(m)
        self._constrain()
        self.history.append(m.copy())              # observed measurements are stored twice

    def _update_state(self, m):                    # base.py:414-459
        S = np.dot(np.dot(_H, self.P), _H.T) + self.R
        S = 0.5 * (S + S.T)
        cf = _safe_cho_factor(S)
        K = scipy.linalg.cho_solve(cf, np.dot(self.P, _H.T).T, check_finite=False).T
        y = m - np.dot(_H, self.x)
        self.x = self.x + np.dot(K, y)
        ikh = np.eye(7) - np.dot(K, _H)
        self.P = np.linalg.multi_dot((ikh, self.P, ikh.T)) + np.linalg.multi_dot((K, self.R, K.T))
        self.P = 0.5 * (self.P + self.P.T)

    def apply_affine(self, m, t):                  # xysr.py:311-366 (axis-aligned branch)
        m = np.asarray(m, dtype=float).reshape((2, 2))
        t = np.asarray(t, dtype=float).reshape((2, 1))
        self.x[:2] = m @ self.x[:2] + t
        self.x[4:6] = m @ self.x[4:6]
        self.P[:2, :2] = m @ self.P[:2, :2] @ m.T
        self.P[4:6, 4:6] = m @ self.P[4:6, 4:6] @ m.T
        if not self.observed and self.saved is not None:
            self.saved["x"][:2] = m @ self.saved["x"][:2] + t
            self.saved["x"][4:6] = m @ self.saved["x"][4:6]
            self.saved["P"][:2, :2] = m @ self.saved["P"][:2, :2] @ m.T
            self.saved["P"][4:6, 4:6] = m @ self.saved["P"][4:6, 4:6] @ m.T
        self._constrain()

    def _unfreeze(self):                           # xysr.py:383-440 (observation-centric re-update)
        if self.saved is None:
            return
        new_history = deepcopy(list(self.history))
        self.__dict__ = self.saved
        self.history = deque(list(self.history)[:-1], maxlen=50)
        idx = np.where(np.array([int(o is None) for o in new_history]) == 0)[0]
        if len(idx) < 2:
            return
        i1, i2 = idx[-2], idx[-1]
        b1 = np.asarray(new_history[i1], dtype=float).reshape(-1)
        b2 = np.asarray(new_history[i2], dtype=float).reshape(-1)
        x1, y1, s1, r1 = b1[:4]
        w1, h1 = np.sqrt(s1 * r1), np.sqrt(s1 / r1)
        x2, y2, s2, r2 = b2[:4]
        w2, h2 = np.sqrt(s2 * r2), np.sqrt(s2 / r2)
        gap = i2 - i1
        if gap <= 0:
            return
        dx, dy = (x2 - x1) / gap, (y2 - y1) / gap
        dw, dh = (w2 - w1) / gap, (h2 - h1) / gap
        for i in range(gap):
            x = x1 + (i + 1) * dx
            y = y1 + (i + 1) * dy
            w = w1 + (i + 1) * dw
            h = h1 + (i + 1) * dh
            self.update(np.array([x, y, w * h, w / float(h)], dtype=float).reshape((4, 1)))
            if i != gap - 1:
                self.predict()
                self.history.pop()
        self.history.pop()


class _Track:
    """KalmanBoxTracker (deepocsort.py:51-233)."""

    def __init__(self, det, tid, delta_t, emb, q_xy, q_s):
        self.conf, self.cls, self.det_ind = det[4], det[5], det[6]
        self.kf = _FilterXYSR(_to_z(det[0:5]), q_xy, q_s)
        self.id = tid
        self.time_since_update = 0
        self.hits = self.hit_streak = self.age = 0
        self.last_observation = np.array([-1, -1, -1, -1, -1])
        self.observations = {}
        self.velocity = None
        self.delta_t = delta_t
        self.emb = emb

    def apply_affine_correction(self, affine):      # :190-209 (last_observation and observations[age] share storage)
        m = affine[:, :2]
        t = affine[:, 2].reshape(2, 1)
        if self.last_observation.sum() > 0:
            ps = self.last_observation[:4].reshape(2, 2).T
            ps = m @ ps + t
            self.last_observation[:4] = ps.T.reshape(-1)
        for dt in range(self.delta_t, -1, -1):
            if self.age - dt in self.observations:
                ps = self.observations[self.age - dt][:4].reshape(2, 2).T
                ps = m @ ps + t
                self.observations[self.age - dt][:4] = ps.T.reshape(-1)
        self.kf.apply_affine(m, t)

    def predict(self):                              # :211-225
        if (self.kf.x[6] + self.kf.x[2]) <= 0:
            self.kf.x[6] *= 0.0
        self.kf.predict()
        self.age += 1
        if self.time_since_update > 0:
            self.hit_streak = 0
        self.time_since_update += 1
        return _to_box(self.kf.x)

    def update(self, det):                          # :143-181
        if det is None:
            self.kf.update(None)
            return
        box = det[0:5]
        self.conf, self.cls, self.det_ind = det[4], det[5], det[6]
        if self.last_observation.sum() >= 0:
            prev = None
            for dt in range(self.delta_t, 0, -1):
                if self.age - dt in self.observations:
                    prev = self.observations[self.age - dt]
                    break
            if prev is None:
                prev = self.last_observation
            cx1, cy1 = (prev[0] + prev[2]) / 2.0, (prev[1] + prev[3]) / 2.0
            cx2, cy2 = (box[0] + box[2]) / 2.0, (box[1] + box[3]) / 2.0
            speed = np.array([cy2 - cy1, cx2 - cx1])
            self.velocity = speed / (np.sqrt((cy2 - cy1) ** 2 + (cx2 - cx1) ** 2) + 1e-6)
        self.last_observation = box
        self.observations[self.age] = box
        self.time_since_update = 0
        self.hits += 1
        self.hit_streak += 1
        self.kf.update(_to_z(box))

    def update_emb(self, emb, alpha):               # :183-185
        self.emb = alpha * self.emb + (1 - alpha) * emb
        self.emb /= np.linalg.norm(self.emb)


def _k_previous_obs(observations, cur_age, k):      # deepocsort.py:17-26
    if len(observations) == 0:
        return [-1, -1, -1, -1, -1]
    for i in range(k):
        if cur_age - (k - i) in observations:
            return observations[cur_age - (k - i)]
    return observations[max(observations.keys())]


def iou_batch(b1, b2):                              # iou.py:134-150
    b2 = np.expand_dims(b2, 0)
    b1 = np.expand_dims(b1, 1)
    xx1 = np.maximum(b1[..., 0], b2[..., 0])
    yy1 = np.maximum(b1[..., 1], b2[..., 1])
    xx2 = np.minimum(b1[..., 2], b2[..., 2])
    yy2 = np.minimum(b1[..., 3], b2[..., 3])
    wh = np.maximum(0.0, xx2 - xx1) * np.maximum(0.0, yy2 - yy1)
    return wh / ((b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
                 + (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1]) - wh)


def _inter_enclose(b1, b2):
    b2 = np.expand_dims(b2, 0)
    b1 = np.expand_dims(b1, 1)
    w = np.maximum(0.0, np.minimum(b1[..., 2], b2[..., 2]) - np.maximum(b1[..., 0], b2[..., 0]))
    h = np.maximum(0.0, np.minimum(b1[..., 3], b2[..., 3]) - np.maximum(b1[..., 1], b2[..., 1]))
    area1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    area2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    wc = np.maximum(b1[..., 2], b2[..., 2]) - np.minimum(b1[..., 0], b2[..., 0])
    hc = np.maximum(b1[..., 3], b2[..., 3]) - np.minimum(b1[..., 1], b2[..., 1])
    return b1, b2, w, h, area1, area2, wc, hc


def hmiou_batch(b1, b2):                            # iou.py:153-203
    b1, b2, w, h, area1, area2, wc, hc = _inter_enclose(b1, b2)
    o = h / np.maximum(1e-10, hc)
    inter = w * h
    return inter / (area1 + area2 - inter + 1e-10) * o


def giou_batch(b1, b2):                             # iou.py:205-244
    b1, b2, w, h, area1, area2, wc, hc = _inter_enclose(b1, b2)
    wh = w * h
    union = area1 + area2 - wh
    iou = wh / union
    assert (wc > 0).all() and (hc > 0).all()
    enclose = wc * hc
    return (iou - (enclose - union) / enclose + 1.0) / 2.0


def _centre_dist2(b1, b2):
    return ((b1[..., 0] + b1[..., 2]) / 2.0 - (b2[..., 0] + b2[..., 2]) / 2.0) ** 2 \
        + ((b1[..., 1] + b1[..., 3]) / 2.0 - (b2[..., 1] + b2[..., 3]) / 2.0) ** 2


def diou_batch(b1, b2):                             # iou.py:346-391
    b1, b2, w, h, area1, area2, wc, hc = _inter_enclose(b1, b2)
    wh = w * h
    iou = wh / (area1 + area2 - wh)
    return (iou - _centre_dist2(b1, b2) / (wc ** 2 + hc ** 2) + 1) / 2.0


def ciou_batch(b1, b2):                             # iou.py:283-344
    eps = 1e-7
    b1, b2, w, h, area1, area2, wc, hc = _inter_enclose(b1, b2)
    wh = w * h
    iou = wh / (area1 + area2 - wh + eps)
    outer = wc ** 2 + hc ** 2 + eps
    w1, h1 = b1[..., 2] - b1[..., 0], b1[..., 3] - b1[..., 1] + eps
    w2, h2 = b2[..., 2] - b2[..., 0], b2[..., 3] - b2[..., 1] + eps
    v = (4 / (np.pi ** 2)) * ((np.arctan(w2 / h2) - np.arctan(w1 / h1)) ** 2)
    alpha = v / ((1 - iou) + v + eps)
    return (iou - (_centre_dist2(b1, b2) / outer) + (alpha * v) + 1) / 2.0


def centroid_batch(b1, b2, w, h):                   # iou.py:253-268
    c1 = np.stack(((b1[..., 0] + b1[..., 2]) / 2, (b1[..., 1] + b1[..., 3]) / 2), axis=-1)
    c2 = np.stack(((b2[..., 0] + b2[..., 2]) / 2, (b2[..., 1] + b2[..., 3]) / 2), axis=-1)
    d = np.sqrt(np.sum((np.expand_dims(c1, 1) - np.expand_dims(c2, 0)) ** 2, axis=-1))
    return 1 - d / np.sqrt(w ** 2 + h ** 2)


def asso_function(name, w=None, h=None):            # AssociationFunction._get_asso_func, iou.py:396-423 (axis-aligned names)
    table = {"iou": iou_batch, "hmiou": hmiou_batch, "giou": giou_batch, "ciou": ciou_batch, "diou": diou_batch,
             "centroid": lambda a, b: centroid_batch(a, b, w, h)}
    if name not in table:
        raise ValueError(f"Invalid association mode: {name}. Choose from {list(table.keys())}")
    return table[name]


def _assign(cost):                                  # association.py:20-24
    _, x, y = oracle_lap.lapjv(cost, extend_cost=True)
    return np.array([[y[i], i] for i in x if i >= 0])


def aw_max_metric(emb_cost, w_emb0, bottom):        # association.py:29-58
    w_emb = np.full_like(emb_cost, w_emb0)
    for i in range(emb_cost.shape[0]):
        inds = np.argsort(-emb_cost[i])
        if len(inds) < 2:
            continue
        if emb_cost[i, inds[0]] == 0:
            rw = 0
        else:
            rw = 1 - max((emb_cost[i, inds[1]] / emb_cost[i, inds[0]]) - bottom, 0) / (1 - bottom)
        w_emb[i] *= rw
    for j in range(emb_cost.shape[1]):
        inds = np.argsort(-emb_cost[:, j])
        if len(inds) < 2:
            continue
        if emb_cost[inds[0], j] == 0:
            cw = 0
        else:
            cw = 1 - max((emb_cost[inds[1], j] / emb_cost[inds[0], j]) - bottom, 0) / (1 - bottom)
        w_emb[:, j] *= cw
    return w_emb * emb_cost


def associate(dets, trks, iou_threshold, velocities, previous_obs, vdc_weight, emb_cost, w_assoc_emb, aw_off, aw_param, asso=iou_batch,
              record=None):
    """association.py:61-152; returns (matches (K,2) [det, trk], unmatched_dets, unmatched_trks).  ``record``: a dict that receives
    the matrices of this call for the cost-value parity tests -- "iou" (dets, trks) and "final_cost" (None when the solver was not
    asked: no matrix, or the already-a-permutation early-out of :104-108)."""
    if record is not None:
        record.update(iou=None, final_cost=None)
    if len(trks) == 0:
        return np.empty((0, 2), dtype=int), np.arange(len(dets)), np.empty((0, 5), dtype=int)
    tr = previous_obs[..., np.newaxis]
    cx1, cy1 = (dets[:, 0] + dets[:, 2]) / 2.0, (dets[:, 1] + dets[:, 3]) / 2.0
    cx2, cy2 = (tr[:, 0] + tr[:, 2]) / 2.0, (tr[:, 1] + tr[:, 3]) / 2.0
    dx, dy = cx1 - cx2, cy1 - cy2
    norm = np.sqrt(dx ** 2 + dy ** 2) + 1e-6
    X, Y = dx / norm, dy / norm                     # (n_trk, n_det)
    inertia_y = np.repeat(velocities[:, 0][:, np.newaxis], Y.shape[1], axis=1)
    inertia_x = np.repeat(velocities[:, 1][:, np.newaxis], X.shape[1], axis=1)
    diff_cos = np.clip(inertia_x * X + inertia_y * Y, a_min=-1, a_max=1)
    diff_angle = (np.pi / 2.0 - np.abs(np.arccos(diff_cos))) / np.pi
    valid = np.ones(previous_obs.shape[0])
    valid[np.where(previous_obs[:, 4] < 0)] = 0
    iou = asso(dets, trks)
    if record is not None:
        record["iou"] = np.array(iou, dtype=np.float64)
    scores = np.repeat(dets[:, -1][:, np.newaxis], trks.shape[0], axis=1)
    valid = np.repeat(valid[:, np.newaxis], X.shape[1], axis=1)
    angle_cost = ((valid * diff_angle) * vdc_weight).T * scores
    if min(iou.shape):
        a = (iou > iou_threshold).astype(np.int32)
        if a.sum(1).max() == 1 and a.sum(0).max() == 1:
            matched = np.stack(np.where(a), axis=1)
        else:
            if emb_cost is None:
                emb_cost = 0
            else:
                emb_cost[iou <= 0] = 0
                if not aw_off:
                    emb_cost = aw_max_metric(emb_cost, w_assoc_emb, aw_param)
                else:
                    emb_cost *= w_assoc_emb
            final_cost = -(iou + angle_cost + emb_cost)
            if record is not None:
                record["final_cost"] = np.array(final_cost, dtype=np.float64)
            matched = _assign(final_cost)
            if matched.size == 0:
                matched = np.empty(shape=(0, 2))
    else:
        matched = np.empty(shape=(0, 2))
    un_d = [d for d in range(len(dets)) if d not in matched[:, 0]]
    un_t = [t for t in range(len(trks)) if t not in matched[:, 1]]
    matches = []
    for m in matched:
        if iou[m[0], m[1]] < iou_threshold:
            un_d.append(m[0])
            un_t.append(m[1])
        else:
            matches.append(m.reshape(1, 2))
    matches = np.concatenate(matches, axis=0) if matches else np.empty((0, 2), dtype=int)
    return matches, np.array(un_d), np.array(un_t)


class DeepOcSortOracle:
    use_byte, min_conf = False, 0.1        # OC-SORT's BYTE branch; only OcSortOracle switches it on

    def __init__(self, reid=None, **kw):
        cfg = dict(DEFAULTS)
        unknown = set(kw) - set(cfg)
        if unknown:
            raise TypeError(f"unknown DeepOCSORT options: {sorted(unknown)}")
        cfg.update(kw)
        if cfg["max_age"] >= cfg["max_obs"]:        # basetracker.py:93-97
            cfg["max_obs"] = cfg["max_age"] + 5
        self.cfg = cfg
        self.reid = reid
        self.frame_count = 0
        self.count = 1                              # KalmanBoxTracker.count = 1 (deepocsort.py:293)
        self.tracks = []
        self.asso = None

    def update(self, dets, img=None, embs=None, warp=None):
        """dets (N,6) [x1,y1,x2,y2,conf,cls] -> what ``DeepOcSort.update`` hands back: rows cast to fp32 by
        ``TrackResults`` (track_results.py:22-31), shape (M,8), or (0,0) when nothing is output.  ``warp``: the 2x3
        matrix ``cmc.apply`` returned (cmc_off=False, deepocsort.py:347-351), applied before the prediction."""
        c = self.cfg
        if self.asso is None:                       # basetracker.py:175-180: the first frame fixes w, h and the function
            wh = c["frame_wh"] if c["frame_wh"] is not None else ((img.shape[1], img.shape[0]) if img is not None else (None, None))
            self.asso = asso_function(c["asso_func"], *wh)
        asso = self.asso
        dets = np.asarray(dets)
        if dets.size == 0:
            dets = np.empty((0, 6), dtype=np.float32)
        self.frame_count += 1
        scores = dets[:, 4]
        dets = np.hstack([dets, np.arange(len(dets)).reshape(-1, 1)])
        keep = scores > c["det_thresh"]
        # OC-SORT's BYTE branch (ocsort.py:393-399): det_thresh > score > min_conf go to a second association
        dets_second = dets[np.logical_and(scores > self.min_conf, scores < c["det_thresh"])] if self.use_byte else dets[:0]
        dets = dets[keep]
        if c["embedding_off"] or dets.shape[0] == 0:
            dets_embs = np.ones((dets.shape[0], 1))
        elif embs is not None:
            dets_embs = embs[keep]
        else:
            dets_embs = self.reid.get_features(dets[:, 0:4], img)
        if warp is not None:
            for trk in self.tracks:
                trk.apply_affine_correction(warp)
        trust = (dets[:, 4] - c["det_thresh"]) / (1 - c["det_thresh"])
        af = c["alpha_fixed_emb"]
        dets_alpha = af + (1 - af) * (1 - trust)

        trks = np.zeros((len(self.tracks), 5))
        trk_embs, to_del = [], []
        for t, row in enumerate(trks):
            pos = self.tracks[t].predict()[0]
            row[:] = [pos[0], pos[1], pos[2], pos[3], 0]
            if np.any(np.isnan(pos)):
                to_del.append(t)
            else:
                trk_embs.append(self.tracks[t].emb)
        trks = np.ma.compress_rows(np.ma.masked_invalid(trks))
        trk_embs = np.vstack(trk_embs) if len(trk_embs) > 0 else np.array(trk_embs)
        for t in reversed(to_del):
            self.tracks.pop(t)
        velocities = np.array([t.velocity if t.velocity is not None else np.array((0, 0)) for t in self.tracks])
        last_boxes = np.array([t.last_observation for t in self.tracks])
        k_obs = np.array([_k_previous_obs(t.observations, t.age, c["delta_t"]) for t in self.tracks])

        if c["embedding_off"] or dets.shape[0] == 0 or trk_embs.shape[0] == 0:
            emb_cost = None
        else:
            emb_cost = dets_embs @ trk_embs.T
        matched, un_d, un_t = associate(dets[:, 0:5], trks, c["iou_threshold"], velocities, k_obs, c["inertia"],
                                        emb_cost, c["w_association_emb"], c["aw_off"], c["aw_param"], asso, record=(rec := {}))
        self.last = {"emb_cost": emb_cost, "matched": matched, **rec}
        for m in matched:
            self.tracks[m[1]].update(dets[m[0], :])
            self.tracks[m[1]].update_emb(dets_embs[m[0]], alpha=dets_alpha[m[0]])

        # OC-SORT only: BYTE association of the low-score detections with the predicted boxes of the unmatched tracks
        # (ocsort.py:456-485)
        if self.use_byte and len(dets_second) > 0 and un_t.shape[0] > 0:
            iou_left = np.array(asso(dets_second, trks[un_t]))
            if iou_left.max() > c["iou_threshold"]:
                rem_t = []
                for m in _assign(-iou_left):
                    di, ti = m[0], un_t[m[1]]
                    if iou_left[m[0], m[1]] < c["iou_threshold"]:
                        continue
                    self.tracks[ti].update(dets_second[di, :])
                    rem_t.append(ti)
                un_t = np.setdiff1d(un_t, np.array(rem_t))

        # second round: observation-centric recovery on the last observations (deepocsort.py:411-450)
        if un_d.shape[0] > 0 and un_t.shape[0] > 0:
            left_dets = dets[un_d]
            left_trks = last_boxes[un_t]
            iou_left = np.array(asso(left_dets, left_trks))
            if iou_left.max() > c["iou_threshold"]:
                rem_d, rem_t = [], []
                for m in _assign(-iou_left):
                    di, ti = un_d[m[0]], un_t[m[1]]
                    if iou_left[m[0], m[1]] < c["iou_threshold"]:
                        continue
                    self.tracks[ti].update(dets[di, :])
                    self.tracks[ti].update_emb(dets_embs[di], alpha=dets_alpha[di])
                    rem_d.append(di)
                    rem_t.append(ti)
                un_d = np.setdiff1d(un_d, np.array(rem_d))
                un_t = np.setdiff1d(un_t, np.array(rem_t))
        for m in un_t:
            self.tracks[m].update(None)
        for i in un_d:
            self.tracks.append(_Track(dets[i], self.count, c["delta_t"], dets_embs[i], c["Q_xy_scaling"], c["Q_s_scaling"]))
            self.count += 1

        ret = []
        i = len(self.tracks)
        for trk in reversed(self.tracks):
            d = _to_box(trk.kf.x)[0] if trk.last_observation.sum() < 0 else trk.last_observation[:4]
            if trk.time_since_update < 1 and (trk.hit_streak >= c["min_hits"] or self.frame_count <= c["min_hits"]):
                ret.append(np.concatenate((d, [trk.id], [trk.conf], [trk.cls], [trk.det_ind])).reshape(1, -1))
            i -= 1
            if trk.time_since_update > c["max_age"]:
                self.tracks.pop(i)
        raw = np.concatenate(ret) if len(ret) > 0 else np.array([])      # deepocsort.py:490-492
        out = np.asarray(raw, dtype=np.float32)
        return out if out.size else out.reshape(0, out.shape[1] if out.ndim == 2 else 0)

    def dump(self):
        t = self.tracks
        return {
            "id": np.array([k.id for k in t], dtype=np.int64),
            "x": np.array([k.kf.x[:, 0] for k in t], dtype=np.float64).reshape(len(t), 7),
            "P": np.array([k.kf.P for k in t], dtype=np.float64).reshape(len(t), 7, 7),
            "age": np.array([k.age for k in t], dtype=np.int64),
            "time_since_update": np.array([k.time_since_update for k in t], dtype=np.int64),
            "hit_streak": np.array([k.hit_streak for k in t], dtype=np.int64),
            "emb": [np.asarray(k.emb, dtype=np.float64) for k in t],
            "count": self.count,
        }


class PerClassDeepOcSortOracle:
    """The reference's per-class fan-out around DeepOcSort (basetracker.py:223-263): one track list per class, the
    frame counter rewound for every class, one shared id counter (KalmanBoxTracker.count)."""

    def __init__(self, nr_classes, **kw):
        self.per_class = [DeepOcSortOracle(**kw) for _ in range(nr_classes)]
        self.frame_count = 0
        self.count = 1

    def update(self, dets, img=None, embs=None):
        dets = np.asarray(dets)
        rows = []
        for c, orc in enumerate(self.per_class):
            idx = np.where(dets[:, 5] == c)[0] if dets.size else np.zeros(0, int)
            orc.frame_count, orc.count = self.frame_count, self.count
            out = orc.update(dets[idx] if dets.size else dets, img, None if embs is None else embs[idx])
            self.count = orc.count
            if out.size > 0:
                rows.append(out)
        self.frame_count += 1
        return np.vstack(rows) if rows else np.empty((0, 8), dtype=np.float32)


class OcSortOracle(DeepOcSortOracle):
    """OC-SORT (boxmot/trackers/bbox/ocsort/ocsort.py:334-555): the reference's ``OcSort`` and its ``DeepOcSort`` with
    ``embedding_off=True, cmc_off=True`` produce identical rows (pinned on the reference classes:
    tests/test_oracle_vs_reference.py, tests/golden/mot17_golden.npz), so the restatement is the DeepOCSORT one with
    those terms off, plus OC-SORT's optional BYTE association (``use_byte=True``, ocsort.py:393-399, 456-485)."""

    def __init__(self, min_conf=0.1, use_byte=False, **kw):
        super().__init__(embedding_off=True, **kw)
        self.min_conf, self.use_byte = min_conf, bool(use_byte)

    def update(self, dets, img=None, embs=None, warp=None):
        return super().update(dets, img, None)
