"""OC-SORT fed ORIENTED detections (7 columns: cx, cy, w, h, angle, conf, cls) -- TEST INFRASTRUCTURE ONLY.

Follows (statement order and NumPy calls kept, so results are bit-identical on one host):
  * boxmot/trackers/bbox/ocsort/ocsort.py:18-31 (k_previous_obs), :49-72 (convert_obb_to_z / convert_x_to_obb), :82-87
    (speed_direction_obb), :91-310 (KalmanBoxTracker with is_obb=True), :363-555 (OcSort._update_impl);
  * boxmot/motion/kalman_filters/xysr.py:18-162 (KalmanFilterXYSR(dim_x=9, dim_z=5): motion matrix, measurement preparation with the
    alignment of the (s, r, theta) parameterisation to the state, state constraints), :368-476 (predict / freeze / unfreeze with the
    interpolated angle / update with the theta-velocity damping) over base.py:116-232 (angle wrap, candidate selection, damping) and
    :366-459 (predict_state / update_state, Joseph form);
  * boxmot/trackers/association/association.py:8-152 (associate -- the restatement of oracle/deepocsort.py is reused: the reference
    runs the SAME column arithmetic on oriented rows, i.e. its "centres" are (cx + w) / 2, (cy + h) / 2 and its "valid previous
    observation" test reads the angle column; reproduced, not corrected) with iou.py:38-115 (iou_obb) or iou.py:263-274
    (centroid_obb) as the association function -- the device step has the first only.
Pinned against the reference class itself (tests/test_oracle_obb.py; fixture tests/golden/obb_golden.npz key "ocsort").  The
rotated-intersection AREA is oracle/obb.py's (the reference: cv2.rotatedRectangleIntersection + contourArea, OpenCV absent offline:
PARITY UNPINNED for that one quantity); ``lap.lapjv`` is the oracle stand-in (oracle/lap.py).  The display-only corner history
(KalmanBoxTracker._state_obb_for_plot, ocsort.py:217-239) is not restated.
"""
from __future__ import annotations

from collections import deque
from copy import deepcopy

import numpy as np
import scipy.linalg

from oracle.deepocsort import _assign, _safe_cho_factor, associate
from oracle.obb import iou_obb_matrix, wrap_angle

DEFAULTS = dict(
    det_thresh=0.3, max_age=30, max_obs=50, min_hits=3, iou_threshold=0.3,           # basetracker.py:19-31
    min_conf=0.1, delta_t=3, inertia=0.2, use_byte=False, Q_xy_scaling=0.01, Q_s_scaling=0.0001,      # ocsort.py:334-344
    asso_func="iou",        # basetracker.py:28 -> "iou_obb" / "centroid_obb" for oriented detections (detection_layout.py:25-26)
    frame_wh=None,          # (w, h) for centroid_obb; None: read off the first image (basetracker.py:175-180)
)

_F = np.eye(9)                                      # xysr.py:54-66: [x, y, s, r, theta, vx, vy, vs, vtheta]
_F[0, 5] = _F[1, 6] = _F[2, 7] = _F[4, 8] = 1.0
_H = np.zeros((5, 9))
_H[:5, :5] = np.eye(5)


def obb_to_z(box):                                  # ocsort.py:49-59
    cx, cy, w, h, theta = np.asarray(box, dtype=float).reshape(-1)
    w = max(float(w), 1e-6)
    h = max(float(h), 1e-6)
    return np.array([cx, cy, w * h, w / h, theta], dtype=float).reshape((5, 1))


def x_to_obb(x):                                    # ocsort.py:62-72 (score=None)
    x = np.asarray(x, dtype=float).reshape(-1)
    w = np.sqrt(max(float(x[2] * x[3]), 1e-12))
    h = float(x[2]) / max(w, 1e-6)
    return np.array([x[0], x[1], w, h, x[4]], dtype=float).reshape((1, 5))


def iou_obb(a, b):                                  # AssociationFunction.iou_batch_obb, iou.py:152-154
    return iou_obb_matrix(np.asarray(a, dtype=float)[:, :5], np.asarray(b, dtype=float)[:, :5])


def centroid_obb(a, b, w, h):                       # AssociationFunction.centroid_batch_obb, iou.py:263-274
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    c1 = np.expand_dims(np.stack((a[..., 0], a[..., 1]), axis=-1), 1)
    c2 = np.expand_dims(np.stack((b[..., 0], b[..., 1]), axis=-1), 0)
    return 1 - np.sqrt(np.sum((c1 - c2) ** 2, axis=-1)) / np.sqrt(w ** 2 + h ** 2)


def align_xysr_measurement(m, ref):                 # xysr.py:96-136 over base.py:131-157
    out = np.asarray(m, dtype=float).copy().reshape((-1,))
    ref = np.asarray(ref, dtype=float).reshape((-1,))
    ref_r, ref_t = max(float(ref[3]), 1e-6), float(ref[4])
    s, r, t = max(float(out[2]), 1e-6), max(float(out[3]), 1e-6), float(out[4])
    best_cost, best = float("inf"), None
    ref_s0 = max(1.0, 1e-6)
    for c0, c1, th in ((1.0, r, t), (1.0, r, t + np.pi), (1.0, 1.0 / r, t + (np.pi / 2.0)), (1.0, 1.0 / r, t - (np.pi / 2.0))):
        s0, s1 = max(float(c0), 1e-6), max(float(c1), 1e-6)
        ta = float(ref_t + wrap_angle(float(th) - float(ref_t)))
        cost = abs(ta - ref_t) + (0.05 * (abs(np.log(s0 / ref_s0)) + abs(np.log(s1 / ref_r))))
        if cost < best_cost:
            best_cost, best = cost, (s0, s1, ta)
    out[2] = max(s, 1e-6)
    out[3] = max(best[1], 1e-6)
    out[4] = best[2]
    return out


class FilterXYSRTheta:
    """KalmanFilterXYSR(dim_x=9, dim_z=5) as KalmanBoxTracker configures it (ocsort.py:121-154)."""

    def __init__(self, z0, q_xy, q_s, q_a):
        self.x = np.zeros((9, 1))
        self.P = np.eye(9)
        self.Q = np.eye(9)
        self.R = np.eye(5)
        self.R[2:, 2:] *= 10.0
        self.P[5:, 5:] *= 1000.0
        self.P *= 10.0
        self.Q[5:7, 5:7] *= q_xy
        self.Q[7, 7] *= q_s
        self.Q[8, 8] *= q_a
        self.x[:5] = z0
        self.history = deque([], maxlen=50)        # KalmanFilterXYSR's own max_obs default is overridden by the tracker's (below)
        self.saved = None
        self.observed = False

    def _constrain(self):                          # xysr.py:154-161 over base.py:160-180 (2-d state)
        self.x[2, :] = np.maximum(self.x[2, :], 1e-6)
        self.x[3, :] = np.maximum(self.x[3, :], 1e-6)
        self.x[4, :] = wrap_angle(self.x[4, :])
        self.P = 0.5 * (self.P + self.P.T)

    def predict(self):                             # xysr.py:368-377, base.py:366-391
        self.x = np.dot(_F, self.x)
        self.P = 1.0 * np.dot(np.dot(_F, self.P), _F.T) + self.Q
        self._constrain()

    def _prepare(self, z):                         # xysr.py:138-152
        m = np.asarray(z, dtype=float)
        if m.shape != (5, 1):
            m = m.reshape((5, 1))
        m[2, 0] = max(float(m[2, 0]), 1e-6)
        m[3, 0] = max(float(m[3, 0]), 1e-6)
        m[4, 0] = float(wrap_angle(m[4, 0]))
        m[:, 0] = align_xysr_measurement(m[:, 0], np.asarray(self.x[:5, 0], dtype=float).copy())
        return m

    def update(self, z):                           # xysr.py:440-476
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
        self.x[8, :] *= float(np.clip(0.8, 0.0, 1.0))        # _damp_theta_velocity, base.py:222-232
        self._constrain()
        self.history.append(m.copy())              # observed measurements are stored twice

    def _update_state(self, m):                    # base.py:414-459
        S = np.dot(np.dot(_H, self.P), _H.T) + self.R
        S = 0.5 * (S + S.T)
        cf = _safe_cho_factor(S)
        K = scipy.linalg.cho_solve(cf, np.dot(self.P, _H.T).T, check_finite=False).T
        y = m - np.dot(_H, self.x)
        self.x = self.x + np.dot(K, y)
        ikh = np.eye(9) - np.dot(K, _H)
        self.P = np.linalg.multi_dot((ikh, self.P, ikh.T)) + np.linalg.multi_dot((K, self.R, K.T))
        self.P = 0.5 * (self.P + self.P.T)

    def _unfreeze(self):                           # xysr.py:383-438
        if self.saved is None:
            return
        new_history = deepcopy(list(self.history))
        self.__dict__ = self.saved
        self.history = deque(list(self.history)[:-1], maxlen=self.history.maxlen)
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
        t1, t2 = b1[4], b2[4]
        dtheta = float(wrap_angle(t2 - t1)) / gap
        for i in range(gap):
            x = x1 + (i + 1) * dx
            y = y1 + (i + 1) * dy
            w = w1 + (i + 1) * dw
            h = h1 + (i + 1) * dh
            theta = float(wrap_angle(t1 + (i + 1) * dtheta))
            self.update(np.array([x, y, w * h, w / float(h), theta], dtype=float).reshape((5, 1)))
            if i != gap - 1:
                self.predict()
                self.history.pop()
        self.history.pop()


class _Track:
    """KalmanBoxTracker(is_obb=True) (ocsort.py:91-310)."""

    def __init__(self, box6, cls, det_ind, tid, delta_t, q_xy, q_s, q_a, max_obs):
        self.det_ind = det_ind
        self.kf = FilterXYSRTheta(obb_to_z(box6[:5]), q_xy, q_s, q_a)
        self.kf.history = deque([], maxlen=max_obs)
        self.id = tid
        self.time_since_update = 0
        self.hits = self.hit_streak = self.age = 0
        self.conf = box6[-1]
        self.cls = cls
        self.last_observation = np.array([-1, -1, -1, -1, -1, -1])
        self.observations = {}
        self.velocity = None
        self.delta_t = delta_t

    def update(self, box6, cls, det_ind):           # ocsort.py:241-278
        self.det_ind = det_ind
        if box6 is None:
            self.kf.update(None)
            return
        self.conf = box6[-1]
        self.cls = cls
        if self.last_observation.sum() >= 0:
            prev = None
            for dt in range(self.delta_t, 0, -1):
                if self.age - dt in self.observations:
                    prev = self.observations[self.age - dt]
                    break
            if prev is None:
                prev = self.last_observation
            cx1, cy1 = prev[0], prev[1]             # speed_direction_obb, ocsort.py:82-87
            cx2, cy2 = box6[0], box6[1]
            speed = np.array([cy2 - cy1, cx2 - cx1])
            self.velocity = speed / (np.sqrt((cy2 - cy1) ** 2 + (cx2 - cx1) ** 2) + 1e-6)
        self.last_observation = box6
        self.observations[self.age] = box6
        self.time_since_update = 0
        self.hits += 1
        self.hit_streak += 1
        self.kf.update(obb_to_z(box6[:5]))

    def predict(self):                              # ocsort.py:280-299
        if (self.kf.x[7] + self.kf.x[2]) <= 0:
            self.kf.x[7] *= 0.0
        self.kf.predict()
        self.age += 1
        if self.time_since_update > 0:
            self.hit_streak = 0
        self.time_since_update += 1
        return x_to_obb(self.kf.x)


def _k_previous_obs(observations, cur_age, k):      # ocsort.py:18-31 (is_obb=True)
    if len(observations) == 0:
        return [-1, -1, -1, -1, -1, -1]
    for i in range(k):
        if cur_age - (k - i) in observations:
            return observations[cur_age - (k - i)]
    return observations[max(observations.keys())]


class OcSortObbOracle:
    def __init__(self, **kw):
        cfg = dict(DEFAULTS)
        unknown = set(kw) - set(cfg)
        if unknown:
            raise TypeError(f"unknown OC-SORT options: {sorted(unknown)}")
        cfg.update(kw)
        if cfg["max_age"] >= cfg["max_obs"]:        # basetracker.py:93-97
            cfg["max_obs"] = cfg["max_age"] + 5
        self.cfg = cfg
        self.frame_count = 0
        self.count = 0                              # KalmanBoxTracker.count = 0 (ocsort.py:358); rows carry id + 1
        self.tracks = []
        self.asso = None

    def update(self, dets, img=None, embs=None):
        """dets (N, 7) [cx, cy, w, h, angle, conf, cls] -> the fp32 rows ``OcSort.update`` hands back, (M, 9)
        [cx, cy, w, h, angle, id, conf, cls, det_ind], or (0, 0) when nothing is output."""
        c = self.cfg
        if self.asso is None:                       # basetracker.py:175-180: the first frame fixes w, h and the function
            if c["asso_func"] == "iou":
                self.asso = iou_obb
            elif c["asso_func"] == "centroid":
                fw, fh = c["frame_wh"] if c["frame_wh"] is not None else (img.shape[1], img.shape[0])
                self.asso = lambda a, b: centroid_obb(a, b, fw, fh)
            else:                                   # AssociationFunction._get_asso_func, iou.py:418-422
                raise ValueError(f"Invalid association mode: {c['asso_func']}_obb")
        asso = self.asso
        dets = np.asarray(dets)
        if dets.size == 0:
            dets = np.empty((0, 7), dtype=np.float32)
        self.frame_count += 1
        dets = np.hstack([dets, np.arange(len(dets), dtype=np.int32).reshape(-1, 1)]) if dets.size else np.empty((0, 8), dtype=dets.dtype)
        confs = dets[:, 5]
        dets_second = dets[np.logical_and(confs > c["min_conf"], confs < c["det_thresh"])]
        dets = dets[confs > c["det_thresh"]]

        trks = np.zeros((len(self.tracks), 6))
        to_del = []
        for t, row in enumerate(trks):
            pos = self.tracks[t].predict()[0]
            row[:] = [pos[i] for i in range(5)] + [0]
            if np.any(np.isnan(pos)):
                to_del.append(t)
        trks = np.ma.compress_rows(np.ma.masked_invalid(trks))
        for t in reversed(to_del):
            self.tracks.pop(t)
        velocities = np.array([t.velocity if t.velocity is not None else np.array((0, 0)) for t in self.tracks])
        last_boxes = np.array([t.last_observation for t in self.tracks])
        k_obs = np.array([_k_previous_obs(t.observations, t.age, c["delta_t"]) for t in self.tracks])

        matched, un_d, un_t = associate(dets[:, 0:6], trks, c["iou_threshold"], velocities, k_obs, c["inertia"],
                                        None, None, None, None, asso)
        for m in matched:
            self.tracks[m[1]].update(dets[m[0], :-2], dets[m[0], -2], dets[m[0], -1])

        if c["use_byte"] and len(dets_second) > 0 and un_t.shape[0] > 0:          # ocsort.py:456-485
            iou_left = np.array(asso(dets_second, trks[un_t]))
            if iou_left.max() > c["iou_threshold"]:
                rem_t = []
                for m in _assign(-iou_left):
                    di, ti = m[0], un_t[m[1]]
                    if iou_left[m[0], m[1]] < c["iou_threshold"]:
                        continue
                    self.tracks[ti].update(dets_second[di, :-2], dets_second[di, -2], dets_second[di, -1])
                    rem_t.append(ti)
                un_t = np.setdiff1d(un_t, np.array(rem_t))

        if un_d.shape[0] > 0 and un_t.shape[0] > 0:                               # ocsort.py:487-517
            iou_left = np.array(asso(dets[un_d], last_boxes[un_t]))
            if iou_left.max() > c["iou_threshold"]:
                rem_d, rem_t = [], []
                for m in _assign(-iou_left):
                    di, ti = un_d[m[0]], un_t[m[1]]
                    if iou_left[m[0], m[1]] < c["iou_threshold"]:
                        continue
                    self.tracks[ti].update(dets[di, :-2], dets[di, -2], dets[di, -1])
                    rem_d.append(di)
                    rem_t.append(ti)
                un_d = np.setdiff1d(un_d, np.array(rem_d))
                un_t = np.setdiff1d(un_t, np.array(rem_t))
        for m in un_t:
            self.tracks[m].update(None, None, None)
        for i in un_d:
            self.tracks.append(_Track(dets[i, :6], dets[i, 6], dets[i, 7], self.count, c["delta_t"], c["Q_xy_scaling"], c["Q_s_scaling"],
                                      c["Q_s_scaling"], c["max_obs"]))           # Q_a_scaling = Q_s_scaling (ocsort.py:530)
            self.count += 1

        ret = []
        i = len(self.tracks)
        for trk in reversed(self.tracks):
            d = x_to_obb(trk.kf.x)[0] if trk.last_observation.sum() < 0 else trk.last_observation[:5]
            if trk.time_since_update < 1 and (trk.hit_streak >= c["min_hits"] or self.frame_count <= c["min_hits"]):
                ret.append(np.concatenate((d, [trk.id + 1], [trk.conf], [trk.cls], [trk.det_ind])).reshape(1, -1))
            i -= 1
            if trk.time_since_update > c["max_age"]:
                self.tracks.pop(i)
        raw = np.concatenate(ret) if len(ret) > 0 else np.array([])
        out = np.asarray(raw, dtype=np.float32)
        return out if out.size else out.reshape(0, out.shape[1] if out.ndim == 2 else 0)

    def dump(self):
        t = self.tracks
        return {
            "id": np.array([k.id + 1 for k in t], dtype=np.int64),
            "x": np.array([k.kf.x[:, 0] for k in t], dtype=np.float64).reshape(len(t), 9),
            "P": np.array([k.kf.P for k in t], dtype=np.float64).reshape(len(t), 9, 9),
            "age": np.array([k.age for k in t], dtype=np.int64),
            "time_since_update": np.array([k.time_since_update for k in t], dtype=np.int64),
            "hit_streak": np.array([k.hit_streak for k in t], dtype=np.int64),
            "count": self.count,
        }
