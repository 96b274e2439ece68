"""Oracle ByteTrack frame step (AABB) -- TEST INFRASTRUCTURE ONLY.

Same slot-table shape as ``oracle/botsort.py`` (BoT-SORT grew out of this tracker, so the stages line up one to one).
Follows:
  * ByteTrack._update_impl                         boxmot/trackers/bbox/bytetrack/bytetrack.py:258-408
  * STrack (fp32 detection wrapper, predict / activate / re_activate / update, xyxy)   bytetrack.py:15-198
  * joint / sub / remove_duplicate_stracks         bytetrack.py:414-447
  * KalmanFilterXYAH over BaseKalmanFilter         boxmot/motion/kalman_filters/xyah.py:16-148, base.py:234-355
  * iou_distance / fuse_score / linear_assignment  boxmot/trackers/association/matching.py
  * xyxy2xywh / xywh2tlwh / tlwh2xyah              boxmot/trackers/common/geometry.py:10-100
Pinned bit-for-bit (rows and fp64 filter state) against the reference class: tests/test_oracle_vs_reference.py,
tests/golden/mot17_golden.npz.  ``lap.lapjv`` is the restated solver of oracle/lap.py (parity unpinned, see there).

Deliberate difference: the track-id counter is per tracker instance; the reference's ``BaseTrack._count``
(bytetrack/basetrack.py:16,37-40) is process-global and never reset by the constructor, so a second tracker in the same
process continues the first one's ids.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

from oracle import matching
from oracle.botsort import LOST, NEW, REMOVED, TRACKED, _boxes, _join, _minus

DEFAULTS = dict(min_conf=0.1, track_thresh=0.45, match_thresh=0.8, track_buffer=25, frame_rate=30)   # bytetrack.py:225-233

STD_POS, STD_VEL = 1.0 / 20, 1.0 / 160       # base.py:60-65
F = np.eye(8)
for _i in range(4):
    F[_i, 4 + _i] = 1.0
H = np.eye(4, 8)


def kf_initiate(xyah):                        # xyah.py:99-105 over base.py:234-244, std xyah.py:22-37
    m = np.asarray(xyah, dtype=float).copy()
    mean = np.r_[m, np.zeros_like(m)]
    std = [2 * STD_POS * m[3], 2 * STD_POS * m[3], 1e-2, 2 * STD_POS * m[3],
           10 * STD_VEL * m[3], 10 * STD_VEL * m[3], 1e-5, 10 * STD_VEL * m[3]]
    cov = np.diag(np.square(std))
    mean[2] = max(float(mean[2]), 1e-4)
    mean[3] = max(float(mean[3]), 1e-4)
    return mean, cov


def kf_multi_predict(mean, cov):              # xyah.py:112-120 over base.py:311-327, std xyah.py:70-89
    std_pos = [STD_POS * mean[:, 3], STD_POS * mean[:, 3], 1e-2 * np.ones_like(mean[:, 3]), STD_POS * mean[:, 3]]
    std_vel = [STD_VEL * mean[:, 3], STD_VEL * mean[:, 3], 1e-5 * np.ones_like(mean[:, 3]), STD_VEL * mean[:, 3]]
    sqr = np.square(np.r_[std_pos, std_vel]).T
    motion_cov = np.asarray([np.diag(sqr[i]) for i in range(len(mean))])
    mean = np.dot(mean, F.T)
    left = np.dot(F, cov).transpose((1, 0, 2))
    cov = np.dot(left, F.T) + motion_cov
    mean[:, 2] = np.maximum(mean[:, 2], 1e-4)
    mean[:, 3] = np.maximum(mean[:, 3], 1e-4)
    return mean, cov


def kf_update(mean, cov, xyah):               # xyah.py:122-148 over base.py:286-355, std xyah.py:57-68
    std = [STD_POS * mean[3], STD_POS * mean[3], 1e-1, STD_POS * mean[3]]
    std = [(1 - 0.0) * x for x in std]
    innovation_cov = np.diag(np.square(std))
    projected_mean = np.dot(H, mean)
    projected_cov = np.linalg.multi_dot((H, cov, H.T)) + innovation_cov
    chol, lower = scipy.linalg.cho_factor(projected_cov, lower=True, check_finite=False)
    gain = scipy.linalg.cho_solve((chol, lower), np.dot(cov, H.T).T, check_finite=False).T
    innovation = xyah - projected_mean
    new_mean = mean + np.dot(innovation, gain.T)
    new_cov = cov - np.linalg.multi_dot((gain, projected_cov, gain.T))
    new_mean[2] = max(float(new_mean[2]), 1e-4)
    new_mean[3] = max(float(new_mean[3]), 1e-4)
    return new_mean, new_cov


class _Rec:
    __slots__ = ("xywh", "xyah", "conf", "cls", "det_ind", "mean", "cov", "is_activated", "tracklet_len", "state", "id",
                 "frame_id", "start_frame")

    def __init__(self, det_row):
        det = np.asarray(det_row, dtype=np.float32)                  # bytetrack.py:21
        self.xywh = matching.xyxy2xywh32(det[:4])                    # :33
        tlwh = self.xywh.copy()                                       # xywh2tlwh, geometry.py:56-60 (fp32)
        tlwh[0] = self.xywh[0] - self.xywh[2] / 2.0
        tlwh[1] = self.xywh[1] - self.xywh[3] / 2.0
        self.xyah = tlwh.copy()                                       # tlwh2xyah, geometry.py:95-99 (fp32)
        self.xyah[0] = tlwh[0] + (tlwh[2] / 2)
        self.xyah[1] = tlwh[1] + (tlwh[3] / 2)
        self.xyah[2] = tlwh[2] / tlwh[3]
        self.conf, self.cls, self.det_ind = det[4], det[5], det[6]
        self.mean = self.cov = None
        self.is_activated = False
        self.tracklet_len = 0
        self.state = NEW
        self.id = 0
        self.frame_id = self.start_frame = 0

    @property
    def xyxy(self):                                                   # bytetrack.py:176-189
        if self.mean is None:
            return matching.xywh2xyxy(self.xywh.copy())
        ret = self.mean[:4].copy()
        ret[2] *= ret[3]
        return matching.xywh2xyxy(ret)

    def absorb(self, det, frame_id, reactivate):                      # update :118-141 / re_activate :100-116
        if reactivate:
            self.tracklet_len = 0
        else:
            self.tracklet_len += 1
        self.frame_id = frame_id
        self.mean, self.cov = kf_update(self.mean, self.cov, det.xyah)
        self.state = TRACKED
        self.is_activated = True
        self.conf, self.cls, self.det_ind = det.conf, det.cls, det.det_ind


class ByteTrackOracle:
    def __init__(self, **kw):
        cfg = dict(DEFAULTS)
        unknown = set(kw) - set(cfg)
        if unknown:
            raise TypeError(f"unknown ByteTrack options: {sorted(unknown)}")
        cfg.update(kw)
        self.cfg = cfg
        self.max_time_lost = int(cfg["frame_rate"] / 30.0 * cfg["track_buffer"])     # bytetrack.py:243-244
        self.frame_count = 0
        self.id_count = 0
        self.active, self.lost = [], []
        self.removed_ids = []              # the reference's removed list is unbounded (bytetrack.py:393)

    def _next_id(self):
        self.id_count += 1
        return self.id_count

    def update(self, dets, img=None, embs=None):
        """dets (N,6) [x1,y1,x2,y2,conf,cls] -> (M,8) fp32 rows; ``embs`` is accepted and ignored like the reference does."""
        c = self.cfg
        dets = np.asarray(dets)
        if dets.size == 0:
            dets = np.empty((0, 6), dtype=np.float32)
        self.frame_count += 1
        fc = self.frame_count
        activated, refound, newly_lost, newly_removed = [], [], [], []
        if len(dets):
            table = np.hstack([dets, np.arange(len(dets), dtype=np.int32).reshape(-1, 1)])    # detection index column
        else:
            table = np.empty((0, 7), dtype=dets.dtype)
        confs = table[:, 4]
        remain = confs > c["track_thresh"]                                                      # :271-278
        second = np.logical_and(confs > c["min_conf"], confs < c["track_thresh"])
        dets_lo, dets_hi = table[second], table[remain]
        cand = [_Rec(d) for d in dets_hi]

        unconfirmed = [t for t in self.active if not t.is_activated]
        tracked = [t for t in self.active if t.is_activated]
        pool = _join(tracked, self.lost)
        if pool:                                                                                # STrack.multi_predict :63-82
            mean = np.asarray([t.mean.copy() for t in pool])
            cov = np.asarray([t.cov for t in pool])
            for i, t in enumerate(pool):
                if t.state != TRACKED:
                    mean[i][7] = 0
            mean, cov = kf_multi_predict(mean, cov)
            for t, m, p in zip(pool, mean, cov):
                t.mean, t.cov = m, p
        d1 = matching.iou_distance(_boxes(pool), _boxes(cand))
        d1 = matching.fuse_score(d1, np.array([d.conf for d in cand]))
        m1, u_trk1, u_det1 = matching.linear_assignment(d1, c["match_thresh"])
        for it, idet in m1:
            t = pool[it]
            if t.state == TRACKED:
                t.absorb(cand[idet], fc, reactivate=False)
                activated.append(t)
            else:
                t.absorb(cand[idet], fc, reactivate=True)
                refound.append(t)

        cand_lo = [_Rec(d) for d in dets_lo]                                                    # second association :320-351
        r_tracked = [pool[i] for i in u_trk1 if pool[i].state == TRACKED]
        d2 = matching.iou_distance(_boxes(r_tracked), _boxes(cand_lo))
        m2, u_trk2, _ = matching.linear_assignment(d2, 0.5)
        for it, idet in m2:
            t = r_tracked[it]
            if t.state == TRACKED:
                t.absorb(cand_lo[idet], fc, reactivate=False)
                activated.append(t)
            else:
                t.absorb(cand_lo[idet], fc, reactivate=True)
                refound.append(t)
        for it in u_trk2:
            t = r_tracked[it]
            if t.state != LOST:
                t.state = LOST
                newly_lost.append(t)

        left = [cand[i] for i in u_det1]                                                        # unconfirmed :353-365
        d3 = matching.iou_distance(_boxes(unconfirmed), _boxes(left))
        d3 = matching.fuse_score(d3, np.array([d.conf for d in left]))
        m3, u_unc, u_det3 = matching.linear_assignment(d3, 0.7)
        for it, idet in m3:
            unconfirmed[it].absorb(left[idet], fc, reactivate=False)
            activated.append(unconfirmed[it])
        for it in u_unc:
            unconfirmed[it].state = REMOVED
            newly_removed.append(unconfirmed[it])

        for inew in u_det3:                                                                     # births :367-373, activate :84-98
            d = left[inew]
            if d.conf < c["track_thresh"]:
                continue
            d.id = self._next_id()
            d.mean, d.cov = kf_initiate(d.xyah)
            d.tracklet_len = 0
            d.state = TRACKED
            if fc == 1:
                d.is_activated = True
            d.frame_id = d.start_frame = fc
            activated.append(d)

        for t in self.lost:                                                                     # :375-378
            if fc - t.frame_id > self.max_time_lost:
                t.state = REMOVED
                newly_removed.append(t)

        self.active = [t for t in self.active if t.state == TRACKED]                            # :380-393
        self.active = _join(self.active, activated)
        self.active = _join(self.active, refound)
        self.lost = _minus(self.lost, [t.id for t in self.active])
        self.lost.extend(newly_lost)
        self.lost = _minus(self.lost, list(self.removed_ids))
        self.removed_ids.extend(t.id for t in newly_removed)
        self.active, self.lost = self._dedup(self.active, self.lost)
        rows = [[*t.xyxy, t.id, t.conf, t.cls, t.det_ind] for t in self.active if t.is_activated]
        return np.asarray(rows, dtype=np.float32) if rows else np.empty((0, 8), dtype=np.float32)

    @staticmethod
    def _dedup(a, b):                                                                           # :431-447
        pd = matching.iou_distance(_boxes(a), _boxes(b))
        drop_a, drop_b = [], []
        for p, q in zip(*np.where(pd < 0.15)):
            tp = a[p].frame_id - a[p].start_frame
            tq = b[q].frame_id - b[q].start_frame
            if tp > tq:
                drop_b.append(q)
            else:
                drop_a.append(p)
        return ([t for i, t in enumerate(a) if i not in drop_a], [t for i, t in enumerate(b) if i not in drop_b])

    def dump(self):
        def pack(recs):
            return dict(
                id=np.array([t.id for t in recs], dtype=np.int64), state=np.array([t.state for t in recs], dtype=np.int64),
                is_activated=np.array([t.is_activated for t in recs], dtype=bool),
                frame_id=np.array([t.frame_id for t in recs], dtype=np.int64),
                start_frame=np.array([t.start_frame for t in recs], dtype=np.int64),
                tracklet_len=np.array([t.tracklet_len for t in recs], dtype=np.int64),
                mean=np.array([t.mean for t in recs], dtype=np.float64).reshape(len(recs), 8),
                cov=np.array([t.cov for t in recs], dtype=np.float64).reshape(len(recs), 8, 8), smooth=None,
                conf=np.array([t.conf for t in recs], dtype=np.float32), cls=np.array([t.cls for t in recs], dtype=np.float32),
                det_ind=np.array([t.det_ind for t in recs], dtype=np.float32))
        return dict(frame_count=self.frame_count, id_count=self.id_count, active=pack(self.active), lost=pack(self.lost),
                    removed_ids=np.array(list(self.removed_ids), dtype=np.int64))


class PerClassByteTrackOracle:
    """``ByteTrack(per_class=True)``: BaseTracker._do_update (basetracker.py:213-271) runs ``_update_impl`` once per class id with
    that class's active list swapped in; the lost list, the removed list, the id counter and the filter are shared, and the frame
    counter is rewound for every class."""

    def __init__(self, nr_classes: int, **kw):
        self.o = ByteTrackOracle(**kw)
        self.n = int(nr_classes)
        self.lists = {c: [] for c in range(self.n)}

    def update(self, dets, img=None, embs=None):
        dets = np.asarray(dets, dtype=np.float32).reshape(-1, 6)
        rows, fc = [], self.o.frame_count
        for c in range(self.n):
            idx = np.where(dets[:, 5] == c)[0]
            self.o.active = self.lists[c]
            self.o.frame_count = fc
            r = np.asarray(self.o.update(dets[idx], img)).reshape(-1, 8)
            # det_ind of a per-class call indexes the class's detections (basetracker.py get_class_dets_n_embs keeps the rows' order)
            self.lists[c] = self.o.active
            if r.size:
                rows.append(r)
        self.o.frame_count = fc + 1
        return np.vstack(rows) if rows else np.empty((0, 8), dtype=np.float32)
