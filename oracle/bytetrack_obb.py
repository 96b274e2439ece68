"""Oracle ByteTrack frame step for ORIENTED detections -- TEST INFRASTRUCTURE ONLY, and so far ORACLE ONLY (no device step takes
7-column detections yet; see oracle/obb.py's header).

The reference runs the same stage sequence as for axis-aligned boxes (boxmot/trackers/bbox/bytetrack/bytetrack.py:258-408) with
  * detections (cx, cy, w, h, angle, conf, cls) + the appended detection index (trackers/common/detection_layout.py:74-84),
  * KalmanFilterXYWH(ndim=5) on the 5-vector itself (bytetrack.py:266, 84-90; zeroed (vw, vh, vtheta) for non-tracked tracks :55-58),
  * iou_distance(..., is_obb=True): rotated IoU of the fp32 `xywha` of the tracks (matching.py:61-76, bytetrack.py:191-198),
  * output rows (cx, cy, w, h, angle, id, conf, cls, det_ind) (bytetrack.py:397).
Pinned on the reference class (rows + fp64 filter state) with oracle/obb.py's intersection area standing where
cv2.rotatedRectangleIntersection / cv2.contourArea stand: tests/test_oracle_obb.py.
"""
from __future__ import annotations

import numpy as np

from oracle import matching, obb
from oracle.botsort import LOST, NEW, REMOVED, TRACKED, _join, _minus
from oracle.bytetrack import ByteTrackOracle


class _RecObb:
    __slots__ = ("xywh", "conf", "cls", "det_ind", "mean", "cov", "is_activated", "tracklet_len", "state", "id", "frame_id", "start_frame")

    def __init__(self, det_row):
        det = np.asarray(det_row, dtype=np.float32)                  # bytetrack.py:24, 45-54
        self.xywh = det[:5].copy()
        self.conf, self.cls, self.det_ind = det[5], det[6], det[7]
        self.mean = self.cov = None
        self.is_activated = False
        self.tracklet_len = 0
        self.state = NEW
        self.id = 0
        self.frame_id = self.start_frame = 0

    @property
    def xywha(self):                                                  # bytetrack.py:191-198
        ret = self.mean[:5].copy() if self.mean is not None else self.xywh.copy()
        return np.asarray(ret, dtype=np.float32)

    def absorb(self, det, frame_id, reactivate):                      # update :118-141 / re_activate :100-116
        if reactivate:
            self.tracklet_len = 0
        else:
            self.tracklet_len += 1
        self.frame_id = frame_id
        self.mean, self.cov = obb.kf5_update(self.mean, self.cov, det.xywh)
        self.state = TRACKED
        self.is_activated = True
        self.conf, self.cls, self.det_ind = det.conf, det.cls, det.det_ind


def _iou_distance_obb(a, b):                                          # matching.py:46-80 with is_obb
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)), dtype=np.float32)
    return 1 - obb.iou_obb_matrix(np.asarray([t.xywha for t in a], dtype=float), np.asarray([t.xywha for t in b], dtype=float))


class ByteTrackObbOracle(ByteTrackOracle):
    def update(self, dets, img=None, embs=None):
        """dets (N,7) [cx,cy,w,h,angle,conf,cls] -> (M,9) fp32 rows [cx,cy,w,h,angle,id,conf,cls,det_ind]."""
        c = self.cfg
        dets = np.asarray(dets)
        if dets.size == 0:
            dets = np.empty((0, 7), dtype=np.float32)
        self.frame_count += 1
        fc = self.frame_count
        activated, refound, newly_lost, newly_removed = [], [], [], []
        table = np.hstack([dets, np.arange(len(dets), dtype=np.int32).reshape(-1, 1)]) if len(dets) else np.empty((0, 8), dtype=dets.dtype)
        confs = table[:, 5]
        remain = confs > c["track_thresh"]
        second = np.logical_and(confs > c["min_conf"], confs < c["track_thresh"])
        dets_lo, dets_hi = table[second], table[remain]
        cand = [_RecObb(d) for d in dets_hi]

        unconfirmed = [t for t in self.active if not t.is_activated]
        tracked = [t for t in self.active if t.is_activated]
        pool = _join(tracked, self.lost)
        if pool:
            mean = np.asarray([t.mean.copy() for t in pool])
            cov = np.asarray([t.cov for t in pool])
            for i, t in enumerate(pool):
                if t.state != TRACKED:
                    mean[i][7:10] = 0
            mean, cov = obb.kf5_multi_predict(mean, cov)
            for t, m, p in zip(pool, mean, cov):
                t.mean, t.cov = m, p
        d1 = matching.fuse_score(_iou_distance_obb(pool, cand), np.array([d.conf for d in cand]))
        m1, u_trk1, u_det1 = matching.linear_assignment(d1, c["match_thresh"])
        for it, idet in m1:
            t = pool[it]
            if t.state == TRACKED:
                t.absorb(cand[idet], fc, reactivate=False)
                activated.append(t)
            else:
                t.absorb(cand[idet], fc, reactivate=True)
                refound.append(t)

        cand_lo = [_RecObb(d) for d in dets_lo]
        r_tracked = [pool[i] for i in u_trk1 if pool[i].state == TRACKED]
        m2, u_trk2, _ = matching.linear_assignment(_iou_distance_obb(r_tracked, cand_lo), 0.5)
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

        left = [cand[i] for i in u_det1]
        d3 = matching.fuse_score(_iou_distance_obb(unconfirmed, left), np.array([d.conf for d in left]))
        m3, u_unc, u_det3 = matching.linear_assignment(d3, 0.7)
        for it, idet in m3:
            unconfirmed[it].absorb(left[idet], fc, reactivate=False)
            activated.append(unconfirmed[it])
        for it in u_unc:
            unconfirmed[it].state = REMOVED
            newly_removed.append(unconfirmed[it])

        for inew in u_det3:
            d = left[inew]
            if d.conf < c["track_thresh"]:
                continue
            d.id = self._next_id()
            d.mean, d.cov = obb.kf5_initiate(d.xywh)
            d.tracklet_len = 0
            d.state = TRACKED
            if fc == 1:
                d.is_activated = True
            d.frame_id = d.start_frame = fc
            activated.append(d)

        for t in self.lost:
            if fc - t.frame_id > self.max_time_lost:
                t.state = REMOVED
                newly_removed.append(t)

        self.active = [t for t in self.active if t.state == TRACKED]
        self.active = _join(self.active, activated)
        self.active = _join(self.active, refound)
        self.lost = _minus(self.lost, [t.id for t in self.active])
        self.lost.extend(newly_lost)
        self.lost = _minus(self.lost, list(self.removed_ids))
        self.removed_ids.extend(t.id for t in newly_removed)
        pd = _iou_distance_obb(self.active, self.lost)                 # remove_duplicate_stracks :431-447
        drop_a, drop_b = [], []
        for p, q in zip(*np.where(pd < 0.15)):
            tp = self.active[p].frame_id - self.active[p].start_frame
            tq = self.lost[q].frame_id - self.lost[q].start_frame
            if tp > tq:
                drop_b.append(q)
            else:
                drop_a.append(p)
        self.active = [t for i, t in enumerate(self.active) if i not in drop_a]
        self.lost = [t for i, t in enumerate(self.lost) if i not in drop_b]
        rows = [[*t.xywha, t.id, t.conf, t.cls, t.det_ind] for t in self.active if t.is_activated]
        return np.asarray(rows, dtype=np.float32) if rows else np.empty((0, 9), dtype=np.float32)
