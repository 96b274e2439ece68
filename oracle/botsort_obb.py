"""Oracle BoT-SORT frame step for ORIENTED detections -- TEST INFRASTRUCTURE ONLY.

BotSortOracle with the pieces the reference switches on `is_obb` (boxmot/trackers/bbox/botsort/botsort.py:105, 267-271, 306, 357, 396,
495; botsort_track.py:16-56, 84-115, 244-330): detections (cx, cy, w, h, angle, conf, cls), KalmanFilterXYWH(ndim=5) with (vw, vh,
vtheta) zeroed for non-tracked tracks, rotated IoU of the fp32 `xywha`, rows (cx, cy, w, h, angle, id, conf, cls, det_ind).  The
appearance path is unchanged.  Camera-motion compensation of oriented boxes (multi_gmc_obb, botsort.py:147-158) goes through
oracle/obb.py's restatements of cv2.transform / cv2.minAreaRect (unpinned); the device step refuses a warp on an oriented handle.
"""
from __future__ import annotations

import numpy as np

from oracle import obb
from oracle.botsort import TRACKED, BotSortOracle, _Rec


class _RecObb(_Rec):
    __slots__ = ()

    def __init__(self, det_row, feat=None):
        det = np.asarray(det_row, dtype=np.float32)        # botsort_track.py:19, 52-56
        self.xywh = det[:5].copy()
        self.conf, self.cls, self.det_ind = det[5], det[6], det[7]
        self.mean = self.cov = None
        self.is_activated = False
        self.tracklet_len = 0
        self.state = 0
        self.id = 0
        self.frame_id = self.start_frame = 0
        self.cls_hist = []
        self.smooth = self.curr = None
        self.vote_cls(self.cls, self.conf)
        if feat is not None:
            self.blend_feature(feat)

    @property
    def xywha(self):                                        # botsort_track.py:319-327
        ret = self.mean[:5].copy() if self.mean is not None else self.xywh.copy()
        return np.asarray(ret, dtype=np.float32)

    def absorb(self, det, frame_id, reactivate):            # botsort_track.py:244-282
        if reactivate:
            self.tracklet_len = 0
        else:
            self.tracklet_len += 1
        self.frame_id = frame_id
        self.mean, self.cov = obb.kf5_update(self.mean, self.cov, det.xywh)
        if det.curr is not None:
            self.blend_feature(det.curr)
        self.state = TRACKED
        self.is_activated = True
        self.conf, self.cls, self.det_ind = det.conf, det.cls, det.det_ind
        self.vote_cls(det.cls, det.conf)


class BotSortObbOracle(BotSortOracle):
    REC = _RecObb
    N_BOX, CONF_COL, N_DET_COLS, N_OUT_COLS = 5, 5, 7, 9
    VEL_ZERO = slice(7, 10)

    @staticmethod
    def _kf_predict(mean, cov):
        return obb.kf5_multi_predict(mean, cov)

    @staticmethod
    def _kf_initiate(z):
        return obb.kf5_initiate(z)

    @staticmethod
    def _warp_tracks(tracks, warp):                         # STrack.multi_gmc_obb, botsort_track.py:197-230
        for t in tracks:
            if t.mean is None or t.cov is None:
                continue
            t.mean, t.cov = obb.gmc_obb(t.mean, t.cov, warp)

    @staticmethod
    def _iou_d(a, b):                                       # matching.py:46-80 with is_obb
        if len(a) == 0 or len(b) == 0:
            return np.zeros((len(a), len(b)), dtype=np.float32)
        return 1 - obb.iou_obb_matrix(np.asarray([t.xywha for t in a], dtype=float), np.asarray([t.xywha for t in b], dtype=float))

    @staticmethod
    def _row(t):
        return [*t.xywha, t.id, t.conf, t.cls, t.det_ind]
