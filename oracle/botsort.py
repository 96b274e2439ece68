"""Oracle BoT-SORT frame step (AABB, CMC off) -- TEST INFRASTRUCTURE ONLY.

A slot-table restatement of the reference tracker, written in the same shape as
the device kernel (track records in slots, ordered index lists for the
active / lost lists, an id ring for the removed deque) so that every phase can
be diffed against ``boxmot_amd`` state dumps.  Follows:
  * BotSort._update_impl and its stages          boxmot/trackers/bbox/botsort/botsort.py:177-500
  * STrack (detection wrapper, features EMA, class histogram, activate /
    re_activate / update)                        boxmot/trackers/bbox/botsort/botsort_track.py:12-282
  * joint / sub / remove_duplicate_stracks       boxmot/trackers/bbox/botsort/botsort_utils.py:10-82
  * TrackState / BaseTrack id counter            boxmot/trackers/bbox/botsort/basetrack.py:6-139
It is pinned bit-for-bit against the reference classes themselves by
tests/test_oracle_golden.py (fixtures produced by tests/golden/make_golden.py).

Deliberate differences from the reference (documented in DESIGN.md):
  * the track-id counter is per tracker instance (the reference's
    ``BaseTrack._count`` is process-global and reset by every constructor);
  * camera-motion compensation is not applied (``use_cmc=False``).
"""
from __future__ import annotations

from collections import deque

import numpy as np

from oracle import kalman, matching

NEW, TRACKED, LOST, LONG_LOST, REMOVED = 0, 1, 2, 3, 4  # basetrack.py:18-22

DEFAULTS = dict(  # botsort.py:66-86 constructor defaults
    track_high_thresh=0.5, track_low_thresh=0.1, new_track_thresh=0.6, track_buffer=30,
    match_thresh=0.8, proximity_thresh=0.5, appearance_thresh=0.25, frame_rate=30,
    fuse_first_associate=False, with_reid=True, second_match_thresh=0.5,
    unconfirmed_match_thresh=0.7, unconfirmed_emb_scale=2.0, removed_stracks_buffer=100,
)
ALPHA = 0.9  # botsort_track.py:39


class _Rec:
    """One STrack-equivalent record (a detection wrapper or a live track)."""

    __slots__ = (
        "xywh", "conf", "cls", "det_ind", "cls_hist", "smooth", "curr", "mean", "cov",
        "is_activated", "tracklet_len", "state", "id", "frame_id", "start_frame",
    )

    def __init__(self, det_row, feat=None):
        det = np.asarray(det_row, dtype=np.float32)        # botsort_track.py:19
        self.xywh = matching.xyxy2xywh32(det[:4])          # :47
        self.conf, self.cls, self.det_ind = det[4], det[5], det[6]
        self.mean = self.cov = None
        self.is_activated = False
        self.tracklet_len = 0
        self.state = NEW
        self.id = 0
        self.frame_id = self.start_frame = 0
        self.cls_hist = []
        self.smooth = self.curr = None
        self.vote_cls(self.cls, self.conf)
        if feat is not None:
            self.blend_feature(feat)

    # botsort_track.py:58-67
    def blend_feature(self, feat):
        feat /= np.linalg.norm(feat)
        self.curr = feat
        if self.smooth is None:
            self.smooth = feat
        else:
            self.smooth = ALPHA * self.smooth + (1 - ALPHA) * feat
        self.smooth /= np.linalg.norm(self.smooth)

    # botsort_track.py:69-82
    def vote_cls(self, cls, conf):
        best = 0
        seen = False
        for entry in self.cls_hist:
            if cls == entry[0]:
                entry[1] += conf
                seen = True
            if entry[1] > best:
                best = entry[1]
                self.cls = entry[0]
        if not seen:
            self.cls_hist.append([cls, conf])
            self.cls = cls

    @property
    def xyxy(self):  # botsort_track.py:310-316
        base = self.mean[:4].copy() if self.mean is not None else self.xywh.copy()
        return matching.xywh2xyxy(base)

    # botsort_track.py:244-282 (shared tail of update / re_activate)
    def absorb(self, det, frame_id, reactivate):
        if reactivate:
            self.tracklet_len = 0
        else:
            self.tracklet_len += 1
        self.frame_id = frame_id
        self.mean, self.cov = kalman.update(self.mean, self.cov, det.xywh)
        if det.curr is not None:
            self.blend_feature(det.curr)
        self.state = TRACKED
        self.is_activated = True
        self.conf, self.cls, self.det_ind = det.conf, det.cls, det.det_ind
        self.vote_cls(det.cls, det.conf)


def _boxes(recs):
    return [r.xyxy for r in recs]


def _join(a, b):
    """joint_stracks: order-preserving union keyed by id (botsort_utils.py:10-31)."""
    seen = {}
    out = []
    for r in a:
        seen[r.id] = 1
        out.append(r)
    for r in b:
        if not seen.get(r.id, 0):
            seen[r.id] = 1
            out.append(r)
    return out


def _minus(a, ids):
    """sub_stracks keyed by id (botsort_utils.py:34-52); ``ids`` is an iterable of ids."""
    keep = {r.id: r for r in a}
    for tid in ids:
        keep.pop(tid, None)
    return list(keep.values())


class BotSortOracle:
    # what the oriented-box variant (oracle/botsort_obb.py) replaces: the detection layout, the record type, the filter, the IoU
    REC = _Rec
    N_BOX, CONF_COL, N_DET_COLS, N_OUT_COLS = 4, 4, 6, 8
    VEL_ZERO = slice(6, 8)                      # velocities zeroed for non-tracked tracks before prediction (botsort_track.py:104-109)

    @staticmethod
    def _kf_predict(mean, cov):
        return kalman.multi_predict(mean, cov)

    @staticmethod
    def _kf_initiate(z):
        return kalman.initiate(z)

    @staticmethod
    def _iou_d(a, b):
        return matching.iou_distance(_boxes(a), _boxes(b))

    @staticmethod
    def _row(t):
        return [*t.xyxy, t.id, t.conf, t.cls, t.det_ind]

    def __init__(self, reid=None, **kw):
        cfg = dict(DEFAULTS)
        unknown = set(kw) - set(cfg)
        if unknown:
            raise TypeError(f"unknown BoT-SORT options: {sorted(unknown)}")
        cfg.update(kw)
        self.cfg = cfg
        self.reid = reid if cfg["with_reid"] else None
        self.max_time_lost = int(cfg["frame_rate"] / 30.0 * cfg["track_buffer"])  # botsort.py:103-104
        self.frame_count = 0
        self.id_count = 0
        self.active = []
        self.lost = []
        self.removed_ids = deque(maxlen=cfg["removed_stracks_buffer"])  # botsort.py:93-95
        self.last = {}

    def _next_id(self):
        self.id_count += 1
        return self.id_count

    def _assoc_cost(self, tracks, dets, emb_scale, fuse):
        """IoU gate + optional score fusion + appearance gate (botsort.py:306-317, 396-413)."""
        c = self.cfg
        iou_d = self._iou_d(tracks, dets)
        gate = iou_d > c["proximity_thresh"]
        self._last_parts = {"iou": np.array(iou_d, dtype=np.float64).reshape(len(tracks), len(dets)), "emb": None}
        if fuse:
            iou_d = matching.fuse_score(iou_d, np.array([d.conf for d in dets]))
        if not c["with_reid"]:
            return iou_d
        emb = matching.embedding_distance([t.smooth for t in tracks], [d.curr for d in dets])
        self._last_parts["emb"] = np.array(emb, dtype=np.float64).reshape(len(tracks), len(dets))
        if emb_scale is not None:
            emb = emb / emb_scale
        emb[emb > c["appearance_thresh"]] = 1.0
        emb[gate] = 1.0
        return np.minimum(iou_d, emb)

    @staticmethod
    def _warp_tracks(tracks, warp):
        """STrack.multi_gmc on the pool, then on the unconfirmed tracks (botsort.py:134-145, botsort_track.py:117-132)."""
        H = np.asarray(warp)
        R8 = np.kron(np.eye(4), H[:2, :2])
        tvec = H[:2, 2]
        for t in tracks:
            m = R8.dot(t.mean)
            m[:2] += tvec
            t.mean = m
            t.cov = R8.dot(t.cov).dot(R8.T)

    def update(self, dets, img=None, embs=None, warp=None):
        """dets (N,6) [x1,y1,x2,y2,conf,cls]; returns (M,8) fp32 rows.  ``warp``: optional 2x3 camera-motion
        matrix (what ``cmc.apply`` returns in the reference), applied after prediction."""
        c = self.cfg
        dets = np.asarray(dets)
        if dets.size == 0:
            dets = np.empty((0, self.N_DET_COLS), dtype=np.float32)
        self.frame_count += 1
        fc = self.frame_count
        activated, refound, newly_lost, newly_removed = [], [], [], []

        # botsort.py:251-261 (det index column promotes the table to fp64)
        if len(dets):
            table = np.hstack([dets, np.arange(len(dets), dtype=np.int32).reshape(-1, 1)])
        else:
            table = np.empty((0, self.N_DET_COLS + 1), dtype=dets.dtype)
        confs = table[:, self.CONF_COL]
        low = np.logical_and(confs > c["track_low_thresh"], confs < c["track_high_thresh"])
        high = confs > c["track_high_thresh"]
        dets_hi, dets_lo = table[high], table[low]

        if c["with_reid"] and embs is None:
            feats = self.reid.get_features(dets_hi[:, :self.N_BOX], img)
        else:
            feats = np.asarray(embs)[high] if embs is not None else None

        if len(dets_hi):
            if c["with_reid"]:
                cand = [self.REC(d, f) for d, f in zip(dets_hi, feats)]
            else:
                cand = [self.REC(d) for d in dets_hi]
        else:
            cand = []

        unconfirmed = [t for t in self.active if not t.is_activated]
        confirmed = [t for t in self.active if t.is_activated]
        pool = _join(confirmed, self.lost)

        # ---- first association (botsort.py:285-333) ----
        if pool:
            mean = np.asarray([t.mean.copy() for t in pool])
            cov = np.asarray([t.cov for t in pool])
            for i, t in enumerate(pool):
                if t.state != TRACKED:
                    mean[i][self.VEL_ZERO] = 0
            mean, cov = self._kf_predict(mean, cov)
            for t, m, p in zip(pool, mean, cov):
                t.mean, t.cov = m, p
        if warp is not None:
            self._warp_tracks(list(pool) + list(unconfirmed), warp)
        dists = self._assoc_cost(pool, cand, None, c["fuse_first_associate"])
        m1, u_trk1, u_det1 = matching.linear_assignment(dists, c["match_thresh"])
        # (the cost matrices of the three associations, kept for the value-parity tests: "dists" is what linear_assignment was given,
        # "iou" = matching.iou_distance before score fusion, "emb" = matching.embedding_distance before scale / gates)
        self.last = {"dists_first": dists, "matches_first": m1,
                     "stages": [dict(dists=np.asarray(dists, dtype=np.float64).reshape(len(pool), len(cand)), **self._last_parts)]}
        for it, idet in m1:
            t = pool[it]
            if t.state == TRACKED:
                t.absorb(cand[idet], fc, reactivate=False)
                activated.append(t)
            else:
                t.absorb(cand[idet], fc, reactivate=True)
                refound.append(t)

        # ---- second association, IoU only (botsort.py:335-378) ----
        cand_lo = [self.REC(d) for d in dets_lo]
        remain = [pool[i] for i in u_trk1 if pool[i].state == TRACKED]
        d2 = self._iou_d(remain, cand_lo)
        m2, u_trk2, _ = matching.linear_assignment(d2, c["second_match_thresh"])
        d2m = np.asarray(d2, dtype=np.float64).reshape(len(remain), len(cand_lo))
        self.last["stages"].append(dict(dists=d2m, iou=d2m, emb=None))
        for it, idet in m2:
            t = remain[it]
            if t.state == TRACKED:
                t.absorb(cand_lo[idet], fc, reactivate=False)
                activated.append(t)
            else:
                t.absorb(cand_lo[idet], fc, reactivate=True)
                refound.append(t)
        for it in u_trk2:
            t = remain[it]
            if t.state != LOST:
                t.state = LOST
                newly_lost.append(t)

        # ---- unconfirmed tracks (botsort.py:380-431) ----
        left = [cand[i] for i in u_det1]
        d3 = self._assoc_cost(unconfirmed, left, c["unconfirmed_emb_scale"], True)
        m3, u_unc, u_det3 = matching.linear_assignment(d3, c["unconfirmed_match_thresh"])
        self.last["stages"].append(dict(dists=np.asarray(d3, dtype=np.float64).reshape(len(unconfirmed), len(left)), **self._last_parts))
        for it, idet in m3:
            unconfirmed[it].absorb(left[idet], fc, reactivate=False)
            activated.append(unconfirmed[it])
        for it in u_unc:
            unconfirmed[it].state = REMOVED
            newly_removed.append(unconfirmed[it])

        # ---- births (botsort.py:433-440, botsort_track.py:232-242) ----
        for inew in u_det3:
            d = left[inew]
            if d.conf < c["new_track_thresh"]:
                continue
            d.id = self._next_id()
            d.mean, d.cov = self._kf_initiate(d.xywh)
            d.tracklet_len = 0
            d.state = TRACKED
            if fc == 1:
                d.is_activated = True
            d.frame_id = d.start_frame = fc
            activated.append(d)

        # ---- lost -> removed by age (botsort.py:472-476) ----
        for t in self.lost:
            if fc - t.frame_id > self.max_time_lost:
                t.state = REMOVED
                newly_removed.append(t)

        # ---- list bookkeeping (botsort.py:478-492) ----
        self.active = [t for t in self.active if t.state == TRACKED]
        self.active = _join(self.active, activated)
        self.active = _join(self.active, refound)
        self.lost = _minus(self.lost, [t.id for t in self.active])
        self.lost.extend(newly_lost)
        self.lost = _minus(self.lost, list(self.removed_ids))
        self.removed_ids.extend(t.id for t in newly_removed)
        self.active, self.lost = self._dedup(self.active, self.lost)

        rows = [self._row(t) for t in self.active if t.is_activated]
        return np.asarray(rows, dtype=np.float32) if rows else np.empty((0, self.N_OUT_COLS), dtype=np.float32)

    @classmethod
    def _dedup(cls, a, b):
        """remove_duplicate_stracks (botsort_utils.py:55-82)."""
        pd = cls._iou_d(a, b)
        drop_a, drop_b = [], []
        for p, q in zip(*np.where(pd < 0.15)):
            tp = a[p].frame_id - a[p].start_frame
            tq = b[q].frame_id - b[q].start_frame
            if tp > tq:
                drop_b.append(q)
            else:
                drop_a.append(p)
        return (
            [t for i, t in enumerate(a) if i not in drop_a],
            [t for i, t in enumerate(b) if i not in drop_b],
        )

    # ---- state dump used by the device-parity tests ----
    def dump(self):
        def pack(recs):
            return dict(
                id=np.array([t.id for t in recs], dtype=np.int64),
                state=np.array([t.state for t in recs], dtype=np.int64),
                is_activated=np.array([t.is_activated for t in recs], dtype=bool),
                frame_id=np.array([t.frame_id for t in recs], dtype=np.int64),
                start_frame=np.array([t.start_frame for t in recs], dtype=np.int64),
                tracklet_len=np.array([t.tracklet_len for t in recs], dtype=np.int64),
                mean=np.array([t.mean for t in recs], dtype=np.float64).reshape(len(recs), 8),
                cov=np.array([t.cov for t in recs], dtype=np.float64).reshape(len(recs), 8, 8),
                smooth=(np.array([t.smooth for t in recs], dtype=np.float32)
                        if recs and recs[0].smooth is not None else None),
                conf=np.array([t.conf for t in recs], dtype=np.float32),
                cls=np.array([t.cls for t in recs], dtype=np.float32),
                det_ind=np.array([t.det_ind for t in recs], dtype=np.float32),
            )
        return dict(frame_count=self.frame_count, id_count=self.id_count,
                    active=pack(self.active), lost=pack(self.lost),
                    removed_ids=np.array(list(self.removed_ids), dtype=np.int64))
