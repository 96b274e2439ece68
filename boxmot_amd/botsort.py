"""BoT-SORT on MI355X behind the reference plugin surface.

``BotSort(...)`` takes the reference constructor's keyword arguments
(boxmot/trackers/bbox/botsort/botsort.py:66-86; YAML keys of
boxmot/configs/trackers/botsort.yaml) and ``update(dets, img, embs=None)``
returns the reference's rows; every per-frame computation runs in HIP kernels
through the C ABI (include/boxmot_hip.h).  Mirrors the shape of the reference's
ctypes wrapper for its native backend (boxmot/native/trackers/botsort.py:94-274).

Camera-motion compensation: applying a warp to the track state runs on the device
(STrack.multi_gmc), and so does *estimating* it from images for both estimators the reference configures BoT-SORT with:
``cmc_method="ecc"`` (the constructor default, boxmot_amd.cmc.HipECC) and ``"sof"`` (configs/trackers/botsort.yaml,
boxmot_amd.cmc.HipSOF); ``cmc=`` accepts any object exposing the reference's ``apply(img, dets) -> 2x3 warp``.

Oriented detections (7 columns, botsort.py:120-131): the frame step has an oriented twin on the device (10-state filter,
rotated-rectangle IoU; ``is_obb`` of the handle's configuration).  The layout is inferred from the first detection table like
the reference does; embeddings of oriented detections come from the caller (``embs``) or from ``reid_model.get_features`` given
the (cx, cy, w, h, angle) boxes, as the reference calls it.  Camera motion: the estimator sees the enclosing axis-aligned boxes of the
oriented detections (botsort.py:147-158) and its warp is applied to the oriented tracks on the device (``STrack.multi_gmc_obb``:
corner warp, minimum-area refit, re-alignment).  Not implemented, and rejected loudly rather than approximated: masks.
"""
from __future__ import annotations

import ctypes
from types import SimpleNamespace
from typing import Any

import numpy as np

from boxmot_amd import _lib
from boxmot_amd.basetracker import OUT_COLS, BaseTracker

TRACK_STATE_NAMES = {0: "New", 1: "Tracked", 2: "Lost", 3: "LongLost", 4: "Removed"}


class BotSort(BaseTracker):
    supports_obb = True

    def __init__(
        self,
        reid_model: Any | None = None,
        track_high_thresh: float = 0.5,
        track_low_thresh: float = 0.1,
        new_track_thresh: float = 0.6,
        track_buffer: int = 30,
        match_thresh: float = 0.8,
        proximity_thresh: float = 0.5,
        appearance_thresh: float = 0.25,
        use_cmc: bool = True,
        cmc_method: str = "ecc",
        frame_rate: int = 30,
        fuse_first_associate: bool = False,
        with_reid: bool = True,
        second_match_thresh: float = 0.5,
        unconfirmed_match_thresh: float = 0.7,
        unconfirmed_emb_scale: float = 2.0,
        removed_stracks_buffer: int = 100,
        # capacity of the device-resident track table (not reference parameters)
        max_tracks: int = 1024,
        max_dets: int = 256,
        emb_dim: int | None = None,
        cmc: Any | None = None,
        _tracker_kind: int = 0,
        _tracker_name: str = "BotSort",
        **kwargs: Any,
    ):
        super().__init__(_tracker_name=_tracker_name, **kwargs)
        if isinstance(cmc, str):
            from boxmot_amd.cmc import get_cmc_method
            cmc = get_cmc_method(cmc)()
        if use_cmc and cmc is None:
            # botsort.py:116-117: get_cmc_method(cmc_method)() -- "ecc" (the constructor default) and "sof" (the YAML default) are
            # estimated on the device (boxmot_amd.cmc.HipECC / HipSOF); anything else raises NotImplementedError there
            from boxmot_amd.cmc import get_cmc_method
            cmc = get_cmc_method(cmc_method)()
        self.track_high_thresh = track_high_thresh
        self.track_low_thresh = track_low_thresh
        self.new_track_thresh = new_track_thresh
        self.match_thresh = match_thresh
        self.buffer_size = int(frame_rate / 30.0 * track_buffer)
        self.max_time_lost = self.buffer_size
        self.proximity_thresh = proximity_thresh
        self.appearance_thresh = appearance_thresh
        self.second_match_thresh = second_match_thresh
        self.unconfirmed_match_thresh = unconfirmed_match_thresh
        self.unconfirmed_emb_scale = unconfirmed_emb_scale
        self.with_reid = with_reid
        self.model = reid_model if with_reid else None
        self.cmc = cmc if use_cmc else None
        self.fuse_first_associate = fuse_first_associate
        self._lib = _lib.load()
        self._emb_dim = emb_dim or getattr(self.model, "feature_dim", None) or 512
        self._max_dets = max_dets
        cfg = _lib.BotSortConfig()
        self._lib.boxmot_hip_botsort_default_config(ctypes.byref(cfg))
        cfg.track_high_thresh = track_high_thresh
        cfg.track_low_thresh = track_low_thresh
        cfg.new_track_thresh = new_track_thresh
        cfg.track_buffer = track_buffer
        cfg.match_thresh = match_thresh
        cfg.proximity_thresh = proximity_thresh
        cfg.appearance_thresh = appearance_thresh
        cfg.cmc_method = None
        cfg.frame_rate = frame_rate
        cfg.fuse_first_associate = int(bool(fuse_first_associate))
        cfg.with_reid = int(bool(with_reid))
        cfg.max_obs = self.max_obs
        cfg.second_match_thresh = second_match_thresh
        cfg.unconfirmed_match_thresh = unconfirmed_match_thresh
        cfg.unconfirmed_emb_scale = unconfirmed_emb_scale
        cfg.removed_stracks_buffer = removed_stracks_buffer
        cfg.n_streams = 1
        cfg.max_tracks = max_tracks
        cfg.max_dets = max_dets
        cfg.emb_dim = self._emb_dim
        cfg.n_class_lists = self.nr_classes if self.per_class else 1
        cfg.tracker_kind = int(_tracker_kind)
        cfg.is_obb = int(self.is_obb)
        self._cfg = cfg
        self._seed_frame_count = False
        self._handle = None
        self._reserved = (0, 0)
        self._max_tracks = max_tracks
        self._check_obb_options()
        self._create_handle()

    def _create_handle(self) -> None:
        self.close()
        self._handle = self._lib.boxmot_hip_botsort_create(ctypes.byref(self._cfg))
        if not self._handle:
            raise RuntimeError(_lib.last_error())
        if any(self._reserved):          # a reserve() made before the layout was known survives the re-creation of the handle
            _lib.check(self._lib.boxmot_hip_botsort_reserve(self._handle, *self._reserved))

    def _check_obb_options(self) -> None:
        pass        # (every BoT-SORT option has an oriented twin now)

    @staticmethod
    def _obb_detections_to_cmc_boxes(dets: np.ndarray) -> np.ndarray:
        """botsort.py:126-132 over STrack.obb_to_xyxy (botsort_track.py:159-174): the enclosing axis-aligned box of every oriented
        detection, from its four corners as cv2.boxPoints lays them out (fp32; the angle in degrees as an fp32 product)."""
        if len(dets) == 0:
            return np.empty((0, 4), dtype=np.float32)
        out = np.empty((len(dets), 4), dtype=np.float32)
        for i, det in enumerate(np.asarray(dets, dtype=np.float32)):
            cx, cy, w, h, ang = (np.float32(v) for v in det[:5])
            w, h = np.float32(max(float(w), 1e-4)), np.float32(max(float(h), 1e-4))
            a = np.float64(float(np.degrees(ang))) * np.pi / 180.0
            b, sn = np.float32(np.cos(a)) * np.float32(0.5), np.float32(np.sin(a)) * np.float32(0.5)
            p0 = (cx - sn * h - b * w, cy + b * h - sn * w)
            p1 = (cx + sn * h - b * w, cy - b * h - sn * w)
            xs = np.array([p0[0], p1[0], np.float32(2) * cx - p0[0], np.float32(2) * cx - p1[0]], dtype=np.float32)
            ys = np.array([p0[1], p1[1], np.float32(2) * cy - p0[1], np.float32(2) * cy - p1[1]], dtype=np.float32)
            out[i] = [xs.min(), ys.min(), xs.max(), ys.max()]
        return out

    def _set_detection_mode(self, is_obb: bool) -> None:
        """The first detection table decides the layout (basetracker.py:163-173).  The device tables are sized for one layout
        (8- or 10-state filter, 6- or 7-column detections): a tracker that has not stepped yet gets a handle of the other kind."""
        changed = bool(is_obb) != bool(self._cfg.is_obb)
        super()._set_detection_mode(is_obb)
        if changed:
            self._check_obb_options()
            self._cfg.is_obb = int(self.is_obb)
            self._create_handle()
            # frames that carried no layout (update(None, img), 1-D empty tables) have already advanced the replaced handle's
            # device frame counter; the reference keeps counting through them, so the new handle starts from the host's count
            self._seed_frame_count = self.frame_count > 0

    # ------------------------------------------------------------------ update
    def _update_impl(self, dets, img, embs=None, masks=None, class_list: int = 0) -> np.ndarray:
        self.check_inputs(dets, img, embs)
        det_arr = np.ascontiguousarray(dets, dtype=np.float32)
        n = int(det_arr.shape[0])
        feats = None
        if self.with_reid:
            if embs is not None:
                feats = np.ascontiguousarray(embs, dtype=np.float32)
            else:
                # same call the reference makes (botsort.py:191-192): boxes of the high-confidence rows
                first = det_arr[:, self.conf_idx].astype(np.float64) > self.track_high_thresh
                feats = np.zeros((n, self._emb_dim), dtype=np.float32)
                if first.any():
                    feats[first] = self.model.get_features(det_arr[first, :self.box_cols], img)
            if n and feats.shape[1] != self._emb_dim:
                raise ValueError(f"embedding width {feats.shape[1]} != emb_dim {self._emb_dim}")
        img_arr = np.ascontiguousarray(img)
        if self.cmc is not None:
            # botsort.py:142: the estimator sees the detection table incl. the index column; the warp is applied
            # on the device after the Kalman prediction (boxmot_hip_botsort_set_warp)
            if self.is_obb:     # botsort.py:147-158: the estimator sees the enclosing boxes of the oriented detections
                table = self._obb_detections_to_cmc_boxes(np.hstack([det_arr, np.arange(n, dtype=np.int32).reshape(-1, 1)])) if n else np.empty((0, 4), np.float32)
            else:
                table = np.hstack([det_arr, np.arange(n, dtype=np.int32).reshape(-1, 1)]) if n else np.empty((0, self.det_cols + 1), det_arr.dtype)
            warp = np.ascontiguousarray(np.asarray(self.cmc.apply(img, table), dtype=np.float64)[:2, :3])
            if warp.shape != (2, 3):
                raise ValueError(f"cmc.apply returned shape {warp.shape}, expected (2, 3)")
            _lib.check(self._lib.boxmot_hip_botsort_set_warp(self._handle, 0, warp.ctypes.data))
        out = np.empty((max(n, 1), 9), dtype=np.float32)
        out_rows = ctypes.c_int(0)
        out_is_obb = ctypes.c_int(0)
        ok = self._lib.boxmot_hip_botsort_update_stream(
            self._handle, 0, int(class_list), int(self.frame_count) if (self.per_class or self._seed_frame_count) else -1,
            det_arr.ctypes.data if n else None, n, self.det_cols,
            feats.ctypes.data if (feats is not None and n) else None, n if feats is not None else 0,
            self._emb_dim if feats is not None else 0,
            img_arr.ctypes.data, int(img_arr.shape[0]), int(img_arr.shape[1]),
            int(img_arr.shape[2]) if img_arr.ndim == 3 else 1,
            out.ctypes.data, int(out.shape[0]), 9, ctypes.byref(out_rows), ctypes.byref(out_is_obb),
        )
        if _lib.step_ran(ok):       # a per-stream status report (capacity, solver) is raised after the step has run
            self.frame_count += 1
            self._seed_frame_count = False
        _lib.check(ok)
        return out[: out_rows.value, :self.output_cols].copy()

    def reset(self) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_reset(self._handle))
        self.frame_count = 0
        self._first_frame_processed = False
        self._first_dets_processed = False

    def capacity(self) -> tuple[int, int, int]:
        """(max_tracks, max_dets, times the device tables grew).  The tables start at the constructor's sizes and grow when a frame
        would not fit, like the reference's lists (include/boxmot_hip.h, boxmot_hip_botsort_reserve)."""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_botsort_capacity(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def reserve(self, max_tracks: int = 0, max_dets: int = 0) -> None:
        _lib.check(self._lib.boxmot_hip_botsort_reserve(self._handle, int(max_tracks), int(max_dets)))
        self._reserved = (max(int(max_tracks), self._reserved[0]), max(int(max_dets), self._reserved[1]))      # re-applied if the handle is re-made

    # ------------------------------------------------- introspection (read-only)
    def state_dump(self, which: int = 0, class_list: int = 0) -> dict:
        """Copy the live tracks back from the device (parity tests / debugging)."""
        cap, dim = self.capacity()[0], self._emb_dim
        ints = np.zeros((cap, 6), dtype=np.int32)
        kf = np.zeros((cap, 110 if self.is_obb else 72), dtype=np.float64)     # mean + covariance: 8 + 64, or 10 + 100 oriented
        smooth = np.zeros((cap, dim), dtype=np.float32)
        misc = np.zeros((cap, 3), dtype=np.float32)
        rows, fc, ic = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_botsort_state_dump(
            self._handle, 0, which, class_list, ints.ctypes.data, kf.ctypes.data, smooth.ctypes.data, misc.ctypes.data,
            ctypes.byref(rows), ctypes.byref(fc), ctypes.byref(ic)))
        n = rows.value
        return dict(n=n, ints=ints[:n], kf=kf[:n], smooth=smooth[:n], misc=misc[:n], frame_count=fc.value,
                    id_count=ic.value)

    def debug_costs_enable(self, on: bool = True) -> None:
        """Keep copies of the association cost matrices of every following update (parity tests; off by default)."""
        _lib.check(self._lib.boxmot_hip_botsort_debug_costs_enable(self._handle, int(bool(on))))

    def debug_costs(self, stage: int, plane: int = 0) -> np.ndarray:
        """(tracks, detections) fp64 cost matrix of the last update: ``stage`` 0 first / 1 second / 2 unconfirmed association;
        ``plane`` 0 the solver's matrix, 1 ``iou_distance``, 2 ``embedding_distance`` where evaluated (NaN elsewhere) --
        include/boxmot_hip.h, boxmot_hip_botsort_debug_costs."""
        cap, nd = self.capacity()[:2]
        buf = np.zeros(cap * nd, dtype=np.float64)
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_botsort_debug_costs(self._handle, 0, int(stage), int(plane), buf.ctypes.data, buf.size,
                                                            ctypes.byref(r), ctypes.byref(c)))
        return buf[: r.value * c.value].reshape(r.value, c.value).copy()

    def _track_views(self, which: int):
        d = self.state_dump(which)
        out = []
        for r in range(d["n"]):
            i = d["ints"][r]
            nm = 10 if self.is_obb else 8
            mean = d["kf"][r, :nm].copy()
            xyxy = np.array([mean[0] - mean[2] / 2, mean[1] - mean[3] / 2, mean[0] + mean[2] / 2, mean[1] + mean[3] / 2])
            out.append(SimpleNamespace(
                id=int(i[0]), state=int(i[1]), is_activated=bool(i[2]), frame_id=int(i[3]), start_frame=int(i[4]),
                tracklet_len=int(i[5]), mean=mean, covariance=d["kf"][r, nm:].reshape(nm, nm).copy(), xyxy=xyxy,
                xywha=mean[:5].astype(np.float32) if self.is_obb else None,
                smooth_feat=d["smooth"][r].copy(), conf=float(d["misc"][r, 0]), cls=float(d["misc"][r, 1]),
                det_ind=float(d["misc"][r, 2])))
        return out

    @property
    def active_tracks(self):
        return self._track_views(0)

    @property
    def lost_stracks(self):
        return self._track_views(1)

    @property
    def removed_stracks(self):
        return []   # ids only live on the device; the reference keeps them for display

    def get_last_reid_time_ms(self) -> float:
        v = ctypes.c_double(0.0)
        _lib.check(self._lib.boxmot_hip_botsort_last_reid_time_ms(self._handle, ctypes.byref(v)))
        return float(v.value)

    def get_last_track_time_ms(self) -> float:
        v = ctypes.c_double(0.0)
        _lib.check(self._lib.boxmot_hip_botsort_last_track_time_ms(self._handle, ctypes.byref(v)))
        return float(v.value)

    def close(self) -> None:
        h = getattr(self, "_handle", None)
        if h:
            self._lib.boxmot_hip_botsort_destroy(h)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
