"""StrongSORT on MI355X behind the reference plugin surface.

``StrongSort(...)`` takes the reference constructor's keyword arguments
(boxmot/trackers/bbox/strongsort/strongsort.py:41-52 plus the BaseTracker ones) and
``update(dets, img, embs=None)`` returns the reference's rows (strongsort.py:69-123).  Per frame two HIP kernels run
through the C ABI (include/boxmot_hip.h): the nearest-neighbour appearance distances of every confirmed track's
sample bank to every detection, and the frame step (camera update, 8-state XYAH Kalman filters with the NSA
confidence scaling, gated appearance stage, IoU stage -- both assigned with a restatement of SciPy's
``linear_sum_assignment`` -- track management, sample banks).

Camera motion: the reference applies an ECC estimate unconditionally (strongsort.py:67,83-86).  ``cmc="ecc"`` (what
``create_tracker("strongsort")`` passes by default) runs that estimator on the device (boxmot_amd.cmc.HipECC); any object with
the reference's ``apply(img, boxes) -> 2x3 warp`` is accepted; ``cmc=None`` applies the identity (static camera).  Rejected loudly: OBB detections, ``nn_budget=None``.
"""
from __future__ import annotations

import ctypes
from typing import Any

import numpy as np

from boxmot_amd import _lib
from boxmot_amd.basetracker import OUT_COLS, BaseTracker


class StrongSort(BaseTracker):
    supports_obb = False

    def __init__(
        self,
        reid_model: Any | None = None,
        min_conf: float = 0.1,
        max_cos_dist: float = 0.2,
        max_iou_dist: float = 0.7,
        n_init: int = 3,
        nn_budget: int = 100,
        mc_lambda: float = 0.98,
        ema_alpha: float = 0.9,
        # not reference parameters: camera-motion provider and capacity of the device-resident track table
        cmc: Any | None = None,
        reid_weights: Any | None = None,
        max_tracks: int = 1024,
        max_dets: int = 256,
        emb_dim: int | None = None,
        **kwargs: Any,
    ):
        super().__init__(_tracker_name="StrongSort", **kwargs)
        # per_class=True needs nothing here: StrongSort keeps its tracks in ``self.tracker.tracks``, which the reference's
        # per-class fan-out (basetracker.py:223-263, it swaps ``self.active_tracks``) never partitions -- the effect is
        # one ``_update_impl`` call per class on the same tracker, which is what BaseTracker._do_update does here too.
        if nn_budget is None:
            raise NotImplementedError("boxmot_amd.StrongSort: nn_budget=None (unbounded sample bank) is not supported")
        self.min_conf = min_conf
        self.model = reid_model
        if isinstance(cmc, str):        # cmc="ecc": the estimator the reference always constructs (strongsort.py:67), on the device
            from boxmot_amd.cmc import get_cmc_method
            cmc = get_cmc_method(cmc)()
        self.cmc = cmc
        self._lib = _lib.load()
        self._emb_dim = emb_dim or getattr(self.model, "feature_dim", None) or 512
        cfg = _lib.StrongSortConfig()
        self._lib.boxmot_hip_strongsort_default_config(ctypes.byref(cfg))
        # reid_weights (state_dict / checkpoint / OSN1 or CLP1 blob): the handle runs crops -> backbone on the device inside
        # update (the C ABI's reid_model_path) instead of asking a Python-side model -- no embedding round trip over PCIe
        self._device_reid = reid_weights is not None
        self._blob_file = None
        if self._device_reid:
            import os
            import tempfile

            from boxmot_amd.reid_weights import load_weights, save_blob
            if isinstance(reid_weights, (str, os.PathLike)) and not str(reid_weights).endswith((".pt", ".pth")):
                path = str(reid_weights)
            else:
                fd, path = tempfile.mkstemp(suffix=".reidblob")
                os.close(fd)
                save_blob(load_weights(reid_weights), path)
                self._blob_file = path
            cfg.reid_model_path = path.encode()
        cfg.max_age, cfg.min_conf, cfg.max_cos_dist, cfg.max_iou_dist = self.max_age, min_conf, max_cos_dist, max_iou_dist
        cfg.n_init, cfg.nn_budget, cfg.mc_lambda, cfg.ema_alpha = n_init, nn_budget, mc_lambda, ema_alpha
        cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim = 1, max_tracks, max_dets, self._emb_dim
        self._cfg = cfg
        self._max_tracks = max_tracks
        self._n_tracks = 0          # len(self.tracker.tracks) after the last update (gates the camera-motion estimator)
        self._handle = self._lib.boxmot_hip_strongsort_create(ctypes.byref(cfg))
        if self._blob_file:
            import os
            os.unlink(self._blob_file)       # read at create
        if not self._handle:
            raise RuntimeError(_lib.last_error())

    def _update_impl(self, dets, img, embs=None, masks=None, class_list: int = 0) -> np.ndarray:
        self.check_inputs(dets, img, embs)
        det_arr = np.ascontiguousarray(dets, dtype=np.float32)
        n = int(det_arr.shape[0])
        keep = det_arr[:, 4].astype(np.float64) >= self.min_conf if n else np.zeros(0, bool)      # strongsort.py:75
        if self.cmc is not None and self._n_tracks >= 1:
            # strongsort.py:83-86: the estimator is asked only while tracks exist (it is stateful: its first call just
            # stores the frame and returns the identity, ecc.py:45-96 -- calling it on other frames changes later warps)
            warp = np.ascontiguousarray(np.asarray(self.cmc.apply(img, det_arr[keep, :4].astype(np.float64)), dtype=np.float64)[:2, :3])
            _lib.check(self._lib.boxmot_hip_strongsort_set_warp(self._handle, 0, warp.ctypes.data))
        feats = None
        if n:
            if embs is not None:
                feats = np.ascontiguousarray(embs, dtype=np.float32)
            elif self._device_reid:
                feats = None            # the handle crops and embeds the detections with conf >= min_conf itself
            else:
                feats = np.zeros((n, self._emb_dim), dtype=np.float32)
                if keep.any():
                    feats[keep] = self.model.get_features(det_arr[keep, :4], img)       # strongsort.py:91
            if feats is not None and feats.shape[1] != self._emb_dim:
                raise ValueError(f"embedding width {feats.shape[1]} != emb_dim {self._emb_dim}")
        img_arr = np.ascontiguousarray(img)
        out = np.empty((max(n, 1), 9), dtype=np.float32)
        out_rows, out_is_obb = ctypes.c_int(0), ctypes.c_int(0)
        ok = self._lib.boxmot_hip_strongsort_update(
            self._handle, det_arr.ctypes.data if n else None, n, 6,
            feats.ctypes.data if feats is not None else None, n if feats is not None else 0,
            self._emb_dim if feats is not None else 0,
            img_arr.ctypes.data, int(img_arr.shape[0]), int(img_arr.shape[1]),
            int(img_arr.shape[2]) if img_arr.ndim == 3 else 1,
            out.ctypes.data, int(out.shape[0]), 9, ctypes.byref(out_rows), ctypes.byref(out_is_obb))
        err = None if ok else _lib.last_error()
        if _lib.step_ran(ok):       # a per-stream status report (capacity, solver) is raised after the step has run
            self.frame_count += 1
            if self.cmc is not None:
                cnt = ctypes.c_int(0)
                _lib.check(self._lib.boxmot_hip_strongsort_track_count(self._handle, 0, ctypes.byref(cnt)))
                self._n_tracks = cnt.value
        if err is not None:
            raise RuntimeError(err)
        return out[: out_rows.value, :OUT_COLS].copy()

    def reset(self) -> None:
        # the reference's StrongSort.reset is a no-op (strongsort.py:125-126); the HIP handle does reset its tracks
        _lib.check(self._lib.boxmot_hip_strongsort_reset(self._handle))
        self.frame_count = 0
        self._n_tracks = 0
        self._first_frame_processed = False
        self._first_dets_processed = False

    def capacity(self) -> tuple[int, int, int]:
        """(max_tracks, max_dets, times the device tables grew): the tables grow when a frame would not fit, like the reference's
        lists (include/boxmot_hip.h, boxmot_hip_botsort_reserve)."""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_strongsort_capacity(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def reserve(self, max_tracks: int = 0, max_dets: int = 0) -> None:
        _lib.check(self._lib.boxmot_hip_strongsort_reserve(self._handle, int(max_tracks), int(max_dets)))

    def state_dump(self) -> dict:
        cap, dim = self.capacity()[0], self._emb_dim
        ints = np.zeros((cap, 6), dtype=np.int32)
        kf = np.zeros((cap, 72), dtype=np.float64)
        feat = np.zeros((cap, dim), dtype=np.float32)
        rows, fc, ni = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_strongsort_state_dump(
            self._handle, 0, ints.ctypes.data, kf.ctypes.data, feat.ctypes.data, ctypes.byref(rows), ctypes.byref(fc),
            ctypes.byref(ni)))
        n = rows.value
        return dict(n=n, ints=ints[:n], kf=kf[:n], feat=feat[:n], frame_count=fc.value, next_id=ni.value)

    def debug_costs_enable(self, on: bool = True) -> None:
        """Keep copies of the two ``min_cost_matching`` cost matrices of every following update (parity tests; off by default)."""
        _lib.check(self._lib.boxmot_hip_strongsort_debug_costs_enable(self._handle, int(bool(on))))

    def debug_costs(self, stage: int, plane: int = 0) -> np.ndarray:
        """(tracks, detections) fp64: ``stage`` 0 gated appearance / 1 IoU; ``plane`` 0 the metric's matrix, 1 after the
        ``max_distance`` clamp (include/boxmot_hip.h, boxmot_hip_strongsort_debug_costs)."""
        cap, nd = self.capacity()[:2]
        big = max(cap, nd)
        buf = np.zeros(big * big, dtype=np.float64)
        r, c = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_strongsort_debug_costs(self._handle, 0, int(stage), int(plane), buf.ctypes.data, buf.size,
                                                               ctypes.byref(r), ctypes.byref(c)))
        return buf[: r.value * c.value].reshape(r.value, c.value).copy()

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.boxmot_hip_strongsort_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
