"""DeepOCSORT on MI355X behind the reference plugin surface.

``DeepOcSort(...)`` takes the reference constructor's keyword arguments
(boxmot/trackers/bbox/deepocsort/deepocsort.py:263-281 plus the BaseTracker ones,
basetracker.py:19-31) and ``update(dets, img, embs=None)`` returns the reference's rows
(deepocsort.py:302-492); the per-frame computation -- per-track 7-state Kalman filters with the
observation-centric re-update, IoU / velocity-direction / adaptive-weighted appearance costs, assignment,
recovery round, bookkeeping -- runs in one HIP kernel through the C ABI (include/boxmot_hip.h).

Camera motion: applying a warp to the tracks runs on the device (``apply_affine_correction``), and so does estimating it:
the sparse-optical-flow estimator the reference constructs (deepocsort.py:297) is ``boxmot_amd.cmc.HipSOF`` (``cmc_off=False``,
the default); ``cmc="ecc"`` or any object exposing ``apply(img, boxes) -> 2x3 warp`` replaces it.  ``per_class=True`` keeps one track list per class on the
device (one stream per class) with the shared id counter and rewound frame counter of the reference's fan-out.
Rejected loudly: OBB detections, ``max_age > 45``.
"""
from __future__ import annotations

import ctypes
from typing import Any

import numpy as np

from boxmot_amd import _lib
from boxmot_amd.basetracker import OUT_COLS, BaseTracker


class DeepOcSort(BaseTracker):
    supports_obb = False

    def __init__(
        self,
        reid_model: Any | None = None,
        delta_t: int = 3,
        inertia: float = 0.2,
        w_association_emb: float = 0.5,
        alpha_fixed_emb: float = 0.95,
        aw_param: float = 0.5,
        embedding_off: bool = False,
        cmc_off: bool = False,
        aw_off: bool = False,
        Q_xy_scaling: float = 0.01,
        Q_s_scaling: float = 0.0001,
        # capacity of the device-resident track table (not reference parameters)
        max_tracks: int = 1024,
        max_dets: int = 256,
        emb_dim: int | None = None,
        cmc: Any | None = None,
        **kwargs: Any,
    ):
        super().__init__(_tracker_name="DeepOcSort", **kwargs)
        if isinstance(cmc, str):            # cmc="ecc": the device ECC estimator instead of the built-in one
            from boxmot_amd.cmc import get_cmc_method
            cmc = get_cmc_method(cmc)()
        if not cmc_off and cmc is None:     # deepocsort.py:297: self.cmc = get_cmc_method("sof")()
            from boxmot_amd.cmc import HipSOF
            cmc = HipSOF()
        self.delta_t, self.inertia = delta_t, inertia
        self.w_association_emb, self.alpha_fixed_emb, self.aw_param = w_association_emb, alpha_fixed_emb, aw_param
        self.Q_xy_scaling, self.Q_s_scaling = Q_xy_scaling, Q_s_scaling
        self.model = reid_model
        self.cmc = None if cmc_off else cmc
        self.embedding_off, self.cmc_off, self.aw_off = embedding_off, cmc_off, aw_off
        self._lib = _lib.load()
        self._emb_dim = 1 if embedding_off else (emb_dim or getattr(self.model, "feature_dim", None) or 512)
        cfg = _lib.DeepOcSortConfig()
        self._lib.boxmot_hip_deepocsort_default_config(ctypes.byref(cfg))
        cfg.det_thresh, cfg.max_age, cfg.max_obs = self.det_thresh, self.max_age, self.max_obs
        cfg.min_hits, cfg.iou_threshold = self.min_hits, self.iou_threshold
        cfg.delta_t, cfg.inertia, cfg.w_association_emb = delta_t, inertia, w_association_emb
        cfg.alpha_fixed_emb, cfg.aw_param = alpha_fixed_emb, aw_param
        cfg.embedding_off, cfg.cmc_off, cfg.aw_off = int(bool(embedding_off)), int(bool(cmc_off)), int(bool(aw_off))
        cfg.Q_xy_scaling, cfg.Q_s_scaling = Q_xy_scaling, Q_s_scaling
        cfg.use_byte, cfg.min_conf = (int(self._byte[0]), self._byte[1]) if hasattr(self, "_byte") else (0, 0.1)   # OcSort only
        # the function behind the step's "iou" matrices (association.py:95, deepocsort.py:420, ocsort.py:457,486); a name
        # outside the table raises on the first frame (BaseTracker._preprocess), where the reference resolves it
        cfg.asso_func = _lib.ASSO_FUNCS.get(self._asso_func_base_name, 0)
        cfg.is_obb = int(self.is_obb)
        cfg.n_streams = self.nr_classes if self.per_class else 1
        cfg.max_tracks, cfg.max_dets, cfg.emb_dim = max_tracks, max_dets, self._emb_dim
        self._ids_issued = ctypes.c_int(0)           # KalmanBoxTracker.count - 1, shared by the per-class lists
        self._cfg = cfg
        self._max_tracks = max_tracks
        self._seed_frame_count = False
        self._handle = None
        self._reserved = (0, 0)
        self._check_obb_options()
        self._create_handle()

    def _create_handle(self) -> None:
        self.close()
        self._handle = self._lib.boxmot_hip_deepocsort_create(ctypes.byref(self._cfg))
        if not self._handle:
            raise RuntimeError(_lib.last_error())
        if any(self._reserved):          # a reserve() made before the layout was known survives the re-creation of the handle
            _lib.check(self._lib.boxmot_hip_deepocsort_reserve(self._handle, *self._reserved))

    def _check_obb_options(self) -> None:
        pass        # iou_obb and centroid_obb are the oriented association functions (iou.py:408-417), both on the device

    def _set_detection_mode(self, is_obb: bool) -> None:
        """The first detection table decides the layout (basetracker.py:163-173); a tracker that has not stepped yet gets a handle
        of the other kind (the device tables are sized for one layout: 7- or 9-state filter, 6- or 7-column detections)."""
        changed = bool(is_obb) != bool(self._cfg.is_obb)
        super()._set_detection_mode(is_obb)
        if changed:
            self._check_obb_options()
            self._cfg.is_obb = int(self.is_obb)
            # the oriented step has one association function, the rotated IoU; a name without an oriented twin (giou_obb, ...) is
            # reported by the first-frame check of BaseTracker._preprocess, as the reference reports it
            oriented = {"iou": _lib.ASSO_FUNCS["iou"], "centroid": _lib.ASSO_FUNCS["centroid"]}
            self._cfg.asso_func = oriented.get(self._asso_func_base_name, 0) if self.is_obb else _lib.ASSO_FUNCS.get(self._asso_func_base_name, 0)
            self._create_handle()
            # frames that carried no layout (update(None, img), 1-D empty tables) have already advanced the replaced handle's
            # device frame counter; the reference keeps counting through them, so the new handle starts from the host's count
            self._seed_frame_count = self.frame_count > 0
            self._ids_issued = ctypes.c_int(0)

    def _update_impl(self, dets, img, embs=None, masks=None, class_list: int = 0) -> np.ndarray:
        self.check_inputs(dets, img, embs)
        det_arr = np.ascontiguousarray(dets, dtype=np.float32)
        n = int(det_arr.shape[0])
        feats = None
        if not self.embedding_off and n:
            keep = det_arr[:, self.conf_idx] > np.float32(self.det_thresh)          # deepocsort.py:334 (fp32 compare)
            if embs is not None:
                feats = np.ascontiguousarray(embs, dtype=np.float32)
            elif keep.any():
                # same call the reference makes (deepocsort.py:345): every detection above det_thresh
                feats = np.zeros((n, self._emb_dim), dtype=np.float32)
                feats[keep] = self.model.get_features(det_arr[keep, :4], img)
            else:
                feats = np.zeros((n, self._emb_dim), dtype=np.float32)
            if feats.shape[1] != self._emb_dim:
                raise ValueError(f"embedding width {feats.shape[1]} != emb_dim {self._emb_dim}")
        img_arr = np.ascontiguousarray(img)
        stream = int(class_list) if self.per_class else 0
        if self.cmc is not None:
            kept = det_arr[det_arr[:, 4] > np.float32(self.det_thresh), :4].astype(np.float64) if n else np.empty((0, 4))
            warp = np.ascontiguousarray(np.asarray(self.cmc.apply(img, kept), dtype=np.float64)[:2, :3])     # deepocsort.py:348-349
            _lib.check(self._lib.boxmot_hip_deepocsort_set_warp(self._handle, stream, warp.ctypes.data))
        out = np.empty((max(n, 1), 9), dtype=np.float32)
        out_rows, out_is_obb = ctypes.c_int(0), ctypes.c_int(0)
        ok = self._lib.boxmot_hip_deepocsort_update_stream(
            self._handle, stream, int(self.frame_count) if (self.per_class or self._seed_frame_count) else -1,
            ctypes.byref(self._ids_issued) if self.per_class else None,
            det_arr.ctypes.data if n else None, n, self.det_cols,
            feats.ctypes.data if feats is not None else None, n if feats is not None else 0,
            self._emb_dim if feats is not None else 0,
            img_arr.ctypes.data, int(img_arr.shape[0]), int(img_arr.shape[1]),
            int(img_arr.shape[2]) if img_arr.ndim == 3 else 1,
            out.ctypes.data, int(out.shape[0]), 9, ctypes.byref(out_rows), ctypes.byref(out_is_obb))
        if _lib.step_ran(ok):       # a per-stream status report (capacity, solver) is raised after the step has run
            self.frame_count += 1
            self._seed_frame_count = False
        _lib.check(ok)
        if out_rows.value == 0:
            return np.array([])                      # deepocsort.py:490-492 -> TrackResults of shape (0, 0)
        return out[: out_rows.value, :self.output_cols].copy()

    def reset(self) -> None:
        _lib.check(self._lib.boxmot_hip_deepocsort_reset(self._handle))
        self._ids_issued = ctypes.c_int(0)
        self.frame_count = 0
        self._first_frame_processed = False
        self._first_dets_processed = False

    def capacity(self) -> tuple[int, int, int]:
        """(max_tracks, max_dets, times the device tables grew): the tables grow when a frame would not fit, like the reference's
        lists (include/boxmot_hip.h, boxmot_hip_botsort_reserve)."""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_deepocsort_capacity(self._handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def reserve(self, max_tracks: int = 0, max_dets: int = 0) -> None:
        _lib.check(self._lib.boxmot_hip_deepocsort_reserve(self._handle, int(max_tracks), int(max_dets)))
        self._reserved = (max(int(max_tracks), self._reserved[0]), max(int(max_dets), self._reserved[1]))      # re-applied if the handle is re-made

    def state_dump(self) -> dict:
        """Copy the live tracks back from the device in list order (parity tests / debugging)."""
        cap, dim = self.capacity()[0], self._emb_dim
        ints = np.zeros((cap, 5), dtype=np.int32)
        kf = np.zeros((cap, 90 if self.is_obb else 72), dtype=np.float64)     # x[8] ++ P[8][8], or x[9] ++ P[9][9] oriented
        emb = np.zeros((cap, dim), dtype=np.float64)
        rows, fc, ic = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_deepocsort_state_dump(
            self._handle, 0, ints.ctypes.data, kf.ctypes.data, emb.ctypes.data, ctypes.byref(rows), ctypes.byref(fc),
            ctypes.byref(ic)))
        n = rows.value
        return dict(n=n, ints=ints[:n], kf=kf[:n], emb=emb[:n], frame_count=fc.value, id_count=ic.value)

    def debug_costs_enable(self, on: bool = True) -> None:
        """Keep copies of ``associate``'s matrices of every following update (parity tests; off by default)."""
        _lib.check(self._lib.boxmot_hip_deepocsort_debug_costs_enable(self._handle, int(bool(on))))

    def debug_costs(self, plane: int = 0):
        """``(matrix, branch)`` of the last update's ``associate`` call: (detections, tracks) fp64; ``plane`` 0 ``final_cost``,
        1 ``iou_matrix``, 2 the weighted ``emb_cost``; ``branch`` 0 no matrix, 1 permutation early-out, 2 solver
        (include/boxmot_hip.h, boxmot_hip_deepocsort_debug_costs)."""
        cap, nd = self.capacity()[:2]
        buf = np.zeros(cap * nd, dtype=np.float64)
        r, c, b = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_deepocsort_debug_costs(self._handle, 0, int(plane), buf.ctypes.data, buf.size,
                                                               ctypes.byref(r), ctypes.byref(c), ctypes.byref(b)))
        return buf[: r.value * c.value].reshape(r.value, c.value).copy(), b.value

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.boxmot_hip_deepocsort_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OcSort(DeepOcSort):
    """OC-SORT behind the reference plugin surface (boxmot/trackers/bbox/ocsort/ocsort.py:334-555).

    The reference's ``OcSort`` and its ``DeepOcSort`` with the appearance and camera-motion terms switched off run the
    same arithmetic -- same ``KalmanBoxTracker`` / ``KalmanFilterXYSR``, ``associate`` with the velocity-direction term,
    observation-centric recovery round and re-update, same output rule (ocsort.py:398-555 vs deepocsort.py:302-492);
    pinned bit-for-bit on the reference classes (tests/golden/mot17_golden.npz, tests/test_oracle_vs_reference.py).  So
    OC-SORT runs on the DeepOCSORT step kernel with those two terms off, plus its own optional BYTE association of the
    detections with ``min_conf < score < det_thresh`` (``use_byte=True``, ocsort.py:393-399, 456-485).

    Oriented detections (7 columns; ``supports_obb``, ocsort.py:332): the step has an oriented twin on the device -- the 9-state
    ``KalmanFilterXYSR(dim_x=9, dim_z=5)`` with the aligned measurement, the interpolated angle of the re-update and the damped
    angular velocity, the rotated IoU, 9-column rows -- chosen by the first detection table like in the reference."""

    supports_obb = True

    def __init__(self, min_conf: float = 0.1, delta_t: int = 3, inertia: float = 0.2, use_byte: bool = False,
                 Q_xy_scaling: float = 0.01, Q_s_scaling: float = 0.0001, max_tracks: int = 1024, max_dets: int = 256,
                 **kwargs: Any):
        for k in ("reid_model", "embedding_off", "cmc_off", "cmc", "emb_dim"):
            if k in kwargs:
                raise TypeError(f"OcSort() got an unexpected keyword argument {k!r}")
        self._byte = (bool(use_byte), float(min_conf))
        super().__init__(reid_model=None, delta_t=delta_t, inertia=inertia, embedding_off=True, cmc_off=True,
                         Q_xy_scaling=Q_xy_scaling, Q_s_scaling=Q_s_scaling, max_tracks=max_tracks, max_dets=max_dets, **kwargs)
        self.min_conf, self.use_byte = min_conf, use_byte
        self.asso_threshold = self.iou_threshold

    def _update_impl(self, dets, img, embs=None, masks=None, class_list: int = 0) -> np.ndarray:
        return super()._update_impl(dets, img, None, masks, class_list)      # appearance is never used (ocsort.py:361-365)
