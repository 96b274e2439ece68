"""Camera-motion compensation providers for the HIP trackers.

``HipECC`` is the reference's ``ECC`` estimator (boxmot/motion/cmc/ecc.py:14-96: MOTION_TRANSLATION, eps 1e-5, 100 iterations,
scale 0.15, grayscale -- what ``get_cmc_method("ecc")()`` and StrongSORT construct) with the estimation in HIP kernels
(csrc/cmc_ecc.hpp) behind the C ABI ``boxmot_hip_ecc_*``: ``apply(img, dets) -> (2, 3) float32`` like ``BaseCMC.apply``
(base_cmc.py:25-28), so it plugs into the ``cmc=`` argument of ``BotSort`` / ``StrongSort`` / ``DeepOcSort``.  The first
call stores the frame and returns the identity; a non-converging pair of frames returns the identity too (ecc.py:67-76).

``HipSOF`` is the reference's ``SOF`` estimator (boxmot/motion/cmc/sof.py:14-147: goodFeaturesToTrack + cornerSubPix keypoints,
pyramidal Lucas-Kanade tracking, RANSAC partial-affine fit; BoT-SORT's YAML default ``cmc_method`` and the estimator DeepOCSORT
constructs, deepocsort.py:297) in HIP kernels (csrc/cmc_sof.hpp) behind ``boxmot_hip_sof_*``; ``apply(img, dets)`` masks the
detections' boxes out of the corner detector like ``BaseCMC.generate_mask`` (base_cmc.py:63-105).

Both restate OpenCV algorithms the reference calls; OpenCV itself is absent offline, so parity with ``cv2`` is UNPINNED (oracle/ecc.py,
oracle/sof.py say what is restated and how); the kernels are tested against those oracles.
"""
from __future__ import annotations

import ctypes

import numpy as np

from boxmot_amd import _lib


class HipECC:
    grayscale = True

    def __init__(self, eps: float = 1e-5, max_iter: int = 100, scale: float = 0.15, warp_mode: int = 0, align: bool = False,
                 grayscale: bool = True):
        if warp_mode != 0:          # cv2.MOTION_TRANSLATION == 0
            raise NotImplementedError("boxmot_amd.HipECC: only MOTION_TRANSLATION (the reference's default) is implemented")
        if align or not grayscale:
            raise NotImplementedError("boxmot_amd.HipECC: align=True / grayscale=False are not implemented")
        self.eps, self.max_iter, self.scale = float(eps), int(max_iter), float(scale)
        self._lib = _lib.load()
        self._handle = None
        self._shape = None
        self.last_iterations = 0

    def _ensure(self, rows: int, cols: int) -> None:
        if self._handle is not None and self._shape == (rows, cols):
            return
        self.close()
        self._handle = self._lib.boxmot_hip_ecc_create(1, rows, cols, self.scale, self.eps, self.max_iter)
        if not self._handle:
            raise RuntimeError(_lib.last_error())
        self._shape = (rows, cols)

    def apply(self, img, dets=None) -> np.ndarray:
        a = np.ascontiguousarray(img)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("Expected img to be an (H, W, 3) uint8 BGR array.")
        self._ensure(int(a.shape[0]), int(a.shape[1]))
        warp = np.zeros(6, dtype=np.float64)
        it = ctypes.c_int(0)
        _lib.check(self._lib.boxmot_hip_ecc_apply(self._handle, 0, a.ctypes.data, a.shape[0], a.shape[1], 3, warp.ctypes.data, ctypes.byref(it)))
        self.last_iterations = it.value
        return warp.reshape(2, 3).astype(np.float32)

    def reset(self) -> None:
        if self._handle is not None:
            _lib.check(self._lib.boxmot_hip_ecc_reset(self._handle, -1))

    def close(self) -> None:
        h = getattr(self, "_handle", None)
        if h:
            self._lib.boxmot_hip_ecc_destroy(h)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipSOF:
    grayscale = True

    def __init__(self, scale: float = 0.15, min_inliers: int = 8, min_inlier_ratio: float = 0.2, ransac_reproj_threshold: float = 3.0):
        self.scale, self.min_inliers = float(scale), int(min_inliers)
        self.min_inlier_ratio, self.ransac_reproj_threshold = float(min_inlier_ratio), float(ransac_reproj_threshold)
        self._lib = _lib.load()
        self._handle = None
        self._shape = None
        self.last_info = {}

    def _ensure(self, rows: int, cols: int) -> None:
        if self._handle is not None and self._shape == (rows, cols):
            return
        self.close()
        self._handle = self._lib.boxmot_hip_sof_create(1, rows, cols, self.scale, self.min_inliers, self.min_inlier_ratio,
                                                       self.ransac_reproj_threshold)
        if not self._handle:
            raise RuntimeError(_lib.last_error())
        self._shape = (rows, cols)

    def apply(self, img, dets=None) -> np.ndarray:
        a = np.ascontiguousarray(img)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("Expected img to be an (H, W, 3) uint8 BGR array.")
        self._ensure(int(a.shape[0]), int(a.shape[1]))
        d = None
        if dets is not None and np.size(dets):
            d = np.asarray(dets)
            d = np.ascontiguousarray(d.reshape(-1, d.shape[-1])[:, :4], dtype=np.float32)       # base_cmc.py:96: det[:4] as float32
        warp = np.zeros(6, dtype=np.float64)
        info = np.zeros(8, dtype=np.int32)
        _lib.check(self._lib.boxmot_hip_sof_apply(self._handle, 0, a.ctypes.data, a.shape[0], a.shape[1], 3,
                                                  d.ctypes.data if d is not None else None, 0 if d is None else len(d), 4,
                                                  warp.ctypes.data, info.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        self.last_info = dict(zip(("mode", "keypoints", "tracked", "inliers", "ransac_iters", "accepted", "detected", "initialized"), info.tolist()))
        return warp.reshape(2, 3).astype(np.float32)

    def keypoints(self) -> np.ndarray:
        """(n, 2) fp32 (x, y) points, in the scaled image, that the next frame will track (``SOF.prev_keypoints``)."""
        out = np.zeros((1000, 2), dtype=np.float32)
        n = ctypes.c_int(0)
        if self._handle is None:
            return out[:0]
        _lib.check(self._lib.boxmot_hip_sof_keypoints(self._handle, 0, out.ctypes.data, 1000, ctypes.byref(n)))
        return out[:n.value]

    def debug_map(self, which: int) -> np.ndarray:
        """The detector's images of the last frame (tests): 0 minimum-eigenvalue map fp32, 1 detection mask, 2 scaled gray frame."""
        h, w = ctypes.c_int(0), ctypes.c_int(0)
        buf = np.zeros(self._shape[0] * self._shape[1], dtype=np.float32 if which == 0 else np.uint8)
        _lib.check(self._lib.boxmot_hip_sof_debug_map(self._handle, 0, int(which), buf.ctypes.data, buf.nbytes, ctypes.byref(h), ctypes.byref(w)))
        return buf[:h.value * w.value].reshape(h.value, w.value).copy()

    def reset(self) -> None:
        if self._handle is not None:
            _lib.check(self._lib.boxmot_hip_sof_reset(self._handle, -1))

    def close(self) -> None:
        h = getattr(self, "_handle", None)
        if h:
            self._lib.boxmot_hip_sof_destroy(h)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_cmc_method(name: str):
    """``boxmot.motion.cmc.get_cmc_method`` for the estimators that exist here."""
    if name == "ecc":
        return HipECC
    if name == "sof":
        return HipSOF
    raise NotImplementedError(f"boxmot_amd: camera-motion estimator '{name}' is not implemented on the device (have: sof, ecc); pass cmc=<object "
                              "with apply(img, dets)> to use a host-side estimator")
