"""Camera-motion compensation providers for the HIP trackers.

``HipECC`` is the reference's ``ECC`` estimator (boxmot/motion/cmc/ecc.py:14-96: MOTION_TRANSLATION, eps 1e-5, 100 iterations,
scale 0.15, grayscale -- what ``get_cmc_method("ecc")()`` and StrongSORT construct) with the estimation in HIP kernels
(csrc/cmc_ecc.hpp) behind the C ABI ``boxmot_hip_ecc_*``: ``apply(img, dets) -> (2, 3) float32`` like ``BaseCMC.apply``
(base_cmc.py:25-28), so it plugs into the ``cmc=`` argument of ``BotSort`` / ``StrongSort`` / ``DeepOcSort``.  The first
call stores the frame and returns the identity; a non-converging pair of frames returns the identity too (ecc.py:67-76).

The sparse-optical-flow estimator (``sof``, BoT-SORT's YAML default) is not built: asking for it raises.
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


def get_cmc_method(name: str):
    """``boxmot.motion.cmc.get_cmc_method`` for the estimators that exist here."""
    if name == "ecc":
        return HipECC
    raise NotImplementedError(f"boxmot_amd: camera-motion estimator '{name}' is not implemented on the device (have: ecc); pass cmc=<object "
                              "with apply(img, dets)> to use a host-side estimator")
