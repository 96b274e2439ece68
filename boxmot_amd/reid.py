"""ReID backend on MI355X with the reference's backend interface.

``HipReID.get_features(xyxys, img)`` has the contract of
``BaseModelBackend.get_features`` (boxmot/reid/backends/base_backend.py:197-217):
boxes ``(N, >=4)`` xyxy + a BGR uint8 frame in, ``(N, 512)`` float32
L2-normalised embeddings out (``np.array([])``-like empty result for no boxes);
``warmup()`` exists.  Two backbones, chosen by the checkpoint's parameter names: OSNet (512-d; x0.25 has two fused MFMA
kernel families: fp16 operands (mode 1) and fp32-grade (mode 2, within 1e-3 of the fp32 CPU path on any weights)) and CLIP-ReID ViT-B/16 (1280-d, make_model.py:95-139; crops normalised with mean = std = 0.5 as
base_backend.py:50-54 does for "clip" models).  Crop / resize / normalise / backbone / L2 all run in HIP
kernels through the ReID C ABI (include/boxmot_hip.h, replacing
boxmot/native/cpp/trackers/base/include/boxmot/trackers/base/reid_capi.h:36-94).
PyTorch is only the container the weights are read from.
"""
from __future__ import annotations

import ctypes

import numpy as np

from boxmot_amd import _lib
from boxmot_amd.reid_weights import load_weights

MODE_FP32_LAYERWISE = 0
MODE_FP16_FUSED = 1
MODE_FP32_FUSED = 2          # fused kernels with fp32-grade arithmetic (fp16 hi + lo operand pairs, fp32 everything else): 1e-3 on any weights


class HipReID:
    input_shape = (256, 128)

    def __init__(self, weights, max_crops: int = 1024, mode: int = MODE_FP32_LAYERWISE, preprocess: str | None = None):
        """``weights``: state_dict, ``.pt`` checkpoint path, OSN1 / CLP1 blob path or blob array.  ``preprocess``: "resize"
        (default) or "resize_pad" (reid/core/preprocessing.py:48-65)."""
        self._lib = _lib.load()
        self.blob = load_weights(weights)
        self.max_crops = max_crops
        self._handle = self._lib.boxmot_hip_reid_create(None, self.blob.ctypes.data, int(self.blob.size), int(max_crops))
        if not self._handle:
            raise RuntimeError(_lib.last_error())
        self.feature_dim = int(self._lib.boxmot_hip_reid_feature_dim(self._handle))
        if mode != MODE_FP32_LAYERWISE:
            self.set_mode(mode)
        if preprocess is not None:
            _lib.check(self._lib.boxmot_hip_reid_set_preprocess(self._handle, preprocess.encode()))

    def set_mode(self, mode: int) -> None:
        _lib.check(self._lib.boxmot_hip_reid_set_mode(self._handle, int(mode)))

    @staticmethod
    def _boxes(xyxys):
        b = np.asarray(xyxys, dtype=np.float32)
        if b.size == 0:
            return b.reshape(0, 4)
        if b.ndim == 1:
            b = b.reshape(1, -1)
        if b.shape[1] in (5, 7, 9):           # oriented boxes [cx, cy, w, h, angle, ...] (base_backend.py:119-122, 157)
            return np.ascontiguousarray(b[:, :5])
        if b.shape[1] < 4:
            raise ValueError("Expected detections with at least 4 coordinates")
        return np.ascontiguousarray(b[:, :4])

    @staticmethod
    def _image(img):
        a = np.ascontiguousarray(img)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("ReID expects an (H, W, 3) uint8 BGR image")
        return a

    def get_features(self, xyxys, img) -> np.ndarray:
        boxes = self._boxes(xyxys)
        n = len(boxes)
        if n == 0:
            return np.empty((0, self.feature_dim), dtype=np.float32)
        a = self._image(img)
        out = np.empty((n, self.feature_dim), dtype=np.float32)
        for i0 in range(0, n, self.max_crops):
            chunk = boxes[i0:i0 + self.max_crops]
            _lib.check(self._lib.boxmot_hip_reid_compute_features(
                self._handle, a.ctypes.data, a.shape[0], a.shape[1], 3, chunk.ctypes.data, len(chunk), chunk.shape[1],
                out[i0:].ctypes.data, len(chunk)))
        return out

    def get_crops(self, xyxys, img) -> np.ndarray:
        """Normalised crops, (N, 3, 256, 128) float32 like base_backend.py:148-195."""
        boxes = self._boxes(xyxys)
        n = len(boxes)
        a = self._image(img)
        out = np.empty((n, 256, 128, 3), dtype=np.float32)
        if n:
            if n > self.max_crops:
                raise ValueError("more boxes than max_crops")
            _lib.check(self._lib.boxmot_hip_reid_preprocess(
                self._handle, a.ctypes.data, a.shape[0], a.shape[1], 3, boxes.ctypes.data, n, boxes.shape[1], out.ctypes.data))
        return np.ascontiguousarray(np.transpose(out, (0, 3, 1, 2)))

    def last_time_ms(self):
        """(preprocess, backbone) device milliseconds of the last ``get_features`` chunk (HIP events)."""
        pre, proc = ctypes.c_double(0), ctypes.c_double(0)
        _lib.check(self._lib.boxmot_hip_reid_last_time_ms(self._handle, ctypes.byref(pre), ctypes.byref(proc)))
        return pre.value, proc.value

    def warmup(self, imgsz=((256, 128, 3),)):
        im = np.random.randint(0, 255, imgsz[0], dtype=np.uint8)
        self.get_features(np.array([[0, 0, 64, 64], [0, 0, 128, 128]], dtype=np.float32), im)

    def close(self):
        h = getattr(self, "_handle", None)
        if h:
            self._lib.boxmot_hip_reid_destroy(h)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
