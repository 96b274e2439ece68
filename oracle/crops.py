"""Oracle restatement of the ReID crop path -- TEST INFRASTRUCTURE ONLY.

Follows boxmot/reid/backends/base_backend.py:148-195 (``get_crops``) and
boxmot/reid/core/preprocessing.py:12-18 (``resize`` = ``cv2.resize(crop,
(W, H), interpolation=cv2.INTER_LINEAR)`` on uint8) plus
``cv2.cvtColor(crop, cv2.COLOR_BGR2RGB)`` (base_backend.py:181).

``cv2`` (opencv-python 4.11.0.86, reference uv.lock:3847-3848) is a third-party
dependency that is absent offline, so ``cv2_resize_linear_u8`` restates
OpenCV's published uint8 INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp:
half-pixel-centre mapping, 11-bit fixed-point coefficients
``INTER_RESIZE_COEF_BITS = 11``, horizontal pass into int32, vertical pass
``(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2``, and the documented
"scale exactly 2 in both axes -> INTER_AREA 2x2 box" special case).
PARITY UNPINNED against real OpenCV: it cannot be imported here.
"""
from __future__ import annotations

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS

# base_backend.py:48-49 (ImageNet statistics, fp32 tensors)
MEAN_RGB = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD_RGB = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _axis_tables(dst_n: int, src_n: int, scale: float | None = None):
    """Per-output-index source offset and the two int16 coefficients.

    resize.cpp: ``f = (float)((d + 0.5) * scale - 0.5); s = cvFloor(f); f -= s``;
    the x axis clamps (s<0 -> s=0,f=0 ; s>=n-1 -> s=n-1,f=0); the y axis keeps
    its coefficients and clips the two row indices instead.
    """
    if scale is None:
        scale = float(src_n) / float(dst_n)  # double, = 1 / inv_scale
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f: np.ndarray):
    """saturate_cast<short>(c * 2048) with round-half-to-even (cvRound)."""
    c0 = np.float32(1.0) - f
    a0 = np.clip(np.rint(c0 * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    a1 = np.clip(np.rint(f * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int32)
    return a0, a1


def cv2_resize_linear_u8(src: np.ndarray, dsize_wh, inv_scale_xy=None) -> np.ndarray:
    """``cv2.resize(src, (W, H), interpolation=cv2.INTER_LINEAR)`` for uint8 HxWxC.  ``inv_scale_xy = (1 / fx, 1 / fy)`` restates
    the ``cv2.resize(src, (0, 0), fx=, fy=)`` form, where the coordinate scale is 1 / fx whatever the rounded output size is."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    sh, sw, _ = src.shape
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    if sh == dh and sw == dw:
        out = src.copy()
        return out[:, :, 0] if squeeze else out
    if sw == 2 * dw and sh == 2 * dh:
        # INTER_LINEAR with an exact 2x2 shrink is dispatched to INTER_AREA (fast path).
        s = src.astype(np.int32)
        out = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
        out = out.astype(np.uint8)
        return out[:, :, 0] if squeeze else out

    sx, fx = _axis_tables(dw, sw, None if inv_scale_xy is None else inv_scale_xy[0])
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1
    fx = np.where(hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(hi, sw - 1, sx)
    ax0, ax1 = _coef(fx)
    sx1 = np.minimum(sx + 1, sw - 1)

    sy, fy = _axis_tables(dh, sh, None if inv_scale_xy is None else inv_scale_xy[1])
    by0, by1 = _coef(fy)
    sy0 = np.clip(sy, 0, sh - 1)
    sy1 = np.clip(sy + 1, 0, sh - 1)

    s32 = src.astype(np.int32)
    # horizontal pass for every source row (int32, scaled by 2^11)
    hbuf = s32[:, sx, :] * ax0[None, :, None] + s32[:, sx1, :] * ax1[None, :, None]
    r0 = hbuf[sy0]
    r1 = hbuf[sy1]
    out = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def crop_boxes_int(xyxys: np.ndarray, w: int, h: int) -> np.ndarray:
    """base_backend.py:172-174: ``box.round().astype(int)`` then clip to the frame."""
    b = np.asarray(xyxys, dtype=np.float32).reshape(-1, xyxys.shape[-1])[:, :4]
    r = np.round(b).astype("int")  # float32 round-half-to-even
    out = np.empty_like(r)
    out[:, 0] = np.maximum(0, r[:, 0])
    out[:, 1] = np.maximum(0, r[:, 1])
    out[:, 2] = np.minimum(w, r[:, 2])
    out[:, 3] = np.minimum(h, r[:, 3])
    return out


IMAGENET_MEAN_BGR = (104, 116, 124)          # reid/core/preprocessing.py:8-9


def resize_pad_u8(crop: np.ndarray, target_shape) -> np.ndarray:
    """reid/core/preprocessing.py:21-45: aspect-preserving resize, constant ImageNet-mean (BGR) border."""
    th, tw = target_shape
    h, w = crop.shape[:2]
    scale = min(tw / w, th / h)
    new_w, new_h = int(w * scale), int(h * scale)
    resized = cv2_resize_linear_u8(crop, (new_w, new_h))
    top = (th - new_h) // 2
    left = (tw - new_w) // 2
    out = np.empty((th, tw, 3), dtype=np.uint8)
    out[:] = np.array(IMAGENET_MEAN_BGR, dtype=np.uint8)
    out[top:top + new_h, left:left + new_w] = resized
    return out


def get_crops_u8(xyxys: np.ndarray, img: np.ndarray, input_shape=(256, 128), preprocess: str = "resize") -> np.ndarray:
    """uint8 RGB crops (N, H, W, 3): slice -> resize (or resize_pad) -> BGR2RGB (blank if empty)."""
    h, w = img.shape[:2]
    xyxys = np.asarray(xyxys, dtype=np.float32)
    if xyxys.size == 0:
        return np.zeros((0, input_shape[0], input_shape[1], 3), dtype=np.uint8)
    boxes = crop_boxes_int(xyxys.reshape(-1, xyxys.shape[-1]), w, h)
    out = np.empty((len(boxes), input_shape[0], input_shape[1], 3), dtype=np.uint8)
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        if x2 > x1 and y2 > y1:
            crop = img[y1:y2, x1:x2]
        else:
            crop = np.zeros((input_shape[0], input_shape[1], 3), dtype=np.uint8)
        if preprocess == "resize_pad":
            crop = resize_pad_u8(crop, input_shape)
        else:
            crop = cv2_resize_linear_u8(crop, (input_shape[1], input_shape[0]))
        out[i] = crop[:, :, ::-1]
    return out


def normalize_crops(crops_u8: np.ndarray, mean=None, std=None) -> np.ndarray:
    """base_backend.py:183-193: NCHW fp32, ``/255.0`` then ``(x - mean) / std``; mean / std default to the ImageNet values,
    "clip" models use 0.5 / 0.5 (base_backend.py:50-54)."""
    m = MEAN_RGB if mean is None else np.asarray(mean, dtype=np.float32)
    s = STD_RGB if std is None else np.asarray(std, dtype=np.float32)
    x = np.ascontiguousarray(np.transpose(crops_u8, (0, 3, 1, 2))).astype(np.float32)
    x = x / np.float32(255.0)
    x = (x - m.reshape(1, 3, 1, 1)) / s.reshape(1, 3, 1, 1)
    return np.ascontiguousarray(x, dtype=np.float32)


def normalization_lut() -> np.ndarray:
    """(3, 256) fp32 table: pixel value -> normalised value per RGB channel."""
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    return ((v[None, :] - MEAN_RGB[:, None]) / STD_RGB[:, None]).astype(np.float32)


def get_crops(xyxys: np.ndarray, img: np.ndarray, input_shape=(256, 128), preprocess: str = "resize", mean=None, std=None) -> np.ndarray:
    return normalize_crops(get_crops_u8(xyxys, img, input_shape, preprocess), mean, std)
