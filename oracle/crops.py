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

import math

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


# ---- oriented boxes: base_backend.py:91-117 (_crop_obb) = cv2.getRotationMatrix2D + cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT 0) ----
# Restated from OpenCV's published algorithm (imgwarp.cpp): the 2x3 matrix is inverted in double precision, destination pixel
# (x, y) samples the source at the fixed-point position X = (cvRound((M01 y + M02) 2^10) + 16 + cvRound(M00 x 2^10)) >> 5 (1/32-pixel
# grid, likewise Y), and the uint8 bilinear kernel is integer: weights (32 - fx)(32 - fy) * 32 ... (they sum to 2^15 exactly, so
# initInterTab2D's sum correction never fires), result (sum + 2^14) >> 15, taps outside the image are the constant border 0.
# PARITY UNPINNED against real OpenCV (absent offline), like cv2_resize_linear_u8.
def obb_crop_geometry(box):
    """(out_w, out_h, inverse 2x3 map as 6 float64) of ``_crop_obb`` for ``box = [cx, cy, w, h, angle]`` (float32 fields)."""
    b = np.asarray(box, dtype=np.float32).reshape(-1)
    cx, cy, bw, bh, angle = (b[k] for k in range(5))
    bw, bh = max(float(bw), 1.0), max(float(bh), 1.0)
    out_w, out_h = max(int(round(bw)), 1), max(int(round(bh)), 1)
    angle_deg = float(np.degrees(angle))                     # float32 degrees -> Python float
    a = angle_deg * (np.pi / 180.0)                          # getRotationMatrix2D: angle *= CV_PI / 180
    alpha, beta = math.cos(a), math.sin(a)                   # libm, as OpenCV's std::cos / std::sin
    fcx, fcy = float(cx), float(cy)
    M = np.array([[alpha, beta, (1 - alpha) * fcx - beta * fcy], [-beta, alpha, beta * fcx + (1 - alpha) * fcy]], dtype=np.float64)
    M[0, 2] += out_w / 2.0 - fcx
    M[1, 2] += out_h / 2.0 - fcy
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]               # warpAffine without WARP_INVERSE_MAP inverts the matrix
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    iM = np.empty(6, dtype=np.float64)
    iM[0], iM[1], iM[3], iM[4] = A11, M[0, 1] * (-D), M[1, 0] * (-D), A22
    iM[2] = -iM[0] * M[0, 2] - iM[1] * M[1, 2]
    iM[5] = -iM[3] * M[0, 2] - iM[4] * M[1, 2]
    return out_w, out_h, iM


def cv2_warp_affine_inverse_linear_u8(img: np.ndarray, iM, out_wh) -> np.ndarray:
    """The remap half of ``cv2.warpAffine(img, M, (w, h), INTER_LINEAR, BORDER_CONSTANT, 0)`` given the INVERTED matrix ``iM``."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape[:2]
    ow, oh = out_wh
    x = np.arange(ow, dtype=np.float64)
    y = np.arange(oh, dtype=np.float64)
    adelta = np.rint(iM[0] * x * 1024.0).astype(np.int64)
    bdelta = np.rint(iM[3] * x * 1024.0).astype(np.int64)
    X0 = np.rint((iM[1] * y + iM[2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((iM[4] * y + iM[5]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)      # saturate_cast<short>
    fx, fy = (X & 31).astype(np.int64), (Y & 31).astype(np.int64)
    w00, w01, w10, w11 = (32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
        return np.where(ok[..., None], v, 0)
    acc = tap(sy, sx) * w00[..., None] + tap(sy, sx + 1) * w01[..., None] + tap(sy + 1, sx) * w10[..., None] + tap(sy + 1, sx + 1) * w11[..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def crop_obb(box, img: np.ndarray) -> np.ndarray:
    """``BaseModelBackend._crop_obb`` (base_backend.py:91-117): the rectified (out_h, out_w, 3) BGR crop of an oriented box."""
    ow, oh, iM = obb_crop_geometry(box)
    return cv2_warp_affine_inverse_linear_u8(img, iM, (ow, oh))


def is_obb(boxes: np.ndarray) -> bool:
    """base_backend.py:119-122: a row of 5, 7 or 9 values is an oriented box."""
    return boxes.ndim == 2 and boxes.shape[0] > 0 and boxes.shape[1] in (5, 7, 9)


def get_crops_u8(xyxys: np.ndarray, img: np.ndarray, input_shape=(256, 128), preprocess: str = "resize") -> np.ndarray:
    """uint8 RGB crops (N, H, W, 3): slice -> resize (or resize_pad) -> BGR2RGB (blank if empty)."""
    h, w = img.shape[:2]
    xyxys = np.asarray(xyxys, dtype=np.float32)
    if xyxys.size == 0:
        return np.zeros((0, input_shape[0], input_shape[1], 3), dtype=np.uint8)
    rows = xyxys.reshape(-1, xyxys.shape[-1])
    obb = is_obb(rows)                                       # base_backend.py:157: decided on the first row
    boxes = crop_boxes_int(rows, w, h) if not obb else np.zeros((len(rows), 4), dtype=int)
    out = np.empty((len(boxes), input_shape[0], input_shape[1], 3), dtype=np.uint8)
    for i, (x1, y1, x2, y2) in enumerate(boxes):
        if obb:
            crop = crop_obb(rows[i, :5], img)
        elif x2 > x1 and y2 > y1:
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
