"""Oracle restatement of the ECC camera-motion estimator -- TEST INFRASTRUCTURE ONLY.

Follows ``ECC.apply`` (boxmot/motion/cmc/ecc.py:45-96) with the arguments StrongSORT and ``get_cmc_method("ecc")`` use
(MOTION_TRANSLATION, eps 1e-5, 100 iterations, scale 0.15, grayscale, no alignment; strongsort.py:63, ecc.py:23-31) and
``BaseCMC.preprocess`` (base_cmc.py:30-61: ``cv2.cvtColor(BGR2GRAY)`` then ``cv2.resize(fx = fy = 0.15, INTER_LINEAR)``).

The numerical core is OpenCV's ``cv2.findTransformECC`` (opencv-python 4.11.0.86, reference uv.lock:3847-3848; third-party,
absent offline): restated from the published algorithm (Evangelidis & Psarakis, PAMI 2008) in the structure of
modules/video/src/ecc.cpp -- float32 images, [-0.5 0 0.5] central-difference gradients (BORDER_REFLECT_101), per iteration
``warpAffine(INTER_LINEAR | WARP_INVERSE_MAP)`` of the image and its two gradients + a nearest-neighbour warp of the all-ones
mask, masked mean / std, zero-mean correlation rho, the 2 x 2 Gauss-Newton system of the translation Jacobian, the lambda of the
illumination model, ``map[:, 2] += deltaP``; stop when ``|rho - last_rho| < eps`` -- and ``cv2.warpAffine``'s fixed-point
coordinate grid (AB_BITS = 10, INTER_BITS = 5: sampling positions rounded to 1/32 pixel, bilinear weights from that fraction,
constant-zero border).  ``COLOR_BGR2GRAY`` is the 14-bit fixed-point ``(B 1868 + G 9617 + R 4899 + 8192) >> 14``.
PARITY UNPINNED against real OpenCV: it cannot be imported here.  The device kernel (csrc/cmc_ecc.hpp) is compared with THIS.
"""
from __future__ import annotations

import numpy as np

from oracle.crops import cv2_resize_linear_u8

AB_BITS, INTER_BITS = 10, 5
AB_SCALE = 1 << AB_BITS
INTER_TAB = 1 << INTER_BITS


def bgr2gray_u8(img: np.ndarray) -> np.ndarray:
    """color.cpp RGB2Gray<uchar>: coefficients B 1868, G 9617, R 4899 at 14 bits, rounded."""
    b, g, r = (img[..., k].astype(np.int32) for k in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def preprocess(img: np.ndarray, scale: float = 0.15) -> np.ndarray:
    gray = bgr2gray_u8(np.asarray(img, dtype=np.uint8))
    h, w = gray.shape
    dw, dh = int(np.rint(w * scale)), int(np.rint(h * scale))          # saturate_cast<int>(ssize * fx)
    return cv2_resize_linear_u8(gray, (dw, dh), (1.0 / scale, 1.0 / scale))


def gradients(im: np.ndarray):
    """filter2D with [-0.5 0 0.5] (and its transpose), BORDER_REFLECT_101."""
    p = np.pad(im, 1, mode="reflect")
    gx = (np.float32(0.5) * p[1:-1, 2:] - np.float32(0.5) * p[1:-1, :-2]).astype(np.float32)
    gy = (np.float32(0.5) * p[2:, 1:-1] - np.float32(0.5) * p[:-2, 1:-1]).astype(np.float32)
    return gx, gy


def _grid(tx: float, ty: float, w: int, h: int, nearest: bool):
    """warpAffine's integer coordinate grid for M = [[1 0 tx] [0 1 ty]] (WARP_INVERSE_MAP: dst(x, y) = src(x + tx, y + ty))."""
    shift = AB_BITS if nearest else AB_BITS - INTER_BITS
    rd = AB_SCALE // 2 if nearest else AB_SCALE // INTER_TAB // 2
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    adelta = np.rint(1.0 * x * AB_SCALE).astype(np.int64)              # saturate_cast<int>(M[0] * x * AB_SCALE)
    X0 = np.rint((0.0 * y + float(tx)) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((1.0 * y + float(ty)) * AB_SCALE).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> shift
    Y = (Y0[:, None] + np.zeros(w, np.int64)[None, :]) >> shift
    return X, Y


def warp_linear(src: np.ndarray, tx, ty) -> np.ndarray:
    h, w = src.shape
    X, Y = _grid(tx, ty, w, h, nearest=False)
    ix, iy = X >> INTER_BITS, Y >> INTER_BITS
    fx = ((X & (INTER_TAB - 1)).astype(np.float32) / np.float32(INTER_TAB)).astype(np.float32)
    fy = ((Y & (INTER_TAB - 1)).astype(np.float32) / np.float32(INTER_TAB)).astype(np.float32)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok, src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], np.float32(0)).astype(np.float32)
    one = np.float32(1)
    w00, w01, w10, w11 = (one - fx) * (one - fy), fx * (one - fy), (one - fx) * fy, fx * fy
    out = tap(iy, ix) * w00 + tap(iy, ix + 1) * w01 + tap(iy + 1, ix) * w10 + tap(iy + 1, ix + 1) * w11
    return out.astype(np.float32)


def warp_mask(w: int, h: int, tx, ty) -> np.ndarray:
    X, Y = _grid(tx, ty, w, h, nearest=True)
    return (X >= 0) & (X < w) & (Y >= 0) & (Y < h)


def find_transform_ecc_translation(template_u8: np.ndarray, image_u8: np.ndarray, eps: float = 1e-5, max_iter: int = 100):
    """Returns (rho, (tx, ty), iterations) or raises RuntimeError for OpenCV's StsNoConv exits."""
    tmpl = template_u8.astype(np.float32)
    img = image_u8.astype(np.float32)
    h, w = tmpl.shape
    gx, gy = gradients(img)
    tx = ty = np.float32(0.0)                                       # the map is CV_32F
    rho, last_rho = -1.0, -float(eps)
    it = 0
    for it in range(1, max_iter + 1):
        if abs(rho - last_rho) < eps:
            it -= 1
            break
        iw = warp_linear(img, tx, ty)
        gxw, gyw = warp_linear(gx, tx, ty), warp_linear(gy, tx, ty)
        m = warp_mask(w, h, tx, ty)
        n = int(m.sum())
        im_mean = float(iw[m].astype(np.float64).mean()) if n else 0.0
        tm_mean = float(tmpl[m].astype(np.float64).mean()) if n else 0.0
        im_std = float(np.sqrt(max((iw[m].astype(np.float64) ** 2).mean() - im_mean ** 2, 0.0))) if n else 0.0
        tm_std = float(np.sqrt(max((tmpl[m].astype(np.float64) ** 2).mean() - tm_mean ** 2, 0.0))) if n else 0.0
        iwz = np.where(m, (iw.astype(np.float64) - im_mean).astype(np.float32), iw).astype(np.float32)
        tz = np.where(m, (tmpl.astype(np.float64) - tm_mean).astype(np.float32), np.float32(0)).astype(np.float32)
        tmp_norm = np.sqrt(n * tm_std * tm_std)
        img_norm = np.sqrt(n * im_std * im_std)

        def dot(a, b):
            return float((a.astype(np.float64) * b.astype(np.float64)).sum())
        H = np.array([[dot(gxw, gxw), dot(gxw, gyw)], [dot(gxw, gyw), dot(gyw, gyw)]], dtype=np.float32)
        det = np.float64(H[0, 0]) * np.float64(H[1, 1]) - np.float64(H[0, 1]) * np.float64(H[1, 0])
        if det == 0.0:
            Hinv = np.zeros((2, 2), np.float32)
        else:
            Hinv = (np.array([[H[1, 1], -H[0, 1]], [-H[1, 0], H[0, 0]]], dtype=np.float64) / det).astype(np.float32)
        corr = dot(tz, iwz)
        last_rho = rho
        rho = corr / (img_norm * tmp_norm) if img_norm * tmp_norm != 0.0 else float("nan")
        if np.isnan(rho):
            raise RuntimeError("ECC: NaN encountered (StsNoConv)")
        ip = np.array([dot(gxw, iwz), dot(gyw, iwz)], dtype=np.float32)
        tp = np.array([dot(gxw, tz), dot(gyw, tz)], dtype=np.float32)
        iph = (Hinv.astype(np.float64) @ ip.astype(np.float64)).astype(np.float32)
        lambda_n = img_norm * img_norm - float(ip.astype(np.float64) @ iph.astype(np.float64))
        lambda_d = corr - float(tp.astype(np.float64) @ iph.astype(np.float64))
        if lambda_d <= 0.0:
            raise RuntimeError("ECC: the correlation is going to be minimized (StsNoConv)")
        lam = lambda_n / lambda_d
        err = (np.float32(lam) * tz - iwz).astype(np.float32)
        ep = np.array([dot(gxw, err), dot(gyw, err)], dtype=np.float32)
        dp = (Hinv.astype(np.float64) @ ep.astype(np.float64)).astype(np.float32)
        tx = np.float32(tx + dp[0])
        ty = np.float32(ty + dp[1])
    return rho, (float(tx), float(ty)), it


class EccOracle:
    """``ECC`` with the reference's default arguments: ``apply(img, dets) -> 2 x 3 float32`` (ecc.py:45-96)."""

    def __init__(self, eps: float = 1e-5, max_iter: int = 100, scale: float = 0.15):
        self.eps, self.max_iter, self.scale = float(eps), int(max_iter), float(scale)
        self.prev = None
        self.last_iterations = 0

    def apply(self, img, dets=None) -> np.ndarray:
        warp = np.eye(2, 3, dtype=np.float32)
        if self.prev is None:
            self.prev = preprocess(img, self.scale)
            return warp
        curr = preprocess(img, self.scale)
        try:
            _, (tx, ty), self.last_iterations = find_transform_ecc_translation(self.prev, curr, self.eps, self.max_iter)
        except RuntimeError:
            self.prev = curr                         # ecc.py:67-76: StsNoConv -> identity
            return warp
        warp[0, 2], warp[1, 2] = np.float32(tx), np.float32(ty)
        if self.scale < 1.0:
            warp[0, 2] /= np.float32(self.scale)
            warp[1, 2] /= np.float32(self.scale)
        self.prev = curr
        return warp
