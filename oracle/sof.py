"""Oracle restatement of the sparse-optical-flow camera-motion estimator -- TEST INFRASTRUCTURE ONLY.

Follows ``SOF.apply`` (boxmot/motion/cmc/sof.py:55-147: BoT-SORT's YAML default ``cmc_method`` and the estimator DeepOCSORT
constructs, deepocsort.py:297) with its fixed arguments: ``BaseCMC.preprocess`` (gray, scale 0.15) and ``generate_mask``
(base_cmc.py:30-105), ``goodFeaturesToTrack(maxCorners 1000, qualityLevel 0.01, minDistance 1, blockSize 3)``, ``cornerSubPix((5, 5),
30 iterations / 0.01)`` on the initialising frame, ``calcOpticalFlowPyrLK(winSize 21 x 21, maxLevel 3, 30 iterations / 0.01)`` and
``estimateAffinePartial2D(RANSAC, threshold 3)`` with the inlier test of sof.py:131-138.

The numerics live in OpenCV (opencv-python 4.11.0.86, reference uv.lock; third-party, absent offline).  They are restated here
from the published algorithms in the structure of OpenCV's sources:

* ``imgproc/src/featureselect.cpp`` + ``corner.cpp``: Sobel 3 x 3 derivatives scaled by 1 / (4 * 3 * 255), products, unnormalised 3 x 3 box
  sums (BORDER_REFLECT_101), minimum eigenvalue ``(a + c) - sqrt((a - c)^2 + b^2)`` of [[a b] [b c]] / 2, masked maximum, THRESH_TOZERO at
  ``0.01 * max``, 3 x 3 dilation local-maximum test on interior pixels, sort by (value, address) descending, first 1000 (with
  ``minDistance = 1`` the grid test ``dx^2 + dy^2 < 1`` never rejects distinct pixels).
* ``imgproc/src/cornersubpix.cpp`` + ``getRectSubPix`` (samplers.cpp): the 11 x 11 Gaussian-weighted gradient normal equations on a
  13 x 13 bilinear patch, fp64 sums, fp32 point update.
* ``video/src/lkpyramid.cpp``: ``pyrDown`` 5-tap pyramids ((sum + 128) >> 8, BORDER_REFLECT_101) while the next level stays larger than
  the window, Scharr 3/10/3 derivatives as int16, per point and level the 14-bit fixed-point bilinear patch (intensities kept at 5
  extra bits), the 2 x 2 structure tensor scaled by 2^-20, the minimum-eigenvalue test 1e-4, <= 30 Newton steps with the 0.01 stop and the
  oscillation rule, the status rules for windows leaving the image.
* ``calib3d/src/ptsetreg.cpp`` + ``levmarq.cpp``: RANSAC over 2-point similarity models with ``cv::RNG(-1)``'s multiply-with-carry
  draws (``getSubset``), squared fp32 reprojection errors against 3^2, ``RANSACUpdateNumIters(0.99)``, then <= 10 Levenberg-Marquardt
  iterations of the 4-parameter model over the inliers (``LMSolverImpl::run``).

Where OpenCV accumulates in fp32 in an order that depends on its SIMD build (box sums, the LK tensor and mismatch vector) this
restatement accumulates exactly (integers) or in a fixed order, and says so at the site.  PARITY UNPINNED against real OpenCV: it cannot be
imported here.  The device kernels (csrc/cmc_sof.hpp) are compared with THIS.
"""
from __future__ import annotations

import math

import numpy as np

from oracle.ecc import preprocess

F32 = np.float32
MAX_CORNERS, QUALITY = 1000, 0.01
WIN = 21                           # LK window
W_BITS = 14
FLT_SCALE = F32(1.0 / (1 << 20))
FLT_EPSILON = float(np.finfo(np.float32).eps)
DBL_EPSILON = float(np.finfo(np.float64).eps)
# exp(-(k / 5)^2), k = 0..5, in fp32: cornerSubPix's separable window (the literal table is shared with csrc/cmc_sof.hpp)
SUBPIX_W = np.array([1.0, 0.96078944, 0.85214376, 0.69767630, 0.52729243, 0.36787945], dtype=np.float32)


# ------------------------------------------------------------------------------------------------------------------------
# base_cmc.py:63-105
def generate_mask(h: int, w: int, dets, scale: float) -> np.ndarray:
    mask = np.zeros((h, w), dtype=np.uint8)
    y1, y2 = int(0.02 * h), int(0.98 * h)
    x1, x2 = int(0.02 * w), int(0.98 * w)
    mask[y1:y2, x1:x2] = 255
    if dets is None:
        return mask
    dets = np.asarray(dets)
    if dets.size == 0:
        return mask
    for det in dets:
        if len(det) < 4:
            continue
        x1b, y1b, x2b, y2b = (np.asarray(det[:4], dtype=np.float32) * np.float32(scale)).astype(int).tolist()
        x1b, x2b = max(0, min(w, x1b)), max(0, min(w, x2b))
        y1b, y2b = max(0, min(h, y1b)), max(0, min(h, y2b))
        if x2b > x1b and y2b > y1b:
            mask[y1b:y2b, x1b:x2b] = 0
    return mask


# ------------------------------------------------------------------------------------------------------------------------
# goodFeaturesToTrack
def _r101(a: np.ndarray, n: int = 1) -> np.ndarray:
    return np.pad(a, n, mode="reflect")


def min_eigen_map(gray: np.ndarray) -> np.ndarray:
    """cornerMinEigenVal(blockSize 3, ksize 3) of an 8-bit image, fp32."""
    s = F32(1.0 / (4.0 * 3.0 * 255.0))
    k0, k1 = F32(2.0) * s, s                                      # the smoothing kernel [1 2 1] carries the scale (Sobel())
    p = _r101(gray.astype(np.float32))
    # Dx: rows [-1 0 1] (exact), columns [1 2 1] * s
    rx = p[:, 2:] - p[:, :-2]
    dx = (rx[:-2] + rx[2:]) * k1 + rx[1:-1] * k0
    # Dy: rows [1 2 1] * s, columns [-1 0 1]
    ry = p[:, 1:-1] * k0 + (p[:, :-2] + p[:, 2:]) * k1
    dy = ry[2:] - ry[:-2]
    dx, dy = dx.astype(np.float32), dy.astype(np.float32)

    def box(c):                                                   # 3 x 3 sum, rows left to right then top to bottom (fixed order)
        q = _r101(c)
        r = (q[:, :-2] + q[:, 1:-1]) + q[:, 2:]
        return ((r[:-2] + r[1:-1]) + r[2:]).astype(np.float32)
    a = box(dx * dx) * F32(0.5)
    b = box(dx * dy)
    c = box(dy * dy) * F32(0.5)
    return ((a + c) - np.sqrt((a - c) * (a - c) + b * b)).astype(np.float32)


def good_features(gray: np.ndarray, mask: np.ndarray):
    """-> (n, 2) fp32 (x, y) corners, strongest first, or None when there is none (what the Python binding returns)."""
    h, w = gray.shape
    eig = min_eigen_map(gray)
    m = mask != 0
    max_val = float(eig[m].max()) if m.any() else 0.0             # minMaxLoc(eig, mask)
    thr = F32(max_val * QUALITY)                                  # threshold() takes a double, compares in the image's type
    eig = np.where(eig > thr, eig, F32(0)).astype(np.float32)     # THRESH_TOZERO
    q = np.pad(eig, 1, mode="constant", constant_values=-np.inf)  # dilate: border pixels do not take part
    dil = np.max(np.stack([q[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)]), axis=0)
    ok = (eig != 0) & (eig == dil) & m
    ok[0, :] = ok[-1, :] = False
    ok[:, 0] = ok[:, -1] = False
    ys, xs = np.nonzero(ok)
    if len(ys) == 0:
        return None
    addr = ys * w + xs
    order = np.lexsort((-addr, -eig[ys, xs].astype(np.float64)))  # value descending, then address descending (greaterThanPtr)
    order = order[:MAX_CORNERS]
    return np.stack([xs[order], ys[order]], axis=1).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------------
# cornerSubPix((5, 5), (-1, -1), 30 iterations / eps 0.01)
def rect_subpix(src: np.ndarray, win_w: int, win_h: int, cx: np.float32, cy: np.float32) -> np.ndarray:
    """getRectSubPix(8U -> 32F): bilinear patch around (cx, cy); outside the image the nearest row / column is repeated with the
    two-row weights only (samplers.cpp, adjustRect)."""
    h, w = src.shape
    cx = F32(cx - F32((win_w - 1) * 0.5))
    cy = F32(cy - F32((win_h - 1) * 0.5))
    ipx, ipy = int(math.floor(cx)), int(math.floor(cy))
    a, b = F32(cx - F32(ipx)), F32(cy - F32(ipy))
    one = F32(1)
    a11, a12, a21, a22 = (one - a) * (one - b), a * (one - b), (one - a) * b, a * b
    b1, b2 = one - b, b
    if 0 <= ipx < w - win_w and 0 <= ipy < h - win_h:
        rx0, rx1, ry0, ry1 = 0, win_w, 0, win_h
    else:
        rx0 = 0 if ipx >= 0 else min(-ipx, win_w)
        rx1 = win_w if ipx < w - win_w else max(w - ipx - 1, 0)
        ry0 = 0 if ipy >= 0 else min(-ipy, win_h)
        ry1 = win_h if ipy < h - win_h else max(h - ipy - 1, 0)
    s = src.astype(np.float32)
    out = np.zeros((win_h, win_w), np.float32)
    for i in range(win_h):
        r0 = min(max(ipy + i, 0), h - 1)
        r1 = r0 + 1 if (ry0 <= i < ry1) else r0
        r1 = min(r1, h - 1)
        for j in range(win_w):
            if j < rx0:
                c = min(max(ipx + rx0, 0), w - 1)
                out[i, j] = s[r0, c] * b1 + s[r1, c] * b2
            elif j < rx1:
                c = ipx + j
                out[i, j] = ((s[r0, c] * a11 + s[r0, c + 1] * a12) + s[r1, c] * a21) + s[r1, c + 1] * a22
            else:
                c = min(max(ipx + rx1, 0), w - 1)
                out[i, j] = s[r0, c] * b1 + s[r1, c] * b2
    return out


def corner_subpix(gray: np.ndarray, pts: np.ndarray, half: int = 5, max_iters: int = 30, eps: float = 0.01) -> np.ndarray:
    h, w = gray.shape
    n = 2 * half + 1
    eps2 = eps * eps
    wt = np.array([SUBPIX_W[abs(k - half)] for k in range(n)], dtype=np.float32)
    maskw = (wt[:, None] * wt[None, :]).astype(np.float32)                    # (float)(vy * exp(-x x))
    px = (np.arange(n) - half).astype(np.float64)[None, :]
    py = (np.arange(n) - half).astype(np.float64)[:, None]
    out = pts.copy()
    for k in range(len(pts)):
        ctx, cty = F32(pts[k, 0]), F32(pts[k, 1])
        cix, ciy = ctx, cty
        it = 0
        while True:
            sp = rect_subpix(gray, n + 2, n + 2, cix, ciy)
            tgx = (sp[1:-1, 2:] - sp[1:-1, :-2]).astype(np.float64)
            tgy = (sp[2:, 1:-1] - sp[:-2, 1:-1]).astype(np.float64)
            m = maskw.astype(np.float64)
            gxx, gxy, gyy = tgx * tgx * m, tgx * tgy * m, tgy * tgy * m
            a, b, c = gxx.sum(), gxy.sum(), gyy.sum()
            bb1 = (gxx * px + gxy * py).sum()
            bb2 = (gxy * px + gyy * py).sum()
            det = a * c - b * b
            if abs(det) <= DBL_EPSILON * DBL_EPSILON:
                break
            sc = 1.0 / det
            c2x = F32(float(cix) + c * sc * bb1 - b * sc * bb2)
            c2y = F32(float(ciy) - b * sc * bb1 + a * sc * bb2)
            err = float((c2x - cix) * (c2x - cix) + (c2y - ciy) * (c2y - ciy))      # fp32 products, as written in the source
            cix, ciy = c2x, c2y
            if cix < 0 or cix >= w or ciy < 0 or ciy >= h:
                break
            it += 1
            if not (it < max_iters and err > eps2):
                break
        if abs(float(cix - ctx)) > half or abs(float(ciy - cty)) > half:
            cix, ciy = ctx, cty
        out[k, 0], out[k, 1] = cix, ciy
    return out


# ------------------------------------------------------------------------------------------------------------------------
# calcOpticalFlowPyrLK
def pyr_down(img: np.ndarray) -> np.ndarray:
    h, w = img.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    p = np.pad(img.astype(np.int32), 2, mode="reflect")                        # BORDER_REFLECT_101 on the source coordinates
    cols = 2 * np.arange(ow)
    r = p[:, cols] + p[:, cols + 4] + 4 * (p[:, cols + 1] + p[:, cols + 3]) + 6 * p[:, cols + 2]
    rows = 2 * np.arange(oh)
    v = r[rows] + r[rows + 4] + 4 * (r[rows + 1] + r[rows + 3]) + 6 * r[rows + 2]
    return ((v + 128) >> 8).astype(np.uint8)


def build_pyramid(img: np.ndarray, max_level: int = 3):
    """buildOpticalFlowPyramid: a level is added while the NEXT size stays larger than the window in both directions."""
    levels = [img]
    h, w = img.shape
    for level in range(max_level + 1):
        if level != 0:
            levels.append(pyr_down(levels[-1]))
        h, w = (h + 1) // 2, (w + 1) // 2
        if w <= WIN or h <= WIN:
            break
    return levels


def scharr_deriv(img: np.ndarray):
    """calcScharrDeriv: int16 (dx, dy), reflected borders."""
    p = np.pad(img.astype(np.int32), 1, mode="reflect")
    t0 = (p[:-2] + p[2:]) * 3 + p[1:-1] * 10                                    # vertical smoothing, all columns of the padded image
    t1 = p[2:] - p[:-2]
    dx = t0[:, 2:] - t0[:, :-2]
    dy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
    return dx.astype(np.int16), dy.astype(np.int16)


def _cv_round(v) -> int:
    return int(np.rint(v))                                                      # cvRound: round half to even


def _weights(a: np.float32, b: np.float32):
    one, sc = F32(1), F32(1 << W_BITS)
    iw00 = _cv_round((one - a) * (one - b) * sc)
    iw01 = _cv_round(a * (one - b) * sc)
    iw10 = _cv_round((one - a) * b * sc)
    return iw00, iw01, iw10, (1 << W_BITS) - iw00 - iw01 - iw10


def _descale(v, n):
    return (v + (1 << (n - 1))) >> n


def lk_track(prev_pyr, next_pyr, pts: np.ndarray, max_count: int = 30, eps: float = 0.01, min_eig: float = 1e-4):
    """-> (next_pts (n, 2) fp32, status (n,) uint8)."""
    n = len(pts)
    L = min(len(prev_pyr), len(next_pyr)) - 1
    eps2 = eps * eps
    half = F32((WIN - 1) * 0.5)
    nxt = np.zeros((n, 2), np.float32)
    status = np.ones(n, np.uint8)
    B = WIN                                                                     # border of the pyramid images
    for level in range(L, -1, -1):
        I = prev_pyr[level]
        J = next_pyr[level]
        rows, cols = I.shape
        Ib = np.pad(I.astype(np.int64), B, mode="reflect")                      # pyrBorder = BORDER_REFLECT_101
        Jb = np.pad(J.astype(np.int64), B, mode="reflect")
        dx, dy = scharr_deriv(I)
        dxb = np.pad(dx.astype(np.int64), B, mode="constant")                   # derivBorder = BORDER_CONSTANT
        dyb = np.pad(dy.astype(np.int64), B, mode="constant")
        inv = F32(1.0 / (1 << level))
        for k in range(n):
            ppx, ppy = F32(pts[k, 0] * inv), F32(pts[k, 1] * inv)
            if level == L:
                nx, ny = ppx, ppy
            else:
                nx, ny = F32(nxt[k, 0] * F32(2)), F32(nxt[k, 1] * F32(2))
            nxt[k] = (nx, ny)
            ppx, ppy = F32(ppx - half), F32(ppy - half)
            ix, iy = int(math.floor(ppx)), int(math.floor(ppy))
            if ix < -WIN or ix >= cols or iy < -WIN or iy >= rows:
                if level == 0:
                    status[k] = 0
                continue
            a, b = F32(ppx - F32(ix)), F32(ppy - F32(iy))
            w00, w01, w10, w11 = _weights(a, b)
            y0, x0 = iy + B, ix + B

            def patch(img, sh):
                v = (img[y0:y0 + WIN, x0:x0 + WIN] * w00 + img[y0:y0 + WIN, x0 + 1:x0 + WIN + 1] * w01
                     + img[y0 + 1:y0 + WIN + 1, x0:x0 + WIN] * w10 + img[y0 + 1:y0 + WIN + 1, x0 + 1:x0 + WIN + 1] * w11)
                return _descale(v, sh)
            Ip = patch(Ib, W_BITS - 5)
            Ix = patch(dxb, W_BITS)
            Iy = patch(dyb, W_BITS)
            # OpenCV adds the products in fp32 (SIMD-order dependent); here the sums are exact integers, converted once
            A11 = F32(F32(int((Ix * Ix).sum())) * FLT_SCALE)
            A12 = F32(F32(int((Ix * Iy).sum())) * FLT_SCALE)
            A22 = F32(F32(int((Iy * Iy).sum())) * FLT_SCALE)
            D = F32(A11 * A22 - A12 * A12)
            me = F32((A22 + A11 - np.sqrt((A11 - A22) * (A11 - A22) + F32(4) * A12 * A12)) / F32(2 * WIN * WIN))
            if float(me) < min_eig or float(D) < FLT_EPSILON:
                if level == 0:
                    status[k] = 0
                continue
            D = F32(F32(1) / D)
            nx, ny = F32(nx - half), F32(ny - half)
            pdx = pdy = F32(0)
            for j in range(max_count):
                jx, jy = int(math.floor(nx)), int(math.floor(ny))
                if jx < -WIN or jx >= cols or jy < -WIN or jy >= rows:
                    if level == 0:
                        status[k] = 0
                    break
                a, b = F32(nx - F32(jx)), F32(ny - F32(jy))
                w00, w01, w10, w11 = _weights(a, b)
                yy, xx = jy + B, jx + B
                Jp = _descale(Jb[yy:yy + WIN, xx:xx + WIN] * w00 + Jb[yy:yy + WIN, xx + 1:xx + WIN + 1] * w01
                              + Jb[yy + 1:yy + WIN + 1, xx:xx + WIN] * w10 + Jb[yy + 1:yy + WIN + 1, xx + 1:xx + WIN + 1] * w11, W_BITS - 5)
                diff = Jp - Ip
                b1 = F32(F32(int((diff * Ix).sum())) * FLT_SCALE)
                b2 = F32(F32(int((diff * Iy).sum())) * FLT_SCALE)
                ddx = F32((A12 * b2 - A22 * b1) * D)
                ddy = F32((A12 * b1 - A11 * b2) * D)
                nx, ny = F32(nx + ddx), F32(ny + ddy)
                nxt[k] = (F32(nx + half), F32(ny + half))
                if float(ddx) * float(ddx) + float(ddy) * float(ddy) <= eps2:
                    break
                if j > 0 and abs(float(F32(ddx + pdx))) < 0.01 and abs(float(F32(ddy + pdy))) < 0.01:
                    nxt[k, 0] = F32(nxt[k, 0] - ddx * F32(0.5))
                    nxt[k, 1] = F32(nxt[k, 1] - ddy * F32(0.5))
                    break
                pdx, pdy = ddx, ddy
            if status[k] and level == 0:                                        # the error pass re-tests the final window
                fx, fy = F32(nxt[k, 0] - half), F32(nxt[k, 1] - half)
                jx, jy = int(math.floor(fx)), int(math.floor(fy))
                if jx < -WIN or jx >= cols or jy < -WIN or jy >= rows:
                    status[k] = 0
    return nxt, status


# ------------------------------------------------------------------------------------------------------------------------
# estimateAffinePartial2D(RANSAC, 3.0, 2000, 0.99, 10)
class CvRng:
    """cv::RNG: multiply-with-carry, coefficient 4164903690."""

    def __init__(self, state: int = 0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self) -> int:
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a: int, b: int) -> int:
        return a if a == b else self.next() % (b - a) + a


def _partial_model(f0, f1, t0, t1):
    """AffinePartial2DEstimatorCallback::runKernel: the similarity through two correspondences, fp64; -> (6,) row-major 2 x 3."""
    x1, y1, x2, y2 = float(f0[0]), float(f0[1]), float(f1[0]), float(f1[1])
    X1, Y1, X2, Y2 = float(t0[0]), float(t0[1]), float(t1[0]), float(t1[1])
    with np.errstate(all="ignore"):
        d = np.float64(1.0) / np.float64((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2))
        S0 = d * ((X1 - X2) * (x1 - x2) + (Y1 - Y2) * (y1 - y2))
        S1 = d * ((Y1 - Y2) * (x1 - x2) - (X1 - X2) * (y1 - y2))
        S2 = d * ((Y1 - Y2) * (x1 * y2 - x2 * y1) - (X1 * y2 - X2 * y1) * (y1 - y2) - (X1 * x2 - X2 * x1) * (x1 - x2))
        S3 = d * (-(X1 - X2) * (x1 * y2 - x2 * y1) - (Y1 * x2 - Y2 * x1) * (x1 - x2) - (Y1 * y2 - Y2 * y1) * (y1 - y2))
    return np.array([S0, -S1, S2, S1, S0, S3], dtype=np.float64)


def _errors(model, frm, to):
    F = model.astype(np.float32)
    with np.errstate(all="ignore"):
        a = ((F[0] * frm[:, 0] + F[1] * frm[:, 1]) + F[2]) - to[:, 0]
        b = ((F[3] * frm[:, 0] + F[4] * frm[:, 1]) + F[5]) - to[:, 1]
        return (a * a + b * b).astype(np.float32)


def _update_iters(p: float, ep: float, model_points: int, max_iters: int) -> int:
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = math.log(num), math.log(denom)
    return max_iters if denom >= 0 or -num >= max_iters * (-denom) else _cv_round(num / denom)


def lm_refine(h4: np.ndarray, src: np.ndarray, dst: np.ndarray, max_iters: int = 10) -> np.ndarray:
    """LMSolverImpl::run with AffinePartial2DRefineCallback; parameters (a, b, tx, ty) of [[a -b tx] [b a ty]], eps = FLT_EPSILON."""
    eps = FLT_EPSILON
    Mx, My = src[:, 0].astype(np.float64), src[:, 1].astype(np.float64)
    mx, my = dst[:, 0].astype(np.float64), dst[:, 1].astype(np.float64)
    n = len(src)
    J = np.zeros((2 * n, 4))
    J[0::2, 0], J[0::2, 1], J[0::2, 2] = Mx, -My, 1.0
    J[1::2, 0], J[1::2, 1], J[1::2, 3] = My, Mx, 1.0

    def resid(h):
        r = np.empty(2 * n)
        r[0::2] = h[0] * Mx - h[1] * My + h[2] - mx
        r[1::2] = h[1] * Mx + h[0] * My + h[3] - my
        return r
    x = h4.astype(np.float64).copy()
    r = resid(x)
    S = float(r @ r)
    A = J.T @ J
    v = J.T @ r
    Dg = np.diag(A).copy()
    Rlo, Rhi = 0.25, 0.75
    lam, lc = 1.0, 0.75
    it = 0
    while True:
        Ap = A + np.diag(lam * Dg)
        d = np.linalg.lstsq(Ap, v, rcond=None)[0]                   # solve(Ap, v, d, DECOMP_EIG)
        xd = x - d
        rd = resid(xd)
        Sd = float(rd @ rd)
        dS = float(d @ (2.0 * v - A @ d))
        R = (S - Sd) / (dS if abs(dS) > DBL_EPSILON else 1.0)
        if R > Rhi:
            lam *= 0.5
            if lam < lc:
                lam = 0.0
        elif R < Rlo:
            t = float(d @ v)
            nu = (Sd - S) / (t if abs(t) > DBL_EPSILON else 1.0) + 2.0
            nu = min(max(nu, 2.0), 10.0)
            if lam == 0.0:
                Ai = np.linalg.pinv(A)
                maxval = max(DBL_EPSILON, float(np.abs(np.diag(Ai)).max()))
                lam = lc = 1.0 / maxval
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S = Sd
            x = xd
            r = resid(x)
            v = J.T @ r
        it += 1
        if not (it < max_iters and float(np.abs(d).max()) >= eps and float(np.abs(r).max()) >= eps):
            break
    return x


def estimate_affine_partial_2d(frm: np.ndarray, to: np.ndarray, thresh: float = 3.0, max_iters: int = 2000, confidence: float = 0.99,
                               refine_iters: int = 10):
    """-> (H (2, 3) fp64 or None, inlier mask (n,) uint8)."""
    frm, to = np.asarray(frm, np.float32), np.asarray(to, np.float32)
    count = len(frm)
    if count < 2:
        return None, np.zeros(count, np.uint8)
    rng = CvRng()
    t2 = F32(thresh * thresh)
    niters = max_iters
    best, best_mask, best_count = None, None, 0
    it = 0
    while it < niters:
        if count > 2:
            idx = [0, 0]
            i = attempts = 0
            while i < 2 and attempts < 10000:
                idx[i] = rng.uniform(0, count)
                if i == 1 and idx[1] == idx[0]:
                    continue
                i += 1
            model = _partial_model(frm[idx[0]], frm[idx[1]], to[idx[0]], to[idx[1]])
        else:
            model = _partial_model(frm[0], frm[1], to[0], to[1])
        err = _errors(model, frm, to)
        m = err <= t2
        good = int(m.sum())
        if good > max(best_count, 1):
            best, best_mask, best_count = model, m, good
            niters = _update_iters(confidence, (count - good) / count, 2, niters)
        it += 1
    if best is None:
        return None, np.zeros(count, np.uint8)
    H = best.copy()
    if count > 2 and refine_iters:
        h4 = lm_refine(np.array([H[0], H[3], H[2], H[5]]), frm[best_mask], to[best_mask], refine_iters)
        H = np.array([h4[0], -h4[1], h4[2], h4[1], h4[0], h4[3]])
    return H.reshape(2, 3), best_mask.astype(np.uint8)


# ------------------------------------------------------------------------------------------------------------------------
class SofOracle:
    """``SOF`` with the reference's default arguments: ``apply(img, dets) -> 2 x 3 float32`` (sof.py:55-129)."""

    def __init__(self, scale: float = 0.15, min_inliers: int = 8, min_inlier_ratio: float = 0.2, ransac_reproj_threshold: float = 3.0):
        self.scale, self.min_inliers, self.min_inlier_ratio = float(scale), int(min_inliers), float(min_inlier_ratio)
        self.thresh = float(ransac_reproj_threshold)
        self.prev_frame = self.prev_pyr = self.prev_keypoints = None
        self.initialized = False
        self.last = {}

    def _detect(self, gray, dets):
        return good_features(gray, generate_mask(gray.shape[0], gray.shape[1], dets, self.scale))

    def apply(self, img, dets=None) -> np.ndarray:
        gray = preprocess(img, self.scale)
        H = np.eye(2, 3, dtype=np.float32)
        self.last = {}
        if not self.initialized or self.prev_frame is None or self.prev_keypoints is None:
            kps = self._detect(gray, dets)
            self.prev_frame = gray
            if kps is None or len(kps) < 4:
                self.prev_keypoints, self.initialized = kps, False
                return H
            self.prev_keypoints, self.initialized = corner_subpix(gray, kps), True
            return H
        nxt, status = lk_track(build_pyramid(self.prev_frame), build_pyramid(gray), self.prev_keypoints)
        pv, nv = self.prev_keypoints[status == 1], nxt[status == 1]
        self.last = dict(next=nxt, status=status)
        if len(pv) < 4:                                             # _reset
            kps = self._detect(gray, dets)
            self.prev_frame, self.prev_keypoints = gray, kps
            self.initialized = kps is not None and len(kps) >= 4
            return H
        He, inl = estimate_affine_partial_2d(pv, nv, self.thresh)
        n_in = int(np.count_nonzero(inl))
        self.last.update(inliers=n_in, matches=len(pv))
        if He is not None and n_in >= self.min_inliers and n_in / len(pv) >= self.min_inlier_ratio:
            H = He.astype(np.float32)
            if self.scale < 1.0:
                H[0, 2] /= np.float32(self.scale)
                H[1, 2] /= np.float32(self.scale)
        kps = self._detect(gray, dets)
        if kps is None or len(kps) < 4:
            kps = nv
        self.prev_frame, self.prev_keypoints, self.initialized = gray, kps.copy(), True
        return H
