"""Oriented-box (OBB) pieces of the tracker path -- TEST INFRASTRUCTURE ONLY: the checker of the oriented frame step
(boxmot_amd/csrc/botsort_step_body.hpp compiled with BM_OBB).  The parts of the reference's OBB path that are
plain arithmetic, restated and pinned on the reference classes.

Follows:
  * KalmanFilterXYWH(ndim=5)       boxmot/motion/kalman_filters/xywh.py:16-206 over base.py:116-355 (initiate, multi_predict, update with
                                   the representation alignment of the measurement, theta-velocity damping, angle wrap)
  * iou_batch_obb / _iou_obb_matrix boxmot/trackers/association/iou.py:5-115 (AABB pre-filter, then the rotated intersection)
  * STrack's OBB accessors         boxmot/trackers/bbox/bytetrack/bytetrack.py:45-54, 147-198 (xywha as fp32 of the filter mean)

Camera-motion compensation of oriented tracks (STrack.multi_gmc_obb, botsort_track.py:134-230) is restated here for the oracle only --
the device steps refuse a warp on an oriented handle; it goes through two more OpenCV calls, cv2.transform and
cv2.minAreaRect, restated below and UNPINNED like the intersection area.

PARITY UNPINNED for one piece: the reference gets the intersection polygon from cv2.rotatedRectangleIntersection + cv2.contourArea
(OpenCV is absent offline).  `rotated_intersection_area` computes the same quantity by clipping one rectangle with the other's four
half-planes (Sutherland-Hodgman) in fp64 and the shoelace formula -- not OpenCV's edge-intersection enumeration in fp32 -- so areas
agree with OpenCV's to its fp32 rounding (~1e-6 relative), not bit for bit.  tests/test_oracle_obb.py pins everything else on the
reference classes with this function standing where the two cv2 calls stand.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

STD_POS, STD_VEL = 1.0 / 20, 1.0 / 160       # base.py:60-65
F10 = np.eye(10)
for _i in range(5):
    F10[_i, 5 + _i] = 1.0
H5 = np.eye(5, 10)


def wrap_angle(a):                               # base.py:116-120
    w = (np.asarray(a, dtype=float) + np.pi) % (2.0 * np.pi) - np.pi
    return float(w) if np.isscalar(a) else w


def _align_to(angle, ref):                       # base.py:122-124
    return float(ref + wrap_angle(float(angle) - float(ref)))


def align_obb_measurement(z, ref):               # xywh.py:85-123, base.py:131-157
    """(w, h, theta) ~ (w, h, theta + pi) ~ (h, w, theta +- pi/2): the parameterisation of the measured rectangle closest to the state."""
    out = np.asarray(z, dtype=float).copy().reshape(-1)
    ref = np.asarray(ref, dtype=float).reshape(-1)
    ref_w, ref_h, ref_t = max(float(ref[2]), 1e-6), max(float(ref[3]), 1e-6), float(ref[4])
    w, h, t = max(float(out[2]), 1e-6), max(float(out[3]), 1e-6), float(out[4])
    best_cost, best = float("inf"), None
    for s0, s1, th in ((w, h, t), (w, h, t + np.pi), (h, w, t + (np.pi / 2.0)), (h, w, t - (np.pi / 2.0))):
        s0, s1 = max(float(s0), 1e-6), max(float(s1), 1e-6)
        ta = _align_to(th, ref_t)
        cost = abs(ta - ref_t) + (0.05 * (abs(np.log(s0 / ref_w)) + abs(np.log(s1 / ref_h))))
        if cost < best_cost:
            best_cost, best = cost, (s0, s1, ta)
    out[2], out[3], out[4] = best
    return out


def _enforce(mean):                              # xywh.py:125-131 over base.py:160-180 (1-d state)
    mean[2] = max(float(mean[2]), 1e-4)
    mean[3] = max(float(mean[3]), 1e-4)
    mean[4] = float(wrap_angle(mean[4]))
    return mean


def kf5_initiate(xywha):                         # xywh.py:133-140 over base.py:234-244, std xywh.py:22-36
    m = np.asarray(xywha, dtype=float).copy()
    m[4] = wrap_angle(m[4])
    mean = np.r_[m, np.zeros_like(m)]
    std = [2 * STD_POS * m[2], 2 * STD_POS * m[3], 2 * STD_POS * m[2], 2 * STD_POS * m[3],
           10 * STD_VEL * m[2], 10 * STD_VEL * m[3], 10 * STD_VEL * m[2], 10 * STD_VEL * m[3]]
    std.insert(4, 1e-2)
    std.append(1e-5)
    return _enforce(mean), np.diag(np.square(std))


def kf5_multi_predict(mean, cov):                # xywh.py:147-160 over base.py:311-327, std xywh.py:66-83
    std_pos = [STD_POS * mean[:, 2], STD_POS * mean[:, 3], STD_POS * mean[:, 2], STD_POS * mean[:, 3], 1e-2 * np.ones_like(mean[:, 2])]
    std_vel = [STD_VEL * mean[:, 2], STD_VEL * mean[:, 3], STD_VEL * mean[:, 2], STD_VEL * mean[:, 3], 1e-5 * np.ones_like(mean[:, 2])]
    sqr = np.square(np.r_[std_pos, std_vel]).T
    motion_cov = np.asarray([np.diag(sqr[i]) for i in range(len(mean))])
    mean = np.dot(mean, F10.T)
    left = np.dot(F10, cov).transpose((1, 0, 2))
    cov = np.dot(left, F10.T) + motion_cov
    mean[:, 2] = np.maximum(mean[:, 2], 1e-4)
    mean[:, 3] = np.maximum(mean[:, 3], 1e-4)
    mean[:, 4] = wrap_angle(mean[:, 4])
    return mean, cov


def kf5_update(mean, cov, xywha):                # xywh.py:162-185 over base.py:286-355, std xywh.py:56-64
    z = align_obb_measurement(np.asarray(xywha, dtype=float).copy().reshape(5), np.asarray(mean, dtype=float))
    std = [STD_POS * mean[2], STD_POS * mean[3], STD_POS * mean[2], STD_POS * mean[3], 1e-1]
    std = [(1 - 0.0) * x for x in std]
    projected_mean = np.dot(H5, mean)
    projected_cov = np.linalg.multi_dot((H5, cov, H5.T)) + np.diag(np.square(std))
    chol, lower = scipy.linalg.cho_factor(projected_cov, lower=True, check_finite=False)
    gain = scipy.linalg.cho_solve((chol, lower), np.dot(cov, H5.T).T, check_finite=False).T
    new_mean = mean + np.dot(z - projected_mean, gain.T)
    new_cov = cov - np.linalg.multi_dot((gain, projected_cov, gain.T))
    new_mean[9] *= float(np.clip(0.8, 0.0, 1.0))             # _damp_theta_velocity, base.py:222-232
    return _enforce(new_mean), new_cov


# ---- rotated rectangles ----
def box_points(cx, cy, w, h, angle_deg):
    """cv2.boxPoints / RotatedRect::points: the four corners (bottom-left, top-left, top-right, bottom-right of the unrotated box),
    fp32 like OpenCV's."""
    a = np.float64(angle_deg) * np.pi / 180.0
    b = np.float32(np.cos(a)) * np.float32(0.5)
    s = np.float32(np.sin(a)) * np.float32(0.5)
    cx, cy, w, h = np.float32(cx), np.float32(cy), np.float32(w), np.float32(h)
    p0 = (cx - s * h - b * w, cy + b * h - s * w)
    p1 = (cx + s * h - b * w, cy - b * h - s * w)
    p2 = (np.float32(2) * cx - p0[0], np.float32(2) * cy - p0[1])
    p3 = (np.float32(2) * cx - p1[0], np.float32(2) * cy - p1[1])
    return np.array([p0, p1, p2, p3], dtype=np.float32)


def rotated_intersection_area(r1, r2):
    """Area of the intersection of two rotated rectangles ((cx, cy), (w, h), angle in degrees) -- the quantity the reference takes from
    cv2.rotatedRectangleIntersection + cv2.contourArea (iou.py:97-101).  Clipping of r1's corners by r2's half-planes, fp64."""
    p = box_points(r1[0][0], r1[0][1], r1[1][0], r1[1][1], r1[2]).astype(np.float64)
    q = box_points(r2[0][0], r2[0][1], r2[1][0], r2[1][1], r2[2]).astype(np.float64)
    if 0.5 * abs(sum(q[i][0] * q[(i + 1) % 4][1] - q[(i + 1) % 4][0] * q[i][1] for i in range(4))) == 0.0:
        return 0.0
    # orientation of the clip polygon decides which side is inside
    orient = np.sign(sum((q[(i + 1) % 4][0] - q[i][0]) * (q[(i + 1) % 4][1] + q[i][1]) for i in range(4)))      # > 0: clockwise (y down)
    poly = [tuple(v) for v in p]
    for i in range(4):
        a, b = q[i], q[(i + 1) % 4]
        ex, ey = b[0] - a[0], b[1] - a[1]

        def side(v, a=a, ex=ex, ey=ey):
            return (ex * (v[1] - a[1]) - ey * (v[0] - a[0])) * (-orient)
        out = []
        for k in range(len(poly)):
            cur, nxt = poly[k], poly[(k + 1) % len(poly)]
            sc, sn = side(cur), side(nxt)
            if sc >= 0:
                out.append(cur)
            if (sc > 0 and sn < 0) or (sc < 0 and sn > 0):
                t = sc / (sc - sn)
                out.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
        poly = out
        if len(poly) < 3:
            return 0.0
    return 0.5 * abs(sum(poly[k][0] * poly[(k + 1) % len(poly)][1] - poly[(k + 1) % len(poly)][0] * poly[k][1] for k in range(len(poly))))


def iou_obb_matrix(b1, b2):                      # iou.py:38-115 (N, 5) x (M, 5) [cx, cy, w, h, angle in radians]
    b1, b2 = np.asarray(b1, dtype=float).reshape(-1, 5), np.asarray(b2, dtype=float).reshape(-1, 5)
    N, M = len(b1), len(b2)
    out = np.zeros((N, M), dtype=np.float64)
    if N == 0 or M == 0:
        return out
    hw1, hh1, c1, s1 = b1[:, 2] / 2, b1[:, 3] / 2, np.abs(np.cos(b1[:, 4])), np.abs(np.sin(b1[:, 4]))
    hw2, hh2, c2, s2 = b2[:, 2] / 2, b2[:, 3] / 2, np.abs(np.cos(b2[:, 4])), np.abs(np.sin(b2[:, 4]))
    ex1, ey1, ex2, ey2 = hw1 * c1 + hh1 * s1, hw1 * s1 + hh1 * c1, hw2 * c2 + hh2 * s2, hw2 * s2 + hh2 * c2
    cand = (np.abs(b1[:, None, 0] - b2[None, :, 0]) < ex1[:, None] + ex2[None, :]) & (np.abs(b1[:, None, 1] - b2[None, :, 1]) < ey1[:, None] + ey2[None, :])
    a1, a2 = b1[:, 2] * b1[:, 3], b2[:, 2] * b2[:, 3]
    d1, d2 = np.degrees(b1[:, 4]), np.degrees(b2[:, 4])
    for i, j in zip(*np.nonzero(cand)):
        inter = rotated_intersection_area(((float(b1[i, 0]), float(b1[i, 1])), (float(b1[i, 2]), float(b1[i, 3])), float(d1[i])),
                                          ((float(b2[j, 0]), float(b2[j, 1])), (float(b2[j, 2]), float(b2[j, 3])), float(d2[j])))
        if inter <= 0.0:
            continue
        union = a1[i] + a2[j] - inter
        if union > 0:
            out[i, j] = inter / union
    return out


# ---- camera-motion compensation of oriented tracks (botsort_track.py:134-230) ----
def transform_points(pts, m):
    """cv2.transform of (N, 1, 2) fp32 points with a 2x3 fp32 matrix: x' = m00 x + m01 y + m02, fp32 (UNPINNED: OpenCV's own
    evaluation order / fused operations are not checkable offline)."""
    p = np.asarray(pts, dtype=np.float32).reshape(-1, 2)
    m = np.asarray(m, dtype=np.float32)
    x, y = p[:, 0], p[:, 1]
    return np.stack([m[0, 0] * x + m[0, 1] * y + m[0, 2], m[1, 0] * x + m[1, 1] * y + m[1, 2]], axis=1).astype(np.float32)


def min_area_rect(pts):
    """cv2.minAreaRect of the four corners of a warped rectangle: ((cx, cy), (w, h), angle in degrees).  A minimum-area enclosing
    rectangle has a side collinear with an edge of the convex hull (rotating calipers): every edge of the quadrilateral is tried, the
    first smallest area wins, `w` runs along that edge.  UNPINNED: OpenCV's fp32 calipers and its angle / side-order convention are
    not checkable offline -- the caller (_corners_to_xywha) re-aligns (w, h, angle) to the track's previous box, which removes the
    convention but not the rounding."""
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    best = None
    for i in range(len(p)):
        e = p[(i + 1) % len(p)] - p[i]
        n = np.hypot(e[0], e[1])
        if n == 0.0:
            continue
        u = e / n
        v = np.array([-u[1], u[0]])
        a, b = p[:, 0] * u[0] + p[:, 1] * u[1], p[:, 0] * v[0] + p[:, 1] * v[1]      # (elementwise: a defined rounding, no BLAS)
        w, h = a.max() - a.min(), b.max() - b.min()
        if best is None or w * h < best[0]:
            ca, cb = (a.max() + a.min()) / 2, (b.max() + b.min()) / 2
            c = np.array([u[0] * ca + v[0] * cb, u[1] * ca + v[1] * cb])
            best = (w * h, c, w, h, np.arctan2(u[1], u[0]) * (180.0 / np.pi))
    if best is None:                                  # all four points coincide
        return (float(p[0, 0]), float(p[0, 1])), (0.0, 0.0), 0.0
    _, c, w, h, ang = best
    return (float(np.float32(c[0])), float(np.float32(c[1]))), (float(np.float32(w)), float(np.float32(h))), float(np.float32(ang))


def corners_to_xywha(corners, reference):            # STrack._corners_to_xywha, botsort_track.py:176-195
    (cx, cy), (w, h), angle_deg = min_area_rect(np.asarray(corners, dtype=np.float32))
    xywha = np.array([cx, cy, max(w, 1e-4), max(h, 1e-4), np.deg2rad(angle_deg)], dtype=np.float32)
    return align_obb_measurement(xywha, reference).astype(np.float32)


def gmc_obb(mean, cov, H):                           # one track of STrack.multi_gmc_obb, botsort_track.py:197-230
    warp = np.asarray(H, dtype=np.float32)
    linear = warp[:2, :2]
    scale_x = max(float(np.linalg.norm(linear[:, 0])), 1e-6)          # _affine_components, botsort_track.py:145-157
    scale_y = max(float(np.linalg.norm(linear[:, 1])), 1e-6)
    transform = np.eye(10, dtype=np.float32)
    transform[:2, :2] = linear
    transform[5:7, 5:7] = linear
    transform[2, 2] = transform[7, 7] = scale_x
    transform[3, 3] = transform[8, 8] = scale_y
    ref_box = np.asarray(mean[:5], dtype=np.float32)
    rect_w, rect_h = max(float(ref_box[2]), 1e-4), max(float(ref_box[3]), 1e-4)
    corners = box_points(float(ref_box[0]), float(ref_box[1]), rect_w, rect_h, float(np.degrees(ref_box[4]))).astype(np.float32)
    warped_box = corners_to_xywha(transform_points(corners, warp), ref_box)
    out = mean.copy()
    out[:5] = warped_box
    out[5:7] = linear @ out[5:7]
    out[7] *= scale_x
    out[8] *= scale_y
    return out, transform @ cov @ transform.T
