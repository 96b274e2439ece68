// Oriented-box geometry shared by the oriented frame steps (botsort_step_body.hpp and deepocsort_step_body.hpp compiled with BM_OBB).
#pragma once

#include "kernel_macros.hpp"

namespace bm {
namespace obb {

constexpr double OBB_PI = 3.141592653589793;

// BaseKalmanFilter._wrap_angle (boxmot/motion/kalman_filters/base.py:116-120): (a + pi) % (2 pi) - pi with NumPy's remainder (the
// sign of the divisor)
__device__ inline double obb_wrap_angle(double a) {
    double r = fmod(a + OBB_PI, 2.0 * OBB_PI);
    if (r != 0.0 && r < 0.0) r += 2.0 * OBB_PI;
    return r - OBB_PI;
}

// Rotated IoU of two (cx, cy, w, h, theta) boxes (boxmot/trackers/association/iou.py:5-115): the reference's enclosing-AABB pre-filter
// (iou.py:38-84: pairs whose AABBs do not overlap are 0), corners in fp32 like RotatedRect::points, then the intersection polygon by
// clipping a's corners with b's four half-planes and the shoelace area, fp64 (the reference: cv2.rotatedRectangleIntersection +
// contourArea; OpenCV is absent offline, so parity is unpinned for this one quantity -- DESIGN.md section 4.6c).
// cv2.boxPoints of ((cx, cy), (w, h), deg): `deg` as the caller's NumPy expression produced it
__device__ inline void obb_corners_deg(const double* r, double deg, double (&p)[4][2]) {
    const double a = deg * OBB_PI / 180.0;
    const float b = (float)cos(a) * 0.5f, s = (float)sin(a) * 0.5f;
    const float cx = (float)r[0], cy = (float)r[1], w = (float)r[2], h = (float)r[3];
    const float p0x = cx - s * h - b * w, p0y = cy + b * h - s * w;
    const float p1x = cx + s * h - b * w, p1y = cy - b * h - s * w;
    p[0][0] = p0x; p[0][1] = p0y; p[1][0] = p1x; p[1][1] = p1y;
    p[2][0] = 2.0f * cx - p0x; p[2][1] = 2.0f * cy - p0y; p[3][0] = 2.0f * cx - p1x; p[3][1] = 2.0f * cy - p1y;
}
__device__ inline void obb_corners(const double* r, double (&p)[4][2]) {
    obb_corners_deg(r, r[4] * (180.0 / OBB_PI), p);          // np.degrees of an fp64 angle (iou.py:14-15)
}
__device__ inline double obb_iou(const double* r1, const double* r2) {
    const double hw1 = r1[2] / 2, hh1 = r1[3] / 2, c1 = fabs(cos(r1[4])), s1 = fabs(sin(r1[4]));
    const double hw2 = r2[2] / 2, hh2 = r2[3] / 2, c2 = fabs(cos(r2[4])), s2 = fabs(sin(r2[4]));
    const double ex1 = hw1 * c1 + hh1 * s1, ey1 = hw1 * s1 + hh1 * c1, ex2 = hw2 * c2 + hh2 * s2, ey2 = hw2 * s2 + hh2 * c2;
    if (!(fabs(r1[0] - r2[0]) < ex1 + ex2 && fabs(r1[1] - r2[1]) < ey1 + ey2)) return 0.0;
    double p[4][2], q[4][2];
    obb_corners(r1, p);
    obb_corners(r2, q);
    double qa = 0.0, orient = 0.0;
    for (int i = 0; i < 4; ++i) {
        const int n = (i + 1) & 3;
        qa += q[i][0] * q[n][1] - q[n][0] * q[i][1];
        orient += (q[n][0] - q[i][0]) * (q[n][1] + q[i][1]);
    }
    if (0.5 * fabs(qa) == 0.0) return 0.0;
    const double sgn = orient > 0.0 ? 1.0 : (orient < 0.0 ? -1.0 : 0.0);
    double poly[2][10][2];
    int n_poly = 4, cur = 0;
    for (int i = 0; i < 4; ++i) { poly[0][i][0] = p[i][0]; poly[0][i][1] = p[i][1]; }
    for (int i = 0; i < 4; ++i) {
        const double ax = q[i][0], ay = q[i][1], ex = q[(i + 1) & 3][0] - ax, ey = q[(i + 1) & 3][1] - ay;
        int n_out = 0;
        for (int k = 0; k < n_poly; ++k) {
            const double* cv = poly[cur][k];
            const double* nv = poly[cur][k + 1 == n_poly ? 0 : k + 1];
            const double sc = (ex * (cv[1] - ay) - ey * (cv[0] - ax)) * (-sgn);
            const double sn = (ex * (nv[1] - ay) - ey * (nv[0] - ax)) * (-sgn);
            if (sc >= 0) { poly[cur ^ 1][n_out][0] = cv[0]; poly[cur ^ 1][n_out][1] = cv[1]; ++n_out; }
            if ((sc > 0 && sn < 0) || (sc < 0 && sn > 0)) {
                const double t = sc / (sc - sn);
                poly[cur ^ 1][n_out][0] = cv[0] + t * (nv[0] - cv[0]);
                poly[cur ^ 1][n_out][1] = cv[1] + t * (nv[1] - cv[1]);
                ++n_out;
            }
        }
        cur ^= 1;
        n_poly = n_out;
        if (n_poly < 3) return 0.0;
    }
    double area = 0.0;
    for (int k = 0; k < n_poly; ++k) {
        const int n = k + 1 == n_poly ? 0 : k + 1;
        area += poly[cur][k][0] * poly[cur][n][1] - poly[cur][n][0] * poly[cur][k][1];
    }
    const double inter = 0.5 * fabs(area);
    if (!(inter > 0.0)) return 0.0;
    const double uni = r1[2] * r1[3] + r2[2] * r2[3] - inter;
    return uni > 0 ? inter / uni : 0.0;
}

}  // namespace obb
}  // namespace bm
