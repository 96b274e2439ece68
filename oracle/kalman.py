"""Oracle XYWH Kalman filter (NumPy/SciPy fp64) -- TEST INFRASTRUCTURE ONLY.

Restates, operation for operation (same NumPy/SciPy calls, same order, so the
result is bit-identical to the reference on the same host):
  * BaseKalmanFilter.initiate          boxmot/motion/kalman_filters/base.py:234-244
  * BaseKalmanFilter.multi_predict     base.py:311-327
  * BaseKalmanFilter.project / update  base.py:286-309, 329-355
  * KalmanFilterXYWH noise models      boxmot/motion/kalman_filters/xywh.py:22-85
  * KalmanFilterXYWH.initiate / multi_predict / update clamps   xywh.py:136-186
Only the AABB (ndim=4) variant used by BoT-SORT is covered.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

STD_POS = 1.0 / 20   # base.py:60-62
STD_VEL = 1.0 / 160  # base.py:63-65
MIN_SIZE = 1e-4      # xywh.py:128-134

F = np.eye(8)
for _i in range(4):
    F[_i, 4 + _i] = 1.0  # base.py:95-101 constant-velocity motion matrix, dt = 1
H = np.eye(4, 8)         # base.py:56


def initiate(xywh):
    """xywh (4,) any float -> mean (8,) f64, cov (8,8) f64.  xywh.py:136-142."""
    m = np.asarray(xywh, dtype=float).copy()
    mean = np.r_[m, np.zeros_like(m)]
    std = [
        2 * STD_POS * m[2], 2 * STD_POS * m[3], 2 * STD_POS * m[2], 2 * STD_POS * m[3],
        10 * STD_VEL * m[2], 10 * STD_VEL * m[3], 10 * STD_VEL * m[2], 10 * STD_VEL * m[3],
    ]
    cov = np.diag(np.square(std))
    mean[2] = max(float(mean[2]), MIN_SIZE)
    mean[3] = max(float(mean[3]), MIN_SIZE)
    return mean, cov


def multi_predict(mean, cov):
    """mean (N,8), cov (N,8,8) -> predicted copies.  base.py:311-327 + xywh.py:149-160."""
    std_pos = [STD_POS * mean[:, 2], STD_POS * mean[:, 3], STD_POS * mean[:, 2], STD_POS * mean[:, 3]]
    std_vel = [STD_VEL * mean[:, 2], STD_VEL * mean[:, 3], STD_VEL * mean[:, 2], STD_VEL * mean[:, 3]]
    sqr = np.square(np.r_[std_pos, std_vel]).T
    motion_cov = np.asarray([np.diag(sqr[i]) for i in range(len(mean))])
    mean = np.dot(mean, F.T)
    left = np.dot(F, cov).transpose((1, 0, 2))
    cov = np.dot(left, F.T) + motion_cov
    mean[:, 2] = np.maximum(mean[:, 2], MIN_SIZE)
    mean[:, 3] = np.maximum(mean[:, 3], MIN_SIZE)
    return mean, cov


def update(mean, cov, measurement, confidence: float = 0.0):
    """One correction step.  base.py:286-355 + xywh.py:162-186 (AABB branch)."""
    std = [STD_POS * mean[2], STD_POS * mean[3], STD_POS * mean[2], STD_POS * mean[3]]
    std = [(1 - confidence) * x for x in std]
    innovation_cov = np.diag(np.square(std))
    projected_mean = np.dot(H, mean)
    projected_cov = np.linalg.multi_dot((H, cov, H.T)) + innovation_cov

    chol, lower = scipy.linalg.cho_factor(projected_cov, lower=True, check_finite=False)
    gain = scipy.linalg.cho_solve((chol, lower), np.dot(cov, H.T).T, check_finite=False).T
    innovation = measurement - projected_mean
    new_mean = mean + np.dot(innovation, gain.T)
    new_cov = cov - np.linalg.multi_dot((gain, projected_cov, gain.T))
    new_mean[2] = max(float(new_mean[2]), MIN_SIZE)
    new_mean[3] = max(float(new_mean[3]), MIN_SIZE)
    return new_mean, new_cov
