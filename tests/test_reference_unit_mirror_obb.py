"""The reference's known-answer tests for the ORIENTED pieces of the path, restated for this repository's counterparts -- same inputs,
same assertions -- and, where /root/reference is mounted, the reference object run beside them bit for bit:
  tests/unit/test_kalman_filters_modes.py:48-74   (KalmanFilterXYWH(ndim=5): wrap-around update)        -> oracle/obb.py kf5_*
  tests/unit/test_kalman_filters_modes.py:117-162 (KalmanFilterXYSR(9, 5): equivalent (r, theta) forms, the re-update across the
                                                   +-pi seam)                                              -> oracle/ocsort_obb.py
  tests/unit/test_mot_utils.py:6-22               (corner order and MMOT rows of equivalent boxes)        -> boxmot_amd/replay.py
(The device filters are compared with these oracles in tests/test_kernel_obb_emu.py, test_docs_obb_emu.py and test_gpu_obb.py.)"""
import numpy as np
import pytest

from oracle import obb, ref_harness
from oracle.ocsort_obb import FilterXYSRTheta

HAVE_REF = ref_harness.reference_available()


def _angle_diff(a, b):
    return float((a - b + np.pi) % (2.0 * np.pi) - np.pi)


def test_xywh_supports_obb_mode():
    init, upd = np.array([100.0, 80.0, 40.0, 20.0, np.pi - 0.01]), np.array([101.0, 79.5, 40.5, 20.5, -np.pi + 0.02])
    mean, cov = obb.kf5_initiate(init)
    m, c = obb.kf5_multi_predict(mean[None].copy(), cov[None].copy())
    new_mean, new_cov = obb.kf5_update(m[0], c[0], upd)
    assert new_mean.shape == (10,) and new_cov.shape == (10, 10)
    assert new_mean[2] > 0 and new_mean[3] > 0
    assert -np.pi <= float(new_mean[4]) < np.pi
    assert abs(_angle_diff(float(new_mean[4]), upd[4])) < 0.2
    if HAVE_REF:
        ref_harness.install_standins()
        from boxmot.motion.kalman_filters.xywh import KalmanFilterXYWH
        kf = KalmanFilterXYWH(ndim=5)
        rm, rc = kf.initiate(init)
        rm, rc = kf.predict(rm, rc)
        rm, rc = kf.update(rm, rc, upd)
        assert np.allclose(rm, new_mean, rtol=1e-12, atol=1e-12) and np.allclose(rc, new_cov, rtol=1e-12, atol=1e-12)


def _default_xysr():
    """KalmanFilterXYSR(dim_x=9, dim_z=5) as constructed (x = 0, P = Q = I, R = I): the tracker's scalings undone."""
    kf = FilterXYSRTheta(np.zeros((5, 1)), 1.0, 1.0, 1.0)
    kf.P, kf.Q, kf.R = np.eye(9), np.eye(9), np.eye(5)
    return kf


def _ref_xysr():
    ref_harness.install_standins()
    from boxmot.motion.kalman_filters.xysr import KalmanFilterXYSR
    return KalmanFilterXYSR(dim_x=9, dim_z=5, max_obs=50)


def test_xysr_obb_aligns_equivalent_ratio_angle_forms():
    theta_ref = 0.35
    kf = _default_xysr()
    # KalmanFilterXYSR.initiate of (300, 200, 50000, 2, theta_ref) is not on the tracker's path (KalmanBoxTracker sets x and P itself):
    # taken from the reference object where it is mounted, a plain state otherwise
    ref = _ref_xysr() if HAVE_REF else None
    if ref is not None:
        mean, cov = ref.initiate(np.array([[300.0], [200.0], [50000.0], [2.0], [theta_ref]]))
        ref.x, ref.P = mean.copy(), cov.copy()
        kf.x, kf.P = mean.copy(), cov.copy()
    else:
        kf.x[:5, 0] = [300.0, 200.0, 50000.0, 2.0, theta_ref]
        kf.P = np.diag([100.0, 100.0, 1e6, 1.0, 1e-2, 1e4, 1e4, 1e6, 1e-4])
    kf.predict()
    equivalent = np.array([[300.5], [199.5], [50050.0], [0.5], [theta_ref + (np.pi / 2.0)]])      # r -> 1 / r, theta -> theta + pi / 2
    kf.update(equivalent.copy())
    assert abs(_angle_diff(float(kf.x[4, 0]), theta_ref)) < 0.25
    assert abs(np.log(float(kf.x[3, 0]) / 2.0)) < 0.35
    assert np.isfinite(float(kf.x[8, 0])) and abs(float(kf.x[8, 0])) < 0.2
    if ref is not None:
        ref.predict()
        ref.update(equivalent.copy())
        assert np.array_equal(ref.x, kf.x) and np.array_equal(ref.P, kf.P)


def test_xysr_obb_unfreeze_handles_angle_wrap():
    obs1 = np.array([[300.0], [200.0], [50000.0], [1.5], [np.pi - 0.05]])
    obs2 = np.array([[320.0], [210.0], [51000.0], [1.4], [-np.pi + 0.04]])
    obs3 = np.array([[350.0], [230.0], [52000.0], [1.3], [np.pi - 0.02]])
    filters = [_default_xysr()] + ([_ref_xysr()] if HAVE_REF else [])
    for kf in filters:
        kf.predict()
        kf.update(obs1.copy())
        kf.predict()
        kf.update(obs2.copy())
        for _ in range(5):
            kf.predict()
            kf.update(None)
        kf.predict()
        kf.update(obs3.copy())
        assert np.all(np.isfinite(kf.x))
        assert -np.pi <= float(kf.x[4, 0]) < np.pi
    kf = filters[0]
    assert kf.observed and len(kf.history) > 0            # thawed: the frozen copy was replayed (unfreeze) before the update
    if HAVE_REF:
        assert abs(float(filters[1].y[4, 0])) < 0.3      # the reference test's innovation bound, on the reference object
        assert np.array_equal(filters[1].x, kf.x) and np.array_equal(filters[1].P, kf.P)


def test_xywha_to_corners_canonicalizes_equivalent_obb_forms():
    from boxmot_amd.replay import xywha_to_corners
    base = np.array([640.0, 512.0, 320.0, 160.0, 0.45], dtype=np.float32)
    equivalent = np.array([640.0, 512.0, 160.0, 320.0, 0.45 + (np.pi / 2.0)], dtype=np.float32)
    np.testing.assert_allclose(xywha_to_corners(base), xywha_to_corners(equivalent), atol=1e-4)


def test_convert_to_mmot_obb_format_matches_equivalent_obb_forms():
    from boxmot_amd.replay import format_for_mmot_obb
    base = np.array([[640.0, 512.0, 320.0, 160.0, 0.45, 3.0, 0.9, 4.0, 7.0]], dtype=np.float32)
    equivalent = np.array([[640.0, 512.0, 160.0, 320.0, 0.45 + (np.pi / 2.0), 3.0, 0.9, 4.0, 7.0]], dtype=np.float32)
    np.testing.assert_allclose(format_for_mmot_obb(base, 12), format_for_mmot_obb(equivalent, 12), atol=1e-4)
    assert format_for_mmot_obb(base, 12).shape == (1, 13)
