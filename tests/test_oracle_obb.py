"""Oriented-box oracle (oracle/obb.py, oracle/bytetrack_obb.py) -- groundwork for the OBB row of the plugin surface; no device step
takes 7-column detections yet.  The reference-class comparisons need /root/reference (build container only)."""
import logging

import numpy as np
import pytest

from oracle import obb, ref_harness


def obb_frames(n_frames, seed):
    """The seeded stress scenes with every detection turned into (cx, cy, w, h, angle, conf, cls): the angle follows the box centre
    smoothly so that tracks see a slowly rotating target, with parameterisation flips (w <-> h, angle + pi / 2) thrown in -- the
    ambiguity KalmanFilterXYWH._align_obb_measurement resolves."""
    from boxmot_amd.scenario import stress_frames
    rng = np.random.default_rng(seed)
    for t, (d, _) in enumerate(stress_frames(n_frames, seed=seed)):
        d = np.asarray(d, dtype=np.float32).reshape(-1, 6)
        cx, cy, w, h = (d[:, 0] + d[:, 2]) / 2, (d[:, 1] + d[:, 3]) / 2, d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]
        ang = 0.6 * np.sin(0.02 * t + 0.004 * cx + 0.006 * cy) + rng.normal(0, 0.01, len(d))
        flip = rng.random(len(d)) < 0.15
        w2, h2, a2 = np.where(flip, h, w), np.where(flip, w, h), np.where(flip, ang + np.pi / 2, ang)
        yield np.stack([cx, cy, w2, h2, a2, d[:, 4], d[:, 5]], axis=1).astype(np.float32)


def test_rotated_intersection_area_known_answers():
    r = ((10.0, 10.0), (4.0, 2.0), 0.0)
    assert obb.rotated_intersection_area(r, r) == pytest.approx(8.0, rel=1e-6)
    assert obb.rotated_intersection_area(r, ((11.0, 10.0), (4.0, 2.0), 0.0)) == pytest.approx(6.0, rel=1e-6)
    assert obb.rotated_intersection_area(r, ((10.0, 10.0), (4.0, 2.0), 90.0)) == pytest.approx(4.0, rel=1e-6)
    assert obb.rotated_intersection_area(r, ((10.0, 10.0), (2.0, 4.0), 90.0)) == pytest.approx(8.0, rel=1e-6)      # the same rectangle, other parameterisation
    assert obb.rotated_intersection_area(r, ((100.0, 10.0), (4.0, 2.0), 30.0)) == 0.0
    # a square rotated by 45 degrees inside a larger square: the octagon's area
    big, small = ((0.0, 0.0), (2.0, 2.0), 0.0), ((0.0, 0.0), (2.0, 2.0), 45.0)
    assert obb.rotated_intersection_area(big, small) == pytest.approx(8.0 * (np.sqrt(2.0) - 1.0), rel=1e-6)
    # against a Monte-Carlo estimate on random pairs
    rng = np.random.default_rng(0)

    def inside(p, q):
        (cx, cy), (w, h), a = q
        a = np.deg2rad(a)
        dx, dy = p[:, 0] - cx, p[:, 1] - cy
        return (np.abs(dx * np.cos(a) + dy * np.sin(a)) <= w / 2) & (np.abs(-dx * np.sin(a) + dy * np.cos(a)) <= h / 2)
    pts = rng.uniform(-10, 20, (400000, 2))
    for _ in range(6):
        r1 = ((rng.uniform(0, 10), rng.uniform(0, 10)), (rng.uniform(2, 8), rng.uniform(2, 8)), rng.uniform(-180, 180))
        r2 = ((rng.uniform(0, 10), rng.uniform(0, 10)), (rng.uniform(2, 8), rng.uniform(2, 8)), rng.uniform(-180, 180))
        mc = (inside(pts, r1) & inside(pts, r2)).mean() * 900.0
        assert abs(obb.rotated_intersection_area(r1, r2) - mc) < 0.25


def test_obb_iou_matrix_is_symmetric_and_bounded():
    rng = np.random.default_rng(1)
    b = np.stack([rng.uniform(0, 50, 12), rng.uniform(0, 50, 12), rng.uniform(5, 20, 12), rng.uniform(5, 20, 12), rng.uniform(-3, 3, 12)], axis=1)
    m = obb.iou_obb_matrix(b, b)
    assert np.allclose(np.diag(m), 1.0, atol=1e-5) and np.allclose(m, m.T, atol=1e-9) and (m >= 0).all() and (m <= 1 + 1e-5).all()      # corners are fp32, like OpenCV's


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")
def test_kalman_filter_ndim5_bit_exact_vs_reference():
    ref_harness.install_standins()
    from boxmot.motion.kalman_filters.xywh import KalmanFilterXYWH
    kf = KalmanFilterXYWH(ndim=5)
    rng = np.random.default_rng(3)
    z = np.array([100.0, 80.0, 40.0, 20.0, 0.3])
    mean, cov = kf.initiate(z)
    om, oc = obb.kf5_initiate(z)
    assert np.array_equal(mean, om) and np.array_equal(cov, oc)
    for t in range(40):
        mean, cov = kf.multi_predict(mean[None].copy(), cov[None].copy())
        om, oc = obb.kf5_multi_predict(om[None].copy(), oc[None].copy())
        mean, cov, om, oc = mean[0], cov[0], om[0], oc[0]
        assert np.array_equal(mean, om) and np.array_equal(cov, oc), t
        z = z + np.array([1.5, -0.7, 0.2, 0.1, 0.05]) + rng.normal(0, 0.3, 5)
        zz = z.copy()
        if t % 7 == 3:                       # the other parameterisation of the same rectangle
            zz = np.array([z[0], z[1], z[3], z[2], z[4] + np.pi / 2])
        if t % 11 == 5:
            zz[4] += np.pi
        mean, cov = kf.update(mean, cov, zz.astype(np.float32))
        om, oc = obb.kf5_update(om, oc, zz.astype(np.float32))
        assert np.array_equal(mean, om) and np.array_equal(cov, oc), t


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")
@pytest.mark.parametrize("kw", [{}, dict(track_buffer=4, track_thresh=0.6, match_thresh=0.7)])
def test_bytetrack_obb_oracle_bit_exact_on_the_reference_class(kw):
    """The reference ByteTrack fed 7-column detections (its OBB mode: KalmanFilterXYWH(ndim=5), rotated IoU, 9-column rows) against
    ByteTrackObbOracle: rows and the fp64 filter state of both lists, with oracle/obb.py's intersection area behind the two cv2 calls."""
    from oracle.bytetrack_obb import ByteTrackObbOracle
    logging.disable(logging.CRITICAL)
    ByteTrack = ref_harness.load_bytetrack()
    ref, orc = ByteTrack(**kw), ByteTrackObbOracle(**kw)
    img = np.zeros((480, 640, 3), np.uint8)
    rows = 0
    for t, d in enumerate(obb_frames(90, seed=4)):
        r = np.asarray(ref.update(d.copy(), img))
        o = orc.update(d.copy(), img)
        assert r.shape == o.shape and np.array_equal(r.reshape(-1, 9), o.reshape(-1, 9)), (kw, t)
        rows += len(o)
    assert ref.is_obb and rows > 300
    for a, b in zip(ref.active_tracks, orc.active):
        assert a.id == b.id and np.array_equal(a.mean, b.mean) and np.array_equal(a.covariance, b.cov)
    assert [t.id for t in ref.lost_stracks] == [t.id for t in orc.lost]
