"""Oriented-box oracles (oracle/obb.py, oracle/bytetrack_obb.py, oracle/botsort_obb.py, oracle/ocsort_obb.py): pinned on the reference
classes fed 7-column detections and on the rows those classes produced (tests/golden/obb_golden.npz).  The reference-class comparisons
need /root/reference (build container only)."""
import logging

import numpy as np
import pytest

from common import obb_frames
from oracle import obb, ref_harness


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


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")
@pytest.mark.parametrize("kw", [dict(with_reid=False), dict(with_reid=True), dict(with_reid=True, track_buffer=4, fuse_first_associate=True)])
def test_botsort_obb_oracle_bit_exact_on_the_reference_class(kw):
    """The reference BotSort (use_cmc=False) fed 7-column detections -- with and without appearance embeddings -- against
    BotSortObbOracle: 9-column rows and the fp64 filter state."""
    from boxmot_amd.scenario import stress_frames
    from oracle.botsort_obb import BotSortObbOracle
    logging.disable(logging.CRITICAL)
    BotSort = ref_harness.load_botsort()
    ref, orc = BotSort(reid_model=None, use_cmc=False, **kw), BotSortObbOracle(**kw)
    img = np.zeros((480, 640, 3), np.uint8)
    embs = [e for _, e in stress_frames(90, seed=4)]
    rows = 0
    for t, d in enumerate(obb_frames(90, seed=4)):
        e = embs[t].copy() if kw["with_reid"] else None
        r = np.asarray(ref.update(d.copy(), img, None if e is None else e.copy()))
        o = orc.update(d.copy(), img, e)
        assert r.shape == o.shape and np.array_equal(r.reshape(-1, 9), o.reshape(-1, 9)), (kw, t)
        rows += len(o)
    assert ref.is_obb and rows > 300
    for a, b in zip(ref.active_tracks, orc.active):
        assert a.id == b.id and np.array_equal(a.mean, b.mean) and np.array_equal(a.covariance, b.cov)
    assert [t.id for t in ref.lost_stracks] == [t.id for t in orc.lost]


@pytest.mark.parametrize("key", ["bytetrack", "botsort_noreid", "botsort_reid"])
def test_obb_oracles_match_reference_golden_rows(key):
    """The OBB oracles against rows the reference classes produced (tests/golden/obb_golden.npz) -- runs without /root/reference."""
    from boxmot_amd.scenario import stress_frames
    from common import obb_golden_rows
    from oracle.botsort_obb import BotSortObbOracle
    from oracle.bytetrack_obb import ByteTrackObbOracle
    want, frames, seed = obb_golden_rows(key)
    orc = ByteTrackObbOracle() if key == "bytetrack" else BotSortObbOracle(with_reid=key == "botsort_reid")
    embs = [e for _, e in stress_frames(frames, seed=seed)]
    img = np.zeros((480, 640, 3), np.uint8)
    for t, d in enumerate(obb_frames(frames, seed=seed)):
        got = np.asarray(orc.update(d.copy(), img, embs[t].copy() if key == "botsort_reid" else None), dtype=np.float32).reshape(-1, 9)
        assert got.shape == want[t].shape and np.array_equal(got, want[t]), (key, t)


OCSORT_CASES = [{}, dict(use_byte=True), dict(max_age=5, min_hits=1, delta_t=2, inertia=0.4, iou_threshold=0.2),
                dict(use_byte=True, max_age=8, min_hits=1), dict(asso_func="centroid", iou_threshold=0.9, use_byte=True)]


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")
@pytest.mark.parametrize("kw", OCSORT_CASES)
def test_ocsort_obb_oracle_bit_exact_on_the_reference_class(kw):
    """The reference OcSort fed 7-column detections (KalmanFilterXYSR(dim_x=9, dim_z=5), rotated IoU, observation-centric re-update with
    the interpolated angle) against OcSortObbOracle: 9-column rows and the fp64 filter state of every track, every frame."""
    from oracle.ocsort_obb import OcSortObbOracle
    logging.disable(logging.CRITICAL)
    ref, orc = ref_harness.load_ocsort()(**kw), OcSortObbOracle(**kw)
    img = np.zeros((480, 640, 3), np.uint8)
    rows, thawed = 0, 0
    for t, d in enumerate(obb_frames(120, seed=4)):
        frozen = {k.id for k in orc.tracks if not k.kf.observed and k.kf.saved is not None}
        r = np.asarray(ref.update(d.copy(), img))
        o = orc.update(d.copy(), img)
        assert np.array_equal(r.reshape(-1, 9).astype(np.float32), o.reshape(-1, 9)), (kw, t)
        rows += len(o)
        thawed += sum(1 for k in orc.tracks if k.id in frozen and k.kf.observed)
        assert len(ref.active_tracks) == len(orc.tracks)
        for a, b in zip(ref.active_tracks, orc.tracks):
            assert a.id == b.id and np.array_equal(a.kf.x, b.kf.x) and np.array_equal(a.kf.P, b.kf.P), (kw, t)
    assert ref.is_obb and rows > 200 and thawed > 5              # the re-update after a gap (unfreeze) is exercised


@pytest.mark.parametrize("key", ["ocsort", "ocsort_byte"])
def test_ocsort_obb_oracle_matches_reference_golden_rows(key):
    from common import obb_golden_rows
    from oracle.ocsort_obb import OcSortObbOracle
    want, frames, seed = obb_golden_rows(key)
    orc = OcSortObbOracle(**({} if key == "ocsort" else dict(use_byte=True, max_age=8, min_hits=1)))
    img = np.zeros((480, 640, 3), np.uint8)
    for t, d in enumerate(obb_frames(frames, seed=seed)):
        got = np.asarray(orc.update(d.copy(), img), dtype=np.float32).reshape(-1, 9)
        assert got.shape == want[t].shape and np.array_equal(got, want[t]), (key, t)


def test_min_area_rect_known_answers():
    """oracle/obb.py's stand-in for cv2.minAreaRect on the shapes it is used for: a rotated rectangle comes back (any of its equivalent
    parameterisations -- the caller re-aligns), a sheared one gets the smaller of the two edge-aligned enclosing rectangles."""
    for ang in (0.0, 17.0, -63.0, 90.0, 134.0):
        pts = obb.box_points(50.0, 40.0, 30.0, 12.0, ang)
        (cx, cy), (w, h), a = obb.min_area_rect(pts)
        assert abs(cx - 50) < 1e-3 and abs(cy - 40) < 1e-3 and abs(w * h - 360.0) < 1e-2 and {round(w), round(h)} == {30, 12}
        back = obb.align_obb_measurement(np.array([cx, cy, w, h, np.deg2rad(a)]), np.array([50.0, 40.0, 30.0, 12.0, np.deg2rad(ang)]))
        assert np.allclose(back, [50.0, 40.0, 30.0, 12.0, obb.wrap_angle(np.deg2rad(ang))], atol=1e-3)
    shear = np.array([[0, 0], [10, 0], [13, 4], [3, 4]], dtype=np.float32)            # base 10, height 4, offset 3
    (_, _), (w, h), a = obb.min_area_rect(shear)
    assert abs(w * h - 13 * 4) < 1e-3 and abs(a) < 1e-4                               # along the long edge: 13 x 4 = 52 < the slanted edge's 65
    assert obb.min_area_rect(np.zeros((4, 2), np.float32)) == ((0.0, 0.0), (0.0, 0.0), 0.0)


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")
@pytest.mark.parametrize("kw", [dict(with_reid=False), dict(with_reid=True)])
def test_botsort_obb_oracle_with_camera_motion_follows_the_reference_class(kw):
    """STrack.multi_gmc_obb (botsort_track.py:197-230) in the oracle against the reference BotSort driven with scheduled warps, the
    cv2.transform / cv2.minAreaRect calls answered by oracle/obb.py's restatements on both sides (those two are unpinned): the flow
    around them -- corner warp, refit, re-alignment to the previous box, velocity / covariance transform -- is bit-exact."""
    from boxmot_amd.scenario import camera_warps, stress_frames
    from oracle.botsort_obb import BotSortObbOracle
    logging.disable(logging.CRITICAL)

    class Scheduled:
        def __init__(self, w):
            self.w, self.k = w, 0

        def apply(self, img, dets):
            self.k += 1
            return self.w[self.k - 1]
    warps = camera_warps(90, seed=4)
    ref, orc = ref_harness.load_botsort()(reid_model=None, use_cmc=False, **kw), BotSortObbOracle(**kw)
    ref.cmc = Scheduled(warps)
    img = np.zeros((480, 640, 3), np.uint8)
    embs = [e for _, e in stress_frames(90, seed=4)]
    rows = 0
    for t, d in enumerate(obb_frames(90, seed=4)):
        e = embs[t].copy() if kw["with_reid"] else None
        r = np.asarray(ref.update(d.copy(), img, None if e is None else e.copy()))
        o = orc.update(d.copy(), img, e, warp=warps[t])
        assert r.shape == o.shape and np.array_equal(r.reshape(-1, 9), o.reshape(-1, 9)), (kw, t)
        rows += len(o)
    assert rows > 300
    for a, b in zip(ref.active_tracks, orc.active):
        assert a.id == b.id and np.array_equal(a.mean, b.mean) and np.array_equal(a.covariance, b.cov)
