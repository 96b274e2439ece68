"""GPU parity tests for ORIENTED detections through BoT-SORT, ByteTrack and OC-SORT (the bm::obb copies of the frame steps: 10-state
KalmanFilterXYWH / 9-state KalmanFilterXYSR, rotated-rectangle IoU; include/boxmot_hip.h `is_obb`), through the C ABI: against the
reference's own rows (tests/golden/obb_golden.npz -- the real BotSort / ByteTrack / OcSort classes fed 7-column detections) and against
the oracles (oracle/botsort_obb.py, oracle/bytetrack_obb.py, oracle/ocsort_obb.py, pinned bit-exact on those classes by
tests/test_oracle_obb.py) including the fp64 filter state.  Rows: id / conf / cls / det_ind and the row order exact; the box (fp32 of the fp64 state) within 2e-4 px / rad."""
import ctypes

import numpy as np
import pytest

from common import obb_frames, obb_golden_rows

pytestmark = pytest.mark.gpu

EMB = 32


def _rows_match(got, want, t):
    got, want = np.asarray(got, dtype=np.float32).reshape(-1, 9), np.asarray(want, dtype=np.float32).reshape(-1, 9)
    assert got.shape == want.shape, (t, got.shape, want.shape)
    assert np.array_equal(got[:, 5:], want[:, 5:]), t
    assert np.allclose(got[:, :5], want[:, :5], rtol=0, atol=2e-4), (t, np.abs(got[:, :5] - want[:, :5]).max())


def _check_state(trk, orc):
    for which, recs in ((0, orc.active), (1, orc.lost)):
        d = trk.state_dump(which)
        assert list(d["ints"][:, 0]) == [r.id for r in recs]
        if d["n"]:
            assert d["kf"].shape[1] == 110
            ref = np.concatenate([np.array([r.mean for r in recs]), np.array([r.cov for r in recs]).reshape(-1, 100)], 1)
            assert np.allclose(d["kf"], ref, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("key", ["bytetrack", "botsort_noreid", "botsort_reid"])
def test_oriented_detections_reproduce_the_reference_rows(key):
    from boxmot_amd import BotSort, ByteTrack
    from boxmot_amd.scenario import stress_frames
    want, frames, seed = obb_golden_rows(key)
    img = np.zeros((480, 640, 3), np.uint8)
    if key == "bytetrack":
        trk = ByteTrack(max_tracks=128, max_dets=64)
    else:
        trk = BotSort(reid_model=None, use_cmc=False, with_reid=key == "botsort_reid", max_tracks=128, max_dets=64, emb_dim=EMB)
    embs = [e for _, e in stress_frames(frames, seed=seed)]
    rows = 0
    for t, d in enumerate(obb_frames(frames, seed=seed)):
        got = trk.update(d, img, embs[t] if key == "botsort_reid" else None)
        assert got.shape[1] == 9 and got.is_obb
        _rows_match(got, want[t], t)
        rows += len(got)
    assert rows > 500 and trk.is_obb
    trk.close()


@pytest.mark.parametrize("kind,kw", [("botsort", dict(with_reid=True, track_buffer=4, fuse_first_associate=True)),
                                     ("botsort", dict(with_reid=False, match_thresh=0.7)),
                                     ("bytetrack", dict(track_buffer=4, track_thresh=0.6, match_thresh=0.7))])
def test_oriented_step_matches_the_oracle_and_its_filter_state(kind, kw):
    """Other seeds and option sets than the golden file's, 150 frames, small initial tables (they grow: 10-state rows carried over)."""
    from boxmot_amd import BotSort, ByteTrack
    from boxmot_amd.scenario import stress_frames
    from oracle.botsort_obb import BotSortObbOracle
    from oracle.bytetrack_obb import ByteTrackObbOracle
    img = np.zeros((480, 640, 3), np.uint8)
    for seed in (9, 13):
        if kind == "bytetrack":
            trk, orc = ByteTrack(max_tracks=16, max_dets=8, **kw), ByteTrackObbOracle(**kw)
        else:
            trk, orc = BotSort(reid_model=None, use_cmc=False, max_tracks=16, max_dets=8, emb_dim=EMB, **kw), BotSortObbOracle(**kw)
        embs = [e for _, e in stress_frames(150, seed=seed)]
        for t, d in enumerate(obb_frames(150, seed=seed)):
            e = embs[t] if kw.get("with_reid") else None
            _rows_match(trk.update(d, img, e), orc.update(d.copy(), None, None if e is None else e.copy()) if kind == "botsort"
                        else orc.update(d.copy(), img), t)
        _check_state(trk, orc)
        assert trk.capacity()[2] >= 1
        trk.close()


def test_reference_unit_tests_for_oriented_boxes():
    """tests/unit/test_trackers.py:297-311, :382-392"""
    from boxmot_amd import BotSort, ByteTrack
    rgb = np.random.default_rng(0).integers(0, 255, size=(640, 640, 3), dtype=np.uint8)
    det = np.array([[320, 240, 80, 40, 0.15, 0.95, 0]], dtype=np.float32)
    for tracker in (BotSort(reid_model=None, with_reid=False, use_cmc=False), ByteTrack()):
        out1 = tracker.update(det, rgb)
        out2 = tracker.update(det, rgb)
        assert out1.shape == (1, 9) and out2.shape == (1, 9)
        np.testing.assert_allclose(out2[0, :5], det[0, :5], atol=1e-2)
        tracker.reset()                                       # after a reset the next table decides the layout again
        assert tracker.update(np.array([[10, 10, 60, 90, 0.9, 0]], dtype=np.float32), rgb).shape == (1, 8)
        tracker.close()


def test_c_abi_guards_of_the_oriented_handle():
    from boxmot_amd import _lib
    lib = _lib.load()
    cfg = _lib.BotSortConfig()
    lib.boxmot_hip_botsort_default_config(ctypes.byref(cfg))
    assert cfg.is_obb == 0
    cfg.with_reid, cfg.max_tracks, cfg.max_dets, cfg.emb_dim, cfg.is_obb = 0, 64, 32, 1, 1
    cfg.reid_model_path = b"/nonexistent/weights.bin"
    cfg.with_reid = 1
    assert not lib.boxmot_hip_botsort_create(ctypes.byref(cfg)) and "oriented" in _lib.last_error()       # in-handle ReID: embeddings come as embs
    cfg.reid_model_path, cfg.with_reid = None, 0
    h = lib.boxmot_hip_botsort_create(ctypes.byref(cfg))
    assert h
    img = np.zeros((64, 64, 3), np.uint8)
    out = np.zeros((4, 9), np.float32)
    rows, obb = ctypes.c_int(0), ctypes.c_int(0)

    def update(d):
        d = np.ascontiguousarray(d, dtype=np.float32)
        return lib.boxmot_hip_botsort_update(h, d.ctypes.data, len(d), d.shape[1], None, 0, 0, img.ctypes.data, 64, 64, 3, out.ctypes.data,
                                             4, 9, ctypes.byref(rows), ctypes.byref(obb))
    assert update(np.array([[32, 32, 20, 10, 0.15, 0.95, 0]])) == 1 and rows.value == 1 and obb.value == 1
    assert np.allclose(out[0], [32, 32, 20, 10, 0.15, 1, 0.95, 0, 0], atol=1e-6)
    assert update(np.array([[22, 27, 42, 37, 0.95, 0]])) == 0 and "oriented" in _lib.last_error()        # 6 columns on an oriented handle
    warp = np.eye(2, 3)
    assert lib.boxmot_hip_botsort_set_warp(h, 0, warp.ctypes.data) == 1          # applied to the oriented tracks by the next step
    assert update(np.array([[32, 32, 20, 10, 0.15, 0.95, 0]])) == 1 and rows.value == 1
    lib.boxmot_hip_botsort_destroy(h)
    # and the other way round: 7 columns on an axis-aligned handle
    cfg.is_obb = 0
    h = lib.boxmot_hip_botsort_create(ctypes.byref(cfg))
    assert update(np.array([[32, 32, 20, 10, 0.15, 0.95, 0]])) == 0 and "is_obb" in _lib.last_error()
    lib.boxmot_hip_botsort_destroy(h)


@pytest.mark.parametrize("with_reid", [False, True])
def test_oriented_tracks_follow_camera_motion_like_the_reference_flow(with_reid):
    """BotSort(cmc=...) on oriented detections: the estimator sees the enclosing boxes, its warp is applied on the device
    (STrack.multi_gmc_obb) -- rows and the fp64 filter state against the oracle, whose flow is pinned on the reference class under
    scheduled warps (tests/test_oracle_obb.py; cv2.transform / cv2.minAreaRect restated on both sides)."""
    from boxmot_amd import BotSort
    from boxmot_amd.scenario import camera_warps, stress_frames
    from oracle.botsort_obb import BotSortObbOracle

    class Scheduled:
        def __init__(self, w):
            self.w, self.k = w, 0

        def apply(self, img, dets):
            assert np.asarray(dets).shape[1:] == (4,)
            self.k += 1
            return self.w[self.k - 1]
    n = 120
    warps = camera_warps(n, seed=7)
    trk = BotSort(reid_model=None, with_reid=with_reid, cmc=Scheduled(warps), max_tracks=128, max_dets=64, emb_dim=EMB)
    orc = BotSortObbOracle(with_reid=with_reid)
    img = np.zeros((480, 640, 3), np.uint8)
    embs = [e for _, e in stress_frames(n, seed=7)]
    for t, d in enumerate(obb_frames(n, seed=7)):
        e = embs[t] if with_reid else None
        _rows_match(trk.update(d, img, e), orc.update(d.copy(), img, None if e is None else e.copy(), warp=warps[t]), t)
    for which, recs in ((0, orc.active), (1, orc.lost)):
        d = trk.state_dump(which)
        assert list(d["ints"][:, 0]) == [r.id for r in recs]
        if d["n"]:
            ref = np.concatenate([np.array([r.mean for r in recs]), np.array([r.cov for r in recs]).reshape(-1, 100)], 1)
            # the refit goes through fp32 corner points (cv2.transform / minAreaRect are fp32): the device's sinf / cosf differ from the
            # host's by an ulp, i.e. ~3e-5 px on a 500-px coordinate, and 120 frames of filter updates carry that along; the rows above
            # are held to 2e-4, the state to the same (the emulated twin, host libm on both sides, holds 2e-10)
            err = np.abs(d["kf"] - ref)
            assert np.allclose(d["kf"], ref, rtol=1e-5, atol=2e-4), (which, float(err.max()), np.unravel_index(err.argmax(), err.shape))
    trk.close()


def test_oriented_streams_in_one_handle_equal_single_stream_trackers():
    """update_batch on an oriented multi-stream handle: every stream's rows equal those of a single-stream ByteTrack-free BoT-SORT fed
    the same oriented detections."""
    from boxmot_amd import BotSort
    from boxmot_amd.streams import MultiStreamBotSort
    S, n = 3, 60
    ms = MultiStreamBotSort(S, max_tracks=128, max_dets=64, emb_dim=1, is_obb=True, with_reid=False)
    singles = [BotSort(reid_model=None, with_reid=False, use_cmc=False, max_tracks=128, max_dets=64) for _ in range(S)]
    frames = [list(obb_frames(n, seed=4 + s)) for s in range(S)]
    img = np.zeros((480, 640, 3), np.uint8)
    for t in range(n):
        got = ms.update_batch([frames[s][t] for s in range(S)])
        for s in range(S):
            want = singles[s].update(frames[s][t], img)
            assert got[s].shape == want.shape and np.array_equal(np.asarray(got[s]), np.asarray(want)), (t, s)
    ms.close()
    for trk in singles:
        trk.close()


@pytest.mark.parametrize("method", ["ecc", "sof"])
def test_in_handle_estimators_on_an_oriented_handle_equal_the_estimator_objects(method):
    """cmc_method = "ecc" / "sof" inside an oriented handle (SOF masked by the enclosing boxes computed on the device) against the same
    estimator as a Python object handing its warp to set_warp (BotSort(cmc=...), which computes the enclosing boxes on the host): the same
    warps, therefore the same rows."""
    from boxmot_amd import BotSort
    from boxmot_amd.cmc import get_cmc_method
    from boxmot_amd.streams import MultiStreamBotSort
    rng = np.random.default_rng(2)
    base = rng.integers(0, 255, (480, 640, 3), dtype=np.uint8)
    n = 40
    frames = list(obb_frames(n, seed=6))
    inside = MultiStreamBotSort(1, max_tracks=128, max_dets=64, emb_dim=1, is_obb=True, with_reid=False, cmc_method=method)
    outside = BotSort(reid_model=None, with_reid=False, cmc=get_cmc_method(method)(), max_tracks=128, max_dets=64)
    for t, d in enumerate(frames):
        img = np.roll(base, (t % 5, 2 * (t % 3)), axis=(0, 1))
        got = inside.update_batch([d], [img])[0]
        want = outside.update(d, img)
        assert got.shape == want.shape and np.array_equal(np.asarray(got), np.asarray(want)), t
    inside.close()
    outside.close()


def test_ocsort_centroid_obb_matches_the_oracle():
    from boxmot_amd import OcSort
    from oracle.ocsort_obb import OcSortObbOracle
    kw = dict(asso_func="centroid", iou_threshold=0.9, use_byte=True)
    trk, orc = OcSort(max_tracks=128, max_dets=64, **kw), OcSortObbOracle(**kw)
    img = np.zeros((480, 640, 3), np.uint8)
    for t, d in enumerate(obb_frames(90, seed=4)):
        got, want = np.asarray(trk.update(d, img)), orc.update(d.copy(), img)
        assert got.size == want.size, t
        if want.size:
            _rows_match(got, want, t)
    assert trk.asso_func_name == "centroid_obb"
    trk.close()


# ---- OC-SORT ----
@pytest.mark.parametrize("key", ["ocsort", "ocsort_byte"])
def test_ocsort_oriented_detections_reproduce_the_reference_rows(key):
    from boxmot_amd import OcSort
    want, frames, seed = obb_golden_rows(key)
    img = np.zeros((480, 640, 3), np.uint8)
    trk = OcSort(max_tracks=128, max_dets=64, **({} if key == "ocsort" else dict(use_byte=True, max_age=8, min_hits=1)))
    rows = 0
    for t, d in enumerate(obb_frames(frames, seed=seed)):
        got = trk.update(d, img)
        if len(want[t]) == 0:
            assert got.size == 0, t
            continue
        assert got.shape[1] == 9 and got.is_obb
        _rows_match(got, want[t], t)
        rows += len(got)
    assert rows > 200 and trk.is_obb
    trk.close()


@pytest.mark.parametrize("kw,seed", [(dict(use_byte=True), 9), (dict(max_age=5, min_hits=1, delta_t=2, inertia=0.4, iou_threshold=0.2), 13)])
def test_ocsort_oriented_step_matches_the_oracle_and_its_filter_state(kw, seed):
    """Other seeds / option sets than the golden file's, 100 frames, small initial tables (they grow: 90-double filter rows carried
    over); the state of every track at the end: 9-state mean and covariance, ages, streaks."""
    from boxmot_amd import OcSort
    from oracle.ocsort_obb import OcSortObbOracle
    img = np.zeros((480, 640, 3), np.uint8)
    for seed in (seed,):
        trk, orc = OcSort(max_tracks=16, max_dets=8, **kw), OcSortObbOracle(**kw)
        for t, d in enumerate(obb_frames(100, seed=seed)):
            want = np.asarray(orc.update(d.copy(), img), dtype=np.float32).reshape(-1, 9)
            got = np.asarray(trk.update(d, img)).reshape(-1, 9)
            _rows_match(got, want, t)
        od, dd = orc.dump(), trk.state_dump()
        assert np.array_equal(dd["ints"][:, 0], od["id"]) and np.array_equal(dd["ints"][:, 1], od["age"])
        assert np.array_equal(dd["ints"][:, 2], od["time_since_update"]) and np.array_equal(dd["ints"][:, 3], od["hit_streak"])
        assert dd["n"] > 0 and dd["kf"].shape[1] == 90
        assert np.allclose(dd["kf"][:, :9], od["x"], rtol=1e-8, atol=1e-9)
        assert np.allclose(dd["kf"][:, 9:].reshape(-1, 9, 9), od["P"], rtol=1e-7, atol=1e-8)
        assert trk.capacity()[2] >= 1
        trk.close()


def test_ocsort_oriented_surface_and_c_abi_guards():
    from boxmot_amd import _lib, OcSort
    rgb = np.zeros((640, 640, 3), np.uint8)
    det = np.array([[320, 240, 80, 40, 0.15, 0.95, 0]], dtype=np.float32)
    tracker = OcSort()
    out1 = tracker.update(det, rgb)
    out2 = tracker.update(det, rgb)
    assert out1.shape == (1, 9) and out2.shape == (1, 9)
    np.testing.assert_allclose(out2[0, :5], det[0, :5], atol=1e-2)
    tracker.reset()
    assert tracker.update(np.array([[10, 10, 60, 90, 0.9, 0]], dtype=np.float32), rgb).shape == (1, 8)
    tracker.close()
    lib = _lib.load()
    cfg = _lib.DeepOcSortConfig()
    lib.boxmot_hip_deepocsort_default_config(ctypes.byref(cfg))
    assert cfg.is_obb == 0
    cfg.max_tracks, cfg.max_dets, cfg.is_obb = 64, 32, 1
    assert not lib.boxmot_hip_deepocsort_create(ctypes.byref(cfg)) and "OC-SORT" in _lib.last_error()       # DeepOCSORT has no oriented mode
    cfg.embedding_off, cfg.cmc_off, cfg.asso_func = 1, 1, 1                   # giou has no oriented twin
    assert not lib.boxmot_hip_deepocsort_create(ctypes.byref(cfg)) and "rotated IoU" in _lib.last_error()
    cfg.asso_func = 0
    h = lib.boxmot_hip_deepocsort_create(ctypes.byref(cfg))
    assert h
    warp = np.eye(2, 3)
    assert lib.boxmot_hip_deepocsort_set_warp(h, 0, warp.ctypes.data) == 0 and "oriented" in _lib.last_error()
    out = np.zeros((4, 9), np.float32)
    rows, obb = ctypes.c_int(0), ctypes.c_int(0)
    d6 = np.array([[22, 27, 42, 37, 0.95, 0]], dtype=np.float32)
    assert lib.boxmot_hip_deepocsort_update(h, d6.ctypes.data, 1, 6, None, 0, 0, rgb.ctypes.data, 640, 640, 3, out.ctypes.data, 4, 9,
                                            ctypes.byref(rows), ctypes.byref(obb)) == 0 and "oriented" in _lib.last_error()
    assert lib.boxmot_hip_deepocsort_update(h, det.ctypes.data, 1, 7, None, 0, 0, rgb.ctypes.data, 640, 640, 3, out.ctypes.data, 4, 9,
                                            ctypes.byref(rows), ctypes.byref(obb)) == 1 and rows.value == 1 and obb.value == 1
    assert np.allclose(out[0], [320, 240, 80, 40, 0.15, 1, 0.95, 0, 0], atol=1e-5)
    lib.boxmot_hip_deepocsort_destroy(h)


@pytest.mark.parametrize("key", ["bytetrack", "botsort_reid", "ocsort"])
def test_oriented_trackers_at_the_configuration_2_shape_reproduce_the_reference_rows(key):
    """64 oriented detections per frame on 256 tracks, 1080p -- BASELINE configuration 2's shape -- against rows of the reference classes
    (tests/golden/obb_config2_golden.npz), all 60 frames, tables starting small (they grow)."""
    from boxmot_amd import BotSort, ByteTrack, OcSort
    from common import GOLDEN, obb_config2_frames
    g = np.load(GOLDEN / "obb_config2_golden.npz")
    rows, counts = g[key + "_rows"], g[key + "_counts"]
    img = np.zeros((1080, 1920, 3), np.uint8)
    if key == "bytetrack":
        trk = ByteTrack(max_tracks=64, max_dets=32)
    elif key == "ocsort":
        trk = OcSort(use_byte=True, max_tracks=64, max_dets=32)
    else:
        trk = BotSort(reid_model=None, use_cmc=False, with_reid=True, max_tracks=64, max_dets=32, emb_dim=EMB)
    o = 0
    for t, (d, e) in enumerate(obb_config2_frames(int(g["frames"]))):
        got = trk.update(d, img, e) if key == "botsort_reid" else trk.update(d, img)
        _rows_match(got, rows[o:o + counts[t]], t)
        o += counts[t]
    assert o == len(rows) and trk.capacity()[2] >= 1
    trk.close()
