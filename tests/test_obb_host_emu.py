"""ORIENTED detections through the real host classes (boxmot_amd.BotSort / ByteTrack) on the build container's CPU: the classes' C-ABI
calls are answered by the emulated device step (tests/emu_lib.py: the bm::obb copy of botsort_step_body.hpp on CPU fibers).  Checks the
host side of the row -- layout inference from the first detection table, handle re-creation, 7 columns in / 9 out, the result view --
against the reference's own rows (tests/golden/obb_golden.npz) and restates the reference's OBB unit tests
(tests/unit/test_trackers.py:286-304, :382-392).  The GPU versions live in tests/test_gpu_obb.py."""
import numpy as np
import pytest

from common import obb_frames, obb_golden_rows

EMB = 32


@pytest.fixture()
def emulated_abi(monkeypatch):
    from boxmot_amd import _lib
    from emu_lib import EmuHipLib
    lib = EmuHipLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(_lib, "last_error", lambda: lib.boxmot_hip_last_error().decode())
    return lib


def _rows_match(got, want, t):
    got, want = np.asarray(got, dtype=np.float32).reshape(-1, 9), np.asarray(want, dtype=np.float32).reshape(-1, 9)
    assert got.shape == want.shape, (t, got.shape, want.shape)
    assert np.array_equal(got[:, 5:], want[:, 5:]), t                   # id, conf, cls, det_ind, row order: exact
    assert np.allclose(got[:, :5], want[:, :5], rtol=0, atol=2e-4), (t, np.abs(got[:, :5] - want[:, :5]).max())


@pytest.mark.parametrize("key", ["bytetrack", "botsort_noreid", "botsort_reid"])
def test_host_classes_reproduce_the_reference_rows_on_oriented_detections(emulated_abi, key):
    from boxmot_amd import BotSort, ByteTrack
    from boxmot_amd.scenario import stress_frames
    from boxmot_amd.track_results import TrackResults
    want, frames, seed = obb_golden_rows(key)
    img = np.zeros((480, 640, 3), np.uint8)
    if key == "bytetrack":
        trk = ByteTrack(max_tracks=128, max_dets=64)
    else:
        trk = BotSort(reid_model=None, use_cmc=False, with_reid=key == "botsort_reid", max_tracks=128, max_dets=64, emb_dim=EMB)
    assert trk.supports_obb and not trk.is_obb
    embs = [e for _, e in stress_frames(frames, seed=seed)]
    for t, d in enumerate(obb_frames(frames, seed=seed)):
        got = trk.update(d, img, embs[t] if key == "botsort_reid" else None)
        assert isinstance(got, TrackResults) and got.shape[1] == 9 and got.is_obb
        _rows_match(got, want[t], t)
        if len(got):
            assert np.array_equal(got.id, want[t][:, 5].astype(int)) and np.array_equal(got.det_ind, want[t][:, 8].astype(int))
            assert got.xywha.shape == (len(got), 5)
    assert trk.is_obb and trk.asso_func_name == "iou_obb"
    views = trk.active_tracks
    assert views and views[0].mean.shape == (10,) and views[0].covariance.shape == (10, 10) and views[0].xywha.shape == (5,)
    trk.close()


# ---- the reference's unit tests, restated ----
def test_botsort_supports_obb_without_reid(emulated_abi):
    """tests/unit/test_trackers.py:297-311 (use_cmc=False here: the default estimator, ECC, needs the device; the test after this one
    drives the oriented camera-motion path with a scheduled estimator)."""
    from boxmot_amd import BotSort
    tracker = BotSort(reid_model=None, with_reid=False, use_cmc=False, max_tracks=64, max_dets=32)
    rgb = np.random.default_rng(0).integers(0, 255, size=(640, 640, 3), dtype=np.uint8)
    det = np.array([[320, 240, 80, 40, 0.15, 0.95, 0]], dtype=np.float32)
    out1 = tracker.update(det, rgb)
    out2 = tracker.update(det, rgb)
    assert out1.shape[1] == 9
    assert out2.shape[1] == 9
    np.testing.assert_allclose(out2[0, :5], det[0, :5], atol=1e-2)
    tracker.close()


def test_botsort_with_an_estimator_tracks_oriented_detections_like_the_reference_flow(emulated_abi):
    """BotSort(cmc=...) on oriented detections: the estimator is handed the enclosing axis-aligned boxes (botsort.py:147-158), its warp
    is applied to the oriented tracks on the device (STrack.multi_gmc_obb) -- against the oracle, whose flow is pinned on the reference
    class under scheduled warps (tests/test_oracle_obb.py)."""
    from boxmot_amd import BotSort
    from boxmot_amd.scenario import camera_warps
    from oracle.botsort_obb import BotSortObbOracle
    from oracle import obb

    class Scheduled:
        def __init__(self, w):
            self.w, self.k, self.seen = w, 0, []

        def apply(self, img, dets):
            self.seen.append(np.asarray(dets).copy())
            self.k += 1
            return self.w[self.k - 1]
    n = 50
    warps = camera_warps(n, seed=4)
    est = Scheduled(warps)
    tracker, orc = BotSort(reid_model=None, with_reid=False, cmc=est, max_tracks=128, max_dets=64), BotSortObbOracle(with_reid=False)
    img = np.zeros((480, 640, 3), np.uint8)
    for t, d in enumerate(obb_frames(n, seed=4)):
        _rows_match(tracker.update(d, img), orc.update(d.copy(), img, None, warp=warps[t]), t)
        boxes = est.seen[-1]
        assert boxes.shape == (len(d), 4)
        for k in range(min(len(d), 3)):            # the enclosing box of a detection = min / max of its cv2.boxPoints corners
            c = obb.box_points(float(d[k, 0]), float(d[k, 1]), max(float(d[k, 2]), 1e-4), max(float(d[k, 3]), 1e-4), float(np.degrees(d[k, 4])))
            assert np.array_equal(boxes[k], np.array([c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()], dtype=np.float32))
    tracker.close()


def test_bytetrack_supports_obb_outputs(emulated_abi):
    """tests/unit/test_trackers.py:382-392"""
    from boxmot_amd import ByteTrack
    tracker = ByteTrack(max_tracks=64, max_dets=32)
    rgb = np.random.default_rng(0).integers(0, 255, size=(640, 640, 3), dtype=np.uint8)
    det = np.array([[320, 240, 80, 40, 0.15, 0.95, 0]], dtype=np.float32)
    out1 = tracker.update(det, rgb)
    out2 = tracker.update(det, rgb)
    assert out1.shape == (1, 9)
    assert out2.shape == (1, 9)
    np.testing.assert_allclose(out2[0, :5], det[0, :5], atol=1e-2)
    tracker.close()


def test_rotating_target_keeps_its_identity(emulated_abi):
    """The scene of test_bytetrack_obb_state_history_follows_rotation_without_flips (tests/unit/test_trackers.py:406-420): a box turning
    through almost a full revolution in steps of 0.32 rad stays one track at a fixed centre; the state (the measurement's closest
    parameterisation, theta-velocity damped) is the oracle's, row for row."""
    from boxmot_amd import ByteTrack
    from oracle.bytetrack_obb import ByteTrackObbOracle
    kw = dict(track_thresh=0.1, min_conf=0.01, match_thresh=0.99)
    tracker, orc = ByteTrack(max_tracks=64, max_dets=32, **kw), ByteTrackObbOracle(**kw)
    rgb = np.zeros((640, 640, 3), np.uint8)
    ids = set()
    for t, angle in enumerate(np.linspace(0.0, 6.1, 20, dtype=np.float32)):
        det = np.array([[320, 240, 90, 40, angle, 0.95, 0]], dtype=np.float32)
        out = tracker.update(det, rgb)
        assert out.shape == (1, 9)
        _rows_match(out, orc.update(det.copy(), rgb), t)
        ids.add(int(out.id[0]))
        assert abs(out[0, 0] - 320) < 1e-2 and abs(out[0, 1] - 240) < 1e-2
    assert len(ids) == 1
    tracker.close()


def test_layout_is_decided_once_and_wrong_widths_are_refused(emulated_abi):
    from boxmot_amd import BotSort, DeepOcSort
    rgb = np.zeros((64, 64, 3), np.uint8)
    tracker = BotSort(reid_model=None, with_reid=False, use_cmc=False, max_tracks=64, max_dets=32)
    assert tracker.update(np.empty((0, 7), np.float32), rgb).shape == (0, 9)          # an empty oriented table decides the layout too
    assert tracker.is_obb
    with pytest.raises(AssertionError, match="valid length is 7"):
        tracker.update(np.array([[10, 10, 30, 30, 0.9, 0]], dtype=np.float32), rgb)
    tracker.reset()                                                                   # reset: the next table decides again
    assert tracker.update(np.array([[10, 10, 30, 30, 0.9, 0]], dtype=np.float32), rgb).shape[1] == 8
    assert not tracker.is_obb and tracker.asso_func_name == "iou"
    tracker.close()
    # a tracker class without an oriented step: the reference's message (basetracker.py:167-171)
    trk = DeepOcSort.__new__(DeepOcSort)
    from boxmot_amd.basetracker import BaseTracker
    BaseTracker.__init__(trk, asso_func="iou")
    with pytest.raises(AssertionError, match="DeepOcSort does not support OBB detections"):
        trk.update(np.array([[32, 32, 20, 10, 0.15, 0.95, 0]], dtype=np.float32), rgb)


@pytest.mark.parametrize("key", ["ocsort", "ocsort_byte"])
def test_ocsort_host_class_reproduces_the_reference_rows_on_oriented_detections(emulated_abi, key):
    from boxmot_amd import OcSort
    from boxmot_amd.track_results import TrackResults
    want, frames, seed = obb_golden_rows(key)
    img = np.zeros((480, 640, 3), np.uint8)
    trk = OcSort(max_tracks=128, max_dets=64, **({} if key == "ocsort" else dict(use_byte=True, max_age=8, min_hits=1)))
    assert trk.supports_obb and not trk.is_obb
    for t, d in enumerate(obb_frames(frames, seed=seed)):
        got = trk.update(d, img)
        assert isinstance(got, TrackResults)
        if len(want[t]) == 0:
            assert got.size == 0, t
            continue
        assert got.shape[1] == 9 and got.is_obb
        _rows_match(got, want[t], t)
    assert trk.is_obb and trk.asso_func_name == "iou_obb"
    d = trk.state_dump()
    assert d["n"] > 0 and d["kf"].shape[1] == 90
    trk.close()


def test_ocsort_supports_obb_outputs_and_refuses_what_it_has_not(emulated_abi):
    """The shape of tests/unit/test_trackers.py:382-392 for OcSort (ocsort.py:332 supports_obb); the other association functions and a
    DeepOcSort are refused like the reference refuses them / loudly."""
    from boxmot_amd import DeepOcSort, OcSort
    rgb = np.zeros((640, 640, 3), np.uint8)
    det = np.array([[320, 240, 80, 40, 0.15, 0.95, 0]], dtype=np.float32)
    tracker = OcSort(max_tracks=64, max_dets=32)
    out1 = tracker.update(det, rgb)
    out2 = tracker.update(det, rgb)
    assert out1.shape == (1, 9) and out2.shape == (1, 9)
    np.testing.assert_allclose(out2[0, :5], det[0, :5], atol=1e-2)
    tracker.reset()
    assert tracker.update(np.array([[10, 10, 60, 90, 0.9, 0]], dtype=np.float32), rgb).shape == (1, 8)
    tracker.close()
    with pytest.raises(ValueError, match="Invalid association mode: giou_obb"):          # AssociationFunction's table has no giou_obb
        OcSort(asso_func="giou", max_tracks=64, max_dets=32).update(det, rgb)
    # centroid_obb (iou.py:263-274), the other oriented association function: against the oracle pinned on the reference class
    from oracle.ocsort_obb import OcSortObbOracle
    kw = dict(asso_func="centroid", iou_threshold=0.9, use_byte=True)
    trk, orc = OcSort(max_tracks=128, max_dets=64, **kw), OcSortObbOracle(**kw)
    img = np.zeros((480, 640, 3), np.uint8)
    for t, d in enumerate(obb_frames(40, seed=4)):
        got, want = np.asarray(trk.update(d, img)), orc.update(d.copy(), img)
        assert got.size == want.size, t
        if want.size:
            _rows_match(got, want, t)
    assert trk.asso_func_name == "centroid_obb"
    trk.close()
    trk = DeepOcSort.__new__(DeepOcSort)
    from boxmot_amd.basetracker import BaseTracker
    BaseTracker.__init__(trk, asso_func="iou")
    with pytest.raises(AssertionError, match="DeepOcSort does not support OBB detections"):
        trk.update(det, rgb)


def test_a_reserve_made_before_the_layout_is_known_survives_the_switch(emulated_abi):
    from boxmot_amd import BotSort
    trk = BotSort(reid_model=None, with_reid=False, use_cmc=False, max_tracks=64, max_dets=32)
    trk.reserve(max_tracks=256, max_dets=96)
    assert trk.capacity()[:2] == (256, 96)
    trk.update(np.array([[32, 32, 20, 10, 0.15, 0.95, 0]], dtype=np.float32), np.zeros((64, 64, 3), np.uint8))      # oriented: the handle is re-made
    assert trk.is_obb and trk.capacity()[:2] == (256, 96)
    trk.close()


def test_track_results_names_the_oriented_columns():
    from boxmot_amd.track_results import TrackResults
    r = TrackResults(np.array([[320, 240, 80, 40, 0.15, 7, 0.95, 2, 0]], dtype=np.float32))
    assert r.is_obb and r.id.tolist() == [7] and r.cls.tolist() == [2] and r.det_ind.tolist() == [0] and np.isclose(r.conf[0], 0.95)
    assert r.summary()[0]["box"] == {"cx": 320.0, "cy": 240.0, "w": 80.0, "h": 40.0, "angle": float(np.float32(0.15))}
    assert r.to_mot_lines(3) == ["3,7,320.00,240.00,80.00,40.00,0.1500,0.950000,2,-1"]
    a = TrackResults(np.array([[1, 2, 3, 4, 7, 0.5, 1, 0]], dtype=np.float32))
    assert not a.is_obb and a.id.tolist() == [7]


def test_oriented_multi_stream_handle_equals_per_stream_oracles(emulated_abi):
    """MultiStreamBotSort(is_obb=True).update_batch over the emulated ABI: every stream's 9-column rows equal its own oracle's."""
    from boxmot_amd.streams import MultiStreamBotSort
    from oracle.botsort_obb import BotSortObbOracle
    S, n = 3, 40
    ms = MultiStreamBotSort(S, max_tracks=128, max_dets=64, emb_dim=1, is_obb=True, with_reid=False)
    orcs = [BotSortObbOracle(with_reid=False) for _ in range(S)]
    frames = [list(obb_frames(n, seed=4 + s)) for s in range(S)]
    for t in range(n):
        got = ms.update_batch([frames[s][t] for s in range(S)])
        for s in range(S):
            assert got[s].shape[1] == 9 and got[s].is_obb
            _rows_match(got[s], orcs[s].update(frames[s][t].copy(), None, None), t)
    ms.close()


@pytest.mark.parametrize("key,lead", [("botsort_noreid", 3), ("bytetrack", 2)])
def test_frames_without_a_layout_before_the_first_oriented_table_keep_the_frame_numbering(emulated_abi, key, lead):
    """`update(None, img)` / empty tables before the first 7-column table: the step runs them as empty frames and the device frame
    counter advances; the handle re-made for the oriented layout starts from the HOST's frame count (round-4 advisor finding:
    a restarted counter turned the first oriented frame into "frame 1" -- immediate activation, shifted start_frame / frame_id).
    The reference keeps counting through such frames (botsort.py:172-176, bytetrack.py:259-275): run beside the oracle that is
    pinned on it, and -- where /root/reference is mounted -- beside the reference class itself."""
    from boxmot_amd import BotSort, ByteTrack
    from oracle import ref_harness
    from oracle.botsort_obb import BotSortObbOracle
    from oracle.bytetrack_obb import ByteTrackObbOracle
    _, frames, seed = obb_golden_rows(key)
    img = np.zeros((480, 640, 3), np.uint8)
    if key == "bytetrack":
        trk, orc = ByteTrack(max_tracks=128, max_dets=64), ByteTrackObbOracle()
        ref = ref_harness.load_bytetrack()() if ref_harness.reference_available() else None
    else:
        trk, orc = BotSort(reid_model=None, use_cmc=False, with_reid=False, max_tracks=128, max_dets=64), BotSortObbOracle(with_reid=False)
        ref = ref_harness.load_botsort()(reid_model=None, use_cmc=False, with_reid=False) if ref_harness.reference_available() else None
    for _ in range(lead):
        got = trk.update(None, img)
        assert len(got) == 0
        orc.update(np.empty((0, 7), np.float32), img)
        if ref is not None:
            ref.update(None, img)
    assert trk.frame_count == lead and not trk.is_obb
    seq = list(obb_frames(frames, seed=seed))[:40]
    for t, d in enumerate(seq):
        got = trk.update(d, img)
        want = orc.update(d, img)
        _rows_match(got, want, t)
        if ref is not None:
            _rows_match(got, np.asarray(ref.update(d, img)), t)
    assert trk.is_obb and trk.frame_count == lead + len(seq)
    # the first oriented frame was NOT frame 1: nothing was activated on it (activation on the first frame only, botsort_track.py:247-250)
    trk.close()
