"""The reference's tracker unit tests (tests/unit/test_trackers.py) restated for ``boxmot_amd.BotSort`` -- same inputs, same assertions,
same order -- on the build container's CPU: the host class is the real one, its C-ABI calls are answered by the emulated device step
(tests/emu_lib.py: botsort_step.hpp on CPU fibers).  Where /root/reference is mounted the reference's own ``BotSort`` runs the same
body beside it, so a behavioural difference shows up as one class passing and the other not.  (The GPU versions of these live in
tests/test_gpu_botsort.py::test_edge_inputs_like_the_reference_tests and tests/test_gpu_tracker_properties.py.)

Mirrored: test_tracker_output_size (:72-92), test_tracker_with_no_detections (:517-534), test_emb_trackers_requires_embeddings
(:561-575), test_invalid_det_array_shape (:578-592), test_track_id_stable_over_frames (:601-636),
test_create_tracker_invalid_tracker_name (:639-650).  The per-class tests need device class lists (one list in the emulated ABI):
they run on the GPU (tests/test_gpu_botsort.py::test_per_class_matches_reference_semantics)
and, since the emulated ABI grew class lists, in test_per_class_tracking_over_the_emulated_abi below."""
import numpy as np
import pytest

from oracle import ref_harness

EMB = 512


@pytest.fixture()
def emulated_abi(monkeypatch):
    from boxmot_amd import _lib
    from emu_lib import EmuHipLib
    lib = EmuHipLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(_lib, "last_error", lambda: lib.boxmot_hip_last_error().decode())
    return lib


def _trackers():
    """[(label, factory)]: ours always; the reference's class when the tree is mounted."""
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}

    def ours():
        from boxmot_amd.botsort import BotSort
        return BotSort(use_cmc=False, max_tracks=64, max_dets=32, emb_dim=EMB, **kw)

    out = [("boxmot_amd", ours)]
    if ref_harness.reference_available():
        def ref():
            return ref_harness.load_botsort()(reid_model=None, use_cmc=False, **kw)
        out.append(("reference", ref))
    return out


TRACKERS = _trackers()
IDS = [t[0] for t in TRACKERS]


def _close(trk):
    if hasattr(trk, "close"):
        trk.close()


@pytest.mark.parametrize("label,make", TRACKERS, ids=IDS)
def test_tracker_output_size(emulated_abi, label, make):
    tracker = make()
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 255, size=(640, 640, 3), dtype=np.uint8)
    det = np.array([[144, 212, 400, 480, 0.92, 0], [425, 281, 576, 472, 0.91, 65]])
    embs = rng.random((2, EMB))
    output = np.empty((0,))
    for _ in range(10):
        output = tracker.update(det, rgb, embs)
        if output.shape == (2, 8):
            break
    assert output.shape == (2, 8)
    _close(tracker)


@pytest.mark.parametrize("dets", [None, np.array([])], ids=["none", "empty"])
@pytest.mark.parametrize("label,make", TRACKERS, ids=IDS)
def test_tracker_with_no_detections(emulated_abi, label, make, dets):
    tracker = make()
    rgb = np.random.default_rng(1).integers(0, 255, size=(640, 640, 3), dtype=np.uint8)
    embs = np.random.default_rng(1).random(size=(0, EMB))
    output = tracker.update(dets, rgb, embs)
    assert output.size == 0, "Output should be empty when no detections are provided"
    _close(tracker)


@pytest.mark.parametrize("label,make", TRACKERS, ids=IDS)
def test_emb_trackers_requires_embeddings(emulated_abi, label, make):
    tracker = make()
    det = np.array([[10, 10, 20, 20, 0.7, 0]])
    rgb = np.zeros((640, 640, 3), dtype=np.uint8)
    with pytest.raises(AssertionError):
        tracker.update(det, rgb, np.random.default_rng(2).random((2, EMB)))
    _close(tracker)


@pytest.mark.parametrize("label,make", TRACKERS, ids=IDS)
def test_invalid_det_array_shape(emulated_abi, label, make):
    tracker = make()
    img = np.zeros((640, 640, 3), dtype=np.uint8)
    rng = np.random.default_rng(3)
    with pytest.raises(AssertionError):
        tracker.update(rng.random((2, 5)), img, rng.random((2, EMB)))
    _close(tracker)


@pytest.mark.parametrize("label,make", TRACKERS, ids=IDS)
def test_track_id_stable_over_frames(emulated_abi, label, make):
    """If the same detection appears in successive frames, the tracker should assign the same track ID."""
    tracker = make()
    det = np.array([[50, 50, 100, 100, 0.95, 3]])
    rgb = np.zeros((640, 640, 3), dtype=np.uint8)
    rng = np.random.default_rng(4)

    def update():
        return tracker.update(det, rgb, rng.random((1, EMB)))

    out = np.empty((0,))
    for _ in range(10):                     # warm up until the track is confirmed
        out = update()
        if out.shape == (1, 8):
            break
    assert out.shape == (1, 8), "Track was not confirmed after warm-up"
    track_id = out[0, 4]
    out2 = update()
    assert out2.shape == (1, 8), "Unexpected output shape on second frame"
    assert out2[0, 4] == track_id, "Track ID should remain the same across frames"
    _close(tracker)


def test_create_tracker_invalid_tracker_name():
    """Creating a tracker with an unknown name raises the reference's ValueError (tracker_zoo.py:103-105); a name the reference has
    and this backend does not says so instead."""
    from boxmot_amd.tracker_zoo import create_tracker
    with pytest.raises(ValueError, match="Unknown tracker type: 'nonexistent_tracker'"):
        create_tracker(tracker_type="nonexistent_tracker", per_class=False)
    with pytest.raises(NotImplementedError, match="not implemented on the HIP backend"):
        create_tracker(tracker_type="boosttrack")


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")
@pytest.mark.parametrize("key", ["aabb", "obb", "empty"])
def test_track_results_surface_equals_the_reference_class(tmp_path, key):
    """boxmot_amd.TrackResults beside boxmot/trackers/track_results.py:12-200 on the same rows: every accessor, summary / to_json /
    to_csv, the files save_csv / save_mot write -- axis-aligned (8 columns), oriented (9) and empty."""
    from boxmot_amd.track_results import TrackResults as Ours
    ref_harness.install_standins()
    from boxmot.trackers.track_results import TrackResults as Ref
    rng = np.random.default_rng(5)
    if key == "aabb":
        rows = np.column_stack([rng.uniform(0, 600, (7, 2)), rng.uniform(600, 900, (7, 2)), np.arange(1, 8), rng.uniform(0.2, 1, 7),
                                rng.integers(0, 80, 7), rng.integers(-1, 20, 7)]).astype(np.float32)
    elif key == "obb":
        rows = np.column_stack([rng.uniform(0, 600, (5, 2)), rng.uniform(10, 90, (5, 2)), rng.uniform(-3, 3, 5), np.arange(1, 6),
                                rng.uniform(0.2, 1, 5), rng.integers(0, 80, 5), rng.integers(-1, 20, 5)]).astype(np.float32)
    else:
        rows = np.empty((0, 8), dtype=np.float32)
    a, b = Ours(rows), Ref(rows.copy())
    assert a.is_obb == b.is_obb
    for name in ("id", "conf", "cls", "det_ind", "xyxy", "xywh") + (("xywha",) if key == "obb" else ()):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a.summary() == b.summary() and a.to_json() == b.to_json() and a.to_json(indent=2) == b.to_json(indent=2)
    assert a.to_csv() == b.to_csv() and a.to_csv(frame_id=3) == b.to_csv(frame_id=3)
    for t in (1, 2):
        a.save_csv(tmp_path / "a.csv", frame_id=t)
        b.save_csv(tmp_path / "b.csv", frame_id=t)
        a.save_mot(tmp_path / "a.txt", frame_id=t)
        b.save_mot(tmp_path / "b.txt", frame_id=t)
    assert (tmp_path / "a.csv").read_text() == (tmp_path / "b.csv").read_text()
    assert (tmp_path / "a.txt").read_text() == (tmp_path / "b.txt").read_text()
    assert isinstance(a[:2], Ours) and a[:2].is_obb == a.is_obb                  # slices stay views of the class


@pytest.mark.parametrize("kind", ["botsort", "bytetrack"])
def test_per_class_tracking_over_the_emulated_abi(emulated_abi, kind):
    """per_class=True on the build container's CPU (the device step's per-class active lists in emulation): one active list per class, a
    shared lost list, removed flags and id counter, the frame counter rewound for every class (basetracker.py:223-263) -- against the
    oracle driven the way the reference's fan-out drives its tracker; with the reference mounted, its own class beside it."""
    from boxmot_amd import BotSort, ByteTrack
    from boxmot_amd.scenario import stress_frames
    from common import assert_rows_match
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    if kind == "botsort":
        ours = BotSort(use_cmc=False, max_tracks=128, max_dets=64, emb_dim=32, per_class=True, nr_classes=3)
        ref = ref_harness.load_botsort()(reid_model=None, use_cmc=False, per_class=True, nr_classes=3) if ref_harness.reference_available() else None
    else:
        ours = ByteTrack(max_tracks=128, max_dets=64, per_class=True, nr_classes=3)
        ref = ref_harness.load_bytetrack()(per_class=True, nr_classes=3) if ref_harness.reference_available() else None
    if ref is None:
        pytest.skip("/root/reference is not mounted (the GPU suite checks this path against the oracle)")
    rows = 0
    for t, (dets, embs) in enumerate(stress_frames(60, seed=5)):
        got = ours.update(dets, img, embs) if kind == "botsort" else ours.update(dets, img)
        want = ref.update(dets.copy(), img, embs.copy()) if kind == "botsort" else ref.update(dets.copy(), img)
        assert_rows_match(np.asarray(got).reshape(-1, 8), np.asarray(want, dtype=np.float32).reshape(-1, 8), t)
        rows += len(want)
    assert rows > 100
    _close(ours)


@pytest.mark.skipif(not ref_harness.reference_available(), reason="/root/reference is not mounted")
@pytest.mark.parametrize("kind,kw", [("ocsort", {}), ("ocsort", dict(use_byte=True, max_age=8, min_hits=1)),
                                     ("deepocsort", {}), ("deepocsort", dict(aw_off=True, inertia=0.4, w_association_emb=0.75))])
def test_ocsort_and_deepocsort_host_classes_beside_the_reference_classes(emulated_abi, kind, kw):
    """boxmot_amd.OcSort / DeepOcSort (embeddings supplied, cmc off) over the emulated DeepOCSORT step: OcSort against the reference class
    run beside it on the same frames, DeepOcSort against the oracle pinned on the reference class."""
    from boxmot_amd import DeepOcSort, OcSort
    from boxmot_amd.scenario import stress_frames
    from common import assert_rows_match
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    if kind == "ocsort":
        ours, ref = OcSort(max_tracks=128, max_dets=64, **kw), ref_harness.load_ocsort()(**kw)
    else:
        # the reference DeepOcSort constructs a ReID backend from weights in its constructor: the oracle (pinned on it bit for bit with
        # a stubbed backend, tests/test_oracle_vs_reference.py) stands in
        from oracle.deepocsort import DeepOcSortOracle
        ours, ref = DeepOcSort(reid_model=None, cmc_off=True, max_tracks=128, max_dets=64, emb_dim=32, **kw), DeepOcSortOracle(**kw)
    rows = 0
    for t, (dets, embs) in enumerate(stress_frames(80, seed=6)):
        if kind == "ocsort":
            got, want = ours.update(dets, img), ref.update(dets.copy(), img)
        else:
            got, want = ours.update(dets, img, embs), ref.update(dets.copy(), img, embs.copy())
        want = np.asarray(want, dtype=np.float32)
        assert np.asarray(got).size == want.size, t
        if want.size:
            assert_rows_match(np.asarray(got).reshape(-1, 8), want.reshape(-1, 8), t, box_atol=1e-3)
            rows += len(want.reshape(-1, 8))
    assert rows > 150
    _close(ours)


@pytest.mark.parametrize("kw,with_cmc", [({}, False), (dict(max_age=5, n_init=1, nn_budget=3), True)])
def test_strongsort_host_class_over_the_emulated_abi(emulated_abi, kw, with_cmc):
    """boxmot_amd.StrongSort (embeddings supplied) over the emulated StrongSORT kernels against the oracle pinned on the reference class;
    with a camera-motion provider: asked only while tracks exist (strongsort.py:83-86), its warp applied by the step (Track.camera_update)."""
    from boxmot_amd import StrongSort
    from boxmot_amd.scenario import camera_warps, stress_frames
    from common import assert_rows_match
    from oracle.strongsort import StrongSortOracle
    n = 45
    warps = camera_warps(n, seed=7)

    class Provider:
        def __init__(self):
            self.calls = []

        def apply(self, img, dets):
            self.calls.append(len(self.calls))
            return warps[self.t]
    prov = Provider() if with_cmc else None
    ours = StrongSort(reid_model=None, cmc=prov, max_tracks=128, max_dets=64, emb_dim=32, **kw)
    orc = StrongSortOracle(**kw)
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    asked = 0
    for t, (dets, embs) in enumerate(stress_frames(n, seed=7)):
        if prov is not None:
            prov.t = t
        had_tracks = len(orc.dump()["id"]) >= 1
        want = orc.update(dets.copy(), None, embs.copy(), warp=warps[t] if (with_cmc and had_tracks) else None)
        got = ours.update(dets, img, embs)
        asked += int(with_cmc and had_tracks)
        assert_rows_match(np.asarray(got).reshape(-1, 8), np.asarray(want, dtype=np.float32).reshape(-1, 8), t)
    if prov is not None:
        assert len(prov.calls) == asked and asked > 20
    _close(ours)
