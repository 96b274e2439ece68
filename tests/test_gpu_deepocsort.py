"""GPU parity tests for DeepOCSORT: the HIP step, called through the C ABI, against the committed golden rows
of the real reference and against the oracle on the same seeded inputs."""
import numpy as np
import pytest

from common import DEEPOCSORT_CASES, assert_rows_match, deepocsort_golden_rows

pytestmark = pytest.mark.gpu


def _tracker(**kw):
    from boxmot_amd.deepocsort import DeepOcSort
    return DeepOcSort(cmc_off=True, **kw)


def _check_state(trk, orc, embedding_off=False):
    od, d = orc.dump(), trk.state_dump()
    assert np.array_equal(d["ints"][:, 0], od["id"])
    assert np.array_equal(d["ints"][:, 1], od["age"])
    assert np.array_equal(d["ints"][:, 2], od["time_since_update"])
    assert np.array_equal(d["ints"][:, 3], od["hit_streak"])
    if d["n"]:
        x = d["kf"][:, :7]
        P = d["kf"][:, 8:].reshape(-1, 8, 8)[:, :7, :7]
        assert np.allclose(x, od["x"], rtol=1e-9, atol=1e-9)
        assert np.allclose(P, od["P"], rtol=1e-8, atol=1e-9)
        if not embedding_off:
            for r, emb in enumerate(od["emb"]):
                assert np.abs(d["emb"][r] - emb).max() < 1e-6
    assert d["id_count"] + 1 == od["count"]


@pytest.mark.parametrize("name", list(DEEPOCSORT_CASES))
def test_hip_deepocsort_matches_reference_golden_and_oracle(name):
    from oracle.deepocsort import DeepOcSortOracle
    make, hw, kw, dim = DEEPOCSORT_CASES[name]
    frames = make()
    want, g = deepocsort_golden_rows(name)
    img = np.zeros((hw[0], hw[1], 3), dtype=np.uint8)
    trk = _tracker(emb_dim=dim, max_tracks=512 if name == "docs_c2" else 256, max_dets=256, **kw)
    orc = DeepOcSortOracle(**kw)
    for t, (dets, embs) in enumerate(frames):
        got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
        assert_rows_match(got, want[t], t)                                            # reference (golden)
        assert_rows_match(got, orc.update(dets, img, embs.copy()).reshape(-1, 8), t)   # oracle, same inputs
    _check_state(trk, orc, embedding_off=kw.get("embedding_off", False))
    assert np.array_equal(trk.state_dump()["ints"][:, 0], g[name + "_final_ids"])
    trk.close()


@pytest.mark.parametrize("seed", [22, 24, 28])
def test_hip_deepocsort_tie_prone_scenes(seed):
    """More detections than tracks with several optimal assignments: the oracle runs with the device solver's tie
    rule (the reference's comes from lapx, which is unpinned), everything else is compared exactly."""
    from boxmot_amd.scenario import stress_frames
    from oracle.deepocsort import DeepOcSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for kw in ({}, dict(max_age=8, min_hits=2, iou_threshold=0.2)):
        trk, orc = _tracker(emb_dim=32, max_tracks=128, max_dets=64, **kw), DeepOcSortOracle(**kw)
        for t, (dets, embs) in enumerate(stress_frames(120, seed=seed, max_objects=30)):
            got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
            assert_rows_match(got, orc.update(dets, img, embs.copy()).reshape(-1, 8), t)
        _check_state(trk, orc)
        trk.close()


@pytest.mark.parametrize("name,seed,kw", [("docs_warp_default", 7, {}), ("docs_warp_short", 11, dict(max_age=6, min_hits=1))])
def test_hip_deepocsort_camera_motion_correction(name, seed, kw):
    """cmc_off=False with a warp provider: apply_affine_correction on the device; golden rows from the reference driven
    with the same scheduled warps."""
    from boxmot_amd.deepocsort import DeepOcSort
    from boxmot_amd.scenario import camera_warps, stress_frames
    from common import GOLDEN
    from oracle.deepocsort import DeepOcSortOracle

    class Scheduled:
        def __init__(self, warps):
            self.warps, self.k = warps, 0

        def apply(self, img, boxes):
            self.k += 1
            return self.warps[self.k - 1]

    g = np.load(GOLDEN / "deepocsort_golden.npz")
    rows, counts = g[name + "_rows"], g[name + "_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    frames = stress_frames(120, seed=seed)
    warps = camera_warps(len(frames), seed=seed)
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    trk = DeepOcSort(cmc_off=False, cmc=Scheduled(warps), emb_dim=32, max_tracks=256, max_dets=64, **kw)
    orc = DeepOcSortOracle(**kw)
    for t, (dets, embs) in enumerate(frames):
        got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
        assert_rows_match(got, rows[offs[t]:offs[t + 1]], t)
        assert_rows_match(got, orc.update(dets, img, embs.copy(), warp=warps[t]).reshape(-1, 8), t)
    _check_state(trk, orc)
    assert np.array_equal(trk.state_dump()["ints"][:, 0], g[name + "_final_ids"])
    trk.close()


def test_hip_deepocsort_per_class_matches_the_reference_fan_out():
    """per_class=True: one track list per class, rewound frame counter, shared id counter (basetracker.py:223-263)."""
    from boxmot_amd.scenario import stress_frames
    from oracle.deepocsort import PerClassDeepOcSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    trk = _tracker(emb_dim=32, max_tracks=128, max_dets=64, per_class=True, nr_classes=3, min_hits=1)
    orc = PerClassDeepOcSortOracle(3, min_hits=1)
    for t, (dets, embs) in enumerate(stress_frames(80, seed=4)):
        got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
        assert_rows_match(got, np.asarray(orc.update(dets, img, embs.copy()), dtype=np.float32).reshape(-1, 8), t)
    # reference test_per_class_isolation (tests/unit/test_trackers.py:537-557): overlapping boxes of two classes -> two ids
    trk2 = _tracker(emb_dim=8, max_tracks=64, max_dets=16, per_class=True, nr_classes=3)
    det = np.array([[100, 100, 150, 150, 0.9, 1], [102, 102, 152, 152, 0.9, 2]])
    out = trk2.update(det, np.zeros((640, 640, 3), np.uint8), np.random.rand(2, 8))
    assert len(set(out[:, 4].tolist())) == 2
    trk.close(); trk2.close()


def test_deepocsort_surface_and_edge_inputs():
    from boxmot_amd import create_tracker
    from boxmot_amd.deepocsort import DeepOcSort
    from boxmot_amd.track_results import TrackResults
    from boxmot_amd.cmc import HipSOF
    dflt = DeepOcSort(embedding_off=True, max_tracks=64, max_dets=32)      # reference default cmc_off=False: the built-in SOF estimator
    assert isinstance(dflt.cmc, HipSOF)
    dflt.close()
    trk = create_tracker("deepocsort", cmc_off=True, embedding_off=True, max_tracks=64, max_dets=32)
    img = np.zeros((240, 320, 3), dtype=np.uint8)
    out = trk.update(np.empty((0, 6), dtype=np.float32), img)
    assert isinstance(out, TrackResults) and out.shape == (0, 0)          # deepocsort.py:490-492
    out = trk.update(None, img)
    assert out.shape == (0, 0)
    d = np.array([[10, 10, 60, 110, 0.9, 0], [100, 50, 150, 160, 0.2, 1]], dtype=np.float32)
    out = trk.update(d, img)
    assert out.shape == (1, 8) and out[0, 4] == 1 and out[0, 7] == 0       # below det_thresh is dropped; ids start at 1
    with pytest.raises(AssertionError):
        trk.update(np.zeros((2, 5), dtype=np.float32), img)
    trk.reset()
    out = trk.update(d, img)
    assert out[0, 4] == 1
    trk.close()


def test_hip_ocsort_matches_oracle_and_surface():
    """OC-SORT = the DeepOCSORT step without appearance / camera terms (boxmot_amd.deepocsort.OcSort); the oracle wrapper
    is pinned on the reference OcSort class (tests/test_oracle_vs_reference.py, tests/golden/mot17_golden.npz)."""
    from boxmot_amd import OcSort, create_tracker
    from boxmot_amd.scenario import stress_frames
    from oracle.deepocsort import OcSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for kw in ({}, dict(max_age=5, min_hits=1, delta_t=2), dict(det_thresh=0.6, inertia=0.1, iou_threshold=0.2), dict(use_byte=True),
               dict(use_byte=True, min_conf=0.2, det_thresh=0.6, inertia=0.1)):
        trk, orc = OcSort(max_tracks=128, max_dets=64, **kw), OcSortOracle(**kw)
        for t, (dets, embs) in enumerate(stress_frames(100, seed=3)):
            got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)          # embeddings are accepted and ignored
            assert_rows_match(got, np.asarray(orc.update(dets, img), dtype=np.float32).reshape(-1, 8), t)
        trk.close()
    with pytest.raises(TypeError):
        OcSort(embedding_off=False)
    trk = create_tracker("ocsort", max_tracks=64, max_dets=32)                    # ocsort.yaml defaults: det_thresh 0.6
    d = np.array([[10, 10, 60, 110, 0.9, 0], [100, 50, 150, 160, 0.5, 1]], dtype=np.float32)
    out = trk.update(d, img)
    assert out.shape == (1, 8) and out[0, 4] == 1 and out[0, 7] == 0
    assert trk.update(np.empty((0, 6), dtype=np.float32), img).shape == (0, 0)
    trk.close()


@pytest.mark.parametrize("asso_func,thr", [("giou", 0.6), ("diou", 0.6), ("ciou", 0.6), ("hmiou", 0.3), ("centroid", 0.9)])
def test_hip_association_functions_match_the_oracle(asso_func, thr):
    """BaseTracker's ``asso_func`` (basetracker.py:28; iou.py:118-423) through the plugin classes: DeepOcSort and OcSort (with its
    BYTE round) constructed with each axis-aligned name, against the oracle whose functions are pinned bit for bit on the reference
    classes (tests/test_oracle_vs_reference.py).  `centroid` takes the frame size from the first image, as the reference does."""
    from boxmot_amd import DeepOcSort, OcSort
    from boxmot_amd.scenario import stress_frames
    from oracle.deepocsort import DeepOcSortOracle, OcSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    trk = DeepOcSort(reid_model=None, cmc_off=True, emb_dim=32, max_tracks=128, max_dets=64, asso_func=asso_func, iou_threshold=thr)
    orc = DeepOcSortOracle(asso_func=asso_func, iou_threshold=thr)
    rows = 0
    for t, (dets, embs) in enumerate(stress_frames(80, seed=5)):
        got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
        assert_rows_match(got, np.asarray(orc.update(dets, img, embs), dtype=np.float32).reshape(-1, 8), t)
        rows += len(got)
    assert rows > 200
    trk.close()
    trk, orc = OcSort(max_tracks=128, max_dets=64, asso_func=asso_func, iou_threshold=thr, use_byte=True), \
        OcSortOracle(asso_func=asso_func, iou_threshold=thr, use_byte=True)
    for t, (dets, _) in enumerate(stress_frames(80, seed=6)):
        got = np.asarray(trk.update(dets, img)).reshape(-1, 8)
        assert_rows_match(got, np.asarray(orc.update(dets, img), dtype=np.float32).reshape(-1, 8), t)
    trk.close()


@pytest.mark.parametrize("name", ["giou", "diou", "ciou", "hmiou", "centroid"])
def test_hip_association_functions_match_reference_golden_rows(name):
    """The same plugin classes against rows the REFERENCE DeepOcSort / OcSort(use_byte=True) produced with each association function
    (tests/golden/asso_golden.npz, tests/golden/make_asso_golden.py): ids, confidences, classes, detection indices and row order exact."""
    from boxmot_amd import DeepOcSort, OcSort
    from common import ASSO_FUNCS, asso_golden_rows
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for tracker in ("deepocsort", "ocsort"):
        want, frames = asso_golden_rows(tracker, name)
        kw = dict(max_tracks=128, max_dets=64, asso_func=name, iou_threshold=ASSO_FUNCS[name])
        trk = DeepOcSort(reid_model=None, cmc_off=True, emb_dim=32, **kw) if tracker == "deepocsort" else OcSort(use_byte=True, **kw)
        for t, (dets, embs) in enumerate(frames()):
            got = np.asarray(trk.update(dets, img, embs) if tracker == "deepocsort" else trk.update(dets, img)).reshape(-1, 8)
            assert_rows_match(got, want[t], t)
        trk.close()


def test_hip_unknown_association_function_raises_on_the_first_frame():
    from boxmot_amd import OcSort
    trk = OcSort(max_tracks=64, max_dets=32, asso_func="nope")          # the reference resolves the name on the first frame too
    with pytest.raises(ValueError, match="Invalid association mode: nope"):
        trk.update(np.empty((0, 6), dtype=np.float32), np.zeros((64, 64, 3), np.uint8))
    trk.close()
