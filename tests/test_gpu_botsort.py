"""GPU parity tests proper: the HIP tracker, called through the C ABI, against the oracle on the
same seeded inputs and against the committed golden rows of the real reference."""
import numpy as np
import pytest

from common import CASES, assert_rows_match, golden_rows

pytestmark = pytest.mark.gpu


def _botsort(**kw):
    from boxmot_amd.botsort import BotSort
    return BotSort(use_cmc=False, **kw)


def _kf_report(dump, od):
    ref = np.concatenate([od["mean"], od["cov"].reshape(-1, 64)], 1)
    rel = np.abs(dump["kf"] - ref) / np.maximum(np.abs(ref), 1e-300)
    rel[ref == 0] = np.abs(dump["kf"] - ref)[ref == 0]
    return float(rel.max()) if rel.size else 0.0


@pytest.mark.parametrize("name", list(CASES))
def test_hip_matches_reference_golden_and_oracle(name):
    from oracle.botsort import BotSortOracle
    make, hw, kw, dim = CASES[name]
    frames = make()
    want, g = golden_rows(name)
    img = np.zeros((hw[0], hw[1], 3), dtype=np.uint8)
    trk = _botsort(emb_dim=dim, max_tracks=1024 if name.startswith("c2") else 256, max_dets=256, **kw)
    orc = BotSortOracle(**kw)
    for t, (dets, embs) in enumerate(frames):
        got = trk.update(dets, img, embs)
        assert_rows_match(got, want[t], t)                       # reference (golden)
        assert_rows_match(got, orc.update(dets, img, embs.copy()), t)   # oracle, same inputs
    # integer state exact, Kalman state to fp64 round-off (different summation order than LAPACK)
    od = orc.dump()
    for which, key in ((0, "active"), (1, "lost")):
        d = trk.state_dump(which)
        assert np.array_equal(d["ints"][:, 0], od[key]["id"])
        assert np.array_equal(d["ints"][:, 1], od[key]["state"])
        assert np.array_equal(d["ints"][:, 2].astype(bool), od[key]["is_activated"])
        assert np.array_equal(d["ints"][:, 3], od[key]["frame_id"])
        assert np.array_equal(d["ints"][:, 4], od[key]["start_frame"])
        assert np.array_equal(d["ints"][:, 5], od[key]["tracklet_len"])
        assert _kf_report(d, od[key]) < 1e-9
        if od[key]["smooth"] is not None and d["n"]:
            assert np.abs(d["smooth"] - od[key]["smooth"]).max() < 1e-5
    assert trk.state_dump(0)["id_count"] == od["id_count"]
    assert np.array_equal(trk.state_dump(0)["ints"][:, 0], g[name + "_final_ids"])
    trk.close()


@pytest.mark.parametrize("name,seed", [("warp_default", 7), ("warp_yaml", 11)])
def test_camera_warp_application_matches_reference(name, seed):
    """use_cmc=True with a warp provider: STrack.multi_gmc runs on the device (botsort_track.py:117-132);
    golden rows come from the reference driven with the same scheduled warps."""
    from boxmot_amd.botsort import BotSort
    from boxmot_amd.scenario import camera_warps, stress_frames
    from common import GOLDEN, YAML
    from oracle.botsort import BotSortOracle

    class Scheduled:
        def __init__(self, warps):
            self.warps, self.k = warps, 0

        def apply(self, img, dets):
            assert dets.ndim == 2 and dets.shape[1] == 7
            self.k += 1
            return self.warps[self.k - 1]

    g = np.load(GOLDEN / "botsort_warp_golden.npz")
    rows, counts = g[name + "_rows"], g[name + "_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    kw = YAML if name == "warp_yaml" else {}
    frames = stress_frames(120, seed=seed)
    warps = camera_warps(len(frames), seed=seed)
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    trk = BotSort(use_cmc=True, cmc=Scheduled(warps), emb_dim=32, max_tracks=256, max_dets=64, **kw)
    orc = BotSortOracle(**kw)
    for t, (dets, embs) in enumerate(frames):
        got = trk.update(dets, img, embs)
        assert_rows_match(got, rows[offs[t]:offs[t + 1]], t)
        assert_rows_match(got, orc.update(dets, img, embs.copy(), warp=warps[t]), t)
    od = orc.dump()
    for which, key in ((0, "active"), (1, "lost")):
        d = trk.state_dump(which)
        assert np.array_equal(d["ints"][:, 0], od[key]["id"])
        assert _kf_report(d, od[key]) < 1e-9
    assert np.array_equal(trk.state_dump(0)["ints"][:, 0], g[name + "_final_ids"])
    trk.close()


def test_without_reid_and_mixed_options():
    from boxmot_amd.scenario import stress_frames
    from oracle.botsort import BotSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for kw in (dict(with_reid=False), dict(fuse_first_associate=True, track_high_thresh=0.7, new_track_thresh=0.75),
               dict(removed_stracks_buffer=0, track_buffer=3)):
        trk, orc = _botsort(emb_dim=32, max_tracks=128, max_dets=64, **kw), BotSortOracle(**kw)
        for t, (dets, embs) in enumerate(stress_frames(120, seed=21)):
            e = None if kw.get("with_reid") is False else embs
            assert_rows_match(trk.update(dets, img, e), orc.update(dets, img, None if e is None else e.copy()), t)
        trk.close()


def test_edge_inputs_like_the_reference_tests():
    img = np.zeros((640, 640, 3), dtype=np.uint8)
    trk = _botsort(emb_dim=8, max_tracks=64, max_dets=16)
    # empty input -> empty (0, 8) output (test_trackers.py:517-534)
    for dets in (np.empty((0, 6), dtype=np.float32), None, np.array([])):
        out = trk.update(dets, img, None if dets is None or np.size(dets) == 0 else np.zeros((0, 8)))
        assert out.shape == (0, 8)
    with pytest.raises(AssertionError):                                  # embs mismatch (:560-574)
        trk.update(np.zeros((2, 6), dtype=np.float32), img, np.zeros((3, 8), dtype=np.float32))
    with pytest.raises(AssertionError):                                  # bad width (:577-591)
        trk.update(np.zeros((2, 5), dtype=np.float32), img)
    # one detection more than max_dets: the tables grow (the reference has no limit; tests/test_gpu_capacity.py)
    out = trk.update(np.tile(np.array([[0, 0, 5, 5, .9, 0]], dtype=np.float32), (17, 1)), img, np.ones((17, 8), dtype=np.float32))
    assert out.shape[1] == 8 and trk.capacity()[1] >= 17 and trk.capacity()[2] == 1
    trk.close()
    # id stability for a repeated detection (test_trackers.py:600-636)
    trk = _botsort(emb_dim=8, max_tracks=64, max_dets=16)
    det = np.array([[100, 100, 200, 300, 0.9, 0], [300, 120, 380, 320, 0.85, 1]], dtype=np.float32)
    emb = np.eye(2, 8, dtype=np.float32) + 0.01
    ids = []
    for _ in range(10):
        out = trk.update(det, img, emb)
        ids.append(out.id.tolist())
    assert ids[0] == [1, 2] and all(i == [1, 2] for i in ids)
    assert out.shape == (2, 8) and out.cls.tolist() == [0, 1] and out.det_ind.tolist() == [0, 1]
    trk.reset()
    assert trk.update(det, img, emb).id.tolist() == [1, 2]              # reset restarts the id counter
    trk.close()
    from boxmot_amd.botsort import BotSort
    from boxmot_amd.cmc import HipECC, HipSOF
    with pytest.raises(NotImplementedError, match="camera-motion"):
        BotSort(cmc_method="orb")                                       # an estimator that is not built: loud
    yaml_dflt = BotSort(cmc_method="sof", with_reid=False)              # configs/trackers/botsort.yaml: sparse optical flow, on the device
    assert isinstance(yaml_dflt.cmc, HipSOF)
    assert yaml_dflt.update(det, img).id.tolist() == [1, 2]
    yaml_dflt.close()
    dflt = BotSort(with_reid=False)                                     # constructor defaults: use_cmc=True, cmc_method="ecc" -> device ECC
    assert isinstance(dflt.cmc, HipECC)
    assert dflt.update(det, img).id.tolist() == [1, 2]
    dflt.close()


def test_per_class_matches_reference_semantics():
    """per_class=True: one active list per class, shared lost list, frame counter not advanced
    between classes (basetracker.py:223-263)."""
    from boxmot_amd.scenario import stress_frames
    from oracle.botsort import BotSortOracle

    class PerClassOracle:
        def __init__(self, n):
            self.o = BotSortOracle()
            self.lists = {c: [] for c in range(n)}
            self.n = n

        def update(self, dets, embs):
            rows, fc = [], self.o.frame_count
            for c in range(self.n):
                idx = np.where(dets[:, 5] == c)[0]
                self.o.active = self.lists[c]
                self.o.frame_count = fc
                r = self.o.update(dets[idx], None, embs[idx].copy())
                self.lists[c] = self.o.active
                if r.size:
                    rows.append(r)
            self.o.frame_count = fc + 1
            return np.vstack(rows) if rows else np.empty((0, 8), dtype=np.float32)

    img = np.zeros((480, 640, 3), dtype=np.uint8)
    trk = _botsort(emb_dim=32, max_tracks=256, max_dets=64, per_class=True, nr_classes=3)
    orc = PerClassOracle(3)
    for t, (dets, embs) in enumerate(stress_frames(90, seed=5)):
        if len(dets) == 0:
            continue
        assert_rows_match(trk.update(dets, img, embs), orc.update(dets, embs), t)
    trk.close()


def test_multi_stream_batch_equals_independent_trackers():
    from boxmot_amd.scenario import stress_frames
    from boxmot_amd.streams import MultiStreamBotSort
    from oracle.botsort import BotSortOracle
    S = 5
    seqs = [stress_frames(60, seed=100 + s) for s in range(S)]
    ms = MultiStreamBotSort(S, max_tracks=128, max_dets=64, emb_dim=32)
    orcs = [BotSortOracle() for _ in range(S)]
    for t in range(60):
        outs = ms.update_batch([seqs[s][t][0] for s in range(S)], None, [seqs[s][t][1] for s in range(S)])
        for s in range(S):
            assert_rows_match(outs[s], orcs[s].update(seqs[s][t][0], None, seqs[s][t][1].copy()), (s, t))
    assert ms.status().tolist() == [0] * S
    ms.close()


def test_full_size_properties_c2():
    """BASELINE config 2 sizes (64 dets x 256 tracks, D=512): size-independent properties."""
    from boxmot_amd.scenario import Scenario
    sc = Scenario(64, 256, random_image=False)
    trk = _botsort(emb_dim=512, max_tracks=1024, max_dets=256)
    seen_ids = set()
    for t in range(45):
        dets, embs = sc.frame(t)
        out = trk.update(dets, np.zeros((1080, 1920, 3), np.uint8), embs)
        ids = out.id
        assert len(set(ids.tolist())) == len(ids)                     # ids unique within a frame
        assert len(out) == len(dets)                                  # every detection continues a track
        assert sorted(out.det_ind.tolist()) == list(range(len(dets))) # det_ind is a permutation
        k = out.det_ind
        iou_center = np.abs((out[:, :2] + out[:, 2:4]) / 2 - (dets[k, :2] + dets[k, 2:4]) / 2).max()
        assert iou_center < 4.0                                       # filtered box stays on its detection
        seen_ids |= set(ids.tolist())
    assert seen_ids == set(range(1, 257))                             # 256 stable identities, never re-issued
    d = trk.state_dump(0)
    cov = d["kf"][:, 8:].reshape(-1, 8, 8)
    assert np.all(np.linalg.eigvalsh((cov + cov.transpose(0, 2, 1)) / 2) > 0)   # covariances stay SPD
    assert d["n"] + trk.state_dump(1)["n"] == 256
    trk.close()
