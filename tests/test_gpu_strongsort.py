"""GPU parity tests for StrongSORT: the HIP kernels, called through the C ABI, against the committed golden rows of
the real reference (identity camera motion) and against the oracle on the same seeded inputs."""
import numpy as np
import pytest

from common import STRONGSORT_CASES, assert_rows_match, strongsort_golden_rows

pytestmark = pytest.mark.gpu


def _check_state(trk, orc):
    od, d = orc.dump(), trk.state_dump()
    assert np.array_equal(d["ints"][:, 0], od["id"])
    assert np.array_equal(d["ints"][:, 1], od["state"])
    assert np.array_equal(d["ints"][:, 2], od["hits"])
    assert np.array_equal(d["ints"][:, 3], od["age"])
    assert np.array_equal(d["ints"][:, 4], od["time_since_update"])
    if d["n"]:
        ref = np.concatenate([od["mean"], od["cov"].reshape(-1, 64)], 1)
        assert np.allclose(d["kf"], ref, rtol=1e-9, atol=1e-10)
        for r, f in enumerate(od["feat"]):
            assert np.abs(d["feat"][r] - f).max() < 1e-5
        assert d["ints"][:, 5].tolist() == [od["bank"].get(int(i), 0) for i in od["id"]]
    assert d["next_id"] == od["next_id"]


@pytest.mark.parametrize("name", list(STRONGSORT_CASES))
def test_hip_strongsort_matches_reference_golden_and_oracle(name):
    from boxmot_amd.strongsort import StrongSort
    from oracle.strongsort import StrongSortOracle
    make, hw, kw, dim = STRONGSORT_CASES[name]
    frames = make()
    want, g = strongsort_golden_rows(name)
    img = np.zeros((hw[0], hw[1], 3), dtype=np.uint8)
    trk = StrongSort(emb_dim=dim, max_tracks=512 if name == "ss_c2" else 256, max_dets=256, **kw)
    orc = StrongSortOracle(**kw)
    for t, (dets, embs) in enumerate(frames):
        got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
        assert_rows_match(got, want[t], t)                                            # reference (golden)
        assert_rows_match(got, orc.update(dets, img, embs.copy()).reshape(-1, 8), t)   # oracle, same inputs
    _check_state(trk, orc)
    assert np.array_equal(trk.state_dump()["ints"][:, 0], g[name + "_final_ids"])
    trk.close()


def test_hip_strongsort_camera_update_and_seed_sweep():
    from boxmot_amd.scenario import camera_warps, stress_frames
    from boxmot_amd.strongsort import StrongSort
    from oracle.strongsort import StrongSortOracle

    class Scheduled:
        def __init__(self, warps):
            self.warps, self.k = warps, 0

        def apply(self, img, boxes):
            self.k += 1
            return self.warps[self.k - 1]

    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for seed, kw in ((5, {}), (22, dict(max_age=8, n_init=2, nn_budget=5, max_iou_dist=0.8)), (24, {})):
        frames = stress_frames(100, seed=seed, max_objects=30)
        warps = camera_warps(len(frames), seed=seed)
        trk, orc = StrongSort(emb_dim=32, max_tracks=128, max_dets=64, cmc=Scheduled(warps), **kw), StrongSortOracle(**kw)
        sched = Scheduled(warps)        # the estimator is asked only while tracks exist (strongsort.py:83-86): same gate on both sides
        for t, (dets, embs) in enumerate(frames):
            got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
            warp = sched.apply(img, None) if len(orc.tracks) >= 1 else None
            assert_rows_match(got, orc.update(dets, img, embs.copy(), warp=warp).reshape(-1, 8), t)
        assert trk.cmc.k == sched.k > 0
        _check_state(trk, orc)
        trk.close()


def test_strongsort_surface_and_edge_inputs():
    from boxmot_amd import create_tracker
    from boxmot_amd.track_results import TrackResults
    trk = create_tracker("strongsort", n_init=1, max_tracks=64, max_dets=32, emb_dim=8)
    img = np.zeros((240, 320, 3), dtype=np.uint8)
    out = trk.update(np.empty((0, 6), dtype=np.float32), img, np.empty((0, 8), dtype=np.float32))
    assert isinstance(out, TrackResults) and out.shape == (0, 8)
    d = np.array([[10, 10, 60, 110, 0.9, 0], [100, 50, 150, 160, 0.2, 1]], dtype=np.float32)
    e = np.random.default_rng(0).standard_normal((2, 8)).astype(np.float32)
    assert trk.update(d, img, e).shape == (0, 8)                # tentative on birth (YAML min_conf 0.6 drops the second)
    out = trk.update(d, img, e)
    assert out.shape == (1, 8) and out[0, 4] == 1 and out[0, 7] == 0
    with pytest.raises(AssertionError):
        trk.update(np.zeros((2, 5), dtype=np.float32), img, e)
    trk.close()


def test_strongsort_per_class_is_one_update_per_class_like_the_reference():
    """basetracker.py:223-263 swaps ``active_tracks`` per class, which StrongSort does not use (its tracks live in
    ``tracker.tracks``): per_class=True therefore means nr_classes consecutive updates of the same tracker."""
    from boxmot_amd.scenario import stress_frames
    from boxmot_amd.strongsort import StrongSort
    from oracle.strongsort import StrongSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    nc = 3
    trk = StrongSort(emb_dim=32, max_tracks=128, max_dets=64, per_class=True, nr_classes=nc, n_init=1, max_age=60)
    orc = StrongSortOracle(n_init=1, max_age=60)
    for t, (dets, embs) in enumerate(stress_frames(40, seed=4)):
        got = np.asarray(trk.update(dets, img, embs)).reshape(-1, 8)
        want = []
        for c in range(nc):
            idx = np.where(dets[:, 5] == c)[0]
            rows = orc.update(dets[idx], img, embs[idx].copy()).reshape(-1, 8)
            if rows.size:
                want.append(rows)
        want = np.vstack(want) if want else np.empty((0, 8), np.float32)
        assert_rows_match(got, want, t)
    trk.close()


@pytest.mark.parametrize("seed", [159, 104, 133])
def test_strongsort_ids_exact_under_the_device_dot_rule(seed):
    """Crowded 300-frame sequences with camera warps (seed 159 is one of the two the round-1 soak lost to a 1-ulp difference of the
    host's BLAS product): against StrongSortOracle(dot_rule="device") -- same algorithm, the kernels' documented fp32 summation
    order -- rows and ids are exact on every frame."""
    from boxmot_amd import StrongSort
    from boxmot_amd.scenario import camera_warps, stress_frames
    from oracle.strongsort import StrongSortOracle

    class Sched:
        def __init__(self, w):
            self.w, self.t = w, 0           # .t = index of the frame being updated, set by the loop: an estimator is not asked on every
                                            # frame (StrongSORT only while tracks exist, strongsort.py:83-86)
        def apply(self, img, d):
            return self.w[self.t]

    n = 300
    frames = stress_frames(n, seed=seed, max_objects=20 + seed % 17)
    warps = camera_warps(n, seed=seed)
    use_w = seed % 2 == 0
    sched = Sched(warps) if use_w else None
    trk = StrongSort(cmc=sched, emb_dim=32, max_tracks=1024, max_dets=64)
    orc = StrongSortOracle(dot_rule="device")
    img = np.zeros((480, 640, 3), np.uint8)
    for t, (d, e) in enumerate(frames):
        if sched is not None:
            sched.t = t
        got = np.asarray(trk.update(d, img, e)).reshape(-1, 8)
        want = np.asarray(orc.update(d, img, e.copy(), warp=warps[t] if use_w else None)).reshape(-1, 8)
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), (seed, t)
        assert np.allclose(got[:, :4], want[:, :4], atol=1e-3), (seed, t)
    trk.close()


def test_strongsort_soak_against_the_unmodified_oracle_reports_the_agreement_rate(capsys):
    """The same crowded sequences against the UNMODIFIED oracle (NumPy -> OpenBLAS products, LAPACK Kalman update: the reference's
    own arithmetic on this host).  Costs are within fp32 rounding of the device's, but crowded frames hold exactly tied clamped
    costs and SciPy's choice among them depends on those last bits, so ids are NOT asserted frame for frame here (that bar is
    test_strongsort_ids_exact_under_the_device_dot_rule): the test REPORTS, per seed, the first frame where the rows differ and the
    fraction of frames with identical rows, and asserts only what must hold on any host -- the row COUNT of every frame agrees up
    to the divergence and a clear majority of sequences is identical end to end."""
    from boxmot_amd import StrongSort
    from boxmot_amd.scenario import stress_frames
    from oracle.strongsort import StrongSortOracle
    img = np.zeros((480, 640, 3), np.uint8)
    seeds = [159, 104, 133, 145, 7, 21, 33, 58]
    n = 200
    report, full = [], 0
    for seed in seeds:
        frames = stress_frames(n, seed=seed, max_objects=20 + seed % 17)
        trk = StrongSort(emb_dim=32, max_tracks=1024, max_dets=64)
        orc = StrongSortOracle()                    # dot_rule default: the host BLAS / LAPACK order
        same, first = 0, None
        for t, (d, e) in enumerate(frames):
            got = np.asarray(trk.update(d, img, e)).reshape(-1, 8)
            want = np.asarray(orc.update(d, img, e.copy())).reshape(-1, 8)
            ok = got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]) and np.allclose(got[:, :4], want[:, :4], atol=1e-3)
            if ok:
                same += 1
            elif first is None:
                first = t
        trk.close()
        full += first is None
        report.append((seed, first, same / n))
    with capsys.disabled():
        print("\nStrongSORT vs the unmodified oracle (host BLAS order), 200-frame crowded sequences:")
        for seed, first, frac in report:
            print(f"  seed {seed:4d}: {'identical' if first is None else f'first difference at frame {first}'}, {100 * frac:.1f} % of frames identical")
        print(f"  {full}/{len(seeds)} sequences identical end to end")
    assert full >= len(seeds) // 2
