"""The device tables grow on demand (include/boxmot_hip.h, boxmot_hip_botsort_reserve): a tracker created with tiny max_tracks /
max_dets returns, frame for frame, what the oracle (which has Python lists, like the reference) returns -- ids, filters and lists
survive every re-allocation -- and what a tracker created large returns."""
import numpy as np
import pytest

from common import CASES, assert_rows_match

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [n for n in CASES if n.startswith("stress")][:2] or list(CASES)[:2])
def test_botsort_grows_from_tiny_tables_and_matches_the_oracle(name):
    from boxmot_amd.botsort import BotSort
    from oracle.botsort import BotSortOracle
    make, hw, kw, dim = CASES[name]
    frames = make()
    img = np.zeros((hw[0], hw[1], 3), dtype=np.uint8)
    trk = BotSort(use_cmc=False, emb_dim=dim, max_tracks=8, max_dets=4, **kw)
    big = BotSort(use_cmc=False, emb_dim=dim, max_tracks=1024, max_dets=256, **kw)
    orc = BotSortOracle(**kw)
    assert trk.capacity() == (8, 4, 0)
    peak_dets = 0
    for t, (dets, embs) in enumerate(frames):
        got = trk.update(dets, img, embs)
        assert_rows_match(got, orc.update(dets, img, embs.copy()), t)
        assert_rows_match(got, big.update(dets, img, embs), t)
        peak_dets = max(peak_dets, len(dets))
    cap, nd, grows = trk.capacity()
    assert grows >= 1 and nd >= peak_dets and cap > 8
    od = orc.dump()
    for which, key in ((0, "active"), (1, "lost")):
        d, b = trk.state_dump(which), big.state_dump(which)
        assert np.array_equal(d["ints"], b["ints"]) and np.array_equal(d["ints"][:, 0], od[key]["id"])
        assert np.array_equal(d["kf"], b["kf"])                 # the copied filters are the same bits a never-grown tracker holds
        assert np.array_equal(d["smooth"], b["smooth"])
    assert big.capacity()[2] == 0
    trk.close(); big.close()


def test_bytetrack_and_per_class_lists_survive_growth():
    from boxmot_amd import ByteTrack
    from boxmot_amd.scenario import stress_frames
    from oracle.bytetrack import ByteTrackOracle, PerClassByteTrackOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    for per_class in (False, True):
        trk = ByteTrack(max_tracks=8, max_dets=4, per_class=True, nr_classes=3) if per_class else ByteTrack(max_tracks=8, max_dets=4)
        orc = PerClassByteTrackOracle(3) if per_class else ByteTrackOracle()
        for t, (d, _) in enumerate(stress_frames(90, seed=9)):
            if len(d) == 0:
                continue
            assert_rows_match(np.asarray(trk.update(d, img)).reshape(-1, 8), orc.update(d.copy(), img), t)
        assert trk.capacity()[2] >= 1
        trk.close()


def test_reid_inside_update_after_the_detection_tables_grew():
    """max_dets growth re-creates the ReID engine for the larger crop count: embeddings computed inside update must not change."""
    from boxmot_amd.botsort import BotSort
    from boxmot_amd.reid import HipReID
    from boxmot_amd.reid_weights import reference_init_state_dict
    from boxmot_amd.scenario import Scenario
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    sc = Scenario(24, 48, width=640, height=480, emb_dim=8, stream=0, random_image=True)
    small = BotSort(reid_model=HipReID(sd, mode=1), use_cmc=False, max_tracks=8, max_dets=4)
    large = BotSort(reid_model=HipReID(sd, mode=1), use_cmc=False, max_tracks=256, max_dets=64)
    for t in range(12):
        dets, _ = sc.frame(t, with_embs=False)
        a, b = small.update(dets, sc.image), large.update(dets, sc.image)
        assert np.array_equal(np.asarray(a), np.asarray(b)), t
    assert small.capacity()[2] >= 1
    small.close(); large.close()


def test_reserve_then_device_resident_steps_and_multi_stream_growth():
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    from oracle.botsort import BotSortOracle
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    S = 3
    ms = MultiStreamBotSort(S, max_tracks=8, max_dets=4, emb_dim=16, **kw)
    scs = [Scenario(12 + 4 * s, 40, emb_dim=16, stream=s, random_image=False) for s in range(S)]
    orcs = [BotSortOracle(**kw) for _ in range(S)]
    img = np.zeros((1080, 1920, 3), dtype=np.uint8)
    for t in range(10):
        fr = [sc.frame(t) for sc in scs]
        got = ms.update_batch([f[0] for f in fr], None, [f[1] for f in fr])
        for s in range(S):
            assert_rows_match(got[s], orcs[s].update(fr[s][0], img, fr[s][1].copy()), t)
        if t == 4:
            before = ms.capacity()
            ms.reserve(max_tracks=before[0] + 100, max_dets=before[1] + 10)        # explicit growth in the middle of a sequence
            after = ms.capacity()
            assert after[0] >= before[0] + 100 and after[1] >= before[1] + 10 and after[2] == before[2] + 1
    assert ms.capacity()[2] >= 2
    ms.close()


@pytest.mark.parametrize("kind", ["deepocsort", "ocsort", "strongsort"])
def test_other_trackers_grow_and_match_their_oracles(kind):
    from boxmot_amd import DeepOcSort, OcSort, StrongSort
    from boxmot_amd.scenario import stress_frames
    from oracle.deepocsort import DeepOcSortOracle, OcSortOracle
    from oracle.strongsort import StrongSortOracle
    img = np.zeros((480, 640, 3), dtype=np.uint8)
    dim = 16
    if kind == "deepocsort":
        trk, orc = DeepOcSort(cmc_off=True, emb_dim=dim, max_tracks=8, max_dets=4), DeepOcSortOracle()
    elif kind == "ocsort":
        trk, orc = OcSort(max_tracks=8, max_dets=4), OcSortOracle()
    else:
        trk, orc = StrongSort(emb_dim=dim, max_tracks=8, max_dets=4), StrongSortOracle()
    for t, (d, e) in enumerate(stress_frames(70, seed=13, emb_dim=dim)):
        got = trk.update(d, img, e) if kind != "ocsort" else trk.update(d, img)
        want = orc.update(d.copy(), img, e.copy()) if kind != "ocsort" else orc.update(d.copy(), img)
        assert_rows_match(np.asarray(got).reshape(-1, 8), np.asarray(want).reshape(-1, 8), t)
    assert trk.capacity()[2] >= 1
    trk.close()


def test_reserve_with_blob_installed_reid_engine_then_device_resident_frames():
    """Weights installed through boxmot_hip_botsort_set_reid_blob (how MultiStreamBotSort(reid_weights=...) and bench.py load
    them) have no file to re-read: growing max_dets rebuilds the ReID engine from the handle's own copy of the blob.  Reserve in
    the middle of a sequence, keep stepping with the frame resident on the device and ReID inside the step: rows equal a handle
    created large, frame for frame."""
    import torch

    from boxmot_amd.reid_weights import random_osnet_state_dict
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    sd = random_osnet_state_dict("osnet_x0_25", seed=0)
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    sc = Scenario(12, 24, width=960, height=540, emb_dim=512, random_image=True)
    dev = torch.device("cuda:0")
    frame = torch.from_numpy(sc.image).to(dev)
    ptrs = torch.tensor([frame.data_ptr()], dtype=torch.int64, device=dev)
    NDBUF = 64
    small = MultiStreamBotSort(1, max_tracks=64, max_dets=24, emb_dim=512, reid_weights=sd, **kw)
    big = MultiStreamBotSort(1, max_tracks=128, max_dets=NDBUF, emb_dim=512, reid_weights=sd, **kw)
    frames = [sc.frame(t, with_embs=False)[0] for t in range(12)]

    def step(ms, nd, dets):
        d = torch.zeros((1, nd, 6), dtype=torch.float32, device=dev)
        d[0, : len(dets)] = torch.from_numpy(dets).to(dev)
        n = torch.tensor([len(dets)], dtype=torch.int32, device=dev)
        out = torch.zeros((1, nd, 8), dtype=torch.float32, device=dev)
        out_n = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ms.step_device(d.data_ptr(), n.data_ptr(), None, ptrs.data_ptr(), sc.height, sc.width, out.data_ptr(), out_n.data_ptr())
        ms.synchronize()
        return out[0, : int(out_n[0])].cpu().numpy()

    for t, dets in enumerate(frames):
        if t == 5:
            before = small.capacity()
            small.reserve(max_dets=NDBUF)              # blob-installed engine: used to throw "cannot open ReID weight blob: "
            after = small.capacity()
            assert after[1] >= NDBUF and after[2] == before[2] + 1
        nd_small = small.capacity()[1]
        got, want = step(small, nd_small, dets), step(big, big.capacity()[1], dets)
        assert_rows_match(got, want, t, box_atol=1e-4)
    assert small.status().tolist() == [0] and big.status().tolist() == [0]
    small.close()
    big.close()
