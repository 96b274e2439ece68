"""Association COST VALUES on the device (north_star: "cost values within 1e-3"), through the C ABI.

``boxmot_hip_*_debug_costs`` (include/boxmot_hip.h) reads the matrices the frame step handed to its assignment solver; they are
compared element by element with the reference's own matrices (tests/golden/cost_golden.npz, recorded inside the reference
classes by tests/golden/make_cost_golden.py) and with the oracle's on the same inputs.  Tolerance: north_star's 1e-3 is the
contract; the fp64 device arithmetic is held to 1e-7 here (observed ~1e-9: the filter state's summation order).
"""
import numpy as np
import pytest

from test_cost_values import COST_CASES, COST_TOL, TIGHT, check_stage, golden_costs, kept_frames

pytestmark = pytest.mark.gpu


@pytest.mark.fast
@pytest.mark.parametrize("name", list(COST_CASES))
def test_botsort_cost_matrices_match_the_reference(name):
    """BoT-SORT at configuration 2's shape (64 x 256, D = 512; YAML and constructor defaults), the scene that forces the dense
    LDS-tiled cosine fallback (every pair ungated: 16 384 > SPARSE_MAX), and the stress scene with score fusion."""
    from boxmot_amd.botsort import BotSort
    from oracle.botsort import DEFAULTS, BotSortOracle

    g = golden_costs()
    frames, kw, dim = COST_CASES[name]
    keep = kept_frames(g, name)
    prox = dict(DEFAULTS, **kw)["proximity_thresh"]
    trk = BotSort(use_cmc=False, emb_dim=dim, max_tracks=512, max_dets=256, **kw)
    trk.debug_costs_enable()
    orc = BotSortOracle(**kw)
    img = np.zeros((64, 64, 3), dtype=np.uint8)
    worst, n_checked = 0.0, 0
    try:
        for t, (d, e) in enumerate(frames()):
            trk.update(d, img, e)
            orc.update(d, None, e.copy())
            for s in range(3):
                dists, iou = trk.debug_costs(s, 0), trk.debug_costs(s, 1)
                emb = trk.debug_costs(s, 2) if s != 1 else None
                st = orc.last["stages"][s]
                assert dists.shape == st["dists"].shape, (t, s)
                if dists.size:          # every frame against the oracle (itself bit-equal to the reference: tests/test_cost_values.py)
                    assert np.abs(dists - st["dists"]).max() <= TIGHT, (t, s)
                    assert np.abs(iou - st["iou"]).max() <= TIGHT, (t, s)
                    assert np.array_equal(dists == 1.0, st["dists"] == 1.0), (t, s)
                if t in keep:           # the recorded frames against the reference's own matrices
                    worst = max(worst, check_stage(g, name, t, s, dists, iou, emb, prox, "device", tol=TIGHT))
                    n_checked += dists.size
                    if name == "c2_dense" and s == 0 and t > 0:
                        # (256 tracks x 256 detections while the scene introduces itself, x 64 from frame 3 on: > SPARSE_MAX ungated pairs)
                        assert emb.shape[0] == 256 and emb.size >= 16384 and np.isfinite(emb).all(), "the dense fallback evaluates every pair"
    finally:
        trk.close()
    assert n_checked > 0 and worst <= TIGHT < COST_TOL
    print(f"{name}: {n_checked} cost values, max |device - reference| = {worst:.3e}")


def test_multistream_cost_matrices_at_the_operating_point_shape():
    """The bench's tracker object (one workgroup per stream): streams 0 and 2 of a 3-stream handle run the golden scene."""
    from boxmot_amd.streams import MultiStreamBotSort

    name = "c2_yaml"
    g = golden_costs()
    frames, kw, dim = COST_CASES[name]
    keep = kept_frames(g, name)
    ms = MultiStreamBotSort(3, max_tracks=512, max_dets=256, emb_dim=dim, use_cmc=False, **kw)
    ms.debug_costs_enable()
    other = list(COST_CASES["c2_default"][0]())
    worst = 0.0
    try:
        for t, (d, e) in enumerate(frames()):
            d1, e1 = other[(t + 3) % len(other)]
            ms.update_batch([d, d1, d], embs_list=[e, e1, e])
            if t in keep:
                for stream in (0, 2):
                    for s in range(3):
                        emb = ms.debug_costs(s, 2, stream=stream) if s != 1 else None
                        worst = max(worst, check_stage(g, name, t, s, ms.debug_costs(s, 0, stream=stream), ms.debug_costs(s, 1, stream=stream),
                                                       emb, 0.5, f"stream {stream}", tol=TIGHT))
    finally:
        ms.close()
    assert worst <= TIGHT


@pytest.mark.parametrize("name", [pytest.param("docs_c3_crowd", marks=pytest.mark.fast), "docs_c3", "docs_stress", "docs_stress_awoff"])
def test_deepocsort_associate_matrices_match_the_reference(name):
    """DeepOCSORT's `associate` (association.py:61-152) at configuration 3's shape (128 x 512, D = 512; the crowded scene makes the
    solver run on the full matrix) and on the stress scenes: iou_matrix and final_cost against the reference's recorded matrices
    on the kept frames, against the oracle (bit-equal to the reference, tests/test_cost_values.py) on every frame."""
    from boxmot_amd.deepocsort import DeepOcSort
    from oracle.deepocsort import DeepOcSortOracle
    from test_cost_values import DOCS_COST_CASES, DOCS_TOL, check_docs_frame

    g = golden_costs()
    frames, kw, dim, step = DOCS_COST_CASES[name]
    keep = kept_frames(g, name)
    big = name.startswith("docs_c3")
    trk = DeepOcSort(cmc_off=True, emb_dim=dim, max_tracks=1024 if big else 128, max_dets=512 if big else 64, **kw)
    trk.debug_costs_enable()
    orc = DeepOcSortOracle(**kw)
    img = np.zeros((64, 64, 3), dtype=np.uint8)
    worst, solved = 0.0, 0
    try:
        for t, (d, e) in enumerate(frames()):
            trk.update(d, img, e)
            orc.update(d.copy(), None, e.copy())
            final, branch = trk.debug_costs(0)
            iou, _ = trk.debug_costs(1)
            if orc.last.get("iou") is not None:
                assert iou.shape == orc.last["iou"].shape, t
                if iou.size:
                    assert np.abs(iou - orc.last["iou"]).max() <= DOCS_TOL, t
                assert (branch == 2) == (orc.last["final_cost"] is not None), t
                if branch == 2:
                    assert np.abs(final - orc.last["final_cost"]).max() <= DOCS_TOL, t
            if t in keep:
                err, s = check_docs_frame(g, name, t, step, final if branch == 2 else None, iou, "device", DOCS_TOL)
                worst, solved = max(worst, err), solved + s
    finally:
        trk.close()
    assert worst <= DOCS_TOL < COST_TOL and (solved > 0 or name == "docs_c3")
    print(f"{name}: max |device - reference| = {worst:.3e} ({solved} solver frames)")


@pytest.mark.parametrize("name", [pytest.param("ss_c5", marks=pytest.mark.fast), "ss_c2", "ss_stress", "ss_stress_loose"])
def test_strongsort_gated_cost_matrices_match_the_reference(name):
    """StrongSORT's gated appearance cost (tracker.py:108-122, linear_assignment.py:145-198) and IoU cost (iou_matching.py:49-87),
    before and after the max_distance clamp, at configuration 5's shape (1024 tracks x 256 detections x 1280-d, bank product on
    the fp32 matrix pipe) and on smaller scenes.  The appearance distances are fp32: north_star's 1e-3 is the tolerance."""
    from boxmot_amd.strongsort import StrongSort
    from oracle.strongsort import StrongSortOracle
    from test_cost_values import SS_COST_CASES, check_ss_frame

    g = golden_costs()
    frames, kw, dim, step = SS_COST_CASES[name]
    keep = kept_frames(g, name)
    cap, nd = {"ss_c5": (2048, 1024), "ss_c2": (512, 256)}.get(name, (128, 64))
    trk = StrongSort(emb_dim=dim, max_tracks=cap, max_dets=nd, **kw)
    trk.debug_costs_enable()
    orc = None if name == "ss_c5" else StrongSortOracle(**kw)        # (the oracle needs seconds per frame at configuration 5's size)
    img = np.zeros((64, 64, 3), dtype=np.uint8)
    worst = 0.0
    try:
        for t, (d, e) in enumerate(frames()):
            trk.update(d, img, e)
            stages = [(trk.debug_costs(s, 0), trk.debug_costs(s, 1)) for s in range(2)]
            if orc is not None:
                orc.update(d.copy(), None, e.copy())
                for s, st in enumerate(orc.last_costs):
                    if st is None:
                        assert stages[s][0].size == 0, (t, s)
                        continue
                    assert stages[s][0].shape == st["raw"].shape, (t, s)
                    assert np.abs(stages[s][0] - st["raw"]).max() <= COST_TOL, (t, s)
                    assert np.abs(stages[s][1] - st["clamped"]).max() <= COST_TOL, (t, s)
            if t in keep:
                worst = max(worst, check_ss_frame(g, name, t, step, stages, "device", COST_TOL))
    finally:
        trk.close()
    assert worst <= COST_TOL
    print(f"{name}: max |device - reference| = {worst:.3e}")
