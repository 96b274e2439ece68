"""Association COST VALUES (north_star: "cost values within 1e-3"), CPU half.

tests/golden/cost_golden.npz holds the reference's own matrices (tests/golden/make_cost_golden.py: the return values of
boxmot/trackers/association/matching.py's ``iou_distance`` / ``embedding_distance`` and the matrix botsort.py:306-317, 396-413 hands
to ``linear_assignment``, recorded inside the reference ``BotSort``).  Here:
  * the oracle's matrices (oracle/botsort.py ``last["stages"]``) equal the reference's bit for bit -- this pins oracle/matching.py;
  * the DEVICE source of the BoT-SORT step, run on CPU threads (tests/host_emu), reproduces them through the debug planes that
    ``boxmot_hip_botsort_debug_costs`` reads (include/boxmot_hip.h), sparse path and dense LDS-tiled fallback.
The GPU half is tests/test_gpu_cost_values.py (same fixture, through the C ABI).
"""
import numpy as np
import pytest

from boxmot_amd.scenario import Scenario, stress_frames
from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
from common import GOLDEN
from emu_util import EmuBotSort
from oracle.botsort import DEFAULTS, BotSortOracle

YAML = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method", "with_reid")}

# keep in step with tests/golden/make_cost_golden.py
COST_CASES = {
    "c2_yaml": (lambda: Scenario(64, 256, random_image=False).frames(24), YAML, 512),
    "c2_default": (lambda: Scenario(64, 256, random_image=False).frames(24), {}, 512),
    "c2_dense": (lambda: Scenario(64, 256, random_image=False).frames(12), dict(YAML, proximity_thresh=1.0), 512),
    "stress_fuse": (lambda: stress_frames(60, seed=7), dict(fuse_first_associate=True, track_buffer=5), 32),
}
COST_TOL = 1e-3         # north_star's tolerance; the fp64 device arithmetic is expected (and asserted) to be < 1e-7
TIGHT = 1e-7          # (the filter state behind the boxes agrees to ~1e-9 relative: summation order of the fp64 products)


def golden_costs():
    return np.load(GOLDEN / "cost_golden.npz")


def kept_frames(g, name):
    return sorted({int(k[len(name) + 2:].split("_")[0]) for k in g.files if k.startswith(name + "_f")})


def check_stage(g, name, t, stage, dists, iou, emb, proximity, where, tol=TIGHT):
    """One association of one frame against the reference's matrices.  ``emb`` may hold NaN where the step did not evaluate the
    cosine (pairs behind the IoU gate on the sparse path); every pair that passes the gate must have been evaluated."""
    key = f"{name}_f{t}_s{stage}_"
    want_d, want_i = g[key + "dists"], g[key + "iou"]
    assert dists.shape == want_d.shape, (where, name, t, stage, dists.shape, want_d.shape)
    if want_d.size == 0:
        return 0.0
    err = float(np.abs(dists - want_d).max())
    assert err <= tol, f"{where} {name} frame {t} stage {stage}: solver matrix differs by {err}"
    assert np.array_equal(dists == 1.0, want_d == 1.0), f"{where} {name} frame {t} stage {stage}: gated entries are not exactly 1.0"
    err_i = float(np.abs(iou - want_i).max())
    assert err_i <= tol, f"{where} {name} frame {t} stage {stage}: iou_distance differs by {err_i}"
    err = max(err, err_i)
    if key + "emb" in g.files and emb is not None:
        want_e = g[key + "emb"]
        have = np.isfinite(emb)
        ungated = ~(want_i > proximity)
        assert have[ungated].all(), f"{where} {name} frame {t} stage {stage}: an ungated pair's cosine was not evaluated"
        err_e = float(np.abs(emb[have] - want_e[have]).max()) if have.any() else 0.0
        assert err_e <= tol, f"{where} {name} frame {t} stage {stage}: embedding_distance differs by {err_e}"
        err = max(err, err_e)
    return err


@pytest.mark.parametrize("name", list(COST_CASES))
def test_oracle_cost_matrices_equal_the_references(name):
    g = golden_costs()
    frames, kw, _ = COST_CASES[name]
    keep = kept_frames(g, name)
    orc = BotSortOracle(**kw)
    n = 0
    for t, (d, e) in enumerate(frames()):
        orc.update(d, None, e.copy())
        if t not in keep:
            continue
        for s, st in enumerate(orc.last["stages"]):
            key = f"{name}_f{t}_s{s}_"
            assert np.array_equal(st["dists"], g[key + "dists"]), (name, t, s)
            assert np.array_equal(st["iou"], g[key + "iou"]), (name, t, s)
            if key + "emb" in g.files:
                assert np.array_equal(st["emb"], g[key + "emb"]), (name, t, s)
            n += st["dists"].size
    assert n > 0


@pytest.mark.parametrize("name,dense", [("stress_fuse", False), ("stress_fuse", True), ("c2_yaml", False), ("c2_dense", False)])
def test_emulated_step_cost_planes_match_the_reference(name, dense):
    """``dense``: the step built with the sparse-pair limit at 0 (every frame takes the dense contraction); "c2_dense" takes it with
    the shipped limit because every one of its 64 x 256 pairs passes the IoU gate."""
    g = golden_costs()
    frames, kw, dim = COST_CASES[name]
    keep = kept_frames(g, name)
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    cap, nd = (512, 256) if name.startswith("c2") else (128, 64)      # (the scene's first frame brings all 256 objects)
    frames = list(frames())
    emu = EmuBotSort(cfg, cap=cap, nd=nd, dim=dim, dense=dense)
    emu.debug_costs_enable()
    worst = 0.0
    try:
        for t, (d, e) in enumerate(frames):
            emu.update(d, e)
            if t not in keep:
                continue
            for s in range(3):
                emb = emu.debug_costs(s, 2) if s != 1 else None
                worst = max(worst, check_stage(g, name, t, s, emu.debug_costs(s, 0), emu.debug_costs(s, 1), emb,
                                               cfg["proximity_thresh"], "emulated step"))
                if emb is not None and emb.size and (dense or name == "c2_dense") and np.isfinite(emb).any():
                    assert np.isfinite(emb).all(), "the dense path evaluates every pair (it is entered when any pair passes the IoU gate)"
    finally:
        emu.close()
    assert worst <= TIGHT


# ---------------------------------------------------------------------------------------------------------------------------
# DeepOCSORT `associate` (association.py:61-152) and StrongSORT's two min_cost_matching matrices (linear_assignment.py:14-79)
# keep in step with tests/golden/make_cost_golden.py (name: frames, kwargs, emb dim, row step of the fixture)
DOCS_COST_CASES = {
    "docs_c3": (lambda: Scenario(128, 512, emb_dim=512, random_image=False).frames(6), {}, 512, 8),
    "docs_c3_crowd": (lambda: Scenario(128, 512, emb_dim=512, random_image=False, crowd=True).frames(12), {}, 512, 4),
    "docs_stress": (lambda: stress_frames(60, seed=7), {}, 32, 1),
    "docs_stress_awoff": (lambda: stress_frames(60, seed=3), dict(aw_off=True, inertia=0.4, w_association_emb=0.75), 32, 1),
}
# emb_cost is `dets_embs @ trk_embs.T` (deepocsort.py:387-390): a float32 BLAS product in the reference (both operands are fp32 there),
# accumulated in fp64 on the device -- the two differ by fp32 rounding of a 512-term sum (observed 2e-7), far inside north_star's 1e-3
DOCS_TOL = 1e-5
SS_COST_CASES = {
    "ss_c5": (lambda: Scenario(256, 1024, emb_dim=1280, random_image=False).frames(6), {}, 1280, 16),
    "ss_c2": (lambda: Scenario(64, 256, emb_dim=128, random_image=False).frames(12), {}, 128, 4),
    "ss_stress": (lambda: stress_frames(60, seed=7), {}, 32, 1),
    "ss_stress_loose": (lambda: stress_frames(60, seed=3),
                        dict(max_cos_dist=0.4, max_iou_dist=0.9, mc_lambda=0.9, ema_alpha=0.8, min_conf=0.3), 32, 1),
}


def check_docs_frame(g, name, t, step, final_cost, iou, where, tol):
    """One frame's `associate` matrices against the reference's (every ``step``-th detection row is in the fixture).
    ``final_cost`` is None when the step did not ask the solver; the reference must not have either."""
    key = f"{name}_f{t}_"
    want_i = g[key + "iou"] if key + "iou" in g.files else None
    err = 0.0
    if want_i is None:
        assert iou is None or iou.size == 0, (where, name, t)
        return 0.0, False
    assert iou[::step].shape == want_i.shape, (where, name, t, iou.shape, want_i.shape)
    err = float(np.abs(iou[::step] - want_i).max()) if want_i.size else 0.0
    assert err <= tol, f"{where} {name} frame {t}: iou_matrix differs by {err}"
    solved = key + "final_cost" in g.files
    assert solved == (final_cost is not None), f"{where} {name} frame {t}: solver branch differs (reference {'ran' if solved else 'skipped'} it)"
    if solved:
        e2 = float(np.abs(final_cost[::step] - g[key + "final_cost"]).max())
        assert e2 <= tol, f"{where} {name} frame {t}: final_cost differs by {e2}"
        err = max(err, e2)
    return err, solved


def check_ss_frame(g, name, t, step, stages, where, tol):
    """``stages``: [(raw, clamped) or None] for stage A and B.  The fixture holds every ``step``-th track row."""
    err = 0.0
    for s, st in enumerate(stages):
        key = f"{name}_f{t}_s{s}_"
        shape = tuple(g[key + "shape"])
        if key + "raw" not in g.files:
            assert st is None or st[0].size == 0, (where, name, t, s)
            continue
        raw, clamped = st
        assert raw.shape == shape, (where, name, t, s, raw.shape, shape)
        for got, k in ((raw, "raw"), (clamped, "clamped")):
            e = float(np.abs(got[::step] - g[key + k]).max())
            assert e <= tol, f"{where} {name} frame {t} stage {s}: {k} cost differs by {e}"
            err = max(err, e)
    return err


@pytest.mark.parametrize("name", list(DOCS_COST_CASES))
def test_deepocsort_oracle_associate_matrices_equal_the_references(name):
    from oracle.deepocsort import DeepOcSortOracle

    g = golden_costs()
    frames, kw, _, step = DOCS_COST_CASES[name]
    keep = kept_frames(g, name)
    orc = DeepOcSortOracle(**kw)
    solved = 0
    for t, (d, e) in enumerate(frames()):
        orc.update(d.copy(), None, e.copy())
        if t in keep:
            key = f"{name}_f{t}_"
            assert np.array_equal(orc.last["iou"][::step], g[key + "iou"]), (name, t)
            assert (orc.last["final_cost"] is not None) == (key + "final_cost" in g.files), (name, t)
            if orc.last["final_cost"] is not None:
                assert np.array_equal(orc.last["final_cost"][::step], g[key + "final_cost"]), (name, t)
                solved += 1
    assert solved > 0 or name == "docs_c3"


@pytest.mark.parametrize("name", list(SS_COST_CASES))
def test_strongsort_oracle_cost_matrices_equal_the_references(name):
    from oracle.strongsort import StrongSortOracle

    g = golden_costs()
    frames, kw, _, step = SS_COST_CASES[name]
    keep = kept_frames(g, name)
    orc = StrongSortOracle(**kw)
    n = 0
    for t, (d, e) in enumerate(frames()):
        orc.update(d.copy(), None, e.copy())
        if t in keep:
            for s, st in enumerate(orc.last_costs):
                key = f"{name}_f{t}_s{s}_"
                assert (st is not None) == (key + "raw" in g.files), (name, t, s)
                if st is not None:
                    assert np.array_equal(st["raw"][::step], g[key + "raw"]), (name, t, s)
                    assert np.array_equal(st["clamped"][::step], g[key + "clamped"]), (name, t, s)
                    n += st["raw"].size
    assert n > 0


@pytest.mark.parametrize("name", ["docs_stress", "docs_stress_awoff", "docs_c3_crowd"])
def test_emulated_deepocsort_step_cost_planes_match_the_reference(name):
    from emu_util import EmuDeepOcSort
    from oracle.deepocsort import DEFAULTS as DOCS_DEFAULTS

    g = golden_costs()
    frames, kw, dim, step = DOCS_COST_CASES[name]
    keep = kept_frames(g, name)
    big = name.startswith("docs_c3")
    frames = list(frames())
    if big:
        frames, keep = frames[:keep[1] + 1], keep[:2]          # (a 128 x 512 x 512-d step takes seconds on CPU threads)
    cfg = dict(DOCS_DEFAULTS)
    cfg.update(kw)
    emu = EmuDeepOcSort(cfg, cap=1024 if big else 128, nd=512 if big else 64, dim=dim)
    emu.debug_costs_enable()
    worst, solved = 0.0, 0
    try:
        for t, (d, e) in enumerate(frames):
            emu.update(d, e)
            if t not in keep:
                continue
            final, branch = emu.debug_costs(0)
            iou, _ = emu.debug_costs(1)
            err, s = check_docs_frame(g, name, t, step, final if branch == 2 else None, iou, "emulated step", DOCS_TOL)
            worst, solved = max(worst, err), solved + s
    finally:
        emu.close()
    print(f"{name}: max |emulated - reference| = {worst:.3e}")
    assert worst <= DOCS_TOL and solved > 0


@pytest.mark.parametrize("name", ["ss_stress", "ss_stress_loose", "ss_c2"])
def test_emulated_strongsort_step_cost_planes_match_the_reference(name):
    from emu_util import EmuStrongSort
    from oracle.strongsort import DEFAULTS as SS_DEFAULTS

    g = golden_costs()
    frames, kw, dim, step = SS_COST_CASES[name]
    keep = kept_frames(g, name)
    cfg = dict(SS_DEFAULTS)
    cfg.update(kw)
    big = name == "ss_c2"
    emu = EmuStrongSort(cfg, cap=512 if big else 128, nd=256 if big else 64, dim=dim)
    emu.debug_costs_enable()
    worst = 0.0
    try:
        for t, (d, e) in enumerate(frames()):
            emu.update(d, e)
            if t in keep:
                stages = [(emu.debug_costs(s, 0), emu.debug_costs(s, 1)) for s in range(2)]
                # the appearance distances are fp32 products: north_star's tolerance applies (observed ~1e-7)
                worst = max(worst, check_ss_frame(g, name, t, step, stages, "emulated step", COST_TOL))
    finally:
        emu.close()
    print(f"{name}: max |emulated - reference| = {worst:.3e}")
    assert worst <= COST_TOL
