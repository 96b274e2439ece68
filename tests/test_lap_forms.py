"""lapx's `extend_cost=True` WITHOUT a cost limit -- the DeepOCSORT / OC-SORT call site (association.py:20-24; lapx 0.9.4 is pinned by
the reference's uv.lock:2522 but absent here) -- can be squared in two ways: the (n_rows + n_cols)^2 matrix filled with max + 1 that
oracle/lap.py and oracle/lapjv.c restate (and the device solver follows tie for tie), or the zero-padded max(n_rows, n_cols)^2 matrix
(SURVEY.md section 8(c)'s "alternative reading"; round-4 review, Weak 3).  Both have the same optimum; they may choose differently among
exactly tied optima.  lapx cannot be consulted offline, so the choice is turned into a TESTED INVARIANCE: every reference-generated
golden of that call site -- the DeepOCSORT stress goldens, configuration 3's 128 x 512 golden, the association-function goldens, the
MOT17-mini OC-SORT / DeepOCSORT rows -- comes out identical under both forms and equal to the reference's rows under both.  The
synthetic tie-prone seeds of tests/test_gpu_deepocsort.py are run under both too and the outcome is recorded: five of six runs are
identical, one (seed 22, second option set) parts at frame 1 -- constructed exact ties are where the form is observable."""
from pathlib import Path

import numpy as np
import pytest

from common import ASSO_FUNCS, DEEPOCSORT_CASES, GOLDEN, asso_golden_rows, deepocsort_golden_rows, mot17_embeddings

FORMS = ("sum_max_plus_one", "zero_pad")


@pytest.fixture()
def lap_form():
    from oracle import lap

    def use(form):
        assert form in FORMS
        lap.NO_LIMIT_FORM = form
    yield use
    lap.NO_LIMIT_FORM = "sum_max_plus_one"


def _run(make_oracle, frames, lap_form, form):
    lap_form(form)
    orc = make_oracle()
    return [np.asarray(orc.update(d.copy(), None, None if e is None else e.copy()), dtype=np.float32).reshape(-1, 8) for d, e in frames]


def _same(a, b, what):
    assert len(a) == len(b)
    for t, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape and np.array_equal(x, y), (what, t)


def test_both_forms_return_an_optimal_assignment(lap_form):
    """Sanity of the switch: both forms return an optimal assignment on random rectangular matrices (cost equal to SciPy's exact
    solver), every row / column pairing is mutual, min(n_rows, n_cols) pairs are made."""
    from scipy.optimize import linear_sum_assignment

    from oracle import lap
    rng = np.random.default_rng(0)
    for nr, nc in ((5, 9), (9, 5), (7, 7), (1, 12), (12, 1), (30, 17)):
        c = rng.uniform(0, 1, (nr, nc))
        r, k = linear_sum_assignment(c)
        best = c[r, k].sum()
        for form in FORMS:
            lap_form(form)
            cost, x, y = lap.lapjv(c, extend_cost=True)
            assert np.isclose(cost, best, rtol=0, atol=1e-12), (form, nr, nc)
            assert (x >= 0).sum() == min(nr, nc)
            for i, j in enumerate(x):
                assert j < 0 or y[j] == i


@pytest.mark.parametrize("name", list(DEEPOCSORT_CASES))
def test_deepocsort_goldens_are_invariant_under_the_extension_form(name, lap_form):
    from oracle.deepocsort import DeepOcSortOracle
    make, _, kw, _ = DEEPOCSORT_CASES[name]
    frames = make()
    want, _ = deepocsort_golden_rows(name)
    if name == "docs_c2":
        frames = frames[:8]
    runs = {f: _run(lambda: DeepOcSortOracle(**kw), frames, lap_form, f) for f in FORMS}
    _same(runs[FORMS[0]], runs[FORMS[1]], name)
    _same(runs[FORMS[1]], want[:len(frames)], name + " vs reference rows")


def test_config3_golden_is_invariant_under_the_extension_form(lap_form):
    from boxmot_amd.scenario import Scenario
    from oracle.deepocsort import DeepOcSortOracle
    g = np.load(GOLDEN / "config3_deepocsort_golden.npz")
    offs = np.concatenate([[0], np.cumsum(g["counts"])])
    runs = {}
    for f in FORMS:
        sc = Scenario(128, 512, emb_dim=512, random_image=False)
        runs[f] = _run(DeepOcSortOracle, [sc.frame(t) for t in range(40)], lap_form, f)
    _same(runs[FORMS[0]], runs[FORMS[1]], "config 3")
    for t, got in enumerate(runs[FORMS[1]]):
        lo, hi = offs[t], offs[t + 1]
        assert len(got) == hi - lo and np.array_equal(got[:, 4].astype(np.int32), g["ids"][lo:hi]) \
            and np.array_equal(got[:, 7].astype(np.int32), g["det_ind"][lo:hi]), t


@pytest.mark.parametrize("name", list(ASSO_FUNCS))
def test_association_function_goldens_are_invariant_under_the_extension_form(name, lap_form):
    from oracle.deepocsort import DeepOcSortOracle, OcSortOracle
    for tracker in ("deepocsort", "ocsort"):
        want, frames = asso_golden_rows(tracker, name)
        make = (lambda: DeepOcSortOracle(asso_func=name, iou_threshold=ASSO_FUNCS[name])) if tracker == "deepocsort" \
            else (lambda: OcSortOracle(asso_func=name, iou_threshold=ASSO_FUNCS[name], use_byte=True))
        fr = list(frames())
        img = np.zeros((480, 640, 3), np.uint8)          # the centroid cost normalises by the frame diagonal
        runs = {}
        for f in FORMS:
            lap_form(f)
            orc = make()
            runs[f] = [np.asarray(orc.update(d.copy(), img, e.copy()), dtype=np.float32).reshape(-1, 8) for d, e in fr]
        _same(runs[FORMS[0]], runs[FORMS[1]], (tracker, name))
        _same(runs[FORMS[1]], want, (tracker, name, "vs reference rows"))


@pytest.mark.parametrize("kind", ["deepocsort", "ocsort", "ocsort_yaml", "ocsort_byte"])
@pytest.mark.parametrize("seq", ["MOT17-02-FRCNN", "MOT17-04-FRCNN"])
def test_mot17_rows_are_invariant_under_the_extension_form(seq, kind, lap_form):
    from oracle.deepocsort import DeepOcSortOracle, OcSortOracle
    g = np.load(Path(GOLDEN) / "mot17_golden.npz")
    rows = g[seq + "_dets"]
    emb = mot17_embeddings(rows)
    n_frames = len(g[f"{seq}_botsort_counts"])
    frames = [(rows[rows[:, 0] == fid, 1:], emb[rows[:, 0] == fid]) for fid in range(1, n_frames + 1)]
    frames = [(d, e) for d, e in frames if len(d)]
    make = {"deepocsort": DeepOcSortOracle, "ocsort": OcSortOracle, "ocsort_yaml": lambda: OcSortOracle(det_thresh=0.6, inertia=0.1),
            "ocsort_byte": lambda: OcSortOracle(use_byte=True)}[kind]
    runs = {f: _run(make, frames, lap_form, f) for f in FORMS}
    _same(runs[FORMS[0]], runs[FORMS[1]], (seq, kind))
    wr, wc = g[f"{seq}_{kind}_rows"], g[f"{seq}_{kind}_counts"]
    want, at = [], 0
    for c in wc:
        if c >= 0:
            want.append(wr[at:at + c])
            at += c
    _same(runs[FORMS[1]], want, (seq, kind, "vs reference rows"))


TIE_KW = ({}, dict(max_age=8, min_hits=2, iou_threshold=0.2))


@pytest.mark.parametrize("seed,kwi,forms_part", [(22, 0, False), (22, 1, True), (24, 0, False), (24, 1, False), (28, 0, False), (28, 1, False)])
def test_tie_prone_scenes_under_both_extension_forms(seed, kwi, forms_part, lap_form):
    """The scenes tests/test_gpu_deepocsort.py::test_hip_deepocsort_tie_prone_scenes runs (synthetic, NOT reference-generated: more
    detections than tracks, several exactly tied optimal assignments) -- the one place where the two forms can part.  Recorded outcome:
    five of the six runs are identical under both forms; seed 22 with (max_age=8, min_hits=2, iou_threshold=0.2) is not -- from
    frame 1 on the zero-padded form picks another of the tied optima (same assignment cost), ids permute and the runs diverge.
    So the tie rule IS observable on constructed ties, and is not on any reference-generated row (the tests above).  The default
    stays the (n_rows + n_cols)^2 / max + 1 form: it is what the gatagat/lap wrapper lapx 0.9.4 descends from does
    (`cost_c_extended[:] = cost_c.max() + 1` for `extend_cost` without a limit), as far as that can be recalled offline."""
    from boxmot_amd.scenario import stress_frames
    from oracle.deepocsort import DeepOcSortOracle
    kw = TIE_KW[kwi]
    frames = list(stress_frames(120, seed=seed, max_objects=30))
    runs = {f: _run(lambda: DeepOcSortOracle(**kw), frames, lap_form, f) for f in FORMS}
    diff = [t for t, (a, b) in enumerate(zip(runs[FORMS[0]], runs[FORMS[1]])) if a.shape != b.shape or not np.array_equal(a, b)]
    assert bool(diff) == forms_part, f"seed {seed} {kw}: frames that differ between the two extension forms: {diff[:10]}"
    if forms_part:
        assert diff[0] == 1
        # same number of output rows per frame until the runs diverge in membership: the forms differ in WHICH tied optimum, not in cost
        assert len(runs[FORMS[0]][1]) == len(runs[FORMS[1]][1])
