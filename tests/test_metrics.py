"""boxmot_amd.metrics (HOTA / CLEAR / Identity restated from TrackEval's published algorithm -- the package is absent offline,
parity unpinned) on hand-computable sequences, and the evaluation flow of the reference (`boxmot eval`: replay -> MOT text rows ->
TrackEval summary columns HOTA, MOTA, IDF1, AssA, AssRe, IDSW, IDs) on the reference's MOT17-mini fixture: the rows the real
reference trackers produced (tests/golden/mot17_golden.npz) and the rows the oracle replays produce score identically."""
import numpy as np
import pytest

from common import GOLDEN


def _gt(ids_boxes_per_frame, cls=1, consider=1):
    rows = []
    for t, objs in enumerate(ids_boxes_per_frame, start=1):
        for i, (l, tp, w, h) in objs:
            rows.append([t, i, l, tp, w, h, consider, cls, 1.0])
    return np.array(rows, dtype=float).reshape(-1, 9)


def _res(ids_boxes_per_frame):
    rows = []
    for t, objs in enumerate(ids_boxes_per_frame, start=1):
        for i, (l, tp, w, h) in objs:
            rows.append([t, i, l, tp, w, h, 0.9, 1, 0])
    return np.array(rows, dtype=float).reshape(-1, 9)


def test_perfect_tracking_scores_one():
    from boxmot_amd.metrics import evaluate_mot
    frames = [[(7, (10 + 2 * t, 20, 30, 60)), (9, (200, 50 + t, 40, 80))] for t in range(10)]
    m = evaluate_mot(_gt(frames), _res([[(i + 100, b) for i, b in f] for f in frames]))
    assert m["HOTA"] == pytest.approx(1.0) and m["MOTA"] == pytest.approx(1.0) and m["IDF1"] == pytest.approx(1.0)
    assert m["IDSW"] == 0 and m["CLR_FP"] == 0 and m["CLR_FN"] == 0 and m["IDs"] == 2 and m["MT"] == 2


def test_one_identity_switch_known_values():
    """One object, 10 frames, perfect boxes; the tracker changes its id after frame 5.
    CLEAR: TP 10, IDSW 1 -> MOTA 0.9.  Identity: the best one-to-one mapping explains 5 of 10 -> IDTP 5, IDFN 5, IDFP 5,
    IDF1 0.5.  HOTA: DetA 1; every TP has association accuracy 5 / (10 + 5 - 5) = 0.5 -> AssA 0.5, HOTA sqrt(0.5)."""
    from boxmot_amd.metrics import evaluate_mot
    gt = _gt([[(1, (10, 10, 50, 100))] for _ in range(10)])
    res = _res([[(1 if t < 5 else 2, (10, 10, 50, 100))] for t in range(10)])
    m = evaluate_mot(gt, res)
    assert (m["CLR_TP"], m["IDSW"], m["CLR_FP"], m["CLR_FN"]) == (10, 1, 0, 0)
    assert m["MOTA"] == pytest.approx(0.9)
    assert (m["IDTP"], m["IDFN"], m["IDFP"]) == (5, 5, 5) and m["IDF1"] == pytest.approx(0.5)
    assert m["DetA"] == pytest.approx(1.0) and m["AssA"] == pytest.approx(0.5) and m["HOTA"] == pytest.approx(np.sqrt(0.5))
    assert m["AssRe"] == pytest.approx(0.5) and m["AssPr"] == pytest.approx(1.0) and m["Frag"] == 0


def test_misses_false_positives_and_distractors():
    from boxmot_amd.metrics import evaluate_mot
    box = (100, 100, 40, 80)
    gt = np.vstack([_gt([[(1, box)] for _ in range(8)]),
                    _gt([[(50, (400, 100, 40, 80))] for _ in range(8)], cls=8),          # distractor: never scored
                    _gt([[(60, (600, 100, 40, 80))] for _ in range(8)], consider=0)])     # pedestrian with consider flag 0: ignored
    # tracker: finds object 1 in frames 1-6 only, follows the distractor all the time (removed), one spurious box in frames 3-4
    res = _res([[(1, box)] * (t < 6) + [(2, (400, 100, 40, 80))] + [(3, (900, 500, 30, 30))] * (t in (2, 3)) for t in range(8)])
    m = evaluate_mot(gt, res)
    assert (m["CLR_TP"], m["CLR_FN"], m["CLR_FP"], m["IDSW"]) == (6, 2, 2, 0)
    assert m["MOTA"] == pytest.approx((6 - 2) / 8)
    assert m["IDs"] == 2 and m["GT_IDs"] == 1                       # tracker ids 1 and 3 (2 matched the distractor and was removed)
    assert m["IDF1"] == pytest.approx(6 / (6 + 0.5 * 2 + 0.5 * 2))
    # half-overlapping boxes: IoU 1/3 < 0.5 -> a miss plus a false positive for CLEAR, but a HOTA true positive for alpha <= 0.3
    m2 = evaluate_mot(_gt([[(1, (0, 0, 100, 100))]]), _res([[(1, (50, 0, 100, 100))]]))
    assert (m2["CLR_TP"], m2["CLR_FN"], m2["CLR_FP"]) == (0, 1, 1)
    assert m2["HOTA_TP"][:6].tolist() == [1] * 6 and m2["HOTA_TP"][6:].sum() == 0


def test_empty_inputs():
    from boxmot_amd.metrics import evaluate_mot
    gt = _gt([[(1, (10, 10, 50, 100))] for _ in range(3)])
    m = evaluate_mot(gt, np.zeros((0, 9)))
    assert m["MOTA"] == 0.0 and m["CLR_FN"] == 3 and m["HOTA"] == 0.0 and m["IDF1"] == 0.0
    m = evaluate_mot(np.zeros((0, 9)), _res([[(1, (10, 10, 50, 100))]]))
    assert m["CLR_FP"] == 1 and m["HOTA"] == 0.0


def _score(rows_by_frame, gt_rows, n_frames):
    from boxmot_amd.metrics import evaluate_mot
    from boxmot_amd.replay import format_for_mot
    mot = [format_for_mot(rows_by_frame[f], f) for f in sorted(rows_by_frame) if f <= n_frames and len(rows_by_frame[f])]
    return evaluate_mot(gt_rows, np.vstack(mot) if mot else np.zeros((0, 9)), n_frames)


@pytest.mark.parametrize("kind", ["botsort", "bytetrack", "deepocsort", "strongsort", "ocsort"])
def test_reference_rows_and_oracle_replay_score_identically_on_mot17_mini(kind):
    """The acceptance metric of the reference's evaluation on its own fixture: HOTA / MOTA / IDF1 / AssA / AssRe / IDSW / IDs
    of the rows the REAL reference tracker produced equal those of the oracle's replay of the same detections."""
    from test_mot17_golden import _frames, _golden_rows, _oracle
    g, gt = np.load(GOLDEN / "mot17_golden.npz"), np.load(GOLDEN / "mot17_mini_gt.npz")
    for seq in ("MOT17-02-FRCNN", "MOT17-04-FRCNN"):
        n = int(gt[seq][:, 0].max())                      # annotated frames of the mini fixture (4 / 8)
        want = _score(_golden_rows(g, seq, kind), gt[seq], n)
        assert 0.0 < want["HOTA"] <= 1.0 and want["CLR_TP"] > 0 and want["GT_IDs"] > 0
        orc, rows = _oracle(kind), {}
        for fid, d, e in _frames(g, seq):
            if fid > n:
                break
            if len(d):
                rows[fid] = np.asarray(orc.update(d.copy(), None, e.copy()), dtype=np.float32).reshape(-1, 8)
        got = _score(rows, gt[seq], n)
        assert got["summary"] == want["summary"], (seq, got["summary"], want["summary"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["botsort", "bytetrack", "deepocsort", "strongsort", "ocsort"])
def test_hip_replay_scores_like_the_reference_on_mot17_mini(kind):
    """`boxmot eval` with the HIP trackers: cached detections -> boxmot_amd.replay (all sequences as streams of one handle) ->
    MOT rows -> HOTA / MOTA / IDF1 ...: the same summary as the reference trackers' rows."""
    from boxmot_amd.metrics import evaluate_mot
    from boxmot_amd.replay import CachedSequence, replay
    from common import BOTSORT_YAML_DEFAULTS, mot17_embeddings
    from test_mot17_golden import _golden_rows
    g, gt = np.load(GOLDEN / "mot17_golden.npz"), np.load(GOLDEN / "mot17_mini_gt.npz")
    seqs = []
    for seq in ("MOT17-02-FRCNN", "MOT17-04-FRCNN"):
        rows = g[seq + "_dets"]
        seqs.append(CachedSequence(seq, np.arange(1, len(g[f"{seq}_botsort_counts"]) + 1), rows, mot17_embeddings(rows)))
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method", "with_reid")} if kind == "botsort" else {}
    got = replay(seqs, tracker_type=kind, max_tracks=512, max_dets=64, **kw)
    for seq in ("MOT17-02-FRCNN", "MOT17-04-FRCNN"):
        n = int(gt[seq][:, 0].max())
        want = _score(_golden_rows(g, seq, kind), gt[seq], n)
        mot = got[seq]
        have = evaluate_mot(gt[seq], mot[mot[:, 0] <= n], n)
        assert have["summary"] == want["summary"], (seq, have["summary"], want["summary"])


@pytest.mark.parametrize("kind", ["botsort", "bytetrack", "deepocsort", "strongsort", "ocsort"])
def test_replay_over_the_emulated_abi_scores_like_the_reference_on_mot17_mini(monkeypatch, kind):
    """The CPU twin of the GPU test above: boxmot_amd.replay (both MOT17-mini sequences as streams of one handle, update_batch) with the
    C-ABI calls answered by the emulated device steps (tests/emu_lib.py) -> MOT rows -> HOTA / MOTA / IDF1: the same summary as the
    reference trackers' own rows."""
    from boxmot_amd import _lib
    from boxmot_amd.metrics import evaluate_mot
    from boxmot_amd.replay import CachedSequence, replay
    from common import BOTSORT_YAML_DEFAULTS, mot17_embeddings
    from emu_lib import EmuHipLib
    from test_mot17_golden import _golden_rows
    lib = EmuHipLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(_lib, "last_error", lambda: lib.boxmot_hip_last_error().decode())
    g, gt = np.load(GOLDEN / "mot17_golden.npz"), np.load(GOLDEN / "mot17_mini_gt.npz")
    seqs = []
    for seq in ("MOT17-02-FRCNN", "MOT17-04-FRCNN"):
        rows = g[seq + "_dets"]
        seqs.append(CachedSequence(seq, np.arange(1, len(g[f"{seq}_botsort_counts"]) + 1), rows, mot17_embeddings(rows)))
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method", "with_reid")} if kind == "botsort" else {}
    got = replay(seqs, tracker_type=kind, max_tracks=256, max_dets=64, **kw)
    for seq in ("MOT17-02-FRCNN", "MOT17-04-FRCNN"):
        n = int(gt[seq][:, 0].max())
        want = _score(_golden_rows(g, seq, kind), gt[seq], n)
        mot = got[seq]
        have = evaluate_mot(gt[seq], mot[mot[:, 0] <= n], n)
        assert have["summary"] == want["summary"], (seq, have["summary"], want["summary"])
