"""bench.py's CPU legs on their own (no GPU): the oracle rows of a stream, the reference classes' rows of the same stream (from
/root/reference here; from the byte-compiled oracle/_ref/ on the GPU box), and the `cpu_baseline` record built from them --
kind = "reference", rows equal to the oracle's, the port's figure attached -- in the embeddings-supplied mode, where a frame costs
milliseconds."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def _runnable():
    from oracle import ref_harness
    return ref_harness.reference_runnable()


@pytest.mark.skipif(not _runnable(), reason="neither /root/reference nor oracle/_ref/ is present")
def test_cpu_baseline_times_the_reference_classes_and_their_rows_equal_the_oracles():
    import bench
    rows_o, t_o, n_o = bench.oracle_rows(None, "embs", 6, 3)
    rows_r, t_r, n_r = bench.reference_rows(None, "embs", 6, 3)
    assert n_o == n_r == 6 and t_o > 0 and t_r > 0 and len(rows_o) == len(rows_r) == 9
    for a, b in zip(rows_o, rows_r):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert sum(len(r) for r in rows_r) > 0
    rec, rows = bench.cpu_baseline(None, "embs", 4)
    assert rec["kind"] == "reference" and rec["rows_equal_oracle_rows"] is True and rec["value"] > 0
    assert rec["port"]["kind"] == "port" and rec["port"]["value"] > 0 and rec["reference_form"] in ("source", "compiled")
    assert "reference boxmot BotSort.update" in rec["sample"] and len(rows) == 7


def test_all_stream_invariants_and_the_oracle_confirmation_of_row_shortfalls():
    """The bench's oracle-free checks over every stream-frame, and the anomaly-directed oracle re-run: rows the oracle itself returns pass
    (a frame with fewer rows than detections counts as a shortfall, not a fault, and is CONFIRMED when the oracle shows the same rows); a
    forged shortfall is refuted; a duplicated id is a hard violation."""
    import bench
    T, S, nd = 8, 2, bench.N_TRACKS
    out_h = np.zeros((T, S, nd, 8), dtype=np.float32)
    out_n = np.zeros((T, S), dtype=np.int32)
    cnt = np.zeros((T, S), dtype=np.int32)
    from boxmot_amd.scenario import Scenario
    for s in range(S):
        rows, _, _ = bench.oracle_rows(None, "embs", T - 3, s)
        sc = Scenario(bench.N_DETS, bench.N_TRACKS, bench.WIDTH, bench.HEIGHT, bench.EMB_DIM, stream=s, random_image=False)
        for t in range(T):
            out_h[t, s, : len(rows[t])] = rows[t]
            out_n[t, s] = len(rows[t])
            cnt[t, s] = len(sc.frame(t, with_embs=False)[0])
    inv = bench.all_stream_invariants(out_h, out_n, cnt, S, T)
    rec = inv["all_streams_invariants"]
    assert rec["all_true"] and rec["first_violations_t_s_rows_dets"] == [] and rec["of_stream_frames"] == S * (T - 3)
    n0 = rec["stream_frames_with_fewer_rows_than_detections_from_frame_3"]
    # forge a shortfall on stream 1, frame 5: drop the last row
    out_n[5, 1] -= 1
    inv = bench.all_stream_invariants(out_h, out_n, cnt, S, T)
    rec = inv["all_streams_invariants"]
    assert rec["all_true"] and rec["stream_frames_with_fewer_rows_than_detections_from_frame_3"] == n0 + 1
    assert [5, 1, int(out_n[5, 1]), int(cnt[5, 1])] in rec["first_shortfalls_t_s_rows_dets"]
    inv = bench.confirm_row_shortfalls(inv, out_h, out_n, None, "embs")
    rec = inv["all_streams_invariants"]
    assert rec["shortfalls_confirmed_by_oracle"] is False and rec["shortfalls_checked_against_oracle"][0]["stream"] == 1
    out_n[5, 1] += 1
    # a duplicated id is a hard violation
    out_h[6, 0, 1, 4] = out_h[6, 0, 0, 4]
    rec = bench.all_stream_invariants(out_h, out_n, cnt, S, T)["all_streams_invariants"]
    assert not rec["all_true"] and not rec["no_duplicate_id_or_det_ind_in_a_frame"] and rec["first_violations_t_s_rows_dets"][0][:2] == [6, 0]
