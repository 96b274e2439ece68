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
