"""The committed final profiles of the round carry the source hash of the library they were taken with (first line, written by
tools/gpu_session.sh: stamp); bench.py reports a profile's figure only while that hash equals the loaded library's.  This test
recomputes the hash from the tree's sources (the way __graft_entry__.build() does) and checks that the three final summaries
and the bench line's traffic source were taken with exactly these sources -- a kernel edit without a new profile session fails here
before it shows up as `traffic_stale` on the bench line."""
import re
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
FINAL = ("r6_kernel_stats_final.txt", "r6_pmc_traffic_final.txt", "r6_mfma_busy_final.txt")
# (round-5 advisor: a stale stamp is reported, not a blocker -- every kernel edit between two GPU sessions would otherwise turn the
# CPU suite red; bench.py carries the same information on its line as `traffic_stale`)


def _tree_hash():
    import __graft_entry__ as g
    sources = sorted(g.CSRC.glob("*.hip")) + sorted(g.CSRC.glob("*.hpp")) + sorted((ROOT / "include").glob("*.h"))
    return g._source_hash(sources, g.HIPCC_FLAGS)


@pytest.mark.parametrize("name", FINAL)
def test_final_profiles_were_taken_with_the_tree_s_sources(name):
    f = ROOT / "profiles" / name
    if not f.exists():
        pytest.skip("no final profile of this round")
    m = re.search(r"source_hash:\s*([0-9a-f]{16})", f.read_text().splitlines()[0])
    assert m, f"{name}: no source-hash stamp on the first line"
    if m.group(1) != _tree_hash():
        pytest.xfail(f"{name} was taken with library {m.group(1)}, the tree's sources hash to {_tree_hash()}: re-run tools/gpu_session.sh trace traffic mfma")


def test_bench_finds_a_current_traffic_profile():
    import json

    import bench
    info = ROOT / "boxmot_amd" / "libboxmot_hip.so.buildinfo"
    if not info.exists():
        pytest.skip("library not built")
    if json.loads(info.read_text()).get("source_hash") != _tree_hash():
        pytest.skip("the built library is not the tree's (build() will rebuild it)")
    per_crop, src, stale = bench.profile_traffic_bytes_per_crop(2)
    if stale or src != "profiles/r6_pmc_traffic_final.txt":
        pytest.xfail(f"bench reports traffic from {src} (stale={stale}): no round-6 traffic profile of the tree's sources yet")
    assert 2.0e6 < per_crop < 3.2e6
