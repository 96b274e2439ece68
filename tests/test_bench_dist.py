"""bench.py's N > 1 control flow on CPU processes (gloo, world 2): rank launch (`--gpus 2` on its own, and the driver's
torch.distributed.run command), stream sharding, the barrier-bracketed timed loop, the MAX all_reduce of the elapsed time, the
timed result gather and the one JSON line of rank 0.  The device handle is replaced by bench.py's documented test stand-in
(`--stub-tracker`): nothing here measures anything -- the GPU run of the same code is the driver's SCALE run."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
ARGS = ["--stub-tracker", "--backend", "gloo", "--streams", "3", "--steps", "4", "--warmup", "2"]


def _json_line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, f"expected ONE JSON line from rank 0, got {len(lines)}:\n{out[-2000:]}"
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


@pytest.mark.parametrize("mode", ["reid", "embs"])
def test_gpus_flag_alone_launches_the_ranks(mode):
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--mode", mode, *ARGS], capture_output=True, text=True,
                       timeout=300, env=_env(), cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["data"] == "stub"
    assert d["config"]["streams_per_gpu"] == 3 and d["config"]["streams_total"] == 6
    assert d["config"]["gather"]["complete_on_rank0"] is True and d["gather_ms"] > 0
    # whole-job value: frames of all ranks over the max-over-ranks time
    assert abs(d["value"] - 2 * 3 * 4 / (d["ms_per_step"] * 4 / 1000.0)) < 1e-6 * d["value"]
    assert "rank 1/2" in p.stderr and "rank 0/2" in p.stderr          # both ranks ran, on different stream shards
    # the N > 1 line's shape (what the driver's SCALE run parses): every contract key, no per-rank CPU baseline, no side lines
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and "workload" in d["config"]
    assert "model" not in d["config"] and "cpu_baseline" not in d and "other_configs" not in d and "tracker_math_m1" not in d
    assert d["config"]["parallelism"].startswith("2 x 3 independent streams")


def test_driver_style_launch_reads_the_environment():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", *ARGS]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_env(), cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["config"]["streams_total"] == 6 and d["config"]["gather"]["backend"] == "gloo"


def test_single_rank_has_no_collective():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *ARGS], capture_output=True, text=True, timeout=300,
                       env=_env(), cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 1 and "gather_ms" not in d and d["config"]["streams_total"] == 3


def test_rank_sharding_is_disjoint_and_by_rank():
    """rank r builds the scenarios of the global streams r*S .. r*S + S - 1 (bench.py main); the stub echoes detections, so the
    gathered blocks differ between ranks exactly when the shards do"""
    sys.path.insert(0, str(ROOT))
    from boxmot_amd.scenario import Scenario
    a = Scenario(64, 256, 1920, 1080, 512, stream=0, random_image=False).frame(5, with_embs=False)[0]
    b = Scenario(64, 256, 1920, 1080, 512, stream=3, random_image=False).frame(5, with_embs=False)[0]
    assert a.shape == b.shape and not (a == b).all()


def test_force_dist_runs_the_collectives_with_one_rank():
    """`--force-dist`: the process group, the barriers, the MAX all_reduce and the result gather with ONE rank -- the switch that lets a
    single-GPU box execute the RCCL branch of the N > 1 run (`--backend nccl` there; gloo + the stub here)."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--force-dist", *ARGS], capture_output=True, text=True, timeout=300,
                       env=_env(), cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["gather"]["complete_on_rank0"] is True and d["gather_ms"] > 0
