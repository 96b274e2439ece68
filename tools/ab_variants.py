"""A/B of compile-time kernel variants on the device (development tool).

Builds `libboxmot_hip` once per variant into tools/_build/ (run this part in the build container -- hipcc cross-compiles),
then, on the GPU box, runs the ReID parity tests and `bench.py --no-cpu-baseline` against each and prints one line per
variant.  The variants are the off-by-default flags of reid_engine.hpp / reid_fused.hpp:

    python tools/ab_variants.py build                      # here
    gpurun -- 'python tools/ab_variants.py run'            # there
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
OUT = ROOT / "tools" / "_build"
VARIANTS = {
    "base": [],
    "dense": ["-DBM_DENSE_LIGHT=1"],                  # stage-0 LightConv as a dense 3x3 on the matrix pipe
    "head_per_crop": ["-DBM_HEAD_PER_CROP=1"],        # the round-1 head instead of k_head_batched
    "no_prefetch": ["-DBM_PREFETCH_CONV1=0", "-DBM_PREFETCH_EPI=0"],     # round-1 load order in conv1 / epilogue
    "pf_epi_all": ["-DBM_PREFETCH_EPI=1"],
    "s1_handover": ["-DBM_STAGE1_HANDOVER=1"],
    "s2_epi_lds": ["-DBM_STAGE2_EPI_LDS=1"],
    "s2_occ4": ["-DBM_STAGE2_OCC4=1"],
    "s2_both": ["-DBM_STAGE2_EPI_LDS=1", "-DBM_STAGE2_OCC4=1"],
    "all": ["-DBM_STAGE1_HANDOVER=1", "-DBM_STAGE2_EPI_LDS=1", "-DBM_STAGE2_OCC4=1"],
}


def build():
    sys.path.insert(0, str(ROOT))
    import __graft_entry__ as g
    OUT.mkdir(parents=True, exist_ok=True)
    for name, flags in VARIANTS.items():
        lib = OUT / f"libboxmot_hip_{name}.so"
        cmd = [os.environ.get("HIPCC", "hipcc"), *g.HIPCC_FLAGS, *flags, "-o", str(lib), str(g.CSRC / "boxmot_hip.hip")]
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)


def run(with_tests: bool = False, rounds: int = 1, only=None):
    """Interleaved rounds (variant order repeated) so that clock / thermal drift does not read as a kernel property."""
    results = {name: [] for name in VARIANTS}
    parity = {}
    for r in range(rounds):
        for name in VARIANTS:
            lib = OUT / f"libboxmot_hip_{name}.so"
            if not lib.exists() or (only and name not in only):
                continue
            env = dict(os.environ, BOXMOT_HIP_LIB=str(lib))
            if with_tests and r == 0:
                t = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_reid.py", "-x", "-q"], cwd=ROOT, env=env,
                                   capture_output=True, text=True)
                parity[name] = t.returncode == 0
            b = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "30", "--warmup", "8"], cwd=ROOT, env=env,
                               capture_output=True, text=True)
            line = b.stdout.strip().splitlines()[-1] if b.stdout.strip() else "{}"
            try:
                d = json.loads(line)
                results[name].append((d["value"], d["roofline"]["launch_ms"], d.get("parity_ids_exact_vs_oracle_stream0"),
                                      d.get("reid_max_abs_err_vs_fp32_oracle")))
            except Exception:
                print(f"{name:12s} bench failed: {b.stderr[-300:]}", flush=True)
    for name, rs in results.items():
        if not rs:
            print(f"{name}: not built / no result")
            continue
        fps = " ".join(f"{v:.0f}" for v, _, _, _ in rs)
        ms = " ".join(f"{m:.3f}" for _, m, _, _ in rs)
        print(f"{name:12s} frames/s [{fps}]  reid_launch_ms [{ms}]  ids_exact={rs[0][2]} emb_err={rs[0][3]}"
              + (f" reid_tests={'ok' if parity.get(name) else 'FAIL'}" if name in parity else ""), flush=True)


if __name__ == "__main__":
    {"build": build, "run": lambda: run(only=sys.argv[2:] or None), "run_tests": lambda: run(True)}[sys.argv[1] if len(sys.argv) > 1 else "run"]()
