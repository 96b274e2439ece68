"""ReID-only microbenchmark (profiling aid): N crops of one random 1080p frame through the ReID C ABI.
Usage: python tools/reid_microbench.py [n_crops] [mode] [iters]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g  # noqa: E402

g.build()
from boxmot_amd.reid import HipReID  # noqa: E402
from boxmot_amd.reid_weights import reference_init_state_dict  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(0)
img = rng.integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
boxes = np.stack([rng.uniform(0, 1800, n), rng.uniform(0, 1000, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
boxes[:, 2] = boxes[:, 0] + rng.uniform(30, 70, n)
boxes[:, 3] = boxes[:, 1] + rng.uniform(35, 75, n)
reid = HipReID(reference_init_state_dict("osnet_x0_25", seed=0), max_crops=n, mode=mode)
reid.get_features(boxes, img)
t0 = time.perf_counter()
for _ in range(iters):
    f = reid.get_features(boxes, img)
dt = (time.perf_counter() - t0) / iters
print(f"n={n} mode={mode}: {dt * 1e3:.3f} ms per call incl. H2D/D2H, {n / dt:.0f} crops/s, checksum {float(f.sum()):.4f}")
