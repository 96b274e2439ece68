"""Profiling aid: per-phase wall-clock breakdown of the tracker step kernel (stream 0)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g  # noqa: E402

g.build()
from boxmot_amd import _lib  # noqa: E402
from boxmot_amd.scenario import Scenario  # noqa: E402
from boxmot_amd.streams import MultiStreamBotSort  # noqa: E402
from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
ms = MultiStreamBotSort(S, max_tracks=512, max_dets=256, emb_dim=512, **kw)
scs = [Scenario(64, 256, stream=s, random_image=False) for s in range(S)]
names = ["det prep", "det feats", "pool lists", "predict", "cost", "assignment", "updates", "second assoc",
         "unconfirmed", "births", "bookkeeping", "output"]
acc = np.zeros(12)
n = 0
for t in range(14):
    fr = [sc.frame(t) for sc in scs]
    ms.update_batch([f[0] for f in fr], None, [f[1] for f in fr])
    clk = np.zeros(16, dtype=np.int64)
    _lib.check(ms._lib.boxmot_hip_botsort_phase_clocks(ms._handle, clk.ctypes.data))
    if t >= 6:
        acc += np.diff(clk[:13])
        n += 1
acc = acc / n / 100.0    # wall_clock64 ticks at 100 MHz -> microseconds
for nm, v in zip(names, acc):
    print(f"{nm:14s} {v:9.1f} us")
print(f"{'total':14s} {acc.sum():9.1f} us  (S={S})")
