"""Measurement aid: what the oriented frame steps cost beside the axis-aligned ones -- one stream, the seeded stress scenes (about 24
objects), host API.  BoT-SORT / ByteTrack: the step kernel's own time (HIP events, boxmot_hip_botsort_last_track_time_ms); OC-SORT: wall
clock of update() (kernel + two small copies + one synchronisation), the same for both layouts.  Medians over the steady-state frames.

    python tools/obb_step_time.py [frames]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as g  # noqa: E402

g.build()
from boxmot_amd import BotSort, ByteTrack, OcSort  # noqa: E402
from boxmot_amd.scenario import stress_frames  # noqa: E402
from common import obb_frames  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
img = np.zeros((480, 640, 3), np.uint8)
aabb = [d for d, _ in stress_frames(N, seed=4)]
obb = list(obb_frames(N, seed=4))


def run(make, frames, kernel_time):
    trk = make()
    ms = []
    for t, d in enumerate(frames):
        t0 = time.perf_counter()
        trk.update(d, img)
        wall = (time.perf_counter() - t0) * 1e3
        if t >= 20:
            ms.append(trk.get_last_track_time_ms() if kernel_time else wall)
    trk.close()
    return float(np.median(ms)), float(np.percentile(ms, 90))


for name, make, kt in (("botsort", lambda: BotSort(reid_model=None, with_reid=False, use_cmc=False, max_tracks=128, max_dets=64), True),
                       ("bytetrack", lambda: ByteTrack(max_tracks=128, max_dets=64), True),
                       ("ocsort", lambda: OcSort(max_tracks=128, max_dets=64), False),
                       ("ocsort+byte", lambda: OcSort(use_byte=True, max_tracks=128, max_dets=64), False)):
    a, o = run(make, aabb, kt), run(make, obb, kt)
    what = "step kernel (HIP events)" if kt else "update() wall clock"
    print(f"{name:12s} {what:26s} axis-aligned {a[0] * 1e3:8.1f} us (p90 {a[1] * 1e3:8.1f})   oriented {o[0] * 1e3:8.1f} us (p90 {o[1] * 1e3:8.1f})   x{o[0] / a[0]:.2f}",
          flush=True)
