"""Golden rows of the REAL reference ByteTrack, BotSort and OcSort classes fed ORIENTED detections (7 columns: cx, cy, w, h, angle, conf, cls),
on the seeded scenes of tests/common.py:obb_frames.  Build container only (/root/reference under the stand-ins of oracle/ref_harness.py;
the rotated-intersection AREA behind cv2.rotatedRectangleIntersection / contourArea is oracle/obb.py's, see there):

    python tests/golden/make_obb_golden.py

-> tests/golden/obb_golden.npz.  tests/test_oracle_obb.py checks the oracles against it without /root/reference, tests/test_gpu_obb.py
the oriented frame step on the device, tests/test_obb_host_emu.py the host classes over the emulated step.

WHAT THESE ROWS PIN, AND WHAT THEY DO NOT (round-4 advisor finding): they are "reference control flow + stand-in rotated IoU".  The
reference CLASSES run unmodified -- the Kalman filter with the angle state, measurement alignment, the association rounds, the
bookkeeping, the output rows -- but the one number they take from OpenCV, the intersection area of two rotated rectangles, is answered
by oracle/obb.py's fp64 polygon clipping (OpenCV is not installed).  cv2.rotatedRectangleIntersection works in fp32 and differs from
that area at ~1e-6 relative; a pair whose IoU sits within that distance of match_thresh / iou_threshold could be assigned differently by
the real library.  So the oriented fixtures are circular for that one quantity and must not be read as pinned on real OpenCV numerics;
regenerating them on a machine with opencv-python (this script, unchanged, picks up the real cv2 when it imports) closes the gap.
"""
from __future__ import annotations

import logging
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from boxmot_amd.scenario import stress_frames  # noqa: E402
from common import obb_frames  # noqa: E402
from oracle import ref_harness  # noqa: E402

FRAMES, SEED = 90, 4
CASES = {"bytetrack": ("bytetrack", {}), "botsort_noreid": ("botsort", dict(with_reid=False)), "botsort_reid": ("botsort", dict(with_reid=True)),
         "ocsort": ("ocsort", {}), "ocsort_byte": ("ocsort", dict(use_byte=True, max_age=8, min_hits=1))}


def main():
    logging.disable(logging.CRITICAL)
    img = np.zeros((480, 640, 3), np.uint8)
    embs = [e for _, e in stress_frames(FRAMES, seed=SEED)]
    out = {}
    for key, (kind, kw) in CASES.items():
        if kind == "bytetrack":
            trk = ref_harness.load_bytetrack()(**kw)
        elif kind == "ocsort":
            trk = ref_harness.load_ocsort()(**kw)
        else:
            trk = ref_harness.load_botsort()(reid_model=None, use_cmc=False, **kw)
        rows, counts = [], []
        for t, d in enumerate(obb_frames(FRAMES, seed=SEED)):
            e = embs[t].copy() if kw.get("with_reid") else None
            r = np.asarray(trk.update(d.copy(), img, e) if kind == "botsort" else trk.update(d.copy(), img), dtype=np.float64).reshape(-1, 9)
            rows.append(r)
            counts.append(len(r))
        out[key + "_counts"] = np.asarray(counts, dtype=np.int32)
        out[key + "_rows"] = np.concatenate(rows).astype(np.float32)
        print(key, sum(counts), "rows", flush=True)
    np.savez_compressed(Path(__file__).resolve().parent / "obb_golden.npz", frames=np.int32(FRAMES), seed=np.int32(SEED), **out)
    # the same three trackers at BASELINE configuration 2's shape (64 detections on 256 tracks, 1080p): tests/golden/obb_config2_golden.npz
    from common import obb_config2_frames
    big, n = {}, 60
    scene = list(obb_config2_frames(n))
    img = np.zeros((1080, 1920, 3), np.uint8)
    for key, (kind, kw) in {"bytetrack": ("bytetrack", {}), "botsort_reid": ("botsort", dict(with_reid=True)), "ocsort": ("ocsort", dict(use_byte=True))}.items():
        if kind == "bytetrack":
            trk = ref_harness.load_bytetrack()(**kw)
        elif kind == "ocsort":
            trk = ref_harness.load_ocsort()(**kw)
        else:
            trk = ref_harness.load_botsort()(reid_model=None, use_cmc=False, **kw)
        rows, counts = [], []
        for d, e in scene:
            r = np.asarray(trk.update(d.copy(), img, e.copy()) if kind == "botsort" else trk.update(d.copy(), img), dtype=np.float64).reshape(-1, 9)
            rows.append(r)
            counts.append(len(r))
        big[key + "_counts"] = np.asarray(counts, dtype=np.int32)
        big[key + "_rows"] = np.concatenate(rows).astype(np.float32)
        print("config 2 shape:", key, sum(counts), "rows", flush=True)
    np.savez_compressed(Path(__file__).resolve().parent / "obb_config2_golden.npz", frames=np.int32(n), **big)


if __name__ == "__main__":
    main()
