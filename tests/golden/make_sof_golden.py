"""Generates tests/golden/sof_golden.npz (build container only): for the preprocessed MOT17-mini frames of tests/golden/ecc_golden.npz
(`small_*`: 0.15-scale grayscale images of the reference's assets/MOT17-mini jpgs), the reference's own detections for those frames scaled to the
small images (det.txt of the two sequences, rows of frames 1..4), and what oracle/sof.py's SofOracle(scale = 1.0) returns fed the small
images as BGR frames: the warp per frame, the number of keypoints kept, the last keypoint set.  It pins the ORACLE against regressions
and feeds the device kernels real image content at the estimator's working resolution -- it is not a cv2 reference (parity unpinned,
see oracle/sof.py)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.sof import SofOracle  # noqa: E402

g = np.load(ROOT / "tests" / "golden" / "ecc_golden.npz")
out = {}
for seq in ("02", "04"):
    small = g[f"small_{seq}"]
    det = np.loadtxt(f"/root/reference/assets/MOT17-mini/train/MOT17-{seq}-FRCNN/det/det.txt", delimiter=",")
    o = SofOracle(scale=1.0)
    warps, nk, dets_all = [], [], []
    nmax = max(int((det[:, 0] == k + 1).sum()) for k in range(len(small)))
    for k in range(len(small)):
        rows = det[det[:, 0] == k + 1]
        tlbr = np.stack([rows[:, 2], rows[:, 3], rows[:, 2] + rows[:, 4], rows[:, 3] + rows[:, 5]], 1) * 0.15
        d = np.zeros((nmax, 4), np.float32)                  # padded with empty boxes (x2 == x1: not masked)
        d[:len(tlbr)] = tlbr.astype(np.float32)
        w = o.apply(np.repeat(small[k][:, :, None], 3, axis=2), d)
        warps.append(w); nk.append(len(o.prev_keypoints)); dets_all.append(d)
        print(seq, k, w.ravel().tolist(), len(o.prev_keypoints), o.last.get("inliers"), o.last.get("matches"))
    out[f"dets_{seq}"] = np.stack(dets_all)
    out[f"warp_{seq}"] = np.stack(warps)
    out[f"nkps_{seq}"] = np.array(nk, np.int32)
    out[f"kps_last_{seq}"] = o.prev_keypoints
np.savez_compressed(ROOT / "tests" / "golden" / "sof_golden.npz", **out)
