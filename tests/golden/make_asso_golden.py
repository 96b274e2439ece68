"""Golden rows of the REAL reference DeepOcSort and OcSort classes constructed with each axis-aligned association function
(BaseTracker's ``asso_func``: giou, diou, ciou, hmiou, centroid; boxmot/trackers/association/iou.py:118-423), on the seeded stress
scenes of boxmot_amd.scenario.  Build container only (/root/reference imported under the stand-ins of oracle/ref_harness.py):

    python tests/golden/make_asso_golden.py

-> tests/golden/asso_golden.npz: per (tracker, function) the per-frame row counts and the rows (boxes fp32, ids / det indices int32).
The GPU box has no /root/reference: its tests compare the device step and the oracle with these rows.
"""
from __future__ import annotations

import logging
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from boxmot_amd.scenario import stress_frames  # noqa: E402
from oracle import ref_harness  # noqa: E402

FUNCS = {"giou": 0.6, "diou": 0.6, "ciou": 0.6, "hmiou": 0.3, "centroid": 0.9}        # name -> iou_threshold used (scores of the first four live in (0.5, 1])
FRAMES, SEED, IMG_HW = 80, 5, (480, 640)


def main():
    logging.disable(logging.CRITICAL)
    img = np.zeros((*IMG_HW, 3), np.uint8)
    out = {}
    for name, thr in FUNCS.items():
        for tracker in ("deepocsort", "ocsort"):
            if tracker == "deepocsort":
                trk = ref_harness.load_deepocsort()(reid_model=None, cmc_off=True, asso_func=name, iou_threshold=thr)
            else:
                trk = ref_harness.load_ocsort()(asso_func=name, iou_threshold=thr, use_byte=True)
            rows, counts = [], []
            for d, e in stress_frames(FRAMES, seed=SEED):
                r = np.asarray(trk.update(d.copy(), img, e.copy()) if tracker == "deepocsort" else trk.update(d.copy(), img), dtype=np.float64).reshape(-1, 8)
                rows.append(r)
                counts.append(len(r))
            allr = np.concatenate(rows) if sum(counts) else np.zeros((0, 8))
            key = f"{tracker}_{name}"
            out[key + "_counts"] = np.asarray(counts, dtype=np.int32)
            out[key + "_rows"] = allr.astype(np.float32)
            print(key, sum(counts), "rows", flush=True)
    np.savez_compressed(Path(__file__).resolve().parent / "asso_golden.npz", frames=np.int32(FRAMES), seed=np.int32(SEED),
                        thresholds=np.asarray(list(FUNCS.values())), **out)


if __name__ == "__main__":
    main()
