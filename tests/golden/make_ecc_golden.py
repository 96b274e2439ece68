"""Generates tests/golden/ecc_golden.npz (build container only): the four frames of each MOT17-mini sequence the reference ships
(assets/MOT17-mini/train/MOT17-{02,04}-FRCNN/img1/*.jpg, decoded with PIL -- cv2 is not installed here), preprocessed by
oracle/ecc.py (BGR2GRAY + 0.15 resize), and the ECC translation oracle/ecc.py finds between consecutive frames.  The fixture
carries the small images so the test needs neither the jpgs nor a jpeg decoder; it pins the ORACLE against regressions and feeds
the device kernels real frames -- it is not a cv2 reference (parity unpinned, see oracle/ecc.py)."""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle.ecc import find_transform_ecc_translation, preprocess  # noqa: E402

out = {}
for seq in ("02", "04"):
    d = Path(f"/root/reference/assets/MOT17-mini/train/MOT17-{seq}-FRCNN/img1")
    frames = [np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]) for p in sorted(d.glob("*.jpg"))]
    small = np.stack([preprocess(f) for f in frames])
    res = [find_transform_ecc_translation(small[k], small[k + 1]) for k in range(len(small) - 1)]
    out[f"small_{seq}"] = small
    out[f"warp_{seq}"] = np.array([r[1] for r in res], dtype=np.float64)
    out[f"iters_{seq}"] = np.array([r[2] for r in res], dtype=np.int32)
    out[f"rho_{seq}"] = np.array([r[0] for r in res], dtype=np.float64)
    print(seq, small.shape, out[f"warp_{seq}"].tolist(), out[f"iters_{seq}"].tolist())
np.savez_compressed(ROOT / "tests" / "golden" / "ecc_golden.npz", **out)
