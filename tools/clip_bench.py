"""CLIP-ReID (ViT-B/16) forward rate on the device: crops of one 4K frame through boxmot_hip_reid_compute_features, backbone time
from the engine's HIP events (boxmot_hip_reid_last_time_ms).  BASELINE.json configuration 5 embeds 256 detections per frame.

    python tools/clip_bench.py [--crops 256] [--iters 10]

Prints one JSON line: crops/s, ms per batch, achieved TFLOP/s (algorithmic FLOPs of the ViT: 2 x MACs of the patch embedding,
the four linear layers and the attention products of the 12 blocks) and the fraction of the 2.5 PFLOP/s dense fp16 MFMA peak.
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def vit_flops(width=768, layers=12, tokens=129, patch_k=768, out_dim=512):
    d, t = width, tokens
    per_layer = 2 * t * d * 3 * d + 2 * 2 * t * t * d + 2 * t * d * d + 2 * 2 * t * d * 4 * d
    return 2 * (t - 1) * patch_k * d + layers * per_layer + 2 * d * out_dim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crops", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from boxmot_amd.clip_weights import pack_clipreid, random_clipreid_state_dict
    from boxmot_amd.reid import HipReID
    blob = pack_clipreid(random_clipreid_state_dict(0))
    reid = HipReID(blob, max_crops=a.crops)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, (2160, 3840, 3), dtype=np.uint8)
    n = a.crops
    boxes = np.stack([rng.uniform(0, 3600, n), rng.uniform(0, 1800, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    boxes[:, 2] = boxes[:, 0] + rng.uniform(40, 200, n)
    boxes[:, 3] = boxes[:, 1] + rng.uniform(80, 340, n)
    reid.get_features(boxes, img)
    pre, proc = [], []
    for _ in range(a.iters):
        reid.get_features(boxes, img)
        p, q = reid.last_time_ms()
        pre.append(p); proc.append(q)
    ms = float(np.median(proc))
    fl = vit_flops() * n
    print(json.dumps({"workload": f"CLIP-ReID ViT-B/16, {n} crops of a 4K frame", "backbone_ms": round(ms, 3),
                      "preprocess_ms": round(float(np.median(pre)), 3), "crops_per_s": round(n / ms * 1e3, 1),
                      "gflop_per_crop": round(vit_flops() / 1e9, 2), "tflops": round(fl / ms / 1e9, 1),
                      "frac_of_fp16_mfma_peak": round(fl / ms / 1e9 / 2500.0, 4)}))
    reid.close()


if __name__ == "__main__":
    main()
