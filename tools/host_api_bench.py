"""PCIe-inclusive throughput of the drop-in host API (development tool): every update hands over host buffers --
detections and a fresh 1080p uint8 frame per stream -- and reads the rows back, ReID inside update.

    python tools/host_api_bench.py [--streams S] [--steps K]

Two numbers: ``update()`` of a single-stream BotSort (the reference-shaped call) and ``update_batch`` of S streams."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=16)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=6)
    a = ap.parse_args()
    from boxmot_amd.botsort import BotSort
    from boxmot_amd.reid import HipReID
    from boxmot_amd.reid_weights import reference_init_state_dict
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    T = a.warmup + a.steps
    # single stream, reference-shaped call
    sc = Scenario(64, 256, stream=0)
    frames = [sc.frame(t, with_embs=False)[0] for t in range(T)]
    img = sc.image
    trk = BotSort(reid_model=HipReID(sd, mode=1), max_tracks=512, max_dets=256, use_cmc=False, **kw)
    for t in range(a.warmup):
        trk.update(frames[t], img)
    t0 = time.perf_counter()
    for t in range(a.warmup, T):
        trk.update(frames[t], img)
    dt1 = time.perf_counter() - t0
    print(json.dumps({"api": "BotSort.update (1 stream, HipReID.get_features + tracker, host buffers)", "frames_per_s": a.steps / dt1,
                      "ms_per_frame": 1e3 * dt1 / a.steps}), flush=True)
    # S streams per call, frames uploaded every step
    S = a.streams
    scs = [Scenario(64, 256, stream=s) for s in range(S)]
    dets = [[scs[s].frame(t, with_embs=False)[0] for s in range(S)] for t in range(T)]
    imgs = [scs[s].image for s in range(S)]
    ms = MultiStreamBotSort(S, max_tracks=512, max_dets=256, emb_dim=512, reid_weights=sd, **kw)
    ms.set_reid_mode(1)
    for t in range(a.warmup):
        ms.update_batch(dets[t], imgs=imgs)
    t0 = time.perf_counter()
    for t in range(a.warmup, T):
        ms.update_batch(dets[t], imgs=imgs)
    dt = time.perf_counter() - t0
    mb = S * imgs[0].nbytes / 1e6
    # the same through the pinned ingest ring: frame t + 1 is copied into page-locked memory and submitted while frame t is tracked
    from boxmot_amd.ingest import FrameRing
    ms.reset()
    ring = FrameRing(3, S, imgs[0].shape[0], imgs[0].shape[1])
    stack = np.stack(imgs)
    for k in range(3):                       # static frames, as in the pageable run above: a decoder would write straight into the slots
        ring.host_view(k)[...] = stack
    ring.submit(0)
    t_ring = 0.0
    for t in range(T):
        if t == a.warmup:
            t_ring = time.perf_counter()
        k, k1 = t % 3, (t + 1) % 3
        ring.submit(k1)                              # upload of frame t + 1 on the copy stream ...
        ms.update_batch(dets[t], ring=ring, slot=k)  # ... while frame t is tracked
    dtr = time.perf_counter() - t_ring
    print(json.dumps({"api": f"update_batch via FrameRing ({S} streams, pinned host slots, upload of t+1 overlapped with tracking of t)",
                      "frames_per_s": S * a.steps / dtr, "ms_per_step": 1e3 * dtr / a.steps, "h2d_mb_per_step": mb}), flush=True)
    ring.close()
    print(json.dumps({"api": f"update_batch ({S} streams, one 1080p frame per stream uploaded per step)", "frames_per_s": S * a.steps / dt,
                      "ms_per_step": 1e3 * dt / a.steps, "h2d_mb_per_step": mb, "h2d_gb_per_s_if_only_copy": mb / (1e3 * dt / a.steps)}), flush=True)


if __name__ == "__main__":
    main()
