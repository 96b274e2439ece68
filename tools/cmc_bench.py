"""Cost of the camera-motion estimators on the device (development tool; the contract benchmark is bench.py).

    python tools/cmc_bench.py [--streams S] [--steps K]

SOF (boxmot_hip_sof_apply_device): S streams of 1080p frames resident in HBM (a panning window over one large texture, distinct
per stream and per step), 64 detections per stream as the mask; one kernel sequence advances all streams; the S warps are read back
per step (that read is inside the timed loop -- it is how a caller gets them).  ECC (boxmot_hip_ecc_apply_device): one stream per call.
Prints one JSON line: ms per step, stream-frames per second, the median state of the estimates (tracked points, inliers, RANSAC iterations)."""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    import torch
    from scipy.ndimage import gaussian_filter

    from boxmot_amd import _lib
    lib = _lib.load()
    S, R, C, nd = a.streams, 1080, 1920, 64
    rng = np.random.default_rng(0)
    tex = gaussian_filter(rng.integers(0, 255, (R + 256, C + 2048, 3)).astype(np.float32), (6, 6, 0))
    tex = torch.from_numpy(((tex - tex.min()) / (tex.max() - tex.min()) * 255).astype(np.uint8)).cuda()
    T = a.warmup + a.steps
    bufs = [torch.empty((S, R, C, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    dets = torch.zeros((S, nd, 6), dtype=torch.float32)
    for s in range(S):
        x = rng.uniform(0, C - 200, nd); y = rng.uniform(0, R - 300, nd)
        dets[s, :, 0], dets[s, :, 1] = torch.from_numpy(x), torch.from_numpy(y)
        dets[s, :, 2], dets[s, :, 3] = torch.from_numpy(x + rng.uniform(40, 160, nd)), torch.from_numpy(y + rng.uniform(80, 280, nd))
    dets = dets.cuda()
    n = torch.full((S,), nd, dtype=torch.int32, device="cuda")

    def fill(t, buf):
        for s in range(S):
            ox, oy = (7 * s) % 1800 + 5 * t, (3 * s) % 200 + 2 * t
            buf[s].copy_(tex[oy:oy + R, ox:ox + C])
    h = lib.boxmot_hip_sof_create(S, R, C, 0.15, 8, 0.2, 3.0)
    assert h, _lib.last_error()
    warps, info = np.zeros((S, 6)), np.zeros((S, 8), np.int32)
    tables = [torch.tensor([b[s].data_ptr() for s in range(S)], dtype=torch.int64, device="cuda") for b in bufs]
    times, infos = [], []
    for t in range(T):
        fill(t, bufs[t % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.boxmot_hip_sof_apply_device(h, tables[t % 2].data_ptr(), dets.data_ptr(), n.data_ptr(), nd, 6, warps.ctypes.data, info.ctypes.data))
        times.append(time.perf_counter() - t0)
        infos.append(info.copy())
        if t >= 1:
            assert np.abs(warps[:, 2] - 5.0).max() < 0.6 and np.abs(warps[:, 5] - 2.0).max() < 0.6, warps[:3]
    lib.boxmot_hip_sof_destroy(h)
    ms = float(np.mean(times[a.warmup:])) * 1e3
    last = np.stack(infos[a.warmup:])
    res = {"estimator": "sof", "streams": S, "steps": a.steps, "ms_per_step": ms, "stream_frames_per_s": S / (ms * 1e-3), "us_per_stream_frame": ms * 1e3 / S,
           "median_tracked": float(np.median(last[:, :, 2])), "median_inliers": float(np.median(last[:, :, 3])),
           "median_ransac_iters": float(np.median(last[:, :, 4])), "accepted_frac": float(last[:, :, 5].mean()),
           "frame_bytes_read_per_step": S * R * C * 3, "frame_read_GBps": S * R * C * 3 / (ms * 1e-3) / 1e9}
    # ECC, one stream per call
    e = lib.boxmot_hip_ecc_create(1, R, C, 0.15, 1e-5, 100)
    w6, it = np.zeros(6), ctypes.c_int(0)
    et = []
    for t in range(T):
        fill(t, bufs[t % 2][:1]) if False else bufs[t % 2][0].copy_(tex[2 * t:2 * t + R, 5 * t:5 * t + C])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.boxmot_hip_ecc_apply_device(e, 0, bufs[t % 2][0].data_ptr(), w6.ctypes.data, ctypes.byref(it)))
        et.append(time.perf_counter() - t0)
    lib.boxmot_hip_ecc_destroy(e)
    res["ecc_single_stream_ms"] = float(np.mean(et[a.warmup:])) * 1e3
    res["ecc_iterations_last"] = it.value
    print(json.dumps(res))


if __name__ == "__main__":
    main()
