#!/usr/bin/env python
"""bench.py -- tracker frames/sec for BASELINE.json config 2:
BoT-SORT + OSNet-x0.25 ReID inside update, 64 detections x 256 live tracks, 1080p frames.

  python bench.py --gpus N --steps K --warmup W [--streams S] [--groups G] [--mode embs|reid] [--reid-mode 0|1]

One "step" = one pass of the hot path over one batch = every one of the S streams of this GPU
advances by one frame (ReID crop/resize/normalise + OSNet + cost matrices + assignment + Kalman +
bookkeeping, all on the device).  Inputs (detections of every frame, one static random frame per
stream -- the reference harness also reuses one image, tests/performance/benchmark_fps.py:186) are
resident in HBM before the timed region.  value = total frames of all streams on all GPUs / wall
time (max over ranks), i.e. whole-job frames/sec; scaling is weak (S streams per GPU).  The S streams
can be split into G groups (--groups, default 1), each with its own handle and HIP stream, so that one
group's tracker step (one workgroup per stream: latency-bound, few CUs) overlaps the other group's ReID
kernels (+4 % at G = 2); the default keeps one group so that the HIP-event launch durations behind
`roofline` are those of kernels running alone and agree with the rocprofv3 per-kernel averages.

Extra objects on the JSON line (N = 1): `roofline`, `cpu_baseline` (the contract), `tracker_math_m1` (embeddings supplied), and
`other_configs` -- short side measurements of BASELINE.json's configurations 3 (DeepOCSORT + OSNet_x1_0, 128 x 512) and 5 (StrongSORT +
CLIP-ReID ViT-B/16, 256 x 1024, 4K) through tools/config_bench.py, each with its ReID roofline fraction and an embedding parity gate;
they are never part of `value`.

N > 1: launched by torch.distributed.run, one rank per GPU; streams are sharded by rank with no
data-path collective; the per-frame result rows are gathered to rank 0 once after the timed loop
(RCCL all_gather of a few hundred KB).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM = 64, 256, 1920, 1080, 512
FLOP_PER_CROP = 2 * 82_314_880          # OSNet x0.25 forward, BASELINE.md section 2
# HBM bytes per crop of the fused ReID launch set: read from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE summary named
# here (separate --pmc passes, gfx950 FETCH correction applied by profiles/summarize_pmc.py) -- counters cannot be collected
# inside a timed run, so the line carries the profile's figure and says which file it came from
TRAFFIC_PROFILES = ("profiles/r2_pmc_traffic.txt", "profiles/r1h_pmc_traffic.txt")


def profile_traffic_bytes_per_crop():
    import re
    for rel in TRAFFIC_PROFILES:
        f = ROOT / rel
        if f.exists():
            m = re.search(r"->\s*([0-9.]+)\s*KB per crop", f.read_text())
            if m:
                return float(m.group(1)) * 1024.0, rel
    return None, None


PEAK_TFLOPS = {0: 157.3, 1: 2500.0}     # dense MFMA peak of the dtype the ReID kernels compute in (fp32 / fp16)
DTYPE = {0: "f32", 1: "f16"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--groups", type=int, default=1,
                    help="stream groups per GPU, each with its own handle and HIP stream (one group's tracker step overlaps "
                         "the other's ReID kernels)")
    ap.add_argument("--steps", type=int, default=200)       # warm-up 40 frames, measure >= 200
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--mode", choices=("reid", "embs"), default="reid",
                    help="reid: ReID inside update (headline, M2); embs: embeddings supplied (tracker math only, M1)")
    ap.add_argument("--reid-mode", type=int, default=int(os.environ.get("BOXMOT_REID_MODE", "1")),
                    help="0: per-layer fp32 kernels, 1: fused fp16 MFMA kernels (default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-m1", action="store_true", help="skip the tracker-math-only (embeddings supplied) side measurement")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the short side measurements of BASELINE.json configurations 3 and 5 (tools/config_bench.py)")
    ap.add_argument("--cpu-frames", type=int, default=10)
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(sd, mode, n_frames):
    """The oracle (a port of the reference Python path) timed on this box's host cores:
    stream 0 of the same workload, 3 confirmation frames untimed, then n_frames timed."""
    import torch

    from boxmot_amd.scenario import Scenario
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    from oracle.botsort import BotSortOracle
    from oracle.osnet import OracleReID

    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    total_cores = os.cpu_count() or 1
    cores = min(total_cores, 32)      # OSNet-x0.25 at batch 64 does not scale past ~32 threads
    torch.set_num_threads(cores)
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=0, random_image=True)
    orc = BotSortOracle(reid=OracleReID(sd) if mode == "reid" else None, **kw)
    rows = []
    t_timed = 0.0
    done = 0
    for t in range(3 + n_frames):
        dets, embs = sc.frame(t, with_embs=(mode != "reid"))
        t0 = time.perf_counter()
        r = orc.update(dets, sc.image, None if mode == "reid" else embs)
        dt = time.perf_counter() - t0
        if t >= 3:
            t_timed += dt
            done += 1
        rows.append(r)
        log(f"cpu baseline frame {t}: {dt:.2f}s")
        if t_timed > 25.0:
            break
    n_frames = max(done, 1)
    return dict(value=n_frames / max(t_timed, 1e-9), unit="frames/s", cores=cores, kind="port",
                cpu_model=cpu_model, host_cores_total=total_cores,
                sample=f"oracle (NumPy/SciPy BoT-SORT, one Python thread + torch-CPU OSNet-x0.25 on {cores} threads), stream 0, "
                       f"{n_frames} steady-state frames after 3 confirmation frames, mode={mode}; host = {cpu_model}, "
                       f"{total_cores} logical cores"), rows


def m1_tracker_only(kw, dev, rank):
    """Side measurement (M1): tracker math only -- embeddings supplied, no ReID -- on 32 streams,
    6 warm-up + 30 timed steps, inputs resident in HBM; reported with the HBM fraction of its algorithmic 1.68 MB per frame."""
    import torch

    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    S1, W1, K1, nd = 32, 6, 30, N_TRACKS
    T1 = W1 + K1
    ms = MultiStreamBotSort(S1, max_tracks=2 * N_TRACKS, max_dets=nd, emb_dim=EMB_DIM, **kw)
    dets_h = np.zeros((T1, S1, nd, 6), dtype=np.float32)
    cnt_h = np.zeros((T1, S1), dtype=np.int32)
    embs_h = np.zeros((T1, S1, nd, EMB_DIM), dtype=np.float32)
    for s in range(S1):
        sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=rank * S1 + s, random_image=False)
        for t in range(T1):
            d, e = sc.frame(t)
            dets_h[t, s, : len(d)] = d
            cnt_h[t, s] = len(d)
            embs_h[t, s, : len(d)] = e
    d_dets, d_cnt, d_embs = (torch.from_numpy(x).to(dev) for x in (dets_h, cnt_h, embs_h))
    d_out = torch.zeros((S1, nd, 8), dtype=torch.float32, device=dev)
    d_out_n = torch.zeros(S1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for t in range(W1):
        ms.step_device(d_dets[t].data_ptr(), d_cnt[t].data_ptr(), d_embs[t].data_ptr(), None, HEIGHT, WIDTH, d_out.data_ptr(), d_out_n.data_ptr())
    ms.synchronize()
    ms.timer_start()
    for t in range(W1, T1):
        ms.step_device(d_dets[t].data_ptr(), d_cnt[t].data_ptr(), d_embs[t].data_ptr(), None, HEIGHT, WIDTH, d_out.data_ptr(), d_out_n.data_ptr())
    dev_ms = ms.timer_stop_ms()
    ms.synchronize()
    assert (ms.status() == 0).all()
    ms.close()
    fps = S1 * K1 / (dev_ms * 1e-3)
    gbs = fps * 1.68e6 / 1e9
    return {"mode": "M1 embs-supplied (tracker math only)", "frames_per_s": fps, "streams": S1, "steps": K1, "ms_per_step": dev_ms / K1,
            "kernel": "botsort_step_kernel", "algorithmic_bytes_per_frame": 1.68e6, "hbm_GBps": gbs, "hbm_frac_of_8TBps": gbs / 8000.0}


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    import __graft_entry__ as g
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        g.build()
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl")
        dist.barrier()
    else:
        torch.cuda.set_device(0)
    if rank != 0:
        g.build()
    dev = torch.device("cuda", local if world > 1 else 0)

    from boxmot_amd.reid_weights import reference_init_state_dict
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS

    S, K, W = a.streams, a.steps, a.warmup
    T = W + K
    log(f"rank {rank}/{world}: S={S} K={K} W={W} mode={a.mode}")
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sd = reference_init_state_dict("osnet_x0_25", seed=0)   # random init as OSNet._init_params does it
    log("weights generated")
    nd = N_TRACKS                       # the 3 confirmation frames show every object
    G = max(1, min(a.groups, S))
    while S % G:
        G -= 1
    Sg = S // G
    groups = [MultiStreamBotSort(Sg, max_tracks=2 * N_TRACKS, max_dets=nd, emb_dim=EMB_DIM,
                                 reid_weights=sd if a.mode == "reid" else None, **kw) for _ in range(G)]
    ms = groups[0]
    if a.mode == "reid":
        for m in groups:
            m.set_reid_mode(a.reid_mode)

    # ---- synthetic inputs, resident in HBM before timing ----
    dets_h = np.zeros((T, S, nd, 6), dtype=np.float32)
    cnt_h = np.zeros((T, S), dtype=np.int32)
    embs_h = np.zeros((T, S, nd, EMB_DIM), dtype=np.float32) if a.mode == "embs" else None
    frames_h = np.zeros((S, HEIGHT, WIDTH, 3), dtype=np.uint8)
    for s in range(S):
        sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=rank * S + s, random_image=a.mode == "reid")
        frames_h[s] = sc.image
        for t in range(T):
            d, e = sc.frame(t, with_embs=(a.mode != "reid"))
            dets_h[t, s, : len(d)] = d
            cnt_h[t, s] = len(d)
            if embs_h is not None:
                embs_h[t, s, : len(d)] = e
    d_dets = torch.from_numpy(dets_h).to(dev)
    d_cnt = torch.from_numpy(cnt_h).to(dev)
    d_embs = torch.from_numpy(embs_h).to(dev) if embs_h is not None else None
    d_frames = torch.from_numpy(frames_h).to(dev)
    d_ptrs = torch.tensor([d_frames[s].data_ptr() for s in range(S)], dtype=torch.int64, device=dev)
    d_out = torch.zeros((T, S, nd, 8), dtype=torch.float32, device=dev)
    d_out_n = torch.zeros((T, S), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    log("inputs resident on the device")

    def step(t):
        for gi, m in enumerate(groups):          # asynchronous launches, one HIP stream per group
            lo = gi * Sg
            m.step_device(d_dets[t, lo:lo + Sg].data_ptr(), d_cnt[t, lo:lo + Sg].data_ptr(),
                          d_embs[t, lo:lo + Sg].data_ptr() if d_embs is not None else None,
                          d_ptrs[lo:lo + Sg].data_ptr() if a.mode == "reid" else None, HEIGHT, WIDTH,
                          d_out[t, lo:lo + Sg].data_ptr(), d_out_n[t, lo:lo + Sg].data_ptr())

    for t in range(W):
        step(t)
    for m in groups:
        m.synchronize()
    log("warm-up done")
    for m in groups:
        m.reid_kernel_ms()              # drop warm-up timings
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ms.timer_start()
    t0 = time.perf_counter()
    for t in range(W, T):
        step(t)
    dev_ms = ms.timer_stop_ms()
    for m in groups:
        m.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    log(f"timed loop done: {elapsed:.3f}s")
    for m in groups:
        status = m.status()
        assert (status == 0).all(), f"tracker status {status}"
    reid_ms, reid_launches = 0.0, 0
    for m in groups:
        r_ms, r_n = m.reid_kernel_ms()
        reid_ms += r_ms
        reid_launches += r_n

    # ---- result gather (the only collective of the path), after the timed loop ----
    out_h, out_n_h = d_out.cpu().numpy(), d_out_n.cpu().numpy()
    if world > 1:
        from boxmot_amd.streams import gather_results
        gather_results(d_out[W:].transpose(0, 1).contiguous(), d_out_n[W:].transpose(0, 1).contiguous(), dst=0)

    if rank == 0:
        total_frames = world * S * K
        fps = total_frames / elapsed
        n_first = int((dets_h[W:, :, :, 4] > kw["track_high_thresh"]).sum())   # crops of this rank in the timed steps
        res = {
            "metric": "tracker frames/sec (64 dets x 256 tracks, 1080p)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1000.0 * elapsed / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[a.reid_mode] if a.mode == "reid" else "f64", "data": "synthetic",
            "config": {"workload": "BoT-SORT + OSNet_x0_25 ReID, 64 dets x 256 tracks, 1080p"
                                   if a.mode == "reid" else "BoT-SORT tracker math only (embeddings supplied), 64 dets x 256 tracks",
                       "streams_per_gpu": S, "stream_groups": G, "mode": "M2 reid-in-update" if a.mode == "reid" else "M1 embs-supplied",
                       "reid_kernels": {0: "per-layer fp32 (v1)", 1: "fused fp16 MFMA"}[a.reid_mode] if a.mode == "reid" else None,
                       "tracker_params": "botsort.yaml defaults, use_cmc=False", "weights": "random-init OSNet-x0.25 (reference _init_params scheme, seed 0)",
                       "device_ms_timed_region": dev_ms},
        }
        if a.mode == "reid" and reid_ms > 0:
            tflops = n_first * FLOP_PER_CROP / (reid_ms * 1e-3) / 1e12
            per_crop, traffic_src = profile_traffic_bytes_per_crop() if a.reid_mode == 1 else (None, None)
            res["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": PEAK_TFLOPS[a.reid_mode], "unit": "TFLOP/s",
                               "frac": tflops / PEAK_TFLOPS[a.reid_mode],
                               "traffic": per_crop * n_first / max(reid_launches, 1) if per_crop else None,
                               "traffic_source": traffic_src,
                               "kernel": "OSNet-x0.25 forward (ReID) region, HIP events on the launch stream",
                               "launch_ms": reid_ms / max(reid_launches, 1), "crops_per_launch": n_first / max(reid_launches, 1)}
        else:
            bytes_per_frame = 1.68e6     # tracker-math algorithmic bytes per frame
            gbs = total_frames / world * bytes_per_frame / (dev_ms * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                               "traffic": None, "kernel": "botsort_step_kernel"}
        if world == 1 and a.mode == "reid" and not a.no_m1:
            for m in groups:
                m.close()
            groups = []
            res["tracker_math_m1"] = m1_tracker_only(kw, dev, rank)
        if world == 1 and a.mode == "reid" and not a.no_side_configs:
            # BASELINE.json's other single-GPU configurations, measured briefly on the same box (not part of `value`): ReID inside
            # update, frames and detections resident in HBM, parity gates against the oracles
            sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
            import config_bench
            side = {}
            for key, kwargs in (("config3", dict(config="c3", streams=8, steps=16, warmup=6, check_frames=0)),
                                ("config5", dict(config="c5", streams=2, steps=6, warmup=3, check_frames=0))):
                try:
                    side[key] = config_bench.run(**kwargs)
                    log(f"side line {key}: {side[key]['frames_per_s']:.1f} frames/s")
                except Exception as exc:                    # a side line never takes the headline down
                    side[key] = {"error": f"{type(exc).__name__}: {exc}"}
            res["other_configs"] = side
        if not a.no_cpu_baseline and world == 1:
            cb, rows = cpu_baseline(sd, a.mode, a.cpu_frames)
            res["cpu_baseline"] = cb
            # parity gate printed with the row: ids / det_ind / row order of stream 0 vs the oracle
            ok = True
            for t in range(min(len(rows), T)):
                got = out_h[t, 0, : out_n_h[t, 0]]
                ok &= got.shape == rows[t].shape and bool(np.array_equal(got[:, 4:], rows[t][:, 4:]))
            res["config"]["parity_ids_exact_vs_oracle_stream0"] = bool(ok)
            if a.mode == "reid":
                # embeddings of the benchmark kernels vs the fp32 oracle on stream 0's steady-state crops
                from boxmot_amd.reid import HipReID
                from oracle.osnet import OracleReID
                sc0 = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=0, random_image=True)
                boxes = sc0.frame(0, with_embs=False)[0][:32, :4]
                hr = HipReID(sd, max_crops=32, mode=a.reid_mode)
                err = float(np.abs(hr.get_features(boxes, sc0.image) - OracleReID(sd).get_features(boxes, sc0.image)).max())
                hr.close()
                res["config"]["reid_max_abs_err_vs_fp32_oracle"] = err
        print(json.dumps(res))
    for m in groups:
        m.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
