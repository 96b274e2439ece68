#!/usr/bin/env python
"""bench.py -- tracker frames/sec for BASELINE.json config 2:
BoT-SORT + OSNet-x0.25 ReID inside update, 64 detections x 256 live tracks, 1080p frames.

  python bench.py --gpus N --steps K --warmup W [--streams S] [--groups G] [--mode embs|reid] [--reid-mode 0|1|2]

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

Extra objects on the JSON line (N = 1): `roofline`, `cpu_baseline` (the contract: kind = "reference" when the reference's own classes
can be imported -- /root/reference in the build container, the byte-compiled oracle/_ref/ (oracle/make_ref.py) on the GPU box -- else
"port" = the oracle; the other one is attached beside it), `cpu_side_baselines` (the reference's tracker math with supplied embeddings
and the reference ByteTrack on BASELINE configuration 1, bounded samples), `tracker_math_m1` (embeddings supplied), and
`other_configs` -- short side measurements of BASELINE.json's configurations 3 (DeepOCSORT + OSNet_x1_0, 128 x 512) and 5 (StrongSORT +
CLIP-ReID ViT-B/16, 256 x 1024, 4K) through tools/config_bench.py, each with its ReID roofline fraction and an embedding parity gate;
they are never part of `value`.

N > 1: one rank per GPU over torch.distributed (backend "nccl" = RCCL).  Either the driver launches the ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, when `--gpus N` is given without WORLD_SIZE, bench.py launches them itself (the same torch.distributed.run
command on 127.0.0.1 with a free port) and relays rank 0's JSON line.  Streams are sharded by rank (rank r owns the global
streams r*S .. r*S+S-1, S = --streams per GPU: weak scaling) with no data-path collective; the timed region is bracketed by a
barrier + device synchronisation on both sides and the elapsed time is the MAX over ranks (all_reduce); the per-frame result
rows are gathered to rank 0 once after the timed loop (`gather_results`: RCCL all_gather of a few hundred KB), timed and
reported separately as `gather_ms`.  Rank 0 keeps the id parity gate at every N -- ids of streams 0, S / 2 and S - 1 of its shard
against the oracle (the first, a middle and the last workgroup / crop range of the timed launch) -- ; the CPU timing baseline runs at
N = 1 only (the contract).

`--stub-tracker --backend gloo` is a TEST seam (tests/test_bench_dist.py): the same launch / sharding / barrier / all_reduce /
gather / JSON code runs on CPU processes with a trivial stand-in for the device handle, so the N > 1 control flow is covered
where no GPU exists.  Its line says `"data": "stub"`; it measures nothing.
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
# one committed profile per kernel family (mode): fused fp32-grade (2), fused fp16 (1)
TRAFFIC_PROFILES = {2: ("profiles/r6_pmc_traffic_final.txt", "profiles/r5_pmc_traffic_final.txt", "profiles/r4_pmc_traffic_final.txt", "profiles/r3_pmc_traffic_final.txt", "profiles/r3_pmc_traffic_hp.txt"),
                    1: ("profiles/r4_pmc_traffic_m1.txt", "profiles/r3_pmc_traffic.txt", "profiles/r2_pmc_traffic.txt", "profiles/r1h_pmc_traffic.txt")}


def library_source_hash():
    """The source hash __graft_entry__.build() recorded next to the library that is loaded (libboxmot_hip.so.buildinfo)."""
    try:
        return json.loads((ROOT / "boxmot_amd" / "libboxmot_hip.so.buildinfo").read_text()).get("source_hash")
    except Exception:
        return None


def profile_traffic_bytes_per_crop(mode):
    """(bytes per crop, profile file, stale): the PMC figure of the newest committed profile of this kernel family.  A profile
    carries the source hash of the library it was taken with (`# source_hash: ...`, written by tools/gpu_session.sh); when that
    differs from the loaded library's -- the kernels changed and nobody re-profiled -- the figure is NOT reported (stale = True)."""
    import re
    have = library_source_hash()
    for rel in TRAFFIC_PROFILES.get(mode, ()):
        f = ROOT / rel
        if f.exists():
            txt = f.read_text()
            m = re.search(r"->\s*([0-9.]+)\s*KB per crop", txt)
            if m:
                h = re.search(r"source_hash:\s*([0-9a-f]+)", txt)
                if not h or have is None or h.group(1) != have:
                    return None, rel, True
                return float(m.group(1)) * 1024.0, rel, False
    return None, None, False


# Dense matrix-pipe peak the ReID region is priced against, per kernel family.  Mode 2 (fp32-grade fused kernels) computes EVERY 1x1
# convolution, the LightConv chains and the stem with fp16 (hi, lo) operand pairs on the fp16 matrix pipe (v_mfma_f32_16x16x32_f16 only:
# three MFMAs per K = 32 product tile, two for a K = 16 layer), fp32 accumulation, and the depthwise taps / gates / shortcuts in fp32 on
# the vector ALU -- there is no fp32 MFMA in the family (its fp32-pipe variant, BM_HP_PW32, measured 14 % slower: profiles/r6_hp_variants_ab.txt).
# It is priced against the fp16 dense peak, like mode 1, on ALGORITHMIC flops (the operand-splitting MFMAs are overhead, not work).
PEAK_TFLOPS = {0: 157.3, 1: 2500.0, 2: 2500.0}
DTYPE = {0: "f32", 1: "f16", 2: "f32-grade: fp16 (hi, lo) operand pairs on the fp16 matrix pipe, fp32 accumulate, fp32 vector taps"}
REID_KERNELS = {0: "per-layer fp32 (v1)", 1: "fused fp16 MFMA", 2: "fused fp32-grade (split-fp16 operand pairs, fp16 MFMA, fp32 accumulate)"}


def parse(argv=None):
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
    ap.add_argument("--reid-mode", type=int, default=int(os.environ.get("BOXMOT_REID_MODE", "2")),
                    help="0: per-layer fp32 kernels, 1: fused fp16 MFMA kernels, 2: fused fp32-grade kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-m1", action="store_true", help="skip the tracker-math-only (embeddings supplied) side measurement")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the short side measurements of BASELINE.json configurations 3 and 5 (tools/config_bench.py)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the barriers / all_reduce / gather even with ONE rank: the RCCL code path of the "
                         "multi-GPU run on a single-GPU box (profiles/r4_nccl_one_rank.txt)")
    ap.add_argument("--side-budget-s", type=float, default=300.0,
                    help="wall budget of the side measurements (configurations 3 and 5, child processes): what does not fit is "
                         "reported as skipped / timed out, the headline line is printed regardless")
    # SURVEY.md section 8(d) asks the id gate over the measured frames; the reference leg costs ~0.4 s per frame on the GPU box's host
    # (ReID inside update on the CPU), so 40 gated frames ~ 16 s: the default.  cpu_baseline() stops at its wall budget on a slow host
    # and the line reports how many frames were actually gated (`parity_id_gate_frames`).
    ap.add_argument("--cpu-frames", type=int, default=40)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--stub-tracker", action="store_true",
                    help="TEST ONLY: run the N > 1 control flow on CPU with a stand-in for the device handle (no measurement)")
    return ap.parse_args(argv)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _host_info():
    total_cores = os.cpu_count() or 1
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return total_cores, cpu_model


def _botsort_kw():
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
    return {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}


def _run_cpu_tracker(trk_update, sc, mode, n_frames, budget_s, tag):
    """3 confirmation frames untimed, then up to n_frames timed (bounded by budget_s of timed work)."""
    rows, t_timed, done = [], 0.0, 0
    for t in range(3 + n_frames):
        dets, embs = sc.frame(t, with_embs=(mode != "reid"))
        t0 = time.perf_counter()
        r = trk_update(dets, sc.image, None if mode == "reid" else embs)
        dt = time.perf_counter() - t0
        if t >= 3:
            t_timed += dt
            done += 1
        rows.append(np.asarray(r))
        log(f"{tag} frame {t}: {dt:.2f}s")
        if t_timed > budget_s:
            break
    return rows, t_timed, max(done, 1)


def oracle_rows(sd, mode, n_frames, stream, budget_s=25.0):
    """The oracle (the pinned port of the reference Python path) on one stream of the workload: the id gate's checker."""
    from boxmot_amd.scenario import Scenario
    from oracle.botsort import BotSortOracle
    from oracle.osnet import OracleReID

    sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=stream, random_image=(mode == "reid"))
    orc = BotSortOracle(reid=OracleReID(sd) if mode == "reid" else None, **_botsort_kw())
    return _run_cpu_tracker(lambda d, img, e: orc.update(d, img, e), sc, mode, n_frames, budget_s, f"oracle stream {stream}")


def reference_rows(sd, mode, n_frames, stream, budget_s=25.0):
    """The REFERENCE classes themselves (oracle/ref_harness.py: /root/reference in the build container, the byte-compiled
    oracle/_ref/ on the GPU box; cv2.resize / lap.lapjv answered by the documented stand-ins) on one stream of the workload."""
    from boxmot_amd.scenario import Scenario
    from oracle import ref_harness

    BotSort = ref_harness.load_botsort()
    trk = BotSort(reid_model=None, use_cmc=False, **_botsort_kw())
    if mode == "reid":
        osnet = ref_harness.load_osnet_module()
        model = osnet.osnet_x0_25(num_classes=1, pretrained=False).eval()
        model.load_state_dict(sd, strict=False)
        trk.model = ref_harness.RefReID(model)      # the constructor builds its ReID through the registry (not importable offline)
    sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=stream, random_image=(mode == "reid"))
    return _run_cpu_tracker(lambda d, img, e: trk.update(d, img, e), sc, mode, n_frames, budget_s, f"reference stream {stream}")


def cpu_baseline(sd, mode, n_frames):
    """The CPU path timed on THIS box's host cores: stream 0 of the same workload, 3 confirmation frames untimed, then n_frames
    timed.  kind = "reference" when the reference classes can be imported (always in the build container; on the GPU box when the
    byte-compiled oracle/_ref/ travelled with the snapshot), else kind = "port" (the oracle).  Returns (record, oracle rows)."""
    import torch

    total_cores, cpu_model = _host_info()
    cores = min(total_cores, 32)      # OSNet-x0.25 at batch 64 does not scale past ~32 threads
    torch.set_num_threads(cores)
    rows, t_port, n_port = oracle_rows(sd, mode, n_frames, 0)
    port = dict(value=n_port / max(t_port, 1e-9), unit="frames/s", cores=cores, kind="port", frames=n_port)
    host = f"host = {cpu_model}, {total_cores} logical cores"
    out = None
    try:
        from oracle import ref_harness
        if ref_harness.reference_runnable():
            r_rows, t_ref, n_ref = reference_rows(sd, mode, n_frames, 0)
            same = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(r_rows, rows))
            out = dict(value=n_ref / max(t_ref, 1e-9), unit="frames/s", cores=cores, kind="reference", cpu_model=cpu_model,
                       host_cores_total=total_cores, reference_form=ref_harness.reference_kind(),
                       rows_equal_oracle_rows=bool(same),
                       sample=f"reference boxmot BotSort.update + reference OSNet-x0.25 (torch CPU, {cores} threads; tracker math one Python "
                              f"thread; cv2.resize / lap.lapjv = the documented stand-ins), stream 0, {n_ref} steady-state frames after 3 "
                              f"confirmation frames, mode={mode}; {host}",
                       port=port)
    except Exception as exc:        # the reference leg never costs the line its baseline
        log(f"reference classes not timed: {type(exc).__name__}: {exc}")
    if out is None:
        out = dict(port, cpu_model=cpu_model, host_cores_total=total_cores,
                   sample=f"oracle (NumPy/SciPy BoT-SORT, one Python thread + torch-CPU OSNet-x0.25 on {cores} threads), stream 0, "
                          f"{n_port} steady-state frames after 3 confirmation frames, mode={mode}; {host}")
    return out, rows


def cpu_side_baselines(budget_s=20.0):
    """Two more CPU figures SURVEY.md section 8(d) asks for, reference classes when runnable (else the oracle port):
    M1 (BoT-SORT tracker math only, embeddings supplied, 64 dets x 256 tracks) and BASELINE.json configuration 1 (ByteTrack, 32
    synthetic detections per frame, 640 x 640: the reference's own CPU-runnable case, tests/performance/benchmark_fps.py:171-220)
    with boxmot_amd.ByteTrack's host-API rate on the same detections beside it."""
    from boxmot_amd.scenario import Scenario
    from oracle import ref_harness

    total_cores, cpu_model = _host_info()
    use_ref = ref_harness.reference_runnable()
    out = {"kind": "reference" if use_ref else "port", "cpu_model": cpu_model, "cores": 1}
    # M1: supplied embeddings
    try:
        fn = reference_rows if use_ref else oracle_rows
        _, t1, n1 = fn(None, "embs", 40, 0, budget_s=budget_s / 2)
        out["m1_tracker_math_64x256"] = {"value": n1 / t1, "unit": "frames/s", "frames": n1}
    except Exception as exc:
        out["m1_tracker_math_64x256"] = {"error": f"{type(exc).__name__}: {exc}"}
    # configuration 1: ByteTrack, 32 dets, 640 x 640
    try:
        sc = Scenario(32, 32, width=640, height=640, emb_dim=8, random_image=False)
        frames = [sc.frame(t)[0] for t in range(20 + 400)]
        img = np.zeros((640, 640, 3), dtype=np.uint8)
        if use_ref:
            trk = ref_harness.load_bytetrack()()
            upd = lambda d: trk.update(d, img)
        else:
            from oracle.bytetrack import ByteTrackOracle
            trk = ByteTrackOracle()
            upd = lambda d: trk.update(d, img)
        for d in frames[:20]:
            upd(d)
        t0 = time.perf_counter()
        n = 0
        for d in frames[20:]:
            upd(d)
            n += 1
            if time.perf_counter() - t0 > budget_s / 2:
                break
        cpu_fps = n / (time.perf_counter() - t0)
        from boxmot_amd.bytetrack import ByteTrack
        dt = ByteTrack(max_tracks=128, max_dets=64)
        for d in frames[:20]:
            dt.update(d, img)
        t0 = time.perf_counter()
        for d in frames[20:20 + n]:
            dt.update(d, img)
        dev_fps = n / (time.perf_counter() - t0)
        dt.close()
        out["config1_bytetrack_32dets_640x640"] = {
            "cpu": {"value": cpu_fps, "unit": "frames/s", "frames": n, "what": ("reference ByteTrack.update" if use_ref else "oracle ByteTrack port") + ", one Python thread"},
            "boxmot_amd_host_api": {"value": dev_fps, "unit": "frames/s", "frames": n,
                                    "what": "boxmot_amd.ByteTrack.update, ONE stream through the synchronous plugin call (upload, one-workgroup frame "
                                            "step, read-back per frame): a latency figure, not the device's throughput"}}
    except Exception as exc:
        out["config1_bytetrack_32dets_640x640"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def m1_tracker_only(kw, dev, rank, streams=256):
    """Side measurement (M1): tracker math only -- embeddings supplied, no ReID -- inputs resident in HBM, at
    the headline's stream count (one workgroup per stream: 256 streams fill the 256 CUs) and at 32 streams (32 CUs busy: the latency
    of the frame step's phase chain); reported with the HBM fraction of its algorithmic 1.68 MB per frame."""
    import torch

    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    W1, K1, nd = 4, 16, N_TRACKS
    T1 = W1 + K1
    S_all = max(int(streams), 32)
    dets_h = np.zeros((T1, S_all, nd, 6), dtype=np.float32)
    cnt_h = np.zeros((T1, S_all), dtype=np.int32)
    embs_h = np.zeros((T1, S_all, nd, EMB_DIM), dtype=np.float32)
    for s in range(S_all):
        sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=rank * S_all + s, random_image=False)
        for t in range(T1):
            d, e = sc.frame(t)
            dets_h[t, s, : len(d)] = d
            cnt_h[t, s] = len(d)
            embs_h[t, s, : len(d)] = e

    def run(S1):
        ms = MultiStreamBotSort(S1, max_tracks=2 * N_TRACKS, max_dets=nd, emb_dim=EMB_DIM, **kw)
        d_dets, d_cnt, d_embs = (torch.from_numpy(np.ascontiguousarray(x[:, :S1])).to(dev) for x in (dets_h, cnt_h, embs_h))
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
        return dev_ms / K1
    ms_full, ms_32 = run(S_all), run(32)
    fps = S_all / (ms_full * 1e-3)
    gbs = fps * 1.68e6 / 1e9
    out = {"mode": "M1 embs-supplied (tracker math only)", "frames_per_s": fps, "streams": S_all, "steps": K1, "ms_per_step": ms_full,
           "kernel": "botsort_step_kernel", "algorithmic_bytes_per_frame": 1.68e6, "hbm_GBps": gbs, "hbm_frac_of_8TBps": gbs / 8000.0,
           "at_32_streams": {"frames_per_s": 32 / (ms_32 * 1e-3), "ms_per_step": ms_32,
                             "note": "32 of 256 CUs busy: the latency of one workgroup's phase chain, not a throughput"},
           # (measured in round 6: 1024 streams take 4 x the 256-stream step, 665 k frames/s -- at 256 registers per lane one 512-thread
           # workgroup fills a CU, so more streams queue behind each other instead of overlapping)
           }
    return out


def side_configs(budget_s):
    """BASELINE.json's other single-GPU configurations, measured briefly on the same box (never part of `value`): ReID inside update,
    frames and detections resident in HBM, each with an embedding gate and a 16-frame id gate against the oracle tracker, with one
    and with two stream groups (both gated against the same oracle rows).  Each configuration runs in a CHILD process
    (tools/config_bench.py --both-groups) under a share of the wall budget: a hang or a crash there costs that side line, not the
    headline."""
    import subprocess
    side = {}
    t_end = time.time() + budget_s
    # the id gate's CPU oracle (an fp32 backbone per frame) gets 35 % of a line's share (the tracker-math gate a third of that again): 16 frames when the host is quick, fewer -- never
    # under 4 -- when it is not; the line reports the count (`id_gate_frames`)
    gate_s = f"{0.35 * budget_s / 2:.0f}"
    plan = (("config3", ["--config", "c3", "--streams", "8", "--steps", "24", "--warmup", "6", "--check-frames", "16", "--reid-mode", "2",
                         "--gate-budget-s", gate_s]),
            # configuration 5: 104 warm-up frames fill every sample bank (nn_budget 100), so the timed steps are steady state
            ("config5", ["--config", "c5", "--streams", "2", "--steps", "24", "--warmup", "104", "--check-frames", "16", "--gate-budget-s", gate_s]))
    for i, (key, args) in enumerate(plan):
        left = t_end - time.time()
        share = left / (len(plan) - i)
        if share < 20:
            side[key] = {"error": f"skipped: {left:.0f} s of the side budget left"}
            continue
        try:
            cp = subprocess.run([sys.executable, str(ROOT / "tools" / "config_bench.py"), *args, "--both-groups"],
                                capture_output=True, text=True, timeout=share)
            lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
            if cp.returncode != 0 or not lines:
                side[key] = {"error": f"rc {cp.returncode}: {cp.stderr.strip()[-300:]}"}
                continue
            side[key] = json.loads(lines[-1])
            log(f"side line {key}: {side[key]['frames_per_s']:.1f} frames/s")
        except subprocess.TimeoutExpired:
            side[key] = {"error": f"timed out after {share:.0f} s (--side-budget-s)"}
        except Exception as exc:                    # a side line never takes the headline down
            side[key] = {"error": f"{type(exc).__name__}: {exc}"}
    return side


def alt_family_line(kw, sd, dev, reid_mode, S=256, W=10, K=40):
    """The same workload through the OTHER fused kernel family (short run, same inputs, same handle type): printed next to the headline so
    that the cost of the precision the headline family carries is on the line."""
    import torch

    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    nd, T = N_TRACKS, W + K
    ms = MultiStreamBotSort(S, max_tracks=2 * N_TRACKS, max_dets=nd, emb_dim=EMB_DIM, reid_weights=sd, **kw)
    ms.set_reid_mode(reid_mode)
    dets_h = np.zeros((T, S, nd, 6), dtype=np.float32)
    cnt_h = np.zeros((T, S), dtype=np.int32)
    frames_h = np.zeros((S, HEIGHT, WIDTH, 3), dtype=np.uint8)
    for s in range(S):
        sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=s, random_image=True)
        frames_h[s] = sc.image
        for t in range(T):
            d, _ = sc.frame(t, with_embs=False)
            dets_h[t, s, : len(d)] = d
            cnt_h[t, s] = len(d)
    d_dets, d_cnt, d_frames = (torch.from_numpy(x).to(dev) for x in (dets_h, cnt_h, frames_h))
    d_ptrs = torch.tensor([d_frames[s].data_ptr() for s in range(S)], dtype=torch.int64, device=dev)
    d_out = torch.zeros((S, nd, 8), dtype=torch.float32, device=dev)
    d_out_n = torch.zeros(S, dtype=torch.int32, device=dev)
    step = lambda t: ms.step_device(d_dets[t].data_ptr(), d_cnt[t].data_ptr(), None, d_ptrs.data_ptr(), HEIGHT, WIDTH,
                                    d_out.data_ptr(), d_out_n.data_ptr())
    for t in range(W):
        step(t)
    ms.synchronize()
    ms.reid_kernel_ms()
    t0 = time.perf_counter()
    for t in range(W, T):
        step(t)
    ms.synchronize()
    dt = time.perf_counter() - t0
    r_ms, r_n = ms.reid_kernel_ms()
    assert (ms.status() == 0).all()
    ms.close()
    crops = int((dets_h[W:, :, :, 4] > kw["track_high_thresh"]).sum())
    tfl = crops * FLOP_PER_CROP / (r_ms * 1e-3) / 1e12 if r_ms > 0 else None
    out = {"reid_kernels": REID_KERNELS[reid_mode], "dtype": DTYPE[reid_mode], "frames_per_s": S * K / dt, "streams": S, "steps": K,
           "warmup": W, "ms_per_step": 1000.0 * dt / K, "reid_launch_ms": r_ms / max(r_n, 1), "reid_tflops": tfl,
           "frac_of_peak": tfl / PEAK_TFLOPS[reid_mode] if tfl else None}
    out.update(reid_parity_gates(sd, reid_mode))
    return out


def all_stream_invariants(out_h, out_n_h, cnt_h, S, T):
    """Oracle-free checks over EVERY stream and EVERY frame of the run (warm-up + timed), so that an indexing fault on a stream the id
    gate does not sample cannot hide.  HARD invariants (a violation is a fault): a frame returns at most one row per detection, ids are
    per-stream track numbers 1 .. N_TRACKS (every object is introduced in the first three frames), no id and no detection index twice in
    a frame, detection indices address the frame's detections.  EXPECTATION, reported with its exceptions: from frame 3 on a stream
    returns exactly one row per detection -- the reference itself leaves a detection without a row now and then (a re-shown object whose
    prediction has drifted past the IoU gate starts an unconfirmed track: no row that frame), so a shortfall is not a fault by itself;
    the caller re-runs the oracle on the first streams that show one (`confirm_row_shortfalls`) and compares whole rows."""
    short_ok = ids_ok = dup_ok = det_ok = True
    bad, shortfalls = [], []
    for t in range(T):
        for s in range(S):
            n = int(out_n_h[t, s])
            r = out_h[t, s, :n]
            ids, det = r[:, 4].astype(np.int64), r[:, 7].astype(np.int64)
            a = n <= int(cnt_h[t, s])
            b = bool(((ids >= 1) & (ids <= N_TRACKS)).all())
            c = len(np.unique(ids)) == n and len(np.unique(det)) == n
            d = bool(((det >= 0) & (det < int(cnt_h[t, s]))).all())
            if not (a and b and c and d) and len(bad) < 4:
                bad.append([t, s, n, int(cnt_h[t, s])])
            if t >= 3 and n != int(cnt_h[t, s]):
                shortfalls.append([t, s, n, int(cnt_h[t, s])])
            short_ok &= a; ids_ok &= b; dup_ok &= c; det_ok &= d
    return {"all_streams_invariants": {"streams": S, "frames": T, "rows_at_most_detections": bool(short_ok),
                                       "ids_within_1_to_n_tracks": bool(ids_ok), "no_duplicate_id_or_det_ind_in_a_frame": bool(dup_ok),
                                       "det_ind_addresses_a_detection": bool(det_ok), "all_true": bool(short_ok and ids_ok and dup_ok and det_ok),
                                       "first_violations_t_s_rows_dets": bad,
                                       "stream_frames_with_fewer_rows_than_detections_from_frame_3": len(shortfalls),
                                       "of_stream_frames": S * max(T - 3, 0), "first_shortfalls_t_s_rows_dets": shortfalls[:4]}}


def confirm_row_shortfalls(inv, out_h, out_n_h, sd, mode, max_streams=2, max_frame=24, budget_s=15.0):
    """The oracle re-run on the first streams whose frames returned fewer rows than detections (the anomaly picks the streams, not the
    sampler): whole rows through the frame of the shortfall must equal the device's.  Adds `shortfalls_confirmed_by_oracle`."""
    rec = inv["all_streams_invariants"]
    seen, checked = set(), []
    for t, s, n, c in rec["first_shortfalls_t_s_rows_dets"]:
        if s in seen or len(seen) >= max_streams or t > max_frame:
            continue
        seen.add(s)
        rows, _, _ = oracle_rows(sd, mode, t - 2, s, budget_s=budget_s)
        ok = len(rows) > t
        for u in range(min(len(rows), t + 1)):
            got = out_h[u, s, : out_n_h[u, s]]
            ok &= got.shape == rows[u].shape and bool(np.array_equal(got[:, 4:], rows[u][:, 4:]))
        checked.append({"stream": s, "through_frame": t, "oracle_rows_at_frame": int(len(rows[t])) if len(rows) > t else None,
                        "device_rows_at_frame": n, "ids_equal_oracle": bool(ok)})
    rec["shortfalls_checked_against_oracle"] = checked
    rec["shortfalls_confirmed_by_oracle"] = bool(all(c["ids_equal_oracle"] for c in checked)) if checked else None
    return inv


def reid_parity_gates(sd, reid_mode):
    """Embeddings of a kernel family against the fp32 oracle (north_star tolerance 1e-3): on the benchmark's weights (the reference's
    own random init, where BatchNorm is the identity) AND on BatchNorm-calibrated random weights of three seeds (non-trivial
    running statistics and affine terms: the case that tells fp32-grade arithmetic from fp16 operands)."""
    from boxmot_amd.reid import HipReID
    from boxmot_amd.reid_weights import random_osnet_state_dict
    from boxmot_amd.scenario import Scenario
    from oracle.osnet import OracleReID
    sc0 = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=0, random_image=True)
    boxes = sc0.frame(0, with_embs=False)[0][:32, :4]
    out = {}
    hr = HipReID(sd, max_crops=32, mode=reid_mode)
    out["reid_max_abs_err_vs_fp32_oracle"] = float(np.abs(hr.get_features(boxes, sc0.image) - OracleReID(sd).get_features(boxes, sc0.image)).max())
    hr.close()
    errs = []
    for seed in (0, 1, 2):
        sdc = random_osnet_state_dict("osnet_x0_25", seed=seed)
        hr = HipReID(sdc, max_crops=16, mode=reid_mode)
        errs.append(float(np.abs(hr.get_features(boxes[:16], sc0.image) - OracleReID(sdc).get_features(boxes[:16], sc0.image)).max()))
        hr.close()
    out["reid_max_abs_err_vs_fp32_oracle_bn_calibrated_seeds012"] = errs
    out["reid_within_1e-3_on_bn_calibrated_weights"] = bool(max(errs) < 1e-3)
    return out


class StubStreams:
    """TEST-ONLY stand-in for MultiStreamBotSort (--stub-tracker): same call surface over host tensors, every detection becomes
    one output row [box, id = index + 1, conf, cls, index].  It exists so that the launch / sharding / barrier / all_reduce /
    gather / JSON code of this file runs under gloo on CPU processes; nothing it does is a measurement or a product path."""

    def __init__(self, n_streams, max_dets):
        self.S, self.nd = n_streams, max_dets
        self._t0 = 0.0

    @staticmethod
    def _view(ptr, shape, dtype):
        import ctypes
        n = int(np.prod(shape))
        buf = (ctypes.c_byte * (n * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)

    def step_device(self, d_dets, d_cnt, d_embs, d_frames, rows, cols, d_out, d_out_n):
        dets = self._view(d_dets, (self.S, self.nd, 6), np.float32)
        cnt = self._view(d_cnt, (self.S,), np.int32)
        out = self._view(d_out, (self.S, self.nd, 8), np.float32)
        out_n = self._view(d_out_n, (self.S,), np.int32)
        idx = np.arange(self.nd, dtype=np.float32)
        out[:, :, :4] = dets[:, :, :4]
        out[:, :, 4] = idx + 1
        out[:, :, 5:7] = dets[:, :, 4:6]
        out[:, :, 7] = idx
        out_n[:] = cnt

    def set_reid_mode(self, mode): pass
    def synchronize(self): pass
    def timer_start(self): self._t0 = time.perf_counter()
    def timer_stop_ms(self): return 1000.0 * (time.perf_counter() - self._t0)
    def reid_kernel_ms(self): return 0.0, 0
    def status(self): return np.zeros(self.S, dtype=np.int32)
    def close(self): pass


def self_launch(a) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves with the command the driver uses
    (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) and relay their output."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    log(f"--gpus {a.gpus} without WORLD_SIZE: launching {a.gpus} ranks: {' '.join(cmd[1:8])} ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this host driver
    return subprocess.call(cmd, env=env)


def main(argv=None):
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        log(f"rank {rank}: --gpus {a.gpus} but WORLD_SIZE {world}: the launcher's world size is what runs")
    stub = bool(a.stub_tracker)
    use_dist = world > 1 or a.force_dist        # collectives on (always with more than one rank)
    if a.force_dist and world == 1:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
            os.environ.setdefault(k, v)
    if stub:
        # no device, no extension: control flow only
        if use_dist:
            dist.init_process_group(a.backend)
            dist.barrier()
        dev = torch.device("cpu")
        dev_sync = lambda: None
    else:
        import __graft_entry__ as g
        if rank == 0:
            g.build()           # rank 0 compiles (or verifies the recorded source hash); the others wait at the barrier and reuse
        if use_dist:
            torch.cuda.set_device(local)
            # bind this rank's communicator to ITS GPU before the first collective (RCCL otherwise picks the device lazily)
            if a.backend == "nccl":
                dist.init_process_group(a.backend, device_id=torch.device("cuda", local))
                dist.barrier(device_ids=[local])
            else:
                dist.init_process_group(a.backend)
                dist.barrier()
        else:
            torch.cuda.set_device(0)
        if rank != 0:
            g.build()
        dev = torch.device("cuda", local if use_dist else 0)
        dev_sync = torch.cuda.synchronize

    from boxmot_amd.scenario import Scenario
    from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS

    S, K, W = a.streams, a.steps, a.warmup
    T = W + K
    log(f"rank {rank}/{world}: S={S} K={K} W={W} mode={a.mode}" + (" [stub tracker: control-flow test]" if stub else ""))
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    nd = N_TRACKS                       # the 3 confirmation frames show every object
    G = max(1, min(a.groups, S))
    while S % G:
        G -= 1
    Sg = S // G
    sd = None
    if stub:
        groups = [StubStreams(Sg, nd) for _ in range(G)]
    else:
        from boxmot_amd.reid_weights import reference_init_state_dict
        from boxmot_amd.streams import MultiStreamBotSort
        sd = reference_init_state_dict("osnet_x0_25", seed=0)   # random init as OSNet._init_params does it
        log("weights generated")
        groups = [MultiStreamBotSort(Sg, max_tracks=2 * N_TRACKS, max_dets=nd, emb_dim=EMB_DIM,
                                     reid_weights=sd if a.mode == "reid" else None, **kw) for _ in range(G)]
    ms = groups[0]
    if a.mode == "reid":
        for m in groups:
            m.set_reid_mode(a.reid_mode)

    # ---- synthetic inputs, resident in HBM before timing: rank r owns the global streams r*S .. r*S + S - 1 ----
    fh, fw = (8, 8) if stub else (HEIGHT, WIDTH)            # the stub never reads pixels
    dets_h = np.zeros((T, S, nd, 6), dtype=np.float32)
    cnt_h = np.zeros((T, S), dtype=np.int32)
    embs_h = np.zeros((T, S, nd, EMB_DIM), dtype=np.float32) if a.mode == "embs" else None
    frames_h = np.zeros((S, fh, fw, 3), dtype=np.uint8)
    for s in range(S):
        sc = Scenario(N_DETS, N_TRACKS, WIDTH, HEIGHT, EMB_DIM, stream=rank * S + s, random_image=a.mode == "reid" and not stub)
        if not stub:
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
    dev_sync()
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
    # ---- the timed region: barrier + device sync on both sides, exactly K steps, MAX over ranks ----
    dev_sync()
    if use_dist:
        dist.barrier()
    ms.timer_start()
    t0 = time.perf_counter()
    for t in range(W, T):
        step(t)
    dev_ms = ms.timer_stop_ms()
    for m in groups:
        m.synchronize()
    dev_sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
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

    # ---- result gather (the only collective of the path), after the timed loop; timed on its own ----
    out_h, out_n_h = d_out.cpu().numpy(), d_out_n.cpu().numpy()
    gather_ms, gathered_ok = None, None
    if use_dist:
        from boxmot_amd.streams import gather_results
        send_rows = d_out[W:].transpose(0, 1).contiguous()           # (S, K, nd, 8)
        send_cnt = d_out_n[W:].transpose(0, 1).contiguous()
        dev_sync()
        dist.barrier()
        tg = time.perf_counter()
        g_rows, g_cnt = gather_results(send_rows, send_cnt, dst=0)
        dev_sync()
        gather_ms = 1000.0 * (time.perf_counter() - tg)
        if rank == 0:
            # every rank's block arrived in rank order with the row counts its scenario implies, and rank 0's own block is intact
            gathered_ok = len(g_rows) == world and all(tuple(r.shape) == tuple(send_rows.shape) for r in g_rows) \
                and bool(torch.equal(g_rows[0], send_rows)) and bool(torch.equal(g_cnt[0], send_cnt)) \
                and all(int(c.sum()) > 0 for c in g_cnt)

    if rank == 0:
        total_frames = world * S * K
        fps = total_frames / elapsed
        n_first = int((dets_h[W:, :, :, 4] > kw["track_high_thresh"]).sum())   # crops of this rank in the timed steps
        res = {
            "metric": "tracker frames/sec (64 dets x 256 tracks, 1080p)", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1000.0 * elapsed / K,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[a.reid_mode] if a.mode == "reid" else "f64", "data": "stub" if stub else "synthetic",
            "config": {"workload": "BoT-SORT + OSNet_x0_25 ReID, 64 dets x 256 tracks, 1080p"
                                   if a.mode == "reid" else "BoT-SORT tracker math only (embeddings supplied), 64 dets x 256 tracks",
                       "streams_per_gpu": S, "streams_total": world * S, "stream_groups": G,
                       "parallelism": f"{world} x {S} independent streams, sharded by rank, no data-path collective",
                       "mode": "M2 reid-in-update" if a.mode == "reid" else "M1 embs-supplied",
                       "reid_kernels": REID_KERNELS[a.reid_mode] if a.mode == "reid" else None,
                       "tracker_params": "botsort.yaml defaults, use_cmc=False", "weights": "random-init OSNet-x0.25 (reference _init_params scheme, seed 0)",
                       "device_ms_timed_region": dev_ms},
        }
        if use_dist:
            res["gather_ms"] = gather_ms
            res["config"]["gather"] = {"backend": a.backend, "bytes_per_rank": int(send_rows.numel() * 4 + send_cnt.numel() * 4),
                                       "complete_on_rank0": gathered_ok}
        if stub:
            pass                                            # control-flow test: nothing below is a measurement
        elif a.mode == "reid" and reid_ms > 0:
            tflops = n_first * FLOP_PER_CROP / (reid_ms * 1e-3) / 1e12
            per_crop, traffic_src, traffic_stale = profile_traffic_bytes_per_crop(a.reid_mode)
            res["roofline"] = {"bound": "mfma", "achieved": tflops, "peak": PEAK_TFLOPS[a.reid_mode], "unit": "TFLOP/s",
                               "frac": tflops / PEAK_TFLOPS[a.reid_mode],
                               "traffic": per_crop * n_first / max(reid_launches, 1) if per_crop else None,
                               "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                               "kernel": "OSNet-x0.25 forward (ReID) region, HIP events on the launch stream",
                               "launch_ms": reid_ms / max(reid_launches, 1), "crops_per_launch": n_first / max(reid_launches, 1)}
        else:
            bytes_per_frame = 1.68e6     # tracker-math algorithmic bytes per frame
            gbs = total_frames / world * bytes_per_frame / (dev_ms * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                               "traffic": None, "kernel": "botsort_step_kernel"}
        for m in groups:
            m.close()
        groups = []
        if stub:
            a.no_m1 = a.no_side_configs = a.no_cpu_baseline = True
        if not a.no_cpu_baseline:
            # the CPU path on stream 0: at N = 1 timed as the CPU baseline (the contract); at every N the id parity gate -- on
            # streams 0, S/2 and S - 1 of rank 0's shard (the first, a middle and the LAST workgroup / crop range of the launch the
            # headline times: an indexing fault at high stream or crop indices fails the gate)
            cb, rows = cpu_baseline(sd, a.mode, a.cpu_frames if world == 1 else min(a.cpu_frames, 4))
            if world == 1:
                res["cpu_baseline"] = cb

            def ids_equal(stream, want):
                ok = True
                for t in range(min(len(want), T)):
                    got = out_h[t, stream, : out_n_h[t, stream]]
                    ok &= got.shape == want[t].shape and bool(np.array_equal(got[:, 4:], want[t][:, 4:]))
                return bool(ok)
            gate = {0: ids_equal(0, rows)}
            for s_chk in sorted({S // 2, S - 1} - {0}):
                r_s, _, _ = oracle_rows(sd, a.mode, min(len(rows) - 3, a.cpu_frames), s_chk, budget_s=12.0)
                gate[s_chk] = ids_equal(s_chk, r_s)
            res["config"]["parity_ids_exact_vs_oracle_stream0"] = gate[0]
            res["config"]["parity_ids_exact_vs_oracle_streams"] = {str(k): v for k, v in gate.items()}
            res["config"]["parity_ids_exact_all_gated_streams"] = all(gate.values())
            res["config"]["parity_id_gate_frames"] = int(min(len(rows), T))
            inv = all_stream_invariants(out_h, out_n_h, cnt_h, S, T)
            try:
                inv = confirm_row_shortfalls(inv, out_h, out_n_h, sd, a.mode)
            except Exception as exc:
                inv["all_streams_invariants"]["shortfalls_confirmed_by_oracle"] = f"not checked: {type(exc).__name__}: {exc}"
            res["config"].update(inv)
            if a.mode == "reid":
                res["config"].update(reid_parity_gates(sd, a.reid_mode))
        # the headline with its gates, as soon as they exist (stderr; the ONE stdout line stays last): what follows are side
        # measurements under a wall budget, none of which can change or lose these fields
        log("headline (gates done): " + json.dumps({k: res[k] for k in res if k not in ("tracker_math_m1", "other_configs", "fp16_family_line")}))
        if world == 1 and a.mode == "reid" and not a.no_m1:
            try:
                res["tracker_math_m1"] = m1_tracker_only(kw, dev, rank, a.streams)
            except Exception as exc:
                res["tracker_math_m1"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and a.mode == "reid" and not a.no_cpu_baseline and not a.no_m1:
            try:
                res["cpu_side_baselines"] = cpu_side_baselines()
            except Exception as exc:
                res["cpu_side_baselines"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and a.mode == "reid" and not a.no_side_configs and a.reid_mode != 1:
            try:
                res["fp16_family_line"] = alt_family_line(kw, sd, dev, 1)
            except Exception as exc:
                res["fp16_family_line"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and a.mode == "reid" and not a.no_side_configs:
            del d_frames, d_out, d_dets                # the child processes get the memory
            torch.cuda.empty_cache()
            res["other_configs"] = side_configs(a.side_budget_s)
        print(json.dumps(res))
    for m in groups:
        m.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
