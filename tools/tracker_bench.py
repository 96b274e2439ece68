"""Tracker-math throughput (mode M1: embeddings supplied, inputs resident in HBM) of the three HIP trackers on the
SURVEY.md section 8(d) scenario shapes.  Development tool; the contract benchmark is bench.py.

    python tools/tracker_bench.py [--tracker botsort|deepocsort|strongsort|all] [--config c2|c3|c5] [--streams S] [--steps K]

Prints one JSON line per tracker: frames/s over all streams, ms per step (one step = one frame of every stream),
and a parity gate (output rows of stream 0 vs the oracle on the first frames)."""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

CONFIGS = {"c2": (64, 256, 512), "c3": (128, 512, 512), "c5": (256, 1024, 1280)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracker", default="all")
    ap.add_argument("--config", default="c2")
    ap.add_argument("--streams", type=int, default=16)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--emb-dim", type=int, default=0)
    ap.add_argument("--check-frames", type=int, default=6)
    a = ap.parse_args()
    import torch

    from boxmot_amd import _lib
    from boxmot_amd.scenario import Scenario
    lib = _lib.load()
    nd, ntr, dim = CONFIGS[a.config]
    dim = a.emb_dim or dim
    S, T = a.streams, a.warmup + a.steps
    dev = torch.device("cuda", 0)
    scen = [Scenario(nd, ntr, emb_dim=dim, stream=s, random_image=False) for s in range(S)]
    cap_nd = ntr                         # frames 1..3 show every object
    dets_h = np.zeros((T, S, cap_nd, 6), np.float32)
    embs_h = np.zeros((T, S, cap_nd, dim), np.float32)
    cnt_h = np.zeros((T, S), np.int32)
    for t in range(T):
        for s in range(S):
            d, e = scen[s].frame(t)
            cnt_h[t, s] = len(d)
            dets_h[t, s, : len(d)] = d
            embs_h[t, s, : len(d)] = e
    d_dets, d_embs, d_cnt = (torch.from_numpy(x).to(dev) for x in (dets_h, embs_h, cnt_h))
    cap = 2 * ntr
    which = ["botsort", "deepocsort", "strongsort"] if a.tracker == "all" else [a.tracker]
    for name in which:
        if name == "botsort":
            from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
            cfg = _lib.BotSortConfig()
            lib.boxmot_hip_botsort_default_config(ctypes.byref(cfg))
            for k, v in BOTSORT_YAML_DEFAULTS.items():
                if hasattr(cfg, k) and k not in ("cmc_method",):
                    setattr(cfg, k, v)
            cfg.cmc_method = None
            cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim, cfg.n_class_lists = S, cap, cap_nd, dim, 1
            h = lib.boxmot_hip_botsort_create(ctypes.byref(cfg))
            out_rows = cap_nd
            step = lambda t, o, on: lib.boxmot_hip_botsort_step_device(h, d_dets[t].data_ptr(), d_cnt[t].data_ptr(), d_embs[t].data_ptr(),
                                                                        None, 1080, 1920, o, on)
            sync, destroy = lib.boxmot_hip_botsort_synchronize, lib.boxmot_hip_botsort_destroy
        elif name == "deepocsort":
            cfg = _lib.DeepOcSortConfig()
            lib.boxmot_hip_deepocsort_default_config(ctypes.byref(cfg))
            cfg.cmc_off = 1
            cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim = S, cap, cap_nd, dim
            h = lib.boxmot_hip_deepocsort_create(ctypes.byref(cfg))
            out_rows = cap
            step = lambda t, o, on: lib.boxmot_hip_deepocsort_step_device(h, d_dets[t].data_ptr(), d_cnt[t].data_ptr(), d_embs[t].data_ptr(), o, on)
            sync, destroy = lib.boxmot_hip_deepocsort_synchronize, lib.boxmot_hip_deepocsort_destroy
        else:
            cfg = _lib.StrongSortConfig()
            lib.boxmot_hip_strongsort_default_config(ctypes.byref(cfg))
            cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim = S, cap, cap_nd, dim
            h = lib.boxmot_hip_strongsort_create(ctypes.byref(cfg))
            out_rows = cap
            step = lambda t, o, on: lib.boxmot_hip_strongsort_step_device(h, d_dets[t].data_ptr(), d_cnt[t].data_ptr(), d_embs[t].data_ptr(), o, on)
            sync, destroy = lib.boxmot_hip_strongsort_synchronize, lib.boxmot_hip_strongsort_destroy
        if not h:
            raise RuntimeError(_lib.last_error())
        d_out = torch.zeros((T, S, out_rows, 8), dtype=torch.float32, device=dev)
        d_out_n = torch.zeros((T, S), dtype=torch.int32, device=dev)
        for t in range(a.warmup):
            _lib.check(step(t, d_out[t].data_ptr(), d_out_n[t].data_ptr()))
        _lib.check(sync(h))
        prof_fn = getattr(lib, "boxmot_hip_debug_ss_prof", None) if name == "strongsort" else None     # a -DBM_SS_PROF build (BOXMOT_HIP_LIB)
        prof = (ctypes.c_ulonglong * 16)()
        if prof_fn:
            prof_fn(prof)                                # clears the warm-up's clocks
        t0 = time.perf_counter()
        for t in range(a.warmup, T):
            _lib.check(step(t, d_out[t].data_ptr(), d_out_n[t].data_ptr()))
        _lib.check(sync(h))
        dt = time.perf_counter() - t0
        phases = {}
        if prof_fn and prof_fn(prof):
            names = ["prologue+predict", "stageA cost build", "-", "stageA matching", "pyset", "stageB", "kf update", "missed+births",
                     "drop+bank feed", "output", "lsa inner iterations", "lsa calls", "clamp pass", "lsa", "-", "-"]
            phases = {"step_phase_kcycles_per_step_wg0": {n: round(prof[i] / a.steps / 1e3, 1) for i, n in enumerate(names) if n != "-" and i < 10 or i in (12, 13)},
                      "lsa_inner_iterations_per_step": prof[10] / a.steps, "lsa_calls_per_step": prof[11] / a.steps}
        # parity gate on stream 0
        if name == "botsort":
            from oracle.botsort import BotSortOracle
            orc = BotSortOracle(**{k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method", "with_reid")})
        elif name == "deepocsort":
            from oracle.deepocsort import DeepOcSortOracle
            orc = DeepOcSortOracle()
        else:
            from oracle.strongsort import StrongSortOracle
            orc = StrongSortOracle()
        ok, out_h, out_n = True, d_out.cpu().numpy(), d_out_n.cpu().numpy()
        for t in range(min(a.check_frames, T)):
            n = cnt_h[t, 0]
            want = np.asarray(orc.update(dets_h[t, 0, :n], None, embs_h[t, 0, :n].copy())).reshape(-1, 8)
            got = out_h[t, 0, : out_n[t, 0]]
            ok = ok and got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]) and np.allclose(got[:, :4], want[:, :4], atol=1e-3)
        extra = {}
        if name == "strongsort":
            # work of the sample-bank distance kernel in the last step: sum over the confirmed tracks of stream 0 of
            # (samples in the bank) x (detections) x dim x 2 FLOP, times the streams (same schedule in every stream)
            ints = np.zeros((cap, 6), np.int32)
            rows, fc, ni = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
            _lib.check(lib.boxmot_hip_strongsort_state_dump(h, 0, ints.ctypes.data, None, None, ctypes.byref(rows), ctypes.byref(fc), ctypes.byref(ni)))
            conf = ints[: rows.value][ints[: rows.value, 1] == 2]
            samples = int(conf[:, 5].sum())
            extra = {"confirmed_tracks": int(len(conf)), "bank_samples_stream0": samples, "banks_full": int((conf[:, 5] >= 100).sum()),
                     "bank_kernel_gflop_per_step": S * samples * int(cnt_h[-1, 0]) * dim * 2 / 1e9,
                     "bank_bytes_per_step_GB": S * samples * dim * 4 / 1e9}
        destroy(h)
        print(json.dumps({"tracker": name, "config": a.config, "streams": S, "steps": a.steps, "warmup": a.warmup, "frames_per_s": S * a.steps / dt,
                          "ms_per_step": 1e3 * dt / a.steps, "rows_stream0_last": int(out_n[-1, 0]),
                          "parity_first_frames_vs_oracle": bool(ok), **extra, **phases}), flush=True)


if __name__ == "__main__":
    main()
