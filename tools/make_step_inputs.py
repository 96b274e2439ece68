"""Writes the inputs of tools/step_prof (the standalone frame-step timer) for one tracker and configuration shape:

    python tools/make_step_inputs.py --tracker deepocsort --config c3 [--streams 8] [--steps 30] [--warmup 8] [--emb-dim 64]
        -> tools/_build/steps_deepocsort_c3.bin   (travels to the GPU box with the snapshot; tools/_build is not tracked)

No GPU needed: the SURVEY.md section 8(d) scenario (boxmot_amd/scenario.py) frame by frame, and the tracker's configuration struct as
bytes -- boxmot_hip_*_default_config over the YAML defaults the benchmark uses -- so that the C++ tool needs no parameter table.
Layout (little endian): "BMSTEP01", int32 kind (0 botsort, 1 deepocsort, 2 strongsort), S, T, warmup, nd (detection rows per
stream-frame slot), dim, cap (track rows), cfg_len, cfg bytes, counts int32 [T][S], then per (t, s) in that order the count's rows
of dets float32 [n][6] followed by embs float32 [n][dim] (compact: the tool expands them into the nd-row slots).  The embedding width defaults to 64 (the assignment and the filters do not depend on it; a
1280-wide configuration-5 file would be 1.3 GB)."""
import argparse
import ctypes
import struct
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

CONFIGS = {"c2": (64, 256), "c3": (128, 512), "c5": (256, 1024)}
KIND = {"botsort": 0, "deepocsort": 1, "strongsort": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracker", default="deepocsort", choices=list(KIND))
    ap.add_argument("--config", default="c3", choices=list(CONFIGS))
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--emb-dim", type=int, default=64)
    a = ap.parse_args()
    from boxmot_amd import _lib
    from boxmot_amd.scenario import Scenario
    lib = _lib.load()
    nd_frame, ntr = CONFIGS[a.config]
    S, T, dim = a.streams, a.warmup + a.steps, a.emb_dim
    nd, cap = ntr, 2 * ntr              # frames 1..3 show every object
    if a.tracker == "botsort":
        from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS
        cfg = _lib.BotSortConfig()
        lib.boxmot_hip_botsort_default_config(ctypes.byref(cfg))
        for k, v in BOTSORT_YAML_DEFAULTS.items():
            if hasattr(cfg, k) and k != "cmc_method":
                setattr(cfg, k, v)
        cfg.cmc_method = None
        cfg.n_class_lists = 1
    elif a.tracker == "deepocsort":
        cfg = _lib.DeepOcSortConfig()
        lib.boxmot_hip_deepocsort_default_config(ctypes.byref(cfg))
        cfg.cmc_off = 1
    else:
        cfg = _lib.StrongSortConfig()
        lib.boxmot_hip_strongsort_default_config(ctypes.byref(cfg))
    cfg.n_streams, cfg.max_tracks, cfg.max_dets, cfg.emb_dim = S, cap, nd, dim
    cfg_bytes = bytes(cfg)
    scen = [Scenario(nd_frame, ntr, emb_dim=dim, stream=s, random_image=False) for s in range(S)]
    cnt = np.zeros((T, S), np.int32)
    rows = []
    for t in range(T):
        for s in range(S):
            d, e = scen[s].frame(t)
            assert len(d) <= nd and d.shape[1] == 6 and e.shape == (len(d), dim)
            cnt[t, s] = len(d)
            rows.append(np.ascontiguousarray(d, dtype=np.float32).tobytes())
            rows.append(np.ascontiguousarray(e, dtype=np.float32).tobytes())
    out = ROOT / "tools" / "_build" / f"steps_{a.tracker}_{a.config}.bin"
    out.parent.mkdir(parents=True, exist_ok=True)
    with open(out, "wb") as f:
        f.write(b"BMSTEP01")
        f.write(struct.pack("<8i", KIND[a.tracker], S, T, a.warmup, nd, dim, cap, len(cfg_bytes)))
        f.write(cfg_bytes)
        f.write(cnt.tobytes())
        for r in rows:
            f.write(r)
    print(f"{out}: {out.stat().st_size / 1e6:.1f} MB ({a.tracker}, {a.config}: {S} streams x {T} frames, {nd_frame} detections x {ntr} tracks, {dim}-d)")


if __name__ == "__main__":
    main()
