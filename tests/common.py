"""Shared helpers for the parity tests (inputs are regenerated from seeds; expected rows come
from tests/golden, produced by the real reference -- see tests/golden/make_golden.py)."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from boxmot_amd.scenario import Scenario, stress_frames
from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS

GOLDEN = Path(__file__).resolve().parent / "golden"
YAML = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method", "with_reid")}

CASES = {
    "stress_default": (lambda: stress_frames(150, seed=7), (480, 640), {}, 32),
    "stress_yaml": (lambda: stress_frames(150, seed=7), (480, 640), YAML, 32),
    "stress_short_buffer": (lambda: stress_frames(150, seed=11), (480, 640), dict(track_buffer=5, removed_stracks_buffer=3), 32),
    "c2_yaml": (lambda: Scenario(64, 256, random_image=False).frames(40), (1080, 1920), YAML, 512),
    "c2_default": (lambda: Scenario(64, 256, random_image=False).frames(40), (1080, 1920), {}, 512),
}


def golden_rows(name):
    g = np.load(GOLDEN / "botsort_golden.npz")
    rows, counts = g[name + "_rows"], g[name + "_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [rows[offs[i]:offs[i + 1]] for i in range(len(counts))], g


def assert_rows_match(got, want, frame, box_atol=1e-4):
    got = np.asarray(got, dtype=np.float32).reshape(-1, 8)
    want = np.asarray(want, dtype=np.float32).reshape(-1, 8)
    assert got.shape == want.shape, f"frame {frame}: {got.shape} vs {want.shape}"
    # ids, conf, cls, det_ind and the row order are exact; boxes come from the fp64 Kalman state
    assert np.array_equal(got[:, 4:], want[:, 4:]), f"frame {frame}: id/conf/cls/det_ind columns differ"
    assert np.allclose(got[:, :4], want[:, :4], rtol=0, atol=box_atol), (
        f"frame {frame}: boxes differ by {np.abs(got[:, :4] - want[:, :4]).max()}")


DEEPOCSORT_CASES = {
    # name: (frames factory, image shape, tracker kwargs, emb dim)  -- keep in step with tests/golden/make_golden.py
    "docs_stress_default": (lambda: stress_frames(150, seed=7), (480, 640), {}, 32),
    "docs_stress_short": (lambda: stress_frames(150, seed=11), (480, 640), dict(max_age=5, min_hits=1), 32),
    "docs_stress_awoff": (lambda: stress_frames(120, seed=3), (480, 640), dict(aw_off=True, inertia=0.4, w_association_emb=0.75), 32),
    "docs_stress_noemb": (lambda: stress_frames(120, seed=5), (480, 640), dict(embedding_off=True), 32),
    "docs_c2": (lambda: Scenario(64, 256, emb_dim=128, random_image=False).frames(30), (1080, 1920), {}, 128),
}


def deepocsort_golden_rows(name):
    g = np.load(GOLDEN / "deepocsort_golden.npz")
    rows, counts = g[name + "_rows"], g[name + "_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [rows[offs[i]:offs[i + 1]] for i in range(len(counts))], g


STRONGSORT_CASES = {
    # keep in step with tests/golden/make_golden.py
    "ss_stress_default": (lambda: stress_frames(150, seed=7), (480, 640), {}, 32),
    "ss_stress_short": (lambda: stress_frames(150, seed=11), (480, 640), dict(max_age=5, n_init=1, nn_budget=3), 32),
    "ss_stress_loose": (lambda: stress_frames(120, seed=3), (480, 640),
                        dict(max_cos_dist=0.4, max_iou_dist=0.9, mc_lambda=0.9, ema_alpha=0.8, min_conf=0.3), 32),
    "ss_c2": (lambda: Scenario(64, 256, emb_dim=128, random_image=False).frames(30), (1080, 1920), {}, 128),
}


def strongsort_golden_rows(name):
    g = np.load(GOLDEN / "strongsort_golden.npz")
    rows, counts = g[name + "_rows"], g[name + "_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    return [rows[offs[i]:offs[i + 1]] for i in range(len(counts))], g


MOT17_DIM = 16


def mot17_embeddings(rows: np.ndarray) -> np.ndarray:
    """Deterministic appearance vectors for the MOT17-mini golden (tests/golden/make_golden.py mot17): sines/cosines of the
    box centre at four spatial frequencies plus seeded noise -- a function of the committed detection rows only."""
    cx, cy = (rows[:, 1] + rows[:, 3]) / 2, (rows[:, 2] + rows[:, 4]) / 2
    feats = []
    for k in range(MOT17_DIM // 4):
        f = np.float32(2 * np.pi * (k + 1) / 1920.0 * 3)
        feats += [np.sin(f * cx), np.cos(f * cx), np.sin(f * cy), np.cos(f * cy)]
    emb = np.stack(feats, 1).astype(np.float64) + np.random.default_rng(17).normal(0, 0.05, (len(rows), MOT17_DIM))
    return emb.astype(np.float32)


def bytetrack_device_config(**kw) -> dict:
    """ByteTrack constructor arguments (bytetrack.py:225-233) -> the BoT-SORT step kernel's configuration in its ByteTrack
    mode (what boxmot_hip_bytetrack_default_config / boxmot_amd.ByteTrack set)."""
    c = dict(min_conf=0.1, track_thresh=0.45, match_thresh=0.8, track_buffer=25, frame_rate=30)
    c.update(kw)
    return dict(track_high_thresh=c["track_thresh"], track_low_thresh=c["min_conf"], new_track_thresh=c["track_thresh"],
                match_thresh=c["match_thresh"], proximity_thresh=0.5, appearance_thresh=0.25, second_match_thresh=0.5,
                unconfirmed_match_thresh=0.7, unconfirmed_emb_scale=2.0, fuse_first_associate=1, with_reid=0,
                frame_rate=c["frame_rate"], track_buffer=c["track_buffer"], removed_stracks_buffer=100, kind=1)


ASSO_FUNCS = {"giou": 0.6, "diou": 0.6, "ciou": 0.6, "hmiou": 0.3, "centroid": 0.9}     # keep in step with tests/golden/make_asso_golden.py


def asso_golden_rows(tracker, name):
    """Per-frame rows of the reference DeepOcSort / OcSort(use_byte=True) constructed with asso_func=name (tests/golden/asso_golden.npz,
    written by tests/golden/make_asso_golden.py from the reference classes) and the frames factory that regenerates the inputs."""
    g = np.load(GOLDEN / "asso_golden.npz")
    rows, counts = g[f"{tracker}_{name}_rows"], g[f"{tracker}_{name}_counts"]
    out, o = [], 0
    for n in counts:
        out.append(rows[o:o + n])
        o += n
    return out, (lambda: stress_frames(int(g["frames"]), seed=int(g["seed"])))


def obb_frames(n_frames, seed):
    """The seeded stress scenes with every detection turned into (cx, cy, w, h, angle, conf, cls): the angle follows the box centre
    smoothly so that tracks see a slowly rotating target, with parameterisation flips (w <-> h, angle + pi / 2) thrown in -- the
    ambiguity KalmanFilterXYWH._align_obb_measurement resolves."""
    from boxmot_amd.scenario import stress_frames
    rng = np.random.default_rng(seed)
    for t, (d, _) in enumerate(stress_frames(n_frames, seed=seed)):
        d = np.asarray(d, dtype=np.float32).reshape(-1, 6)
        cx, cy, w, h = (d[:, 0] + d[:, 2]) / 2, (d[:, 1] + d[:, 3]) / 2, d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]
        ang = 0.6 * np.sin(0.02 * t + 0.004 * cx + 0.006 * cy) + rng.normal(0, 0.01, len(d))
        flip = rng.random(len(d)) < 0.15
        w2, h2, a2 = np.where(flip, h, w), np.where(flip, w, h), np.where(flip, ang + np.pi / 2, ang)
        yield np.stack([cx, cy, w2, h2, a2, d[:, 4], d[:, 5]], axis=1).astype(np.float32)


def obb_config2_frames(n_frames, emb_dim=32):
    """BASELINE configuration 2's shape with ORIENTED detections: 64 detections per frame on 256 tracks, 1920 x 1080 (all 256 in the first
    three frames), every box turned into (cx, cy, w, h, angle) with an angle that drifts slowly per object and occasional equivalent
    re-parameterisations (w <-> h, angle + pi / 2).  Yields (dets (n, 7) fp32, embs (n, emb_dim) fp32)."""
    from boxmot_amd.scenario import Scenario
    sc = Scenario(64, 256, emb_dim=emb_dim, random_image=False)
    rng = np.random.default_rng(11)
    for t in range(n_frames):
        d, e = sc.frame(t)
        cx, cy, w, h = (d[:, 0] + d[:, 2]) / 2, (d[:, 1] + d[:, 3]) / 2, d[:, 2] - d[:, 0], d[:, 3] - d[:, 1]
        ang = 0.5 * np.sin(0.03 * t + 0.002 * cx + 0.003 * cy) + rng.normal(0, 0.01, len(d))
        flip = rng.random(len(d)) < 0.1
        w2, h2, a2 = np.where(flip, h, w), np.where(flip, w, h), np.where(flip, ang + np.pi / 2, ang)
        yield np.stack([cx, cy, w2, h2, a2, d[:, 4], d[:, 5]], axis=1).astype(np.float32), e


def obb_golden_rows(key):
    """Per-frame 9-column rows of the reference ByteTrack / BotSort fed oriented detections (tests/golden/obb_golden.npz, written by
    tests/golden/make_obb_golden.py) and the frame count / seed that regenerate the inputs with obb_frames."""
    g = np.load(GOLDEN / "obb_golden.npz")
    rows, counts = g[key + "_rows"], g[key + "_counts"]
    out, o = [], 0
    for n in counts:
        out.append(rows[o:o + n])
        o += n
    return out, int(g["frames"]), int(g["seed"])
