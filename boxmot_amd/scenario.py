"""Synthetic multi-object scenario used by bench.py, smoke() and the parity tests.

The reference's FPS harness (tests/performance/benchmark_fps.py:60-94, 171-196)
draws ``n`` random boxes, jitters them every frame and reuses ONE static random
image; its generator yields "tracks == dets".  BASELINE.json's metric needs a
256-track pool with 64 detections per frame, so the generator is extended as
specified in SURVEY.md section 8(d) / Appendix A.2:

  * N_t objects sit at the cell centres of a grid covering the frame, sized
    U(0.4, 0.8) of a cell so boxes never overlap;
  * frames 0..2 show every object (all tracks get confirmed);
  * later frames show N_d = 0.75*N_d persistent objects + 0.25*N_d slots that
    rotate round-robin over the remaining objects (each re-seen well inside
    ``track_buffer``), so the pool stays at N_t and update / re_activate /
    mark_lost are all exercised;
  * per-frame jitter N(0, 1 px), conf U(0.70, 0.95), cls 0;
  * each object has a fixed unit appearance vector (+ N(0, 0.01) per frame)
    for the "embeddings supplied" mode; seed = 42 + N_d (+ 1000 * stream).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class Scenario:
    n_dets: int = 64
    n_tracks: int = 256
    width: int = 1920
    height: int = 1080
    emb_dim: int = 512
    stream: int = 0
    random_image: bool = True
    crowd: bool = False       # every odd object sits a third of a box beside its even neighbour (overlapping pairs: IoU ~ 0.4-0.6)

    def __post_init__(self):
        self.seed = 42 + self.n_dets + 1000 * self.stream
        rng = np.random.default_rng(self.seed)
        nt, w, h = self.n_tracks, self.width, self.height
        cols = math.ceil(math.sqrt(nt * w / h))
        rows = math.ceil(nt / cols)
        cell_w, cell_h = w / cols, h / rows
        idx = np.arange(nt)
        self.cx = (idx % cols + 0.5) * cell_w
        self.cy = (idx // cols + 0.5) * cell_h
        self.bw = rng.uniform(0.4, 0.8, nt) * cell_w
        self.bh = rng.uniform(0.4, 0.8, nt) * cell_h
        if self.crowd:          # (drawn from no RNG: the other sequences of a scene stay what they are without the flag)
            odd = np.arange(1, nt, 2)
            self.cx[odd] = self.cx[odd - 1] + 0.3 * self.bw[odd - 1]
            self.cy[odd] = self.cy[odd - 1]
            self.bw[odd], self.bh[odd] = self.bw[odd - 1], self.bh[odd - 1]
        emb = rng.standard_normal((nt, self.emb_dim))
        self.emb = (emb / np.linalg.norm(emb, axis=1, keepdims=True)).astype(np.float32)
        self.perm = rng.permutation(nt)
        self.n_persist = int(0.75 * self.n_dets)
        self.n_rot = self.n_dets - self.n_persist
        rest = nt - self.n_persist
        self.n_groups = max(1, math.ceil(rest / max(self.n_rot, 1)))
        if self.random_image:
            self.image = rng.integers(0, 255, (h, w, 3), dtype=np.uint8)
        else:
            self.image = np.zeros((h, w, 3), dtype=np.uint8)
        self._rng = rng

    def visible(self, t: int) -> np.ndarray:
        if t < 3 or self.n_tracks <= self.n_dets:
            return self.perm
        g = t % self.n_groups
        rot = self.perm[self.n_persist:][g * self.n_rot:(g + 1) * self.n_rot]
        return np.concatenate([self.perm[: self.n_persist], rot])

    def frame(self, t: int, with_embs: bool = True):
        """Returns (dets (n,6) fp32, embs (n,D) fp32) for 0-based frame ``t`` (call in order).
        ``with_embs=False`` skips drawing the appearance noise (embs is None); the detection
        sequence then differs from the with_embs=True sequence (one RNG), so use one setting
        consistently on both sides of a comparison."""
        rng = self._rng
        idx = self.visible(t)
        n = len(idx)
        jx = rng.normal(0.0, 1.0, n)
        jy = rng.normal(0.0, 1.0, n)
        conf = rng.uniform(0.70, 0.95, n)
        x1 = self.cx[idx] - self.bw[idx] / 2 + jx
        y1 = self.cy[idx] - self.bh[idx] / 2 + jy
        x2 = self.cx[idx] + self.bw[idx] / 2 + jx
        y2 = self.cy[idx] + self.bh[idx] / 2 + jy
        dets = np.stack([x1, y1, x2, y2, conf, np.zeros(n)], axis=1).astype(np.float32)
        if not with_embs:
            return dets, None
        embs = (self.emb[idx] + rng.normal(0.0, 0.01, (n, self.emb_dim))).astype(np.float32)
        return dets, embs

    def frames(self, n_frames: int):
        return [self.frame(t) for t in range(n_frames)]


def stress_frames(n_frames: int, seed: int = 7, width: int = 640, height: int = 480,
                  max_objects: int = 24, emb_dim: int = 32):
    """Adversarial small sequence for parity tests: overlapping random-walk objects,
    births / deaths / occlusions, low-confidence detections, duplicate-prone boxes,
    empty frames.  Exercises every BoT-SORT branch (second association, unconfirmed
    handling, removal by age, duplicate removal, removed-deque overflow)."""
    rng = np.random.default_rng(seed)
    n_obj = max_objects
    pos = np.stack([rng.uniform(40, width - 40, n_obj), rng.uniform(40, height - 40, n_obj)], 1)
    vel = rng.normal(0, 3.0, (n_obj, 2))
    size = np.stack([rng.uniform(20, 70, n_obj), rng.uniform(30, 110, n_obj)], 1)
    emb = rng.standard_normal((n_obj, emb_dim))
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    alive = rng.uniform(size=n_obj) < 0.6
    out = []
    for t in range(n_frames):
        pos += vel + rng.normal(0, 0.7, pos.shape)
        vel += rng.normal(0, 0.3, vel.shape)
        flip = rng.uniform(size=n_obj) < 0.04
        alive = np.where(flip, ~alive, alive)
        if t % 37 == 36:
            vis = np.zeros(n_obj, dtype=bool)  # empty frame
        else:
            vis = alive & (rng.uniform(size=n_obj) > 0.15)
        idx = np.nonzero(vis)[0]
        n = len(idx)
        conf = rng.uniform(0.05, 0.98, n)
        jit = rng.normal(0, 1.5, (n, 2))
        c = pos[idx] + jit
        s = size[idx] * rng.uniform(0.9, 1.1, (n, 2))
        dets = np.concatenate([c - s / 2, c + s / 2, conf[:, None], rng.integers(0, 3, (n, 1))], 1)
        # occasional duplicate detection of the same object (slightly shifted)
        if n and rng.uniform() < 0.3:
            k = rng.integers(0, n)
            dup = dets[k].copy()
            dup[:4] += rng.normal(0, 2.0, 4)
            dup[4] = rng.uniform(0.3, 0.9)
            dets = np.vstack([dets, dup])
            idx = np.append(idx, idx[k])
        e = emb[idx] + rng.normal(0, 0.05, (len(idx), emb_dim))
        order = rng.permutation(len(idx))
        out.append((dets[order].astype(np.float32), e[order].astype(np.float32)))
    return out


def camera_warps(n_frames: int, seed: int = 5, every: int = 1):
    """A seeded schedule of small 2x3 fp32 camera-motion warps (rotation/scale jitter ~1e-2, translation
    ~3 px), the kind ``cmc.apply`` returns in the reference (motion/cmc/ecc.py:45-96).  Frames where
    ``t % every != 0`` get the identity."""
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_frames):
        H = np.eye(2, 3, dtype=np.float32)
        a = rng.normal(0.0, 0.01, (2, 2)).astype(np.float32)
        tr = rng.normal(0.0, 3.0, 2).astype(np.float32)
        if t % every == 0:
            H[:, :2] += a
            H[:, 2] = tr
        out.append(H)
    return out
