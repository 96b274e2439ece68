"""Golden ASSOCIATION COST MATRICES from the REAL reference (build container only; /root/reference imported through
oracle/ref_harness.py):

    python tests/golden/make_cost_golden.py [botsort] [docs] [ss]       (default: all three groups; a group is re-made in place)

For each case the reference ``BotSort`` (boxmot/trackers/bbox/botsort/botsort.py) runs the scene frame by frame, embeddings
supplied; the module's own names ``iou_distance`` / ``embedding_distance`` / ``fuse_score`` / ``linear_assignment``
(boxmot/trackers/association/matching.py:28-147, bound in botsort.py's namespace) are wrapped so that every call's RETURN VALUE --
and the matrix handed to ``linear_assignment`` -- is recorded; nothing is restated.  Per recorded frame the fixture holds, for the
three associations (first :285-333, second :335-378, unconfirmed :380-431):
    <case>_f<frame>_s<stage>_dists   what linear_assignment was given               (tracks, dets) fp64
    <case>_f<frame>_s<stage>_iou     iou_distance's return value                    (tracks, dets)
    <case>_f<frame>_s<stage>_emb     embedding_distance's return value (stages 0, 2; before the / unconfirmed_emb_scale and the gates)
Inputs are regenerated from the seed by the tests (boxmot_amd.scenario), so the fixture is outputs only.
The lap stand-in (oracle/lap.py, lapx absent) decides the assignment and thereby the NEXT frame's track lists, as everywhere in
tests/golden; the recorded matrices themselves are pure reference arithmetic (NumPy + SciPy cdist).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from boxmot_amd.scenario import Scenario, stress_frames  # noqa: E402
from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS  # noqa: E402
from oracle import ref_harness  # noqa: E402

OUT = Path(__file__).resolve().parent
YAML = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method", "with_reid")}

# name: (frames factory, image shape, tracker kwargs, emb dim, frames whose matrices are kept) -- keep in step with tests/test_cost_values.py
CASES = {
    # BASELINE.json configuration 2's shape: 64 detections x 256 tracks, 512-d
    "c2_yaml": (lambda: Scenario(64, 256, random_image=False).frames(24), (1080, 1920), YAML, 512, (0, 1, 2, 9, 23)),
    "c2_default": (lambda: Scenario(64, 256, random_image=False).frames(24), (1080, 1920), {}, 512, (1, 9, 23)),
    # every pair passes the IoU gate (proximity_thresh = 1: iou_distance > 1 never holds): 64 x 256 ungated pairs > 4096,
    # the device takes its dense LDS-tiled cosine fallback
    "c2_dense": (lambda: Scenario(64, 256, random_image=False).frames(12), (1080, 1920), dict(YAML, proximity_thresh=1.0), 512, (1, 5, 11)),
    # births / losses / re-activations / unconfirmed tracks every few frames, score fusion in the first association
    "stress_fuse": (lambda: stress_frames(60, seed=7), (480, 640), dict(fuse_first_associate=True, track_buffer=5), 32, tuple(range(2, 60, 3))),
}


def record_case(name):
    factory, shape, kw, dim, keep = CASES[name]
    BotSort = ref_harness.load_botsort()
    mod = sys.modules[BotSort.__module__]
    log = []

    def wrap(fn_name):
        real = getattr(mod, fn_name)

        def inner(*a, **k):
            out = real(*a, **k)
            log.append((fn_name, np.array(a[0], dtype=np.float64) if fn_name == "linear_assignment" else np.array(out, dtype=np.float64)))
            return out

        setattr(mod, fn_name, inner)
        return real

    saved = {n: wrap(n) for n in ("iou_distance", "embedding_distance", "fuse_score", "linear_assignment")}
    out = {}
    try:
        img = np.zeros((*shape, 3), dtype=np.uint8)
        trk = BotSort(reid_model=None, use_cmc=False, **kw)
        for t, (d, e) in enumerate(factory()):
            log.clear()
            trk.update(d, img, e.copy())
            # split the call log at the linear_assignment calls: one segment per association, in order
            stages, cur = [], {}
            for fn, val in log:
                if fn == "linear_assignment":
                    cur["dists"] = val
                    stages.append(cur)
                    cur = {}
                elif fn == "iou_distance":
                    cur["iou"] = val
                elif fn == "embedding_distance":
                    cur["emb"] = val
            # (_remove_duplicate_stracks calls iou_distance after the third association: left in `cur`, not recorded)
            assert len(stages) == 3, (name, t, len(stages))
            if t in keep:
                for s, st in enumerate(stages):
                    for k, v in st.items():
                        out[f"{name}_f{t}_s{s}_{k}"] = v
    finally:
        for n, real in saved.items():
            setattr(mod, n, real)
    return out


# DeepOCSORT: the matrices of `associate` (boxmot/trackers/association/association.py:61-152) -- iou_matrix = the association
# function's return value, final_cost = what its linear_assignment is handed (absent on the frames where the already-a-permutation
# early-out of :104-108 answers).  name: (frames, shape, kwargs, emb dim, kept frames, row step): at configuration 3's size only
# every `row step`-th DETECTION row of the (128, 512) matrices is kept (fixture size); the tests compare the same rows.
DOCS_CASES = {
    # (configuration 3's own scene is a grid of separate objects: every frame is answered by the early-out, only iou_matrix exists)
    "docs_c3": (lambda: Scenario(128, 512, emb_dim=512, random_image=False).frames(6), (1080, 1920), {}, 512, (1, 5), 8),
    # the same size with overlapping pairs of objects: the solver runs on the full 128 x 512 final_cost
    "docs_c3_crowd": (lambda: Scenario(128, 512, emb_dim=512, random_image=False, crowd=True).frames(12), (1080, 1920), {}, 512, (1, 2, 5, 11), 4),
    "docs_stress": (lambda: stress_frames(60, seed=7), (480, 640), {}, 32, tuple(range(1, 60, 2)), 1),
    "docs_stress_awoff": (lambda: stress_frames(60, seed=3), (480, 640), dict(aw_off=True, inertia=0.4, w_association_emb=0.75), 32,
                          tuple(range(1, 60, 2)), 1),
}


def record_docs_case(name):
    factory, shape, kw, dim, keep, step = DOCS_CASES[name]
    DeepOcSort = ref_harness.load_deepocsort()
    mod = sys.modules[DeepOcSort.__module__]
    import boxmot.trackers.association.association as assoc_mod

    real_associate, real_la = mod.associate, assoc_mod.linear_assignment
    rec = {}

    def la(cost):
        rec["final_cost"] = np.array(cost, dtype=np.float64)
        return real_la(cost)

    def associate(detections, trackers, asso_func, *a, **k):
        def asso(d, t):
            out = asso_func(d, t)
            rec["iou"] = np.array(out, dtype=np.float64)
            return out
        return real_associate(detections, trackers, asso, *a, **k)

    mod.associate, assoc_mod.linear_assignment = associate, la
    out = {}
    try:
        img = np.zeros((*shape, 3), dtype=np.uint8)
        trk = DeepOcSort(reid_model=None, cmc_off=True, **kw)
        for t, (d, e) in enumerate(factory()):
            rec.clear()
            trk.update(d.copy(), img, e.copy())
            if t in keep:
                for k, v in rec.items():
                    out[f"{name}_f{t}_{k}"] = v[::step]
    finally:
        mod.associate, assoc_mod.linear_assignment = real_associate, real_la
    return out


# StrongSORT: the cost matrices of the two min_cost_matching calls of a frame (sort/linear_assignment.py:14-79): what the
# distance metric returned ("raw": the gated appearance metric of sort/tracker.py:108-122, or iou_cost) and what
# linear_sum_assignment is given after the max_distance clamp ("clamped").  At configuration 5's size (1024 tracks x 256
# detections x 1280-d; the reference needs ~17 s per frame) every 16th TRACK row of two frames is kept.
SS_CASES = {
    "ss_c5": (lambda: Scenario(256, 1024, emb_dim=1280, random_image=False).frames(6), (2160, 3840), {}, 1280, (4, 5), 16),
    "ss_c2": (lambda: Scenario(64, 256, emb_dim=128, random_image=False).frames(12), (1080, 1920), {}, 128, (3, 4, 7, 11), 4),
    "ss_stress": (lambda: stress_frames(60, seed=7), (480, 640), {}, 32, tuple(range(3, 60, 3)), 1),
    "ss_stress_loose": (lambda: stress_frames(60, seed=3), (480, 640),
                        dict(max_cos_dist=0.4, max_iou_dist=0.9, mc_lambda=0.9, ema_alpha=0.8, min_conf=0.3), 32, tuple(range(3, 60, 3)), 1),
}


def record_ss_case(name):
    factory, shape, kw, dim, keep, step = SS_CASES[name]
    StrongSort = ref_harness.load_strongsort()
    import boxmot.trackers.bbox.strongsort.sort.linear_assignment as la_mod

    real_mcm, real_lsa = la_mod.min_cost_matching, la_mod.linear_sum_assignment
    calls = []

    def mcm(distance_metric, max_distance, tracks, detections, track_indices=None, detection_indices=None):
        cur = {}
        calls.append(cur)

        def metric(*a):
            out = distance_metric(*a)
            cur["raw"] = np.array(out, dtype=np.float64)
            return out
        return real_mcm(metric, max_distance, tracks, detections, track_indices, detection_indices)

    def lsa(cost):
        calls[-1]["clamped"] = np.array(cost, dtype=np.float64)
        return real_lsa(cost)

    la_mod.min_cost_matching, la_mod.linear_sum_assignment = mcm, lsa
    out = {}
    try:
        img = np.zeros((*shape, 3), dtype=np.uint8)
        trk = StrongSort(reid_model=None, **kw)
        trk.cmc = ref_harness.IdentityCMC()
        for t, (d, e) in enumerate(factory()):
            calls.clear()
            trk.update(d.copy(), img, e.copy())
            assert len(calls) == 2, (name, t, len(calls))
            if t in keep:
                for s, cur in enumerate(calls):
                    for k, v in cur.items():
                        out[f"{name}_f{t}_s{s}_{k}"] = v[::step]
                    out[f"{name}_f{t}_s{s}_shape"] = np.array(cur["raw"].shape if "raw" in cur else (0, 0), dtype=np.int32)
            print(f"  {name} frame {t}", flush=True)
    finally:
        la_mod.min_cost_matching, la_mod.linear_sum_assignment = real_mcm, real_lsa
    return out


def main():
    import logging

    logging.disable(logging.CRITICAL)
    which = sys.argv[1:] or ["botsort", "docs", "ss"]
    path = OUT / "cost_golden.npz"
    allv = dict(np.load(path)) if path.exists() else {}
    for group, cases, fn in (("botsort", CASES, record_case), ("docs", DOCS_CASES, record_docs_case), ("ss", SS_CASES, record_ss_case)):
        if group not in which:
            continue
        for name in cases:
            allv = {k: v for k, v in allv.items() if not k.startswith(name + "_f")}
            got = fn(name)
            allv.update(got)
            print(f"{name}: {len(got)} arrays, {sum(v.size for v in got.values())} values", flush=True)
    np.savez_compressed(path, **allv)
    print("wrote", path, path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
