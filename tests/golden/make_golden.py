"""Generate the committed golden fixtures by running the REAL reference
(/root/reference, imported under the stand-ins documented in oracle/ref_harness.py).

Run in the build container only:  python tests/golden/make_golden.py
Outputs (small, committed):
  tests/golden/botsort_golden.npz  per-frame output rows of the reference BotSort on seeded
                                   scenarios (inputs are regenerated from the seed by the tests)
  tests/golden/botsort_warp_golden.npz   the same with a scheduled camera-motion warp (STrack.multi_gmc)
  tests/golden/deepocsort_golden.npz     per-frame rows + final Kalman state of the reference DeepOcSort
  tests/golden/strongsort_golden.npz     the same for the reference StrongSort (identity camera motion)
  tests/golden/mot17_golden.npz          the reference BotSort / ByteTrack / DeepOcSort / OcSort / StrongSort replayed over the reference's MOT17-mini det.txt files
  tests/golden/reid_golden.npz     a seeded OSNet-x0.25 state_dict, test boxes, and the reference
                                   BaseModelBackend.get_features / get_crops results for them
The lap / cv2 stand-ins make those two boundaries "parity unpinned" (see oracle/__init__.py).
"""
from __future__ import annotations

import logging
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from boxmot_amd.reid_weights import random_osnet_state_dict  # noqa: E402
from boxmot_amd.scenario import Scenario, camera_warps, stress_frames  # noqa: E402
from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS  # noqa: E402
from oracle import ref_harness  # noqa: E402
from tests.common import mot17_embeddings  # noqa: E402

OUT = Path(__file__).resolve().parent
YAML = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method", "with_reid")}

CASES = {
    # name: (frames factory, image shape, tracker kwargs)
    "stress_default": (lambda: stress_frames(150, seed=7), (480, 640), {}),
    "stress_yaml": (lambda: stress_frames(150, seed=7), (480, 640), YAML),
    "stress_short_buffer": (lambda: stress_frames(150, seed=11), (480, 640), dict(track_buffer=5, removed_stracks_buffer=3)),
    "c2_yaml": (lambda: Scenario(64, 256, random_image=False).frames(40), (1080, 1920), YAML),
    "c2_default": (lambda: Scenario(64, 256, random_image=False).frames(40), (1080, 1920), {}),
}
REID_BOXES = np.array([
    [30.2, 40.7, 90.1, 200.3], [-10.0, -5.0, 60.0, 120.0], [1800.5, 900.5, 1990.0, 1100.0],
    [100.0, 100.0, 100.0, 150.0], [400.0, 300.0, 656.0, 812.0], [10.5, 10.5, 138.5, 266.5],
    [700.49, 200.5, 752.51, 254.5], [5.0, 5.0, 8.0, 9.0],
], dtype=np.float32)


class ScheduledCMC:
    """Stands where the reference's ECC/SOF object stands (botsort.py:116-117): returns a scheduled warp."""

    def __init__(self, warps):
        self.warps, self.k = warps, 0

    def apply(self, img, dets):
        self.k += 1
        return self.warps[self.k - 1]


def warp_golden():
    """botsort_warp_golden.npz: the reference BotSort driven with a scheduled camera-motion warp."""
    logging.disable(logging.CRITICAL)
    BotSort = ref_harness.load_botsort()
    out = {}
    for name, seed, kw in (("warp_default", 7, {}), ("warp_yaml", 11, YAML)):
        frames = stress_frames(120, seed=seed)
        img = np.zeros((480, 640, 3), dtype=np.uint8)
        trk = BotSort(reid_model=None, with_reid=True, use_cmc=False, **kw)
        trk.cmc = ScheduledCMC(camera_warps(len(frames), seed=seed))
        rows, counts = [], []
        for dets, embs in frames:
            r = np.asarray(trk.update(dets.copy(), img, embs.copy()), dtype=np.float32).reshape(-1, 8)
            rows.append(r)
            counts.append(len(r))
        out[name + "_rows"] = np.concatenate(rows, 0)
        out[name + "_counts"] = np.array(counts, dtype=np.int32)
        act = trk.active_tracks
        out[name + "_final_mean"] = np.array([t.mean for t in act], dtype=np.float64).reshape(len(act), 8)
        out[name + "_final_cov"] = np.array([t.covariance for t in act], dtype=np.float64).reshape(len(act), 8, 8)
        out[name + "_final_ids"] = np.array([t.id for t in act], dtype=np.int64)
        print(name, "rows", sum(counts))
    np.savez_compressed(OUT / "botsort_warp_golden.npz", **out)


DEEPOCSORT_CASES = {
    # name: (frames factory, image shape, tracker kwargs)
    "docs_stress_default": (lambda: stress_frames(150, seed=7), (480, 640), {}),
    "docs_stress_short": (lambda: stress_frames(150, seed=11), (480, 640), dict(max_age=5, min_hits=1)),
    "docs_stress_awoff": (lambda: stress_frames(120, seed=3), (480, 640), dict(aw_off=True, inertia=0.4, w_association_emb=0.75)),
    "docs_stress_noemb": (lambda: stress_frames(120, seed=5), (480, 640), dict(embedding_off=True)),
    "docs_c2": (lambda: Scenario(64, 256, emb_dim=128, random_image=False).frames(30), (1080, 1920), {}),
}


def deepocsort_golden():
    """deepocsort_golden.npz: the reference DeepOcSort (cmc_off=True, embeddings supplied) on seeded scenarios."""
    logging.disable(logging.CRITICAL)
    DeepOcSort = ref_harness.load_deepocsort()
    out = {}
    for name, (make, hw, kw) in DEEPOCSORT_CASES.items():
        frames = make()
        img = np.zeros((hw[0], hw[1], 3), dtype=np.uint8)
        trk = DeepOcSort(reid_model=None, cmc_off=True, **kw)
        rows, counts = [], []
        for dets, embs in frames:
            r = np.asarray(trk.update(dets.copy(), img, embs.copy()), dtype=np.float32).reshape(-1, 8)
            rows.append(r)
            counts.append(len(r))
        out[name + "_rows"] = np.concatenate(rows, 0)
        out[name + "_counts"] = np.array(counts, dtype=np.int32)
        act = trk.active_tracks
        out[name + "_final_x"] = np.array([t.kf.x[:, 0] for t in act], dtype=np.float64).reshape(len(act), 7)
        out[name + "_final_P"] = np.array([t.kf.P for t in act], dtype=np.float64).reshape(len(act), 7, 7)
        out[name + "_final_ids"] = np.array([t.id for t in act], dtype=np.int64)
        print(name, "frames", len(frames), "rows", sum(counts), "tracks", len(act))
    # camera-motion correction: cmc_off=False with a scheduled stand-in for the reference's SOF object (deepocsort.py:296)
    for name, seed, kw in (("docs_warp_default", 7, {}), ("docs_warp_short", 11, dict(max_age=6, min_hits=1))):
        frames = stress_frames(120, seed=seed)
        img = np.zeros((480, 640, 3), dtype=np.uint8)
        trk = DeepOcSort(reid_model=None, cmc_off=False, **kw)
        trk.cmc = ScheduledCMC(camera_warps(len(frames), seed=seed))
        rows, counts = [], []
        for dets, embs in frames:
            r = np.asarray(trk.update(dets.copy(), img, embs.copy()), dtype=np.float32).reshape(-1, 8)
            rows.append(r)
            counts.append(len(r))
        out[name + "_rows"] = np.concatenate(rows, 0)
        out[name + "_counts"] = np.array(counts, dtype=np.int32)
        out[name + "_final_ids"] = np.array([t.id for t in trk.active_tracks], dtype=np.int64)
        print(name, "rows", sum(counts))
    np.savez_compressed(OUT / "deepocsort_golden.npz", **out)


STRONGSORT_CASES = {
    "ss_stress_default": (lambda: stress_frames(150, seed=7), (480, 640), {}),
    "ss_stress_short": (lambda: stress_frames(150, seed=11), (480, 640), dict(max_age=5, n_init=1, nn_budget=3)),
    "ss_stress_loose": (lambda: stress_frames(120, seed=3), (480, 640),
                        dict(max_cos_dist=0.4, max_iou_dist=0.9, mc_lambda=0.9, ema_alpha=0.8, min_conf=0.3)),
    "ss_c2": (lambda: Scenario(64, 256, emb_dim=128, random_image=False).frames(30), (1080, 1920), {}),
}


def strongsort_golden():
    """strongsort_golden.npz: the reference StrongSort with an identity camera-motion object (its ECC estimator is
    unconditional, strongsort.py:67,83-86), embeddings supplied, on seeded scenarios."""
    logging.disable(logging.CRITICAL)
    StrongSort = ref_harness.load_strongsort()
    out = {}
    for name, (make, hw, kw) in STRONGSORT_CASES.items():
        frames = make()
        img = np.zeros((hw[0], hw[1], 3), dtype=np.uint8)
        trk = StrongSort(reid_model=None, **kw)
        trk.cmc = ref_harness.IdentityCMC()
        rows, counts = [], []
        for dets, embs in frames:
            r = np.asarray(trk.update(dets.copy(), img, embs.copy()), dtype=np.float32).reshape(-1, 8)
            rows.append(r)
            counts.append(len(r))
        out[name + "_rows"] = np.concatenate(rows, 0)
        out[name + "_counts"] = np.array(counts, dtype=np.int32)
        act = trk.tracker.tracks
        out[name + "_final_mean"] = np.array([t.mean for t in act], dtype=np.float64).reshape(len(act), 8)
        out[name + "_final_cov"] = np.array([t.covariance for t in act], dtype=np.float64).reshape(len(act), 8, 8)
        out[name + "_final_ids"] = np.array([t.id for t in act], dtype=np.int64)
        print(name, "frames", len(frames), "rows", sum(counts), "tracks", len(act))
    np.savez_compressed(OUT / "strongsort_golden.npz", **out)


MOT17_FRAMES = 200


def mot17_inputs(seq: str):
    """First MOT17_FRAMES frames of the reference's own detection fixture (assets/MOT17-mini/train/<seq>/det/det.txt, public
    FRCNN detections with their real confidences 0.05..1) as dets_n_embs rows [frame, x1, y1, x2, y2, conf, cls], plus a
    deterministic appearance vector per detection: sines/cosines of the box centre at four spatial frequencies (neighbouring
    people look alike, as real embeddings of a crowd do) with a little seeded noise."""
    path = Path("/root/reference/assets/MOT17-mini/train") / seq / "det" / "det.txt"
    d = np.loadtxt(path, delimiter=",")
    d = d[d[:, 0] <= MOT17_FRAMES]
    d = d[np.argsort(d[:, 0], kind="stable")]
    rows = np.zeros((len(d), 7), dtype=np.float32)
    rows[:, 0] = d[:, 0]
    rows[:, 1:3] = d[:, 2:4]
    rows[:, 3:5] = d[:, 2:4] + d[:, 4:6]
    rows[:, 5] = d[:, 6]
    return rows, mot17_embeddings(rows)


def mot17_golden():
    """mot17_golden.npz: the reference BotSort (with and without appearance), DeepOcSort and StrongSort replayed over the
    reference's MOT17-mini detection files the way process_sequence does (frames without detections are skipped)."""
    logging.disable(logging.CRITICAL)
    BotSort, DeepOcSort, StrongSort = ref_harness.load_botsort(), ref_harness.load_deepocsort(), ref_harness.load_strongsort()
    OcSort = ref_harness.load_ocsort()
    img = np.zeros((1080, 1920, 3), dtype=np.uint8)

    def byte():
        return ref_harness.load_bytetrack()()            # rewinds the process-global id counter first

    out = {}

    def strong():
        t = StrongSort(reid_model=None)
        t.cmc = ref_harness.IdentityCMC()
        return t

    makers = {
        "botsort": lambda: BotSort(reid_model=None, with_reid=True, use_cmc=False, **YAML),
        "botsort_noreid": lambda: BotSort(reid_model=None, with_reid=False, use_cmc=False, **YAML),
        "deepocsort": lambda: DeepOcSort(reid_model=None, cmc_off=True),
        "strongsort": strong,
        "ocsort": lambda: OcSort(),
        "ocsort_yaml": lambda: OcSort(det_thresh=0.6, inertia=0.1),       # configs/trackers/ocsort.yaml defaults
        "bytetrack": byte,
        "ocsort_byte": lambda: OcSort(use_byte=True),
    }
    for seq in ("MOT17-02-FRCNN", "MOT17-04-FRCNN"):
        rows, emb = mot17_inputs(seq)
        out[seq + "_dets"] = rows           # the embeddings are a function of these rows (tests/common.py mot17_embeddings)
        for name, mk in makers.items():
            trk = mk()
            res, counts = [], []
            for fid in range(1, MOT17_FRAMES + 1):
                m = rows[:, 0] == fid
                if not m.any():
                    counts.append(-1)
                    continue
                r = np.asarray(trk.update(rows[m, 1:].copy(), img, emb[m].copy()), dtype=np.float32).reshape(-1, 8)
                res.append(r)
                counts.append(len(r))
            out[f"{seq}_{name}_rows"] = np.concatenate(res, 0)
            out[f"{seq}_{name}_counts"] = np.array(counts, dtype=np.int32)
            print(seq, name, "dets", len(rows), "rows", len(out[f"{seq}_{name}_rows"]), "max id", int(out[f"{seq}_{name}_rows"][:, 4].max()))
    np.savez_compressed(OUT / "mot17_golden.npz", **out)


def main():
    logging.disable(logging.CRITICAL)
    BotSort = ref_harness.load_botsort()
    out = {}
    for name, (make, hw, kw) in CASES.items():
        frames = make()
        img = np.zeros((hw[0], hw[1], 3), dtype=np.uint8)
        trk = BotSort(reid_model=None, with_reid=True, use_cmc=False, **kw)
        rows, counts = [], []
        for dets, embs in frames:
            r = np.asarray(trk.update(dets.copy(), img, embs.copy()), dtype=np.float32).reshape(-1, 8)
            rows.append(r)
            counts.append(len(r))
        out[name + "_rows"] = np.concatenate(rows, 0)
        out[name + "_counts"] = np.array(counts, dtype=np.int32)
        act = trk.active_tracks
        out[name + "_final_mean"] = np.array([t.mean for t in act], dtype=np.float64).reshape(len(act), 8)
        out[name + "_final_cov"] = np.array([t.covariance for t in act], dtype=np.float64).reshape(len(act), 8, 8)
        out[name + "_final_ids"] = np.array([t.id for t in act], dtype=np.int64)
        print(name, "frames", len(frames), "rows", sum(counts))
    np.savez_compressed(OUT / "botsort_golden.npz", **out)

    import torch

    sd = random_osnet_state_dict("osnet_x0_25", seed=1234)
    mod = ref_harness.load_osnet_module()
    model = mod.osnet_x0_25(num_classes=1041, pretrained=False).eval()
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("classifier") for k in missing.missing_keys), missing
    img = np.random.default_rng(99).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
    ref = ref_harness.RefReID(model)
    feats = ref.get_features(REID_BOXES, img)
    crops = ref.get_crops(REID_BOXES, img).numpy()
    blob = {("sd/" + k): v.numpy() for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    np.savez_compressed(OUT / "reid_golden.npz", boxes=REID_BOXES, feats=feats.astype(np.float32),
                        crop0=crops[0], crop_sums=crops.reshape(len(crops), -1).astype(np.float64).sum(1),
                        image_seed=np.array(99), **blob)
    print("reid feats", feats.shape, "crop sums", crops.reshape(len(crops), -1).sum(1)[:3])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "warp":
        warp_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "deepocsort":
        deepocsort_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "strongsort":
        strongsort_golden()
        mot17_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "mot17":
        mot17_golden()
    else:
        main()
        warp_golden()
        deepocsort_golden()
        strongsort_golden()
        mot17_golden()
