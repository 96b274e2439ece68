"""Golden rows at the sizes of BASELINE.json's large configurations, from the REAL reference classes
(/root/reference imported under the stand-ins of oracle/ref_harness.py).  Build container only:

    python tests/golden/make_config_golden.py c5 [frames]     # StrongSORT, 256 dets x 1024 tracks x 1280-d, ~17 s per frame
    python tests/golden/make_config_golden.py c2 [frames]     # BoT-SORT, 64 dets x 256 tracks, YAML defaults, embeddings supplied
    python tests/golden/make_config_golden.py c2reid init|calib [frames]
    python tests/golden/make_config_golden.py c3reid [frames] [init|calib]
                                                              # the same with ReID INSIDE update: the reference BotSort asks the
                                                              # reference OSNet-x0.25 (random-init / BN-calibrated weights) per frame
    python tests/golden/make_config_golden.py c3 [frames]     # DeepOCSORT, 128 dets x 512 tracks x 512-d, embeddings supplied
    python tests/golden/make_config_golden.py c3reid [frames] # the same with ReID inside update: reference DeepOcSort + reference OSNet-x1.0
    python tests/golden/make_config_golden.py c5reid [frames] # StrongSORT + the reference CLIP-ReID ViT-B/16 modules inside update, 4K frame

The CPU reference needs about a quarter of a minute per frame at config 5 (a Python loop per track), far too slow to run beside
the GPU test, so its per-frame output rows are committed instead (tests/golden/config5_strongsort_golden.npz: ids / det
indices as int32, boxes and confidences as fp32; the inputs are regenerated from the seed by the test).  StrongSORT's
assignment is SciPy's own (no stand-in on its path), so these rows are reference-exact.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from boxmot_amd.scenario import Scenario  # noqa: E402
from boxmot_amd.tracker_zoo import BOTSORT_YAML_DEFAULTS  # noqa: E402
from oracle import ref_harness  # noqa: E402

OUT = Path(__file__).resolve().parent


def _save(name, rows_per_frame, extra):
    counts = np.array([len(r) for r in rows_per_frame], dtype=np.int32)
    rows = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1, 8) for r in rows_per_frame]) if counts.sum() else np.zeros((0, 8))
    np.savez_compressed(OUT / name, counts=counts, boxes=rows[:, :4].astype(np.float32), ids=rows[:, 4].astype(np.int32),
                        conf=rows[:, 5].astype(np.float32), cls=rows[:, 6].astype(np.int32), det_ind=rows[:, 7].astype(np.int32),
                        **extra)


def config5(frames: int):
    StrongSort = ref_harness.load_strongsort()
    sc = Scenario(256, 1024, emb_dim=1280, random_image=False)
    img = np.zeros((2160, 3840, 3), dtype=np.uint8)
    trk = StrongSort(reid_model=None)                 # constructor defaults: n_init 3, nn_budget 100, max_age 30
    trk.cmc = ref_harness.IdentityCMC()               # stands where its unconditional ECC object stands (strongsort.py:67)
    out = []
    t0 = time.time()
    for t in range(frames):
        d, e = sc.frame(t)
        r = np.asarray(trk.update(d, img, e.copy()), dtype=np.float64).reshape(-1, 8)
        out.append(r)
        if t % 10 == 0:
            print(f"c5 frame {t}: {len(r)} rows, {time.time() - t0:.0f} s", flush=True)
            _save("config5_strongsort_golden.npz", out, dict(frames=np.int32(len(out))))
    _save("config5_strongsort_golden.npz", out, dict(frames=np.int32(len(out))))


def config2(frames: int):
    BotSort = ref_harness.load_botsort()
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    sc = Scenario(64, 256, emb_dim=512, random_image=False)
    img = np.zeros((1080, 1920, 3), dtype=np.uint8)
    trk = BotSort(reid_model=None, use_cmc=False, **kw)
    out = []
    for t in range(frames):
        d, e = sc.frame(t)
        out.append(np.asarray(trk.update(d, img, e.copy()), dtype=np.float64).reshape(-1, 8))
    _save("config2_botsort_golden.npz", out, dict(frames=np.int32(frames)))


def config2_reid(weights: str, frames: int):
    """BASELINE.json config 2 exactly as bench.py runs it (64 dets x 256 tracks, 1080p random frame of stream 0, YAML
    defaults, use_cmc=False, embs=None -> ReID inside update), on the reference classes: BotSort + BaseModelBackend
    get_features + OSNet-x0.25 (cv2 / lap stand-ins as documented in oracle/ref_harness.py)."""
    import torch

    from boxmot_amd.reid_weights import random_osnet_state_dict, reference_init_state_dict

    torch.set_num_threads(8)
    sd = reference_init_state_dict("osnet_x0_25", seed=0) if weights == "init" else random_osnet_state_dict("osnet_x0_25", seed=0)
    mod = ref_harness.load_osnet_module()
    model = mod.osnet_x0_25(num_classes=1041, pretrained=False).eval()
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("classifier") for k in missing.missing_keys), missing
    BotSort = ref_harness.load_botsort()
    kw = {k: v for k, v in BOTSORT_YAML_DEFAULTS.items() if k not in ("use_cmc", "cmc_method")}
    sc = Scenario(64, 256, emb_dim=512, stream=0, random_image=True)
    trk = BotSort(reid_model=ref_harness.RefReID(model), use_cmc=False, **kw)
    out = []
    t0 = time.time()
    for t in range(frames):
        d, _ = sc.frame(t, with_embs=False)
        out.append(np.asarray(trk.update(d, sc.image), dtype=np.float64).reshape(-1, 8))
        if t % 20 == 0:
            print(f"c2reid[{weights}] frame {t}: {len(out[-1])} rows, {time.time() - t0:.0f} s", flush=True)
    _save(f"config2_reid_{weights}_golden.npz", out, dict(frames=np.int32(frames)))


def config3(frames: int):
    """BASELINE.json config 3's tracker (DeepOCSORT, 128 dets x 512 tracks, 512-d embeddings supplied) on the reference class;
    its assignment goes through the lap stand-in (oracle/lapjv.c: the Jonker-Volgenant code restated)."""
    DeepOcSort = ref_harness.load_deepocsort()
    sc = Scenario(128, 512, emb_dim=512, random_image=False)
    img = np.zeros((1080, 1920, 3), dtype=np.uint8)
    trk = DeepOcSort(reid_model=None, cmc_off=True)
    out = []
    t0 = time.time()
    for t in range(frames):
        d, e = sc.frame(t)
        out.append(np.asarray(trk.update(d, img, e.copy()), dtype=np.float64).reshape(-1, 8))
        if t % 20 == 0:
            print(f"c3 frame {t}: {len(out[-1])} rows, {time.time() - t0:.0f} s", flush=True)
    _save("config3_deepocsort_golden.npz", out, dict(frames=np.int32(frames)))


def config3_reid(frames: int, weights: str = "init"):
    """Config 3 as tools/config_bench.py runs it: reference DeepOcSort (cmc_off) asking the reference OSNet-x1.0 (random init, seed 0;
    `calib`: BatchNorm-calibrated random weights, the case that tells fp32-grade kernels from fp16 operands) for every detection
    above det_thresh, stream 0's random 1080p frame."""
    import torch

    from boxmot_amd.reid_weights import random_osnet_state_dict, reference_init_state_dict

    torch.set_num_threads(8)
    sd = reference_init_state_dict("osnet_x1_0", seed=0) if weights == "init" else random_osnet_state_dict("osnet_x1_0", seed=0)
    fname = "config3_reid_golden.npz" if weights == "init" else "config3_reid_calib_golden.npz"
    mod = ref_harness.load_osnet_module()
    model = mod.osnet_x1_0(num_classes=1041, pretrained=False).eval()
    missing = model.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("classifier") for k in missing.missing_keys), missing
    DeepOcSort = ref_harness.load_deepocsort()
    sc = Scenario(128, 512, width=1920, height=1080, emb_dim=8, stream=0, random_image=True)
    trk = DeepOcSort(reid_model=ref_harness.RefReID(model), cmc_off=True)
    out = []
    t0 = time.time()
    for t in range(frames):
        d, _ = sc.frame(t, with_embs=False)
        out.append(np.asarray(trk.update(d, sc.image), dtype=np.float64).reshape(-1, 8))
        if t % 5 == 0:
            print(f"c3reid frame {t}: {len(out[-1])} rows, {time.time() - t0:.0f} s", flush=True)
            _save(fname, out, dict(frames=np.int32(len(out))))
    _save(fname, out, dict(frames=np.int32(frames)))


class _RefClipReid:
    """The reference's CLIP-ReID image path (make_model.py:95-139, NECK_FEAT "after") assembled from its own modules:
    VisionTransformer (clip/model.py) + the two BatchNorm necks, evaluated in fp32 on the CPU."""

    def __init__(self, sd):
        import importlib.util

        import torch
        ref_harness.install_standins()               # clip/model.py imports boxmot.utils.logger
        spec = importlib.util.spec_from_file_location("_ref_clip_model", ref_harness.REFERENCE_ROOT / "boxmot/reid/backbones/clip/clip/model.py")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        self.vt = m.VisionTransformer(h_resolution=16, w_resolution=8, patch_size=16, stride_size=16, width=768, layers=12, heads=12,
                                      output_dim=512).eval()
        self.vt.load_state_dict({k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}, strict=True)
        self.bn, self.bnp = torch.nn.BatchNorm1d(768).eval(), torch.nn.BatchNorm1d(512).eval()
        self.bn.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("bottleneck.")}, strict=False)
        self.bnp.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("bottleneck_proj.")}, strict=False)

    def __call__(self, x):
        import torch
        outs = []
        with torch.no_grad():
            for i in range(0, len(x), 64):
                _, x12, xproj = self.vt(x[i:i + 64])
                outs.append(torch.cat([self.bn(x12[:, 0]), self.bnp(xproj[:, 0])], dim=1))
        return torch.cat(outs)


def config5_reid(frames: int):
    """Config 5 as tools/config_bench.py runs it (without camera motion: identity CMC): reference StrongSort asking the reference
    CLIP-ReID modules (random weights, seed 0; crops normalised with mean = std = 0.5, base_backend.py:50-54), stream 0's random
    4K frame, 256 dets x 1024 tracks."""
    import torch

    from boxmot_amd.clip_weights import random_clipreid_state_dict

    torch.set_num_threads(8)
    sd = random_clipreid_state_dict(0)
    reid = ref_harness.RefReID(_RefClipReid(sd))
    reid._b.mean_array = torch.tensor([0.5, 0.5, 0.5]).view(1, 3, 1, 1)
    reid._b.std_array = torch.tensor([0.5, 0.5, 0.5]).view(1, 3, 1, 1)
    StrongSort = ref_harness.load_strongsort()
    sc = Scenario(256, 1024, width=3840, height=2160, emb_dim=8, stream=0, random_image=True)
    trk = StrongSort(reid_model=reid)
    trk.cmc = ref_harness.IdentityCMC()
    out = []
    t0 = time.time()
    for t in range(frames):
        d, _ = sc.frame(t, with_embs=False)
        out.append(np.asarray(trk.update(d, sc.image), dtype=np.float64).reshape(-1, 8))
        print(f"c5reid frame {t}: {len(out[-1])} rows, {time.time() - t0:.0f} s", flush=True)
        if t % 4 == 0:
            _save("config5_reid_golden.npz", out, dict(frames=np.int32(len(out))))
    _save("config5_reid_golden.npz", out, dict(frames=np.int32(frames)))


def mot17_mini_gt():
    """tests/golden/mot17_mini_gt.npz: the ground-truth rows of the reference's MOT17-mini fixture (assets/MOT17-mini/train/
    <seq>/gt/gt.txt -- 4 and 8 annotated frames) for the metric tests (boxmot_amd.metrics)."""
    out = {}
    for seq in ("MOT17-02-FRCNN", "MOT17-04-FRCNN"):
        out[seq] = np.loadtxt(ref_harness.REFERENCE_ROOT / "assets" / "MOT17-mini" / "train" / seq / "gt" / "gt.txt", delimiter=",").astype(np.float32)
    np.savez_compressed(OUT / "mot17_mini_gt.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1]
    if which == "gt":
        mot17_mini_gt()
    elif which == "c2reid":
        config2_reid(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 240)
    elif which == "c3reid":
        config3_reid(int(sys.argv[2]) if len(sys.argv) > 2 else 64, sys.argv[3] if len(sys.argv) > 3 else "init")
    elif which == "c5reid":
        config5_reid(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
    else:
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
        {"c5": config5, "c2": config2, "c3": config3}[which](n)
