"""GPU parity tests for the ReID path (crop -> resize -> normalise -> OSNet -> L2) through the C ABI."""
import numpy as np
import pytest

from common import GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-3   # BASELINE.json north_star: embeddings within 1e-3 (fp32 reference)


def _golden():
    import torch
    g = np.load(GOLDEN / "reid_golden.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    img = np.random.default_rng(int(g["image_seed"])).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
    return g, sd, img


def test_crops_bit_exact_vs_oracle_and_reference():
    from boxmot_amd.reid import HipReID
    from oracle.crops import get_crops
    g, sd, img = _golden()
    reid = HipReID(sd, max_crops=64)
    boxes = g["boxes"]
    got = reid.get_crops(boxes, img)
    want = get_crops(boxes, img)
    assert got.shape == want.shape == (len(boxes), 3, 256, 128)
    assert np.array_equal(got, want)                     # integer resize + table lookup: bit-exact
    assert np.array_equal(got[0], g["crop0"])            # reference get_crops (golden)
    # ragged / degenerate boxes: outside the frame, 1-pixel, exact 2x shrink, identity size, huge
    rng = np.random.default_rng(0)
    extra = np.array([[-50, -50, -10, -10], [0, 0, 1, 1], [100, 100, 356, 612], [200, 50, 328, 306],
                      [0, 0, 1920, 1080], [1919.4, 1079.4, 1925, 1085], [10.5, 20.5, 11.5, 300.5]], dtype=np.float32)
    rand = np.stack([rng.uniform(-40, 1900, 40), rng.uniform(-40, 1000, 40), np.zeros(40), np.zeros(40)], 1)
    rand[:, 2] = rand[:, 0] + rng.uniform(1, 300, 40)
    rand[:, 3] = rand[:, 1] + rng.uniform(1, 500, 40)
    allb = np.concatenate([extra, rand.astype(np.float32)])
    assert np.array_equal(reid.get_crops(allb, img), get_crops(allb, img))
    reid.close()


def test_features_vs_oracle_and_reference_golden():
    from boxmot_amd.reid import HipReID
    from oracle.osnet import OracleReID
    g, sd, img = _golden()
    reid = HipReID(sd, max_crops=16)
    got = reid.get_features(g["boxes"], img)
    assert got.dtype == np.float32 and got.shape == (len(g["boxes"]), 512)
    assert np.abs(got - g["feats"]).max() < TOL                       # reference (golden)
    want = OracleReID(sd).get_features(g["boxes"], img)
    err = np.abs(got - want).max()
    assert err < TOL, err
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    cos = (got * want).sum(1)
    assert cos.min() > 0.9999                                          # cf. test_reid_capi.py:158-171 (> 0.99)
    # chunking over max_crops and the empty case
    many = np.tile(g["boxes"], (5, 1))
    f2 = reid.get_features(many, img)
    assert np.array_equal(f2[:8], f2[8:16]) and np.abs(f2[:8] - got).max() == 0
    assert reid.get_features(np.empty((0, 4), dtype=np.float32), img).shape == (0, 512)
    reid.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_botsort_with_reid_in_the_loop_matches_oracle_ids(mode):
    """embs=None: the tracker asks the ReID model itself (botsort.py:191-192).  Both kernel families, BN-calibrated weights."""
    from boxmot_amd.botsort import BotSort
    from boxmot_amd.reid import HipReID
    from boxmot_amd.scenario import Scenario
    from oracle.botsort import BotSortOracle
    from oracle.osnet import OracleReID
    _, sd, _ = _golden()
    sc = Scenario(16, 32, width=960, height=540, random_image=True)
    reid = HipReID(sd, max_crops=64, mode=mode)
    trk = BotSort(reid_model=reid, use_cmc=False, max_tracks=128, max_dets=64)
    orc = BotSortOracle(reid=OracleReID(sd))
    for t in range(10):
        dets, _ = sc.frame(t)
        got = np.asarray(trk.update(dets, sc.image))
        want = orc.update(dets, sc.image)
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
        assert np.allclose(got[:, :4], want[:, :4], atol=1e-3)
    assert trk.get_last_track_time_ms() > 0
    trk.close()
    reid.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_device_resident_multistream_reid_step(mode):
    """step_device with frames resident on the GPU: ReID crop list, OSNet and the tracker step all
    run without host buffers; ids must equal the per-stream oracle.  Both kernel families, BN-calibrated weights."""
    import torch

    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    from oracle.botsort import BotSortOracle
    from oracle.osnet import OracleReID
    _, sd, _ = _golden()
    S, nd = 3, 32
    scs = [Scenario(12, 24, width=640, height=480, random_image=True, stream=s) for s in range(S)]
    ms = MultiStreamBotSort(S, max_tracks=64, max_dets=nd, emb_dim=512, reid_weights=sd)
    ms.set_reid_mode(mode)
    orcs = [BotSortOracle(reid=OracleReID(sd)) for _ in range(S)]
    dev = torch.device("cuda:0")
    frames = torch.stack([torch.from_numpy(sc.image) for sc in scs]).to(dev)
    ptrs = torch.tensor([frames[s].data_ptr() for s in range(S)], dtype=torch.int64, device=dev)
    d_dets = torch.zeros((S, nd, 6), dtype=torch.float32, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    d_out = torch.zeros((S, nd, 8), dtype=torch.float32, device=dev)
    d_out_n = torch.zeros(S, dtype=torch.int32, device=dev)
    for t in range(8):
        per = [sc.frame(t)[0] for sc in scs]
        for s in range(S):
            d_dets[s, : len(per[s])] = torch.from_numpy(per[s]).to(dev)
            d_n[s] = len(per[s])
        torch.cuda.synchronize()
        ms.step_device(d_dets.data_ptr(), d_n.data_ptr(), None, ptrs.data_ptr(), 480, 640, d_out.data_ptr(), d_out_n.data_ptr())
        ms.synchronize()
        out, cnt = d_out.cpu().numpy(), d_out_n.cpu().numpy()
        for s in range(S):
            want = orcs[s].update(per[s], scs[s].image)
            got = out[s, : cnt[s]]
            assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), (s, t)
    assert ms.status().tolist() == [0] * S
    ms.close()


def test_fused_fp16_mode_reference_init_within_tolerance():
    """mode 1 (fused fp16 MFMA kernels) on random-init weights of the architecture (the benchmark's weights):
    embeddings within 1e-3 of the fp32 oracle; mode 0 and mode 1 agree on the same handle."""
    from boxmot_amd.reid import MODE_FP16_FUSED, MODE_FP32_LAYERWISE, HipReID
    from boxmot_amd.reid_weights import reference_init_state_dict
    from oracle.osnet import OracleReID
    g, _, img = _golden()
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    rng = np.random.default_rng(1)
    boxes = np.stack([rng.uniform(0, 1800, 48), rng.uniform(0, 900, 48), np.zeros(48), np.zeros(48)], 1).astype(np.float32)
    boxes[:, 2] = boxes[:, 0] + rng.uniform(20, 120, 48)
    boxes[:, 3] = boxes[:, 1] + rng.uniform(40, 180, 48)
    boxes = np.concatenate([g["boxes"], boxes]).astype(np.float32)
    want = OracleReID(sd).get_features(boxes, img)
    reid = HipReID(sd, max_crops=64, mode=MODE_FP16_FUSED)
    got16 = reid.get_features(boxes, img)
    err16 = np.abs(got16 - want).max()
    assert err16 < TOL, err16
    assert (got16 * want).sum(1).min() > 0.99999
    reid.set_mode(MODE_FP32_LAYERWISE)
    got32 = reid.get_features(boxes, img)
    assert np.abs(got32 - want).max() < 1e-4
    # crops are produced by the same integer pipeline in both modes
    assert np.array_equal(reid.get_crops(boxes[:4], img), __import__("oracle.crops", fromlist=["get_crops"]).get_crops(boxes[:4], img))
    reid.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fused_families_on_calibrated_weights(seed):
    """BatchNorm-calibrated random networks (non-trivial running statistics and affine terms; noise-amplifying by construction):
    the case that separates fp32-grade arithmetic from fp16 operands.  Asserted per seed, measured numbers printed:
    (a) mode 2 -- the fused fp32-grade family, the one bench.py reports -- is within north_star's 1e-3 of the fp32 oracle
        (measured ~1e-5), and so is mode 0 (per-layer fp32);
    (b) mode 1 (fp16 operands) is NOT required to meet 1e-3 here: it stays a unit-norm embedding with cosine > 0.999 to the
        oracle and within 2e-2 max-abs -- the error class of the reference's own half=True path (base_backend.py:162,185,223),
        printed beside it for comparison (it is sometimes above, sometimes below that path: no ordering is claimed)."""
    import torch

    from boxmot_amd.reid import MODE_FP16_FUSED, MODE_FP32_FUSED, MODE_FP32_LAYERWISE, HipReID
    from boxmot_amd.reid_weights import random_osnet_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import OracleReID, osnet_forward
    sd = random_osnet_state_dict("osnet_x0_25", seed=seed)
    img = np.random.default_rng(5).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
    rng = np.random.default_rng(1)
    n = 16
    boxes = np.stack([rng.uniform(0, 1800, n), rng.uniform(0, 900, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    boxes[:, 2] = boxes[:, 0] + rng.uniform(20, 120, n)
    boxes[:, 3] = boxes[:, 1] + rng.uniform(40, 180, n)
    want = OracleReID(sd).get_features(boxes, img)
    reid = HipReID(sd, max_crops=16, mode=MODE_FP32_FUSED)
    got_hp = reid.get_features(boxes, img)
    reid.set_mode(MODE_FP16_FUSED)
    got16 = reid.get_features(boxes, img)
    reid.set_mode(MODE_FP32_LAYERWISE)
    got32 = reid.get_features(boxes, img)
    reid.close()
    with torch.no_grad():       # the reference's half=True arithmetic: weights and crops in fp16, torch CPU
        sd16 = {k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}
        half = osnet_forward(sd16, torch.from_numpy(get_crops(boxes, img)).half()).float().numpy()
    half = half / np.linalg.norm(half, axis=1, keepdims=True)
    err_hp, err16, err32, err_half = (float(np.abs(x - want).max()) for x in (got_hp, got16, got32, half))
    print(f"calibrated seed {seed}: max|diff| vs fp32 oracle -- mode 2 (fused fp32-grade): {err_hp:.2e}, mode 0: {err32:.2e}, "
          f"mode 1 (fused fp16): {err16:.2e}, reference half path: {err_half:.2e}; min cosine mode 1: {(got16 * want).sum(1).min():.6f}")
    assert err_hp < TOL, err_hp
    assert err32 < TOL, err32
    assert np.allclose(np.linalg.norm(got_hp, axis=1), 1.0, atol=1e-5)
    assert err16 < 2e-2, err16
    assert np.allclose(np.linalg.norm(got16, axis=1), 1.0, atol=1e-3)
    assert (got16 * want).sum(1).min() > 0.999


def test_fused_fp32_grade_mode_many_crops_and_resize_pad():
    """mode 2 on more crops than one head workgroup, odd boxes (clipped, degenerate, identity- and 2x-sized), both preprocess
    modes: within 1e-3 of the oracle (measured ~1e-5)."""
    from boxmot_amd.reid import MODE_FP32_FUSED, HipReID
    from boxmot_amd.reid_weights import random_osnet_state_dict
    from oracle.osnet import OracleReID
    sd = random_osnet_state_dict("osnet_x0_25", seed=3)
    img = np.random.default_rng(9).integers(0, 255, (720, 1281, 3), dtype=np.uint8)
    rng = np.random.default_rng(2)
    n = 45
    boxes = np.stack([rng.uniform(-20, 1200, n), rng.uniform(-20, 650, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    boxes[:, 2] = boxes[:, 0] + rng.uniform(5, 200, n)
    boxes[:, 3] = boxes[:, 1] + rng.uniform(5, 300, n)
    boxes[0] = [100, 100, 228, 356]         # identity-sized
    boxes[1] = [10, 10, 266, 522]           # exact 2x
    boxes[2] = [50, 60, 50, 200]            # empty crop
    for pre in ("resize", "resize_pad"):
        want = OracleReID(sd, preprocess=pre).get_features(boxes, img)
        r = HipReID(sd, max_crops=64, mode=MODE_FP32_FUSED, preprocess=pre)
        got = r.get_features(boxes, img)
        r.close()
        err = float(np.abs(got - want).max())
        print(f"mode 2, {pre}: max|diff| {err:.2e}")
        assert err < TOL, (pre, err)


def test_botsort_multistream_fused_reid_ids_match_oracle():
    import torch

    from boxmot_amd.reid_weights import reference_init_state_dict
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.streams import MultiStreamBotSort
    from oracle.botsort import BotSortOracle
    from oracle.osnet import OracleReID
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    S, nd = 2, 32
    scs = [Scenario(12, 24, width=640, height=480, random_image=True, stream=s) for s in range(S)]
    ms = MultiStreamBotSort(S, max_tracks=64, max_dets=nd, emb_dim=512, reid_weights=sd)
    ms.set_reid_mode(1)
    orcs = [BotSortOracle(reid=OracleReID(sd)) for _ in range(S)]
    dev = torch.device("cuda:0")
    frames = torch.stack([torch.from_numpy(sc.image) for sc in scs]).to(dev)
    ptrs = torch.tensor([frames[s].data_ptr() for s in range(S)], dtype=torch.int64, device=dev)
    d_dets = torch.zeros((S, nd, 6), dtype=torch.float32, device=dev)
    d_n = torch.zeros(S, dtype=torch.int32, device=dev)
    d_out = torch.zeros((S, nd, 8), dtype=torch.float32, device=dev)
    d_out_n = torch.zeros(S, dtype=torch.int32, device=dev)
    for t in range(8):
        per = [sc.frame(t)[0] for sc in scs]
        for s in range(S):
            d_dets[s, : len(per[s])] = torch.from_numpy(per[s]).to(dev)
            d_n[s] = len(per[s])
        torch.cuda.synchronize()
        ms.step_device(d_dets.data_ptr(), d_n.data_ptr(), None, ptrs.data_ptr(), 480, 640, d_out.data_ptr(), d_out_n.data_ptr())
        ms.synchronize()
        out, cnt = d_out.cpu().numpy(), d_out_n.cpu().numpy()
        for s in range(S):
            want = orcs[s].update(per[s], scs[s].image)
            got = out[s, : cnt[s]]
            assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), (s, t)
    ms.close()


def test_resize_pad_preprocess_matches_oracle_in_both_modes():
    """reid_preprocess="resize_pad" (reid/core/preprocessing.py:21-45): crops bit-exact, features within tolerance."""
    import torch

    from boxmot_amd.reid import MODE_FP16_FUSED, MODE_FP32_LAYERWISE, HipReID
    from boxmot_amd.reid_weights import reference_init_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import OracleReID
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    img = np.random.default_rng(11).integers(0, 255, (480, 641, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [-20, -10, 40, 60], [100, 100, 100, 150], [10, 10, 138, 266], [0, 0, 641, 480],
                      [600.4, 430.2, 700, 500], [50, 20, 200, 330], [5, 5, 300, 40], [7, 3, 9, 400]], dtype=np.float32)
    want = OracleReID(sd, preprocess="resize_pad").get_features(boxes, img)
    for mode, tol in ((MODE_FP32_LAYERWISE, 2e-5), (MODE_FP16_FUSED, 1e-3)):
        r = HipReID(sd, mode=mode, preprocess="resize_pad")
        if mode == MODE_FP32_LAYERWISE:
            crops = r.get_crops(boxes, img)
            assert np.array_equal(np.asarray(crops), get_crops(boxes, img, preprocess="resize_pad"))
        assert np.abs(r.get_features(boxes, img) - want).max() < tol
    with pytest.raises(RuntimeError):
        HipReID(sd, preprocess="letterbox")


@pytest.mark.parametrize("cols", [5, 7, 9])
def test_oriented_box_crops_bit_exact_and_features_in_both_modes(cols):
    """Rows of 5 / 7 / 9 values are oriented boxes [cx, cy, w, h, angle, ...] (base_backend.py:119-122, 157): the rectified
    crop of _crop_obb (getRotationMatrix2D + warpAffine, base_backend.py:91-117) is sampled on the device -- crops bit for
    bit against the restated warpAffine, features within the ReID tolerance in the fp32 and the fused fp16 kernels, for
    both preprocess modes; more boxes than one engine chunk."""
    from boxmot_amd.reid import MODE_FP16_FUSED, MODE_FP32_LAYERWISE, HipReID
    from boxmot_amd.reid_weights import reference_init_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import OracleReID
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    rng = np.random.default_rng(21)
    img = rng.integers(0, 255, (480, 641, 3), dtype=np.uint8)
    fixed = np.array([[320, 240, 80, 160, 0.3], [10, 20, 60, 90, -1.2], [630, 470, 50, 120, 2.0], [300, 200, 33.4, 71.6, 0.77],
                      [200, 200, 128, 256, 0.0], [200.5, 100.5, 256, 512, 0.0], [100, 300, 0.2, 40, 0.5], [400, 100, 300.5, 20.5, 1.5708],
                      [-50, -60, 40, 40, 0.1]], dtype=np.float32)
    rand = np.stack([rng.uniform(-20, 660, 30), rng.uniform(-20, 500, 30), rng.uniform(2, 200, 30), rng.uniform(2, 400, 30),
                     rng.uniform(-np.pi, np.pi, 30)], 1).astype(np.float32)
    boxes = np.concatenate([fixed, rand])
    if cols > 5:
        boxes = np.concatenate([boxes, rng.uniform(0, 1, (len(boxes), cols - 5)).astype(np.float32)], 1)
    for pre in ("resize", "resize_pad"):
        want_crops = get_crops(boxes, img, preprocess=pre)
        want = OracleReID(sd, preprocess=pre).get_features(boxes, img)
        for mode, tol in ((MODE_FP32_LAYERWISE, 2e-5), (MODE_FP16_FUSED, 1e-3)):
            r = HipReID(sd, mode=mode, preprocess=pre, max_crops=16)
            if mode == MODE_FP32_LAYERWISE:
                got = np.concatenate([r.get_crops(boxes[i:i + 16], img) for i in range(0, len(boxes), 16)])
                assert np.array_equal(got, want_crops)
            err = np.abs(r.get_features(boxes, img) - want).max()
            assert err < tol, (pre, mode, err)
            # an axis-aligned call right after an oriented one does not see stale geometry
            aabb = np.array([[136, 72, 264, 328]], dtype=np.float32)
            assert np.abs(r.get_features(aabb, img) - OracleReID(sd, preprocess=pre).get_features(aabb, img)).max() < tol
            r.close()
