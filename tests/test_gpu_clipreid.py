"""GPU parity tests for the CLIP-ReID (ViT-B/16) backbone of BASELINE.json configuration 5, through the ReID C ABI:
crop -> resize -> normalise (mean = std = 0.5) -> ViT-B/16 -> BatchNorm necks -> concat (1280-d) -> L2, against
oracle/clipreid.py (pinned on the reference modules, tests/test_oracle_clipreid.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-3   # BASELINE.json north_star: embeddings within 1e-3 (fp32 reference)


def _boxes(n, seed=1, w=1900, h=1000):
    rng = np.random.default_rng(seed)
    b = np.stack([rng.uniform(0, w - 130, n), rng.uniform(0, h - 200, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    b[:, 2] = b[:, 0] + rng.uniform(20, 120, n)
    b[:, 3] = b[:, 1] + rng.uniform(40, 180, n)
    return b


@pytest.fixture(scope="module")
def vitb16():
    from boxmot_amd.clip_weights import pack_clipreid, random_clipreid_state_dict
    sd = random_clipreid_state_dict(0)                       # ViT-B/16: width 768, 12 layers, 12 heads, 512-d projection
    return sd, pack_clipreid(sd)


def test_vitb16_features_and_crops_vs_oracle(vitb16):
    from boxmot_amd.reid import HipReID
    from oracle.clipreid import OracleClipReID
    from oracle.crops import get_crops
    sd, blob = vitb16
    img = np.random.default_rng(5).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
    boxes = np.concatenate([_boxes(9), np.array([[-20, -10, 40, 60], [100, 100, 356, 612], [1900.4, 1000.2, 1990, 1200]], np.float32)])
    reid = HipReID(blob, max_crops=8)                        # 12 boxes through an 8-crop engine: two chunks
    assert reid.feature_dim == 1280
    got = reid.get_features(boxes, img)
    want = OracleClipReID(sd).get_features(boxes, img)
    err = float(np.abs(got - want).max())
    print(f"CLIP-ReID ViT-B/16 on device: max|diff| vs fp32 oracle = {err:.2e}, min cosine {(got * want).sum(1).min():.6f}")
    assert got.shape == (12, 1280) and err < TOL
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    # the crops are the integer resize + the "clip" normalisation table: bit-exact
    assert np.array_equal(reid.get_crops(boxes[:8], img), get_crops(boxes[:8], img, mean=(0.5,) * 3, std=(0.5,) * 3))
    assert reid.get_features(np.empty((0, 4), np.float32), img).shape == (0, 1280)
    reid.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_vitb16_features_on_gain_randomised_weights(seed):
    """The fp16 GEMM operands of the CLIP-ReID kernels on something harder than the initialisation-like set (round-4 review, Weak 5):
    LayerNorm gains U(0.4, 2.5) / biases N(0, 0.3) and neck BatchNorm variances U(0.02, 0.2) -- the tolerance is north_star's 1e-3 --
    and, characterised rather than promised, the same set with 2.5 x larger query / key projections (attention logits ~6 x larger):
    there a rounding error in a logit moves attention mass and fp16 operands reach 1-2e-3 in a torch simulation; the device figure is
    printed and held to 5e-3 (an fp32-grade (hi, lo) operand path for the ViT would cost 3 x the matrix-pipe time of a kernel family
    that is matrix-pipe bound: not built)."""
    from boxmot_amd.clip_weights import pack_clipreid, random_clipreid_state_dict
    from boxmot_amd.reid import HipReID
    from oracle.clipreid import OracleClipReID
    img = np.random.default_rng(5).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
    boxes = _boxes(8, seed=3 + seed)
    errs = {}
    for name, kw in (("gain_randomised", dict(gain_randomised=True)), ("gain_randomised+sharp_attention", dict(gain_randomised=True, sharp_attention=True))):
        sd = random_clipreid_state_dict(seed, **kw)
        reid = HipReID(pack_clipreid(sd), max_crops=8)
        got = reid.get_features(boxes, img)
        reid.close()
        want = OracleClipReID(sd).get_features(boxes, img)
        errs[name] = float(np.abs(got - want).max())
        cos = want @ want.T
        print(f"CLIP-ReID ViT-B/16, {name}, seed {seed}: max|diff| vs fp32 oracle = {errs[name]:.2e} (mean cosine between crops {cos[np.triu_indices(8, 1)].mean():.3f})")
    assert errs["gain_randomised"] < TOL, errs
    assert errs["gain_randomised+sharp_attention"] < 5e-3, errs


def test_strongsort_with_clipreid_inside_update_matches_oracle_ids(vitb16):
    """StrongSORT asks the ReID model itself (strongsort.py:95-99): BASELINE configuration 5's pairing at a small scene."""
    from boxmot_amd.reid import HipReID
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.strongsort import StrongSort
    from oracle.clipreid import OracleClipReID
    from oracle.strongsort import StrongSortOracle
    sd, blob = vitb16
    sc = Scenario(8, 12, width=960, height=540, random_image=True)
    reid = HipReID(blob, max_crops=16)
    trk = StrongSort(reid_model=reid, max_tracks=64, max_dets=16, emb_dim=1280)
    orc = StrongSortOracle(reid=OracleClipReID(sd))
    for t in range(6):
        dets, _ = sc.frame(t)
        got = np.asarray(trk.update(dets, sc.image))
        want = orc.update(dets, sc.image)
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
        assert np.allclose(got[:, :4], want[:, :4], atol=1e-3)
    trk.close()
    reid.close()


def test_strongsort_handle_with_clp1_weights_runs_reid_on_device(vitb16, tmp_path):
    """reid_model_path of the C ABI takes a CLP1 blob: crops -> ViT-B/16 -> bank distance -> step, all on the device."""
    from boxmot_amd.reid_weights import save_blob
    from boxmot_amd.scenario import Scenario
    from boxmot_amd.strongsort import StrongSort
    from oracle.clipreid import OracleClipReID
    from oracle.strongsort import StrongSortOracle
    sd, blob = vitb16
    path = save_blob(blob, tmp_path / "clip_vitb16.clp1")
    sc = Scenario(8, 12, width=960, height=540, random_image=True, stream=3)
    trk = StrongSort(reid_weights=str(path), max_tracks=64, max_dets=16, emb_dim=1280)
    orc = StrongSortOracle(reid=OracleClipReID(sd))
    for t in range(5):
        dets, _ = sc.frame(t)
        got = np.asarray(trk.update(dets, sc.image))
        want = orc.update(dets, sc.image)
        assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
    trk.close()
