"""oracle/clipreid.py against the reference modules themselves (build container only: imports
/root/reference/boxmot/reid/backbones/clip/clip/model.py by path -- it needs only torch), and the CLP1 packer's layout."""
import importlib.util

import numpy as np
import pytest

from oracle import ref_harness


@pytest.mark.skipif(not ref_harness.reference_available(), reason="needs /root/reference")
@pytest.mark.parametrize("width,out_dim", [(128, 64), (192, 96)])
def test_oracle_equals_reference_vision_transformer_and_necks(width, out_dim):
    import torch

    from boxmot_amd.clip_weights import random_clipreid_state_dict
    from oracle.clipreid import clipreid_forward
    ref_harness.install_standins()
    spec = importlib.util.spec_from_file_location("_ref_clip_model", ref_harness.REFERENCE_ROOT / "boxmot/reid/backbones/clip/clip/model.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    # the reference forward indexes resblocks[:11] and [11] (model.py:285-286): 12 layers, reduced width / input for speed
    sd = random_clipreid_state_dict(3, width=width, layers=12, out_dim=out_dim, input_hw=(64, 32))
    vt = m.VisionTransformer(h_resolution=4, w_resolution=2, patch_size=16, stride_size=16, width=width, layers=12,
                             heads=width // 64, output_dim=out_dim).eval()
    vt.load_state_dict({k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}, strict=True)
    bn, bnp = torch.nn.BatchNorm1d(width).eval(), torch.nn.BatchNorm1d(out_dim).eval()
    bn.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("bottleneck.")}, strict=False)
    bnp.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("bottleneck_proj.")}, strict=False)
    x = torch.randn(3, 3, 64, 32, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        _, x12, xproj = vt(x)
        want = torch.cat([bn(x12[:, 0]), bnp(xproj[:, 0])], dim=1)       # make_model.py:119-137, NECK_FEAT "after"
    got = clipreid_forward(sd, x)
    assert got.shape == (3, width + out_dim) and torch.equal(got, want)


def test_clp1_blob_layout():
    from boxmot_amd.clip_weights import HEADER_INTS, MAGIC, pack_clipreid, random_clipreid_state_dict
    sd = random_clipreid_state_dict(0, width=128, layers=2, out_dim=64, input_hw=(32, 16))
    blob = pack_clipreid(sd, (32, 16))
    hdr = blob[:HEADER_INTS].view(np.int32)
    assert hdr[0] == MAGIC and hdr[1:10].tolist() == [128, 2, 2, 16, 2, 1, 64, 32, 16] and hdr[10] == blob.size - HEADER_INTS
    w = 128
    per_layer = 2 * w + 3 * w * w + 3 * w + w * w + w + 2 * w + 4 * w * w + 4 * w + 4 * w * w + w
    assert hdr[10] == w * 768 + w + 3 * w + 2 * w + 2 * per_layer + 2 * w + w * 64 + 2 * w + 2 * 64
    # patch-embedding rows are (ky, kx, c)-ordered
    conv = sd["image_encoder.conv1.weight"].numpy()
    body = blob[HEADER_INTS:]
    assert np.array_equal(body[: 768].reshape(16, 16, 3), conv[0].transpose(1, 2, 0))
    with pytest.raises(ValueError):
        pack_clipreid(sd, (64, 16))
