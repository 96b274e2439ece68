"""CLIP-ReID (ViT-B/16) weight preparation for the HIP ReID engine.

* ``pack_clipreid(state_dict)`` -- serialise the tensors of the reference's ``build_transformer`` (parameter names of
  boxmot/reid/backbones/clip/make_model.py:35-139 + clip/model.py:229-295, e.g. ``clip_market1501.pt``; ``module.`` prefix and
  the two training-only classifiers ignored) into the "CLP1" fp32 blob consumed by the C ABI (csrc/clip_engine.hpp):
  patch-embedding weights reordered to the NHWC gather order (ky, kx, c), the two BatchNorm1d necks folded to scale / shift
  (eval semantics, TEST.NECK_FEAT = "after").
* ``random_clipreid_state_dict`` -- seeded random weights with the reference's names and shapes and CLIP's initialisation
  scales (clip/model.py:253-263, CLIP.initialize_parameters), non-trivial neck statistics; there is no network access for
  the real checkpoints.
"""
from __future__ import annotations

import numpy as np

MAGIC = 0x434C5031          # "CLP1"
HEADER_INTS = 16
BN_EPS = 1e-5


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64) if hasattr(t, "detach") else np.asarray(t, dtype=np.float64)


def _clean(sd):
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd else sd
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def pack_clipreid(state_dict, input_hw=(256, 128)) -> np.ndarray:
    sd = _clean(state_dict)
    e = "image_encoder."
    conv = _np(sd[e + "conv1.weight"])                       # (width, 3, patch, patch)
    width, patch = conv.shape[0], conv.shape[-1]
    layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(e + "transformer.resblocks."))
    heads = width // 64
    gh, gw = input_hw[0] // patch, input_hw[1] // patch
    tokens = gh * gw + 1
    proj = _np(sd[e + "proj"])                               # (width, out_dim)
    out_dim = proj.shape[1]
    pos = _np(sd[e + "positional_embedding"])
    if pos.shape != (tokens, width):
        raise ValueError(f"positional embedding {pos.shape} does not match a {input_hw} input with patch {patch}")
    parts = []

    def put(a):
        parts.append(np.ascontiguousarray(a, dtype=np.float64).reshape(-1))

    put(np.transpose(conv, (0, 2, 3, 1)))                    # (width, ky, kx, c): the k order of the NHWC patch gather
    put(_np(sd[e + "class_embedding"])); put(pos)
    put(_np(sd[e + "ln_pre.weight"])); put(_np(sd[e + "ln_pre.bias"]))
    for i in range(layers):
        p = f"{e}transformer.resblocks.{i}."
        put(_np(sd[p + "ln_1.weight"])); put(_np(sd[p + "ln_1.bias"]))
        put(_np(sd[p + "attn.in_proj_weight"])); put(_np(sd[p + "attn.in_proj_bias"]))         # (3 width, width), q | k | v
        put(_np(sd[p + "attn.out_proj.weight"])); put(_np(sd[p + "attn.out_proj.bias"]))
        put(_np(sd[p + "ln_2.weight"])); put(_np(sd[p + "ln_2.bias"]))
        put(_np(sd[p + "mlp.c_fc.weight"])); put(_np(sd[p + "mlp.c_fc.bias"]))                 # (4 width, width)
        put(_np(sd[p + "mlp.c_proj.weight"])); put(_np(sd[p + "mlp.c_proj.bias"]))             # (width, 4 width)
    put(_np(sd[e + "ln_post.weight"])); put(_np(sd[e + "ln_post.bias"]))
    put(proj)
    for name in ("bottleneck", "bottleneck_proj"):            # BatchNorm1d necks -> y = x * scale + shift
        scale = _np(sd[name + ".weight"]) / np.sqrt(_np(sd[name + ".running_var"]) + BN_EPS)
        put(scale); put(_np(sd[name + ".bias"]) - _np(sd[name + ".running_mean"]) * scale)
    body = np.concatenate(parts).astype(np.float32)
    header = np.zeros(HEADER_INTS, dtype=np.int32)
    header[:10] = [MAGIC, width, layers, heads, patch, gh, gw, out_dim, input_hw[0], input_hw[1]]
    header[10] = body.size
    return np.concatenate([header.view(np.float32), body])


def random_clipreid_state_dict(seed: int = 0, width: int = 768, layers: int = 12, out_dim: int = 512, patch: int = 16,
                               input_hw=(256, 128), gain_randomised: bool = False, sharp_attention: bool = False):
    """ViT-B/16 defaults; smaller widths (multiples of 128) / depths give the reduced models the emulation tests run.
    ``gain_randomised``: the noise-amplifying twin of the initialisation-like set (what the BatchNorm-calibrated OSNets are to the
    OSNet init; round-4 review, Weak 5): LayerNorm gains U(0.4, 2.5) and biases N(0, 0.3), neck BatchNorm variances U(0.02, 0.2)
    (x 2 .. 7 gain on the final features) and means N(0, 0.05).  ``sharp_attention`` additionally makes the query / key projections
    2.5 x larger (logits ~6 x larger: a rounding error in a logit moves attention mass) -- the operand precision of the device
    kernels is gated on the first and characterised on the second (tests/test_gpu_clipreid.py, tools/config_bench.py c5;
    a torch simulation of fp16 GEMM operands gives 4e-5 / 9e-5 / 1-2e-3 for init-like / gain-randomised / + sharp attention)."""
    import torch

    g = torch.Generator().manual_seed(seed)
    heads = width // 64
    tokens = (input_hw[0] // patch) * (input_hw[1] // patch) + 1
    rn = lambda *shape, std=1.0: torch.randn(*shape, generator=g) * std
    scale = width ** -0.5
    proj_std, attn_std, fc_std = scale * (2 * layers) ** -0.5, scale, (2 * width) ** -0.5      # CLIP.initialize_parameters
    sd = {}
    e = "image_encoder."
    sd[e + "conv1.weight"] = rn(width, 3, patch, patch, std=(3 * patch * patch) ** -0.5)
    sd[e + "class_embedding"] = rn(width, std=scale)
    sd[e + "positional_embedding"] = rn(tokens, width, std=scale)
    for name in ("ln_pre", "ln_post"):
        sd[e + name + ".weight"] = 1.0 + rn(width, std=0.1)
        sd[e + name + ".bias"] = rn(width, std=0.05)
    for i in range(layers):
        p = f"{e}transformer.resblocks.{i}."
        for ln in ("ln_1", "ln_2"):
            sd[p + ln + ".weight"] = 1.0 + rn(width, std=0.1)
            sd[p + ln + ".bias"] = rn(width, std=0.05)
        sd[p + "attn.in_proj_weight"] = rn(3 * width, width, std=attn_std)
        sd[p + "attn.in_proj_bias"] = rn(3 * width, std=0.02)
        sd[p + "attn.out_proj.weight"] = rn(width, width, std=proj_std)
        sd[p + "attn.out_proj.bias"] = rn(width, std=0.02)
        sd[p + "mlp.c_fc.weight"] = rn(4 * width, width, std=fc_std)
        sd[p + "mlp.c_fc.bias"] = rn(4 * width, std=0.02)
        sd[p + "mlp.c_proj.weight"] = rn(width, 4 * width, std=proj_std)
        sd[p + "mlp.c_proj.bias"] = rn(width, std=0.02)
    sd[e + "proj"] = rn(width, out_dim, std=scale)
    for name, c in (("bottleneck", width), ("bottleneck_proj", out_dim)):
        sd[name + ".weight"] = torch.empty(c).uniform_(0.5, 1.5, generator=g)
        sd[name + ".bias"] = rn(c, std=0.2)
        sd[name + ".running_mean"] = rn(c, std=0.3)
        sd[name + ".running_var"] = torch.empty(c).uniform_(0.5, 2.0, generator=g)
    if gain_randomised:
        for k in list(sd):
            if ".ln_" in k or "ln_pre" in k or "ln_post" in k:
                n = sd[k].numel()
                sd[k] = torch.empty(n).uniform_(0.4, 2.5, generator=g) if k.endswith(".weight") else rn(n, std=0.3)
            elif k.endswith(".running_var"):
                sd[k] = torch.empty(sd[k].numel()).uniform_(0.02, 0.2, generator=g)
            elif k.endswith(".running_mean"):
                sd[k] = rn(sd[k].numel(), std=0.05)
    if sharp_attention:
        for k in list(sd):
            if k.endswith("attn.in_proj_weight"):
                sd[k] = sd[k].clone()
                sd[k][: 2 * width] *= 2.5                       # q and k rows
    return {k: v.contiguous() for k, v in sd.items()}
