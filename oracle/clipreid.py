"""Oracle CLIP-ReID (ViT-B/16) forward (torch CPU fp32) -- TEST INFRASTRUCTURE ONLY.

A functional restatement, driven by a state_dict with the reference's parameter names, of the eval path of
  * build_transformer.forward, ViT-B-16 branch, TEST.NECK_FEAT = "after"   boxmot/reid/backbones/clip/make_model.py:95-139
    (config/defaults.py:60,72,227: stride 16, input 256 x 128, neck feature after the BatchNorm bottlenecks)
  * VisionTransformer.forward                                              boxmot/reid/backbones/clip/clip/model.py:265-295
  * ResidualAttentionBlock (nn.MultiheadAttention, LayerNorm in fp32, QuickGELU)   clip/model.py:169-211
Output: cat(bottleneck(ln_post(x12)[:, 0]), bottleneck_proj((ln_post(x12) @ proj)[:, 0])) = 768 + 512 = 1280 values per crop
(the feature width of BASELINE.json configuration 5).  The reference backend normalises crops with mean = std = 0.5 for
"clip" models (reid/backends/base_backend.py:52-54).  Pinned bit-for-bit against the reference modules themselves
(tests/test_oracle_clipreid.py imports clip/model.py by path: it needs only torch).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LN_EPS = 1e-5
BN_EPS = 1e-5


def _ln(x, w, b):
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, LN_EPS)


def _block(sd, p, x, heads):
    """x: (L, N, D) as nn.MultiheadAttention takes it (model.py:283)."""
    h = _ln(x, sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"])
    a, _ = F.multi_head_attention_forward(
        h, h, h, x.shape[-1], heads, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"], None, None, False, 0.0,
        sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"], training=False, need_weights=False)
    x = x + a
    h = _ln(x, sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"])
    h = F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)                                   # QuickGELU, model.py:181-183
    return x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])


@torch.no_grad()
def clipreid_forward(sd, x: torch.Tensor, return_tokens: bool = False):
    """Eval-mode forward: (N, 3, H, W) fp32 -> (N, width + out_dim) fp32 (not L2-normalised)."""
    e = "image_encoder."
    w = sd[e + "conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    heads = width // 64                                                 # model.py:330
    layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(e + "transformer.resblocks."))
    x = F.conv2d(x, w, None, patch)                                     # stride = patch (config STRIDE_SIZE 16)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)
    cls = sd[e + "class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
    x = torch.cat([cls, x], dim=1) + sd[e + "positional_embedding"].to(x.dtype)
    x = _ln(x, sd[e + "ln_pre.weight"], sd[e + "ln_pre.bias"])
    x = x.permute(1, 0, 2)
    for i in range(layers):
        x = _block(sd, f"{e}transformer.resblocks.{i}", x, heads)
    x12 = _ln(x.permute(1, 0, 2), sd[e + "ln_post.weight"], sd[e + "ln_post.bias"])
    xproj = x12 @ sd[e + "proj"]
    feat = F.batch_norm(x12[:, 0], sd["bottleneck.running_mean"], sd["bottleneck.running_var"], sd["bottleneck.weight"],
                        sd["bottleneck.bias"], False, 0.0, BN_EPS)
    feat_proj = F.batch_norm(xproj[:, 0], sd["bottleneck_proj.running_mean"], sd["bottleneck_proj.running_var"],
                             sd["bottleneck_proj.weight"], sd["bottleneck_proj.bias"], False, 0.0, BN_EPS)
    out = torch.cat([feat, feat_proj], dim=1)
    return (out, x12) if return_tokens else out


class OracleClipReID:
    """Oracle of ``BaseModelBackend.get_features`` for a "clip" model (base_backend.py:52-54, 197-207)."""

    def __init__(self, state_dict, input_shape=(256, 128), preprocess: str = "resize"):
        self.sd = {k: v.detach().to(torch.float32) for k, v in state_dict.items()}
        self.input_shape, self.preprocess = input_shape, preprocess

    def get_features(self, xyxys, img):
        import numpy as np

        from oracle.crops import get_crops

        xyxys = np.asarray(xyxys)
        if xyxys.size == 0:
            return np.array([])
        crops = get_crops(xyxys, img, self.input_shape, self.preprocess, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5))
        feats = clipreid_forward(self.sd, torch.from_numpy(crops)).numpy()
        return feats / np.linalg.norm(feats, axis=-1, keepdims=True)
