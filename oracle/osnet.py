"""Oracle OSNet forward (torch CPU fp32) -- TEST INFRASTRUCTURE ONLY.

A functional restatement of boxmot/reid/backbones/osnet.py driven directly by
a state_dict with the reference's parameter names:
  * OSNet.featuremaps / forward (eval)         osnet.py:380-405
  * OSBlock.forward                            osnet.py:246-260
  * LightConv3x3 (1x1 linear -> dw3x3 -> BN -> ReLU)   osnet.py:127-155
  * ChannelGate (GAP -> fc1 -> ReLU -> fc2 -> sigmoid)  osnet.py:161-209
  * head: GAP -> Linear -> BatchNorm1d -> ReLU  osnet.py:311-315, 393-396
This is the fp32 reference the HIP ReID kernels are compared with (tolerance
1e-3 on L2-normalised embeddings, BASELINE.json north_star).  It is validated
bit-for-bit against the reference nn.Module by tests/golden/make_golden.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

# (stem, stage1, stage2, stage3) channel widths; osnet.py:488-545
ARCH_CHANNELS = {
    "osnet_x1_0": (64, 256, 384, 512),
    "osnet_x0_75": (48, 192, 288, 384),
    "osnet_x0_5": (32, 128, 192, 256),
    "osnet_x0_25": (16, 64, 96, 128),
}
BN_EPS = 1e-5
# number of stacked LightConv3x3 in branches a..d (osnet.py:223-241)
BRANCH_DEPTHS = (("conv2a", 1), ("conv2b", 2), ("conv2c", 3), ("conv2d", 4))


def _bn(sd, p, x):
    return F.batch_norm(
        x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
        False, 0.0, BN_EPS,
    )


def _conv_bn(sd, p, x, relu, stride=1, padding=0):
    x = F.conv2d(x, sd[p + ".conv.weight"], None, stride, padding)
    x = _bn(sd, p + ".bn", x)
    return F.relu(x) if relu else x


def _light(sd, p, x):
    c = x.shape[1]
    x = F.conv2d(x, sd[p + ".conv1.weight"])
    x = F.conv2d(x, sd[p + ".conv2.weight"], None, 1, 1, 1, c)
    return F.relu(_bn(sd, p + ".bn", x))


def _gate(sd, p, x):
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(F.conv2d(g, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"]))
    g = torch.sigmoid(F.conv2d(g, sd[p + ".fc2.weight"], sd[p + ".fc2.bias"]))
    return x * g


def _osblock(sd, p, x):
    identity = x
    x1 = _conv_bn(sd, p + ".conv1", x, relu=True)
    outs = []
    for name, depth in BRANCH_DEPTHS:
        t = x1
        if depth == 1:
            t = _light(sd, f"{p}.{name}", t)
        else:
            for k in range(depth):
                t = _light(sd, f"{p}.{name}.{k}", t)
        outs.append(t)
    x2 = _gate(sd, p + ".gate", outs[0]) + _gate(sd, p + ".gate", outs[1]) \
        + _gate(sd, p + ".gate", outs[2]) + _gate(sd, p + ".gate", outs[3])
    x3 = _conv_bn(sd, p + ".conv3", x2, relu=False)
    if (p + ".downsample.conv.weight") in sd:
        identity = _conv_bn(sd, p + ".downsample", identity, relu=False)
    return F.relu(x3 + identity)


@torch.no_grad()
def osnet_forward(sd, x: torch.Tensor, return_stages: bool = False):
    """Eval-mode forward: (N,3,256,128) fp32 -> (N,512) fp32 (not L2-normalised)."""
    stages = {}
    x = _conv_bn(sd, "conv1", x, relu=True, stride=2, padding=3)
    stages["conv1"] = x
    x = F.max_pool2d(x, 3, 2, 1)
    stages["maxpool"] = x
    for stage, reduce in (("conv2", True), ("conv3", True), ("conv4", False)):
        x = _osblock(sd, stage + ".0", x)
        stages[stage + ".0"] = x
        x = _osblock(sd, stage + ".1", x)
        stages[stage + ".1"] = x
        if reduce:
            x = _conv_bn(sd, stage + ".2.0", x, relu=True)
            x = F.avg_pool2d(x, 2, 2)
            stages[stage + ".2"] = x
    x = _conv_bn(sd, "conv5", x, relu=True)
    stages["conv5"] = x
    v = F.adaptive_avg_pool2d(x, 1).flatten(1)
    v = F.linear(v, sd["fc.0.weight"], sd["fc.0.bias"])
    v = F.batch_norm(v, sd["fc.1.running_mean"], sd["fc.1.running_var"], sd["fc.1.weight"],
                     sd["fc.1.bias"], False, 0.0, BN_EPS)
    v = F.relu(v)
    return (v, stages) if return_stages else v


class OracleReID:
    """Oracle of ``BaseModelBackend.get_features`` (base_backend.py:197-207)."""

    def __init__(self, state_dict, input_shape=(256, 128), threads: int | None = None, preprocess: str = "resize"):
        self.sd = {k: v.detach().to(torch.float32) for k, v in state_dict.items()}
        self.input_shape = input_shape
        self.preprocess = preprocess
        if threads:
            torch.set_num_threads(threads)

    def get_features(self, xyxys, img):
        import numpy as np

        from oracle.crops import get_crops

        xyxys = np.asarray(xyxys)
        if xyxys.size != 0:
            crops = get_crops(xyxys, img, self.input_shape, self.preprocess)
            feats = osnet_forward(self.sd, torch.from_numpy(crops)).numpy()
        else:
            feats = np.array([])
        feats = feats / np.linalg.norm(feats, axis=-1, keepdims=True)
        return feats
