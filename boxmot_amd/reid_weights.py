"""OSNet weight preparation for the HIP ReID engine (PyTorch is used here only as
the container / source of backbone weights).

* ``pack_osnet(state_dict)``   -- fold every BatchNorm into the preceding
  convolution / linear layer (eval semantics, eps = 1e-5) and serialise the
  tensors in the order fixed by ``csrc/reid_layout.hpp`` ("OSN1" blob).
  Accepts the reference's checkpoints as they are (parameter names of
  boxmot/reid/backbones/osnet.py, e.g. ``osnet_x0_25_msmt17.pt``), with or
  without a ``module.`` prefix; the classifier head is ignored (eval forward
  never uses it, osnet.py:397-398).
* ``random_osnet_state_dict`` -- seeded random-init weights of the named
  architecture for benchmarking without checkpoints (there is no network
  access); BatchNorm statistics are calibrated on noise so activations have a
  trained-network-like scale (matters for the fp16 kernels).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

ARCH_CHANNELS = {
    "osnet_x1_0": (64, 256, 384, 512),
    "osnet_x0_75": (48, 192, 288, 384),
    "osnet_x0_5": (32, 128, 192, 256),
    "osnet_x0_25": (16, 64, 96, 128),
}
FEATURE_DIM = 512
BN_EPS = 1e-5
MAGIC = 0x4F534E31
HEADER_INTS = 16
LIGHT_NAMES = ("conv2a", "conv2b.0", "conv2b.1", "conv2c.0", "conv2c.1", "conv2c.2",
               "conv2d.0", "conv2d.1", "conv2d.2", "conv2d.3")
BLOCKS = ("conv2.0", "conv2.1", "conv3.0", "conv3.1", "conv4.0", "conv4.1")


def _np(t):
    return t.detach().cpu().numpy().astype(np.float64) if hasattr(t, "detach") else np.asarray(t, dtype=np.float64)


def _clean(sd):
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd else sd
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def infer_arch(sd) -> str:
    c0 = int(_clean(sd)["conv1.conv.weight"].shape[0])
    for name, ch in ARCH_CHANNELS.items():
        if ch[0] == c0:
            return name
    raise ValueError(f"unsupported OSNet stem width {c0}")


def infer_channels(sd) -> tuple:
    """(stem, stage 1, stage 2, stage 3) output widths read off the tensors themselves (osnet.py:281-309)."""
    sd = _clean(sd)
    return (int(sd["conv1.conv.weight"].shape[0]), int(sd["conv2.0.conv3.conv.weight"].shape[0]),
            int(sd["conv3.0.conv3.conv.weight"].shape[0]), int(sd["conv4.0.conv3.conv.weight"].shape[0]))


def _channels(arch) -> tuple:
    return tuple(int(c) for c in arch) if isinstance(arch, (tuple, list)) else ARCH_CHANNELS[arch]


def _fold(w, sd, bn):
    """conv/linear weight (out, ...) + BN named ``bn`` -> (W', b')."""
    scale = _np(sd[bn + ".weight"]) / np.sqrt(_np(sd[bn + ".running_var"]) + BN_EPS)
    shift = _np(sd[bn + ".bias"]) - _np(sd[bn + ".running_mean"]) * scale
    return w * scale.reshape((-1,) + (1,) * (w.ndim - 1)), shift


def pack_osnet(state_dict) -> np.ndarray:
    """Returns the fp32 blob (header viewed as int32) consumed by the C ABI."""
    sd = _clean(state_dict)
    ch = infer_channels(sd)
    parts = []

    def put(a):
        parts.append(np.ascontiguousarray(a, dtype=np.float64).reshape(-1))

    w, b = _fold(_np(sd["conv1.conv.weight"]), sd, "conv1.bn")
    put(np.transpose(w, (0, 2, 3, 1)))            # (co, ky, kx, ci)
    put(b)
    cin = ch[0]
    for bi, p in enumerate(BLOCKS):
        cout = ch[bi // 2 + 1]
        w, b = _fold(_np(sd[p + ".conv1.conv.weight"])[:, :, 0, 0], sd, p + ".conv1.bn")
        put(w); put(b)
        for ln in LIGHT_NAMES:
            q = f"{p}.{ln}"
            put(_np(sd[q + ".conv1.weight"])[:, :, 0, 0])
            w, b = _fold(_np(sd[q + ".conv2.weight"])[:, 0], sd, q + ".bn")
            put(w); put(b)
        put(_np(sd[p + ".gate.fc1.weight"])[:, :, 0, 0]); put(_np(sd[p + ".gate.fc1.bias"]))
        put(_np(sd[p + ".gate.fc2.weight"])[:, :, 0, 0]); put(_np(sd[p + ".gate.fc2.bias"]))
        w, b = _fold(_np(sd[p + ".conv3.conv.weight"])[:, :, 0, 0], sd, p + ".conv3.bn")
        put(w); put(b)
        if cin != cout:
            w, b = _fold(_np(sd[p + ".downsample.conv.weight"])[:, :, 0, 0], sd, p + ".downsample.bn")
            put(w); put(b)
        cin = cout
        if bi in (1, 3):
            t = p.rsplit(".", 1)[0] + ".2.0"
            w, b = _fold(_np(sd[t + ".conv.weight"])[:, :, 0, 0], sd, t + ".bn")
            put(w); put(b)
    w, b = _fold(_np(sd["conv5.conv.weight"])[:, :, 0, 0], sd, "conv5.bn")
    put(w); put(b)
    scale = _np(sd["fc.1.weight"]) / np.sqrt(_np(sd["fc.1.running_var"]) + BN_EPS)
    put(_np(sd["fc.0.weight"]) * scale[:, None])
    put((_np(sd["fc.0.bias"]) - _np(sd["fc.1.running_mean"])) * scale + _np(sd["fc.1.bias"]))
    body = np.concatenate(parts).astype(np.float32)
    header = np.zeros(HEADER_INTS, dtype=np.int32)
    header[0] = MAGIC
    header[1:5] = ch
    header[5] = FEATURE_DIM
    header[6] = body.size
    return np.concatenate([header.view(np.float32), body])


def save_blob(blob: np.ndarray, path) -> Path:
    path = Path(path)
    np.ascontiguousarray(blob, dtype=np.float32).tofile(path)
    return path


def _pack_state_dict(sd):
    """OSNet or CLIP-ReID by the parameter names of the checkpoint (the reference picks the architecture from the file name,
    reid/core/registry.py; the names are what the file actually holds)."""
    inner = sd.get("state_dict", sd) if isinstance(sd, dict) and "state_dict" in sd else sd
    if any(k.replace("module.", "", 1).startswith("image_encoder.") for k in inner):
        from boxmot_amd.clip_weights import pack_clipreid

        return pack_clipreid(sd)
    return pack_osnet(sd)


def load_weights(weights):
    """Accepts a state_dict, a ``.pt`` checkpoint path, an OSN1 / CLP1 blob path or a blob array."""
    if isinstance(weights, np.ndarray):
        return np.ascontiguousarray(weights, dtype=np.float32)
    if isinstance(weights, (str, Path)):
        path = Path(weights)
        if path.suffix == ".pt" or path.suffix == ".pth":
            import torch

            return _pack_state_dict(torch.load(path, map_location="cpu", weights_only=False))
        return np.fromfile(path, dtype=np.float32)
    return _pack_state_dict(weights)


def reference_init_state_dict(arch: str = "osnet_x0_25", seed: int = 0):
    """Random-init weights exactly as the reference architecture initialises itself
    (OSNet._init_params, osnet.py:360-378): Kaiming-normal (fan_out, ReLU gain) convolutions,
    identity BatchNorm (weight 1, bias 0, running mean 0 / var 1), Linear ~ N(0, 0.01), zero biases.
    This is the "random-init weights of that architecture" the benchmark uses."""
    import torch

    ch = _channels(arch)
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k, groups=1):
        sd[name] = torch.randn(cout, cin // groups, k, k, generator=g) * (2.0 / (cout * k * k)) ** 0.5

    def bn(name, c):
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)

    conv("conv1.conv.weight", ch[0], 3, 7)
    bn("conv1.bn", ch[0])
    cin = ch[0]
    for bi, p in enumerate(BLOCKS):
        cout = ch[bi // 2 + 1]
        mid = cout // 4
        conv(p + ".conv1.conv.weight", mid, cin, 1)
        bn(p + ".conv1.bn", mid)
        for ln in LIGHT_NAMES:
            conv(f"{p}.{ln}.conv1.weight", mid, mid, 1)
            conv(f"{p}.{ln}.conv2.weight", mid, mid, 3, groups=mid)
            bn(f"{p}.{ln}.bn", mid)
        hid = mid // 16
        conv(p + ".gate.fc1.weight", hid, mid, 1)
        sd[p + ".gate.fc1.bias"] = torch.zeros(hid)
        conv(p + ".gate.fc2.weight", mid, hid, 1)
        sd[p + ".gate.fc2.bias"] = torch.zeros(mid)
        conv(p + ".conv3.conv.weight", cout, mid, 1)
        bn(p + ".conv3.bn", cout)
        if cin != cout:
            conv(p + ".downsample.conv.weight", cout, cin, 1)
            bn(p + ".downsample.bn", cout)
        cin = cout
        if bi in (1, 3):
            t = p.rsplit(".", 1)[0] + ".2.0"
            conv(t + ".conv.weight", cout, cout, 1)
            bn(t + ".bn", cout)
    conv("conv5.conv.weight", ch[3], ch[3], 1)
    bn("conv5.bn", ch[3])
    sd["fc.0.weight"] = torch.randn(FEATURE_DIM, ch[3], generator=g) * 0.01
    sd["fc.0.bias"] = torch.zeros(FEATURE_DIM)
    bn("fc.1", FEATURE_DIM)
    return sd


def random_osnet_state_dict(arch: str = "osnet_x0_25", seed: int = 0, num_classes: int = 0, calib_batch: int = 4):
    """Seeded random weights with the reference's parameter names and shapes ("calibrated" variant:
    deliberately non-trivial BatchNorm statistics to exercise the folding; it is a noise-amplifying
    network, used for the fp32 parity tests -- see reference_init_state_dict for the benchmark init).

    Convolutions: Kaiming-normal (fan_out), as osnet.py:360-378 initialises them.
    BatchNorm: gamma ~ U(0.5, 1.5), beta ~ N(0, 0.2); running statistics are set
    from a forward pass over noise crops (then perturbed by U(0.8, 1.25)), i.e.
    the network is "trained-scale" without being trained.
    """
    import torch
    import torch.nn.functional as F

    ch = _channels(arch)
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv_w(cout, cin, k, groups=1):
        std = (2.0 / (cout * k * k)) ** 0.5
        return torch.randn(cout, cin // groups, k, k, generator=g) * std

    def bn_fit(name, y):
        c = y.shape[1]
        dims = [d for d in range(y.dim()) if d != 1]
        mean = y.mean(dim=dims)
        var = y.var(dim=dims, unbiased=False) + 1e-3
        sd[name + ".weight"] = torch.empty(c).uniform_(0.5, 1.5, generator=g)
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.2
        sd[name + ".running_mean"] = mean + torch.randn(c, generator=g) * 0.1 * var.sqrt()
        sd[name + ".running_var"] = var * torch.empty(c).uniform_(0.8, 1.25, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(1)
        return F.batch_norm(y, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, 0.0, BN_EPS)

    def conv_bn(name, x, cout, k=1, stride=1, pad=0, relu=True):
        sd[name + ".conv.weight"] = conv_w(cout, x.shape[1], k)
        y = bn_fit(name + ".bn", F.conv2d(x, sd[name + ".conv.weight"], None, stride, pad))
        return F.relu(y) if relu else y

    def light(name, x):
        c = x.shape[1]
        sd[name + ".conv1.weight"] = conv_w(c, c, 1)
        sd[name + ".conv2.weight"] = conv_w(c, c, 3, groups=c) * 2.0
        y = F.conv2d(F.conv2d(x, sd[name + ".conv1.weight"]), sd[name + ".conv2.weight"], None, 1, 1, 1, c)
        return F.relu(bn_fit(name + ".bn", y))

    def gate(name, x):
        c = x.shape[1]
        hid = c // 16
        if name + ".fc1.weight" not in sd:
            sd[name + ".fc1.weight"] = torch.randn(hid, c, 1, 1, generator=g) * (1.0 / c) ** 0.5
            sd[name + ".fc1.bias"] = torch.randn(hid, generator=g) * 0.1
            sd[name + ".fc2.weight"] = torch.randn(c, hid, 1, 1, generator=g)
            sd[name + ".fc2.bias"] = torch.randn(c, generator=g) * 0.5
        v = F.adaptive_avg_pool2d(x, 1)
        v = F.relu(F.conv2d(v, sd[name + ".fc1.weight"], sd[name + ".fc1.bias"]))
        return x * torch.sigmoid(F.conv2d(v, sd[name + ".fc2.weight"], sd[name + ".fc2.bias"]))

    def osblock(name, x, cout):
        mid = cout // 4
        x1 = conv_bn(name + ".conv1", x, mid)
        outs = []
        for bname, depth in (("conv2a", 1), ("conv2b", 2), ("conv2c", 3), ("conv2d", 4)):
            t = x1
            for k in range(depth):
                t = light(f"{name}.{bname}" if depth == 1 else f"{name}.{bname}.{k}", t)
            outs.append(t)
        x2 = sum(gate(name + ".gate", o) for o in outs)
        x3 = conv_bn(name + ".conv3", x2, cout, relu=False)
        idn = x if x.shape[1] == cout else conv_bn(name + ".downsample", x, cout, relu=False)
        return F.relu(x3 + idn)

    with torch.no_grad():
        x = torch.randn(calib_batch, 3, 256, 128, generator=g)
        x = conv_bn("conv1", x, ch[0], 7, 2, 3)
        x = F.max_pool2d(x, 3, 2, 1)
        for si, stage in enumerate(("conv2", "conv3", "conv4")):
            x = osblock(stage + ".0", x, ch[si + 1])
            x = osblock(stage + ".1", x, ch[si + 1])
            if si < 2:
                x = F.avg_pool2d(conv_bn(stage + ".2.0", x, ch[si + 1]), 2, 2)
        x = conv_bn("conv5", x, ch[3])
        v = F.adaptive_avg_pool2d(x, 1).flatten(1)
        sd["fc.0.weight"] = torch.randn(FEATURE_DIM, ch[3], generator=g) * (2.0 / ch[3]) ** 0.5
        sd["fc.0.bias"] = torch.randn(FEATURE_DIM, generator=g) * 0.05
        bn_fit("fc.1", F.linear(v, sd["fc.0.weight"], sd["fc.0.bias"]))
        if num_classes:
            sd["classifier.weight"] = torch.randn(num_classes, FEATURE_DIM, generator=g) * 0.01
            sd["classifier.bias"] = torch.zeros(num_classes)
    return {k: v.contiguous() for k, v in sd.items()}
