"""Where the fp16 error of the fused ReID kernels enters, and what precision an OSNet needs for 1e-3 embeddings.

A torch-CPU simulator of the fused pipeline's arithmetic (fp16-rounded MFMA operands with fp32 accumulation, fp16 LDS image,
sequential packed-fp16 depthwise accumulation, fp16 gate sum) with one precision switch per piece and per stage.  It is a
development / evidence tool: the numbers it prints are in profiles/r2_reid_error_budget.txt and DESIGN.md section 4.2.

    python tools/reid_error_budget.py            # ~3 minutes on 8 cores

Findings (3 BN-calibrated seeds + the reference random init, 16 crops of a noise frame; max |embedding - fp32 oracle|):
  * the reference's own half-precision path (torch fp16 end to end = `half=True`, base_backend.py:162,185,223) is at 0.9-1.5e-2
    on the calibrated networks and 1.5e-4 on the random init; the fused kernels' arithmetic is at 6-10e-3 / 1.2e-4;
  * the calibrated networks are "whitened": BatchNorm1d after the FC divides by the (small) spread of the pooled features over
    the calibration crops, so the embedding is an amplified deviation from the mean feature -- exactly what makes it
    discriminative (inter-crop cosine 0.6-0.8 instead of 0.99) -- and rounding noise is amplified with it.  The same weights
    on out-of-distribution crops (inter-crop cosine 0.98) give 7-10e-4;
  * no single piece is the culprit: with fp32-grade weights AND depthwise AND gate but fp16 activation storage the error is still
    5e-3; with fp32 activations but fp16 weights 4-6e-3; stem alone in fp16 (input, weights, output rounding) 4-6e-3, stage 0
    alone 8e-3, stage 1 alone 2e-3, stage 2 alone 0.7-1.2e-3, head alone 3e-4;
  * hence 1e-3 on such weights needs fp32-grade operands (weights and activations) in the stem and in all three stages: only the
    head may stay fp16.  That is what mode 0 (per-layer fp32 kernels, 4e-6) provides; a fused fp16-operand kernel cannot.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from boxmot_amd.reid_weights import random_osnet_state_dict, reference_init_state_dict  # noqa: E402
from oracle import osnet as O  # noqa: E402
from oracle.crops import get_crops  # noqa: E402

EPS = 1e-5


def h(x):
    return x.to(torch.float16).to(torch.float32)


def q(x, bits):
    return h(x) if bits == 16 else x


def fold(sd, w, bn):
    scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + EPS)
    shift = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * scale
    return (w.double() * scale.reshape((-1,) + (1,) * (w.dim() - 1))).float(), shift.float()


def dwconv(x, w, b, mode):
    """'h': 9 sequential fp16 FMAs (v_pk_fma_f16) as the kernels do; 'f': fp32 accumulate, fp16 weights; 'F': fp32."""
    if mode == "h":
        xp = F.pad(x, (1, 1, 1, 1))
        wq = h(w)
        o = h(b).view(1, -1, 1, 1).expand_as(x).clone()
        H, W = x.shape[2:]
        for t in range(9):
            dy, dx = t // 3, t % 3
            o = h(o + xp[:, :, dy:dy + H, dx:dx + W] * wq[:, dy, dx].view(1, -1, 1, 1))
        return o
    return F.conv2d(x, (h(w) if mode == "f" else w).unsqueeze(1), b, 1, 1, 1, x.shape[1])


def osblock(sd, p, x, c):
    wq, aq, dwm, gq = c["w"], c["a"], c["dw"], c["g"]
    w1, b1 = fold(sd, sd[p + ".conv1.conv.weight"], p + ".conv1.bn")
    x1 = q(F.relu(F.conv2d(x, q(w1, wq), b1)), aq)
    x2 = None
    for name, depth in O.BRANCH_DEPTHS:
        t = x1
        for k in range(depth):
            pp = f"{p}.{name}" if depth == 1 else f"{p}.{name}.{k}"
            t = q(F.conv2d(t, q(sd[pp + ".conv1.weight"], wq)), aq)          # 1x1 -> LDS image
            wd, bd = fold(sd, sd[pp + ".conv2.weight"][:, 0], pp + ".bn")
            t = q(F.relu(dwconv(t, wd, bd, dwm)), aq)
        g = F.adaptive_avg_pool2d(t, 1)
        g = F.relu(F.conv2d(g, sd[p + ".gate.fc1.weight"], sd[p + ".gate.fc1.bias"]))
        g = torch.sigmoid(F.conv2d(g, sd[p + ".gate.fc2.weight"], sd[p + ".gate.fc2.bias"]))
        if gq == 16:
            g = h(g)
            x2 = h(t * g) if x2 is None else h(x2 + t * g)
        else:
            x2 = t * g if x2 is None else x2 + t * g
    x2 = q(x2, aq)
    w3, b3 = fold(sd, sd[p + ".conv3.conv.weight"], p + ".conv3.bn")
    y = F.conv2d(x2, q(w3, wq), b3)
    if (p + ".downsample.conv.weight") in sd:
        wd, bd = fold(sd, sd[p + ".downsample.conv.weight"], p + ".downsample.bn")
        y = y + F.conv2d(x, q(wd, wq), bd)
    else:
        y = y + x
    return q(F.relu(y), aq)


def forward(sd, crops, cfg):
    c = cfg["stem"]
    w, b = fold(sd, sd["conv1.conv.weight"], "conv1.bn")
    x = F.relu(F.conv2d(q(crops, c["in"]), q(w, c["w"]), b, 2, 3))
    x = q(F.max_pool2d(x, 3, 2, 1), c["a"])
    for si, (stage, reduce) in enumerate((("conv2", True), ("conv3", True), ("conv4", False))):
        c = cfg["s%d" % si]
        x = osblock(sd, stage + ".0", x, c)
        x = osblock(sd, stage + ".1", x, c)
        if reduce:
            w, b = fold(sd, sd[stage + ".2.0.conv.weight"], stage + ".2.0.bn")
            x = F.relu(F.conv2d(x, q(w, c["w"]), b))
            x = q(F.avg_pool2d(x, 2, 2), cfg["s%d" % (si + 1)]["a"])
    c = cfg["head"]
    w, b = fold(sd, sd["conv5.conv.weight"], "conv5.bn")
    x = F.relu(F.conv2d(x, q(w, c["w"]), b))
    v = F.adaptive_avg_pool2d(x, 1).flatten(1)
    scale = sd["fc.1.weight"] / torch.sqrt(sd["fc.1.running_var"] + EPS)
    v = F.relu(F.linear(v, q(sd["fc.0.weight"] * scale[:, None], c["w"]),
                        (sd["fc.0.bias"] - sd["fc.1.running_mean"]) * scale + sd["fc.1.bias"]))
    return v / v.norm(dim=1, keepdim=True)


def mk(stem, s0, s1, s2, head):
    return dict(stem=stem, s0=s0, s1=s1, s2=s2, head=head)


H16 = dict(w=16, a=16, dw="h", g=16)                    # what the fused kernels compute
F32 = dict(w=32, a=32, dw="F", g=32)
IMP = dict(w=32, a=16, dw="F", g=32)                    # everything fp32-grade except the stored activations
A32 = dict(w=16, a=32, dw="F", g=32)                    # fp32 activations, fp16 weights
ST16, ST32 = dict(**{"in": 16}, w=16, a=16), dict(**{"in": 32}, w=32, a=32)
SCHEMES = {
    "fused kernels' arithmetic (all fp16 operands)": mk(ST16, H16, H16, H16, dict(w=16)),
    "fp32 weights/depthwise/gate, fp16 activations": mk(dict(**{"in": 32}, w=32, a=16), IMP, IMP, IMP, dict(w=32)),
    "fp32 activations, fp16 weights (stem..stage 1)": mk(dict(**{"in": 32}, w=16, a=32), A32, A32, H16, dict(w=16)),
    "only the stem in fp16": mk(ST16, F32, F32, F32, dict(w=32)),
    "only the stem INPUT in fp16": mk(dict(**{"in": 16}, w=32, a=32), F32, F32, F32, dict(w=32)),
    "only stage 0 in fp16": mk(ST32, H16, F32, F32, dict(w=32)),
    "only stage 1 in fp16": mk(ST32, F32, H16, F32, dict(w=32)),
    "only stage 2 in fp16": mk(ST32, F32, F32, H16, dict(w=32)),
    "only the head in fp16": mk(ST32, F32, F32, F32, dict(w=16)),
    "stem + stages 0-1 fp32, stage 2 + head fp16": mk(ST32, F32, F32, H16, dict(w=16)),
}


def main():
    torch.set_num_threads(8)
    img = np.random.default_rng(5).integers(0, 255, (1080, 1920, 3), dtype=np.uint8)
    rng = np.random.default_rng(1)
    n = 16
    boxes = np.stack([rng.uniform(0, 1800, n), rng.uniform(0, 900, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
    boxes[:, 2] = boxes[:, 0] + rng.uniform(20, 120, n)
    boxes[:, 3] = boxes[:, 1] + rng.uniform(40, 180, n)
    crops = torch.from_numpy(get_crops(boxes, img))
    nets = {f"calibrated seed {s}": random_osnet_state_dict("osnet_x0_25", seed=s) for s in (0, 1, 2)}
    nets["reference random init"] = reference_init_state_dict("osnet_x0_25", 0)
    with torch.no_grad():
        for nm, sd in nets.items():
            sd32 = {k: v.float() for k, v in sd.items()}
            ref = O.osnet_forward(sd32, crops)
            ref = ref / ref.norm(dim=1, keepdim=True)
            cosm = ref @ ref.T
            print(f"{nm}: mean inter-crop cosine {((cosm.sum() - n) / (n * n - n)).item():.3f}")
            sd16 = {k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}
            out = O.osnet_forward(sd16, crops.half()).float()
            out = out / out.norm(dim=1, keepdim=True)
            print(f"   {'reference half=True path (torch fp16 end to end)':52s} max|err| {(out - ref).abs().max().item():.2e}")
            for sn, cfg in SCHEMES.items():
                print(f"   {sn:52s} max|err| {(forward(sd32, crops, cfg) - ref).abs().max().item():.2e}")


if __name__ == "__main__":
    main()
