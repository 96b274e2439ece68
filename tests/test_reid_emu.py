"""Runs the fused fp16 MFMA ReID kernels (device source, unchanged) on CPU threads with an emulated
MFMA (tests/host_emu) and compares every stage with the torch fp32 oracle.  Validates the weight
packing / fragment layouts / data flow without a GPU.  Not a product path."""
import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent / "host_emu"
CLANG = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")


def _build(no_vm_wait=False, immediate=False):
    """The kernels' asynchronous global -> LDS copies are emulated in their latest-completion form (EMU_DEFER_GLDS, hip_shim.hpp):
    the data lands when the issuing thread executes BM_WAIT_VM0, not at the copy and not at a barrier (a missing wait shows).
    immediate: the earliest-completion form -- the data lands at the copy (a copy issued while another wave still reads the
    destination shows).  no_vm_wait: the wait does nothing -- the negative control of the first check."""
    out = HERE / ("libemu_reid_nowait.so" if no_vm_wait else ("libemu_reid_immediate.so" if immediate else "libemu_reid.so"))
    deps = [HERE / "emu_reid.cpp", HERE / "hip_shim.hpp"] + list((HERE.parent.parent / "boxmot_amd" / "csrc").glob("*.hpp"))
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
                               *([] if immediate else ["-DEMU_DEFER_GLDS=1"]), *(["-DEMU_NO_VM_WAIT=1"] if no_vm_wait else []),
                               "-o", str(out), str(HERE / "emu_reid.cpp")])
    return out


@pytest.mark.parametrize("stage1_handover", [0, 1])
@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
def test_fused_reid_kernels_emulated_vs_oracle(stage1_handover):
    """stage1_handover = 1: stage 1 also runs as an EMIT / RECON pair (engine build flag BM_STAGE1_HANDOVER, off by default
    until it has been measured on the device)."""
    import torch

    from boxmot_amd.reid_weights import pack_osnet, reference_init_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import osnet_forward

    lib = ctypes.CDLL(str(_build()))
    lib.emu_reid_forward.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    blob = pack_osnet(sd)
    img = np.random.default_rng(5).integers(0, 255, (480, 640, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3]], dtype=np.float32)
    crops = get_crops(boxes, img)
    nhwc = np.ascontiguousarray(np.transpose(crops, (0, 2, 3, 1)))
    n = len(boxes)
    feats = np.zeros((n, 512), np.float32)
    shapes = [(2048, 16), (2048, 64), (2048, 64), (512, 64), (512, 96), (512, 96), (128, 96), (128, 128), (128, 128)]
    bufs = [np.zeros((n,) + s, np.float32) for s in shapes]
    ptrs = (ctypes.c_void_p * 9)(*[b.ctypes.data for b in bufs])
    lib.emu_reid_set_stage1_handover(stage1_handover)
    try:
        assert lib.emu_reid_forward(blob.ctypes.data, blob.size, nhwc.ctypes.data, n, feats.ctypes.data, ptrs) == 0
    finally:
        lib.emu_reid_set_stage1_handover(0)
    want, st = osnet_forward(sd, torch.from_numpy(crops), return_stages=True)
    names = ["maxpool", "conv2.0", "conv2.1", "conv2.2", "conv3.0", "conv3.1", "conv3.2", "conv4.0", "conv4.1"]
    for nm, buf in zip(names, bufs):
        if nm in ("conv2.0", "conv2.1", "conv3.1") or (stage1_handover and nm == "conv3.0"):
            # consumed in registers (EMIT/RECON hand-over, fused transitions), never stored
            assert not buf.any()
            continue
        ref = st[nm].numpy().transpose(0, 2, 3, 1).reshape(buf.shape)
        assert np.abs(buf - ref).max() < 3e-3 * np.abs(ref).max(), nm      # fp16 storage, fp32 accumulate
    w = want.numpy()
    w = w / np.linalg.norm(w, axis=1, keepdims=True)
    assert np.abs(feats - w).max() < 1e-3
    assert (feats * w).sum(1).min() > 0.99999


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
def test_fused_crop_stem_kernel_equals_separate_kernels_emulated():
    """k_stem_resize_fused (crop + cv2-style resize + normalise + conv7x7 + maxpool in one kernel, LDS row ring)
    must reproduce the resize-kernel + stem-kernel pair bit for bit; odd frame width exercises the
    per-row dword alignment of the staged source rows, the box list the special cases."""
    import torch

    from boxmot_amd.reid_weights import pack_osnet, reference_init_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import osnet_forward

    lib = ctypes.CDLL(str(_build()))
    lib.emu_reid_forward.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.emu_stem_from_frame.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    sd = reference_init_state_dict("osnet_x0_25", seed=0)
    blob = pack_osnet(sd)
    wd, hd = 641, 480
    img = np.random.default_rng(5).integers(0, 255, (hd, wd, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [-20, -10, 40, 60], [100, 100, 100, 150], [10, 10, 138, 266],
                      [0, 0, 641, 480], [600.4, 430.2, 700, 500], [50, 20, 200, 330], [300, 5, 420, 300]], dtype=np.float32)
    n = len(boxes)
    crops = get_crops(boxes, img)
    nhwc = np.ascontiguousarray(np.transpose(crops, (0, 2, 3, 1)))
    feats = np.zeros((n, 512), np.float32)
    st_sep = np.zeros((n, 2048, 16), np.float32)
    ptrs = (ctypes.c_void_p * 9)(st_sep.ctypes.data, *[None] * 8)
    assert lib.emu_reid_forward(blob.ctypes.data, blob.size, nhwc.ctypes.data, n, feats.ctypes.data, ptrs) == 0
    st_fused = np.zeros((n, 2048, 16), np.float32)
    assert lib.emu_stem_from_frame(blob.ctypes.data, blob.size, img.ctypes.data, wd, hd, boxes.ctypes.data, n,
                                   st_fused.ctypes.data) == 0
    assert np.array_equal(st_fused, st_sep)
    _, st = osnet_forward(sd, torch.from_numpy(crops), return_stages=True)
    ref = st["maxpool"].numpy().transpose(0, 2, 3, 1).reshape(n, 2048, 16)
    assert np.abs(st_fused - ref).max() < 1e-3 * np.abs(ref).max()


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
@pytest.mark.parametrize("pad", [0, 1])
def test_crop_kernel_resize_and_resize_pad_equal_the_oracle_emulated(pad):
    """k_crop_resize (device source on CPU threads) vs oracle.crops for both preprocess modes: "resize" and the
    aspect-preserving "resize_pad" with the ImageNet-mean border (reid/core/preprocessing.py:12-45) -- bit for bit."""
    from oracle.crops import get_crops

    lib = ctypes.CDLL(str(_build()))
    lib.emu_crop_resize.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    wd, hd = 641, 480
    img = np.random.default_rng(5).integers(0, 255, (hd, wd, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [-20, -10, 40, 60], [100, 100, 100, 150], [10, 10, 138, 266],
                      [0, 0, 641, 480], [600.4, 430.2, 700, 500], [50, 20, 200, 330], [5, 5, 300, 40], [7, 3, 9, 400]], dtype=np.float32)
    out = np.zeros((len(boxes), 256, 128, 3), np.float32)
    assert lib.emu_crop_resize(img.ctypes.data, wd, hd, boxes.ctypes.data, len(boxes), pad, out.ctypes.data) == 0
    want = get_crops(boxes, img, preprocess="resize_pad" if pad else "resize").transpose(0, 2, 3, 1)
    assert np.array_equal(out, want)


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
@pytest.mark.parametrize("pad", [0, 1])
def test_oriented_box_crop_kernel_equals_the_oracle_emulated(pad):
    """k_crop_resize_obb (device source on CPU threads) vs oracle.crops for oriented boxes [cx, cy, w, h, angle]
    (base_backend.py:91-117): boxes inside the frame, hanging over its edges, degenerate sizes, angle 0 -- bit for bit."""
    from oracle.crops import get_crops, obb_crop_geometry

    lib = ctypes.CDLL(str(_build()))
    lib.emu_crop_resize_obb.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    wd, hd = 641, 480
    img = np.random.default_rng(6).integers(0, 255, (hd, wd, 3), dtype=np.uint8)
    boxes = np.array([[320, 240, 80, 160, 0.3], [10, 20, 60, 90, -1.2], [630, 470, 50, 120, 2.0], [300, 200, 33.4, 71.6, 0.77],
                      [200, 200, 128, 256, 0.0], [200.5, 100.5, 256, 512, 0.0], [100, 300, 0.2, 40, 0.5], [400, 100, 300.5, 20.5, 1.5708],
                      [-50, -60, 40, 40, 0.1]], dtype=np.float32)
    geo = np.empty((len(boxes), 8), dtype=np.float64)
    for i, b in enumerate(boxes):
        ow, oh, im = obb_crop_geometry(b)
        geo[i, 0], geo[i, 1], geo[i, 2:] = ow, oh, im
    out = np.zeros((len(boxes), 256, 128, 3), np.float32)
    assert lib.emu_crop_resize_obb(img.ctypes.data, wd, hd, geo.ctypes.data, len(boxes), pad, out.ctypes.data) == 0
    want = get_crops(boxes, img, preprocess="resize_pad" if pad else "resize").transpose(0, 2, 3, 1)
    assert np.array_equal(out, want)
    # an unrotated oriented box with integer corners inside the frame is the axis-aligned crop of the same rectangle
    aabb = get_crops(np.array([[136, 72, 264, 328]], dtype=np.float32), img).transpose(0, 2, 3, 1)
    assert np.array_equal(out[4], aabb[0]) or pad


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
def test_batched_head_equals_per_crop_head_and_oracle_emulated():
    """k_head_batched (16 crops per workgroup, conv5 from an LDS-staged swizzled crop, FC as hi + lo MFMAs) against the
    per-crop head and against torch on random stage-2 activations: 37 crops = two full batches + a ragged one, an output
    row map, and a device-side crop count that cuts the last batch short."""
    import torch
    import torch.nn.functional as F

    from boxmot_amd.reid_weights import pack_osnet, random_osnet_state_dict

    lib = ctypes.CDLL(str(_build()))
    lib.emu_head_pair.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_void_p]
    sd = random_osnet_state_dict("osnet_x0_25", seed=3)
    blob = pack_osnet(sd)
    n, count = 37, 35
    rng = np.random.default_rng(0)
    act = np.maximum(rng.normal(0.3, 1.0, (n, 128, 128)), 0).astype(np.float16).astype(np.float32)     # post-ReLU, fp16-exact
    rows = rng.permutation(n).astype(np.int32)
    fb = np.full((n, 512), np.nan, np.float32)
    fp = np.full((n, 512), np.nan, np.float32)
    assert lib.emu_head_pair(blob.ctypes.data, blob.size, act.ctypes.data, n, count, rows.ctypes.data, fb.ctypes.data, fp.ctypes.data) == 0
    live = rows[:count]
    dead = rows[count:]
    assert np.isnan(fb[dead]).all() and np.isnan(fp[dead]).all()            # crops beyond the device-side count are not touched
    assert np.isfinite(fb[live]).all()
    assert np.abs(fb[live] - fp[live]).max() < 2e-6                           # same fp16 operands; only the fp32 summation order differs
    with torch.no_grad():
        x = torch.from_numpy(act).permute(0, 2, 1).reshape(n, 128, 16, 8)     # (n, C, H, W)
        sdf = {k: v.float() for k, v in sd.items()}
        y = F.relu(F.batch_norm(F.conv2d(x, sdf["conv5.conv.weight"]), sdf["conv5.bn.running_mean"], sdf["conv5.bn.running_var"],
                                sdf["conv5.bn.weight"], sdf["conv5.bn.bias"], False, 0.0, 1e-5))
        v = F.linear(y.mean((2, 3)), sdf["fc.0.weight"], sdf["fc.0.bias"])
        v = F.relu(F.batch_norm(v, sdf["fc.1.running_mean"], sdf["fc.1.running_var"], sdf["fc.1.weight"], sdf["fc.1.bias"], False, 0.0, 1e-5))
        want = (v / v.norm(dim=1, keepdim=True)).numpy()
    assert np.abs(fb[rows[:count]] - want[:count]).max() < 1e-3


@pytest.mark.parametrize("weights,fused_stem,immediate", [("init", 1, False), ("calib0", 1, False), ("calib1", 0, False), ("calib2", 1, False),
                                                          ("calib0", 1, True)])
@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
def test_fp32_grade_fused_family_emulated_vs_oracle(weights, fused_stem, immediate):
    """fused_stem = 1: crop + resize + stem in one kernel on raw pixel values with the normalisation folded into (hi, lo) weights
    (k_stem_resize_fused_hp, the engine's default); 0: crop kernel + k_stem_hp on (hi, lo) normalised planes (resize_pad path).
    The fp32-grade fused family (reid_hp.hpp, ReID mode 2: fp16 (hi, lo) operand pairs, fp32 image / depthwise / gates) on CPU
    threads, every stored stage and the embeddings against the torch fp32 oracle -- on the reference's own initialisation and on
    BatchNorm-calibrated random networks (the case the fp16-operand family misses by 6-12x): 1e-3 is north_star's tolerance, the
    arithmetic delivers ~1e-5."""
    import torch

    from boxmot_amd.reid_weights import pack_osnet, random_osnet_state_dict, reference_init_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import osnet_forward

    lib = ctypes.CDLL(str(_build(immediate=immediate)))
    lib.emu_reid_forward_hp.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    sd = reference_init_state_dict("osnet_x0_25", seed=0) if weights == "init" else random_osnet_state_dict("osnet_x0_25", seed=int(weights[-1]))
    blob = pack_osnet(sd)
    img = np.random.default_rng(5).integers(0, 255, (480, 641, 3), dtype=np.uint8)
    # a general box, a box clipped at the frame border, an identity-sized and an exact-2x box (the resampler's special cases)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [-12.0, 300.0, 120.5, 500.0], [100, 100, 228, 356], [300, 10, 556, 522]], dtype=np.float32)[: 4 if fused_stem else 2]
    crops = get_crops(boxes, img)
    n = len(boxes)
    feats = np.zeros((n, 512), np.float32)
    shapes = [(2048, 16), (2048, 64), (2048, 64), (512, 64), (512, 96), (512, 96), (128, 96), (128, 128), (128, 128)]
    bufs = [np.zeros((n,) + s, np.float32) for s in shapes]
    ptrs = (ctypes.c_void_p * 9)(*[b.ctypes.data for b in bufs])
    assert lib.emu_reid_forward_hp(blob.ctypes.data, blob.size, img.ctypes.data, 641, 480, boxes.ctypes.data, n, feats.ctypes.data, ptrs,
                                   fused_stem) == 0
    want, st = osnet_forward(sd, torch.from_numpy(crops), return_stages=True)
    names = ["maxpool", "conv2.0", "conv2.1", "conv2.2", "conv3.0", "conv3.1", "conv3.2", "conv4.0", "conv4.1"]
    for nm, buf in zip(names, bufs):
        if nm in ("conv2.0", "conv2.1", "conv3.1"):
            assert not buf.any()            # never stored (hand-over, fused transitions)
            continue
        ref = st[nm].numpy().transpose(0, 2, 3, 1).reshape(buf.shape)
        err = np.abs(buf - ref).max() / np.abs(ref).max()
        print(f"{weights} {nm}: rel max err {err:.2e}")
        assert err < 1e-4, (nm, err)            # fp32 round-off of two summation orders (the fp16-operand family: 3e-3)
    w = want.numpy()
    w = w / np.linalg.norm(w, axis=1, keepdims=True)
    err = np.abs(feats - w).max()
    print(f"{weights} embeddings: max|diff| {err:.2e}")
    assert err < 1e-4                       # north_star tolerance 1e-3; fp32-grade arithmetic
    assert np.allclose(np.linalg.norm(feats, axis=1), 1.0, atol=1e-5)


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
def test_missing_wait_for_an_asynchronous_lds_copy_is_caught_by_the_emulation():
    """Negative control of the deferred-copy emulation: the same fp32-grade kernels with BM_WAIT_VM0 compiled to nothing publish
    their staged weights / source rows through barriers without waiting for them -- the embeddings must come out wrong (the launcher
    poisons LDS).  With the waits in place the same build flags give the oracle's embeddings
    (test_fp32_grade_fused_family_emulated_vs_oracle runs on that library)."""
    import torch

    from boxmot_amd.reid_weights import pack_osnet, random_osnet_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import osnet_forward

    lib = ctypes.CDLL(str(_build(no_vm_wait=True)))
    lib.emu_reid_forward_hp.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    sd = random_osnet_state_dict("osnet_x0_25", seed=0)
    blob = pack_osnet(sd)
    img = np.random.default_rng(5).integers(0, 255, (480, 641, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3]], dtype=np.float32)
    feats = np.zeros((1, 512), np.float32)
    shapes = [(2048, 16), (2048, 64), (2048, 64), (512, 64), (512, 96), (512, 96), (128, 96), (128, 128), (128, 128)]
    bufs = [np.zeros((1,) + s, np.float32) for s in shapes]
    ptrs = (ctypes.c_void_p * 9)(*[b.ctypes.data for b in bufs])
    assert lib.emu_reid_forward_hp(blob.ctypes.data, blob.size, img.ctypes.data, 641, 480, boxes.ctypes.data, 1, feats.ctypes.data, ptrs, 1) == 0
    want = osnet_forward(sd, torch.from_numpy(get_crops(boxes, img))).numpy()
    want = want / np.linalg.norm(want, axis=1, keepdims=True)
    err = np.abs(feats - want).max()
    assert not (err < 1e-3), f"the emulation did not notice the missing waits (max|diff| {err})"


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
def test_barrier_free_layer_loop_and_depthwise_prefetch_are_bit_identical_emulated(tmp_path):
    """The two structural switches of reid_hp.hpp measured in round 5 (profiles/r5_hp_s0_ab.txt; default off): BM_HP_NBR_SYNC -- the
    LightConv layer loop of stages 0 / 1 synchronised by neighbour flags in LDS instead of two workgroup barriers per layer -- and
    BM_HP_DW_PREFETCH -- the depthwise pass's LDS reads issued a row ahead.  On CPU threads both together return the default build's
    embeddings and stored stages BIT FOR BIT; with the flag waits compiled out (EMU_NO_FLAG_WAIT: the negative control) the same
    kernels read rows their neighbours have not written and come out wrong -- the emulation does check the protocol.  The round's
    other switches ride along: conv1 of the first stage-0 block computed once (BM_HP_S0_RECOMP = 0), the depthwise pass instantiated
    per "another layer follows" (BM_HP_DW_STATIC_MORE), and the persistent launch form."""
    import sys
    sys.path.insert(0, str(HERE.parent.parent / "tools"))
    import hp_variant_check as hv
    from boxmot_amd.reid_weights import pack_osnet, random_osnet_state_dict
    blob = pack_osnet(random_osnet_state_dict("osnet_x0_25", seed=0))
    img = np.random.default_rng(5).integers(0, 255, (480, 641, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [-12.0, 300.0, 120.5, 500.0]], dtype=np.float32)
    base = hv.forward(hv.build([], tmp_path / "base.so"), blob, img, boxes)
    both = hv.forward(hv.build(["-DBM_HP_NBR_SYNC=1", "-DBM_HP_DW_PREFETCH=1"], tmp_path / "both.so"), blob, img, boxes)
    assert np.array_equal(base[0], both[0]) and all(np.array_equal(a, b) for a, b in zip(base[1], both[1]))
    assert np.allclose(np.linalg.norm(both[0], axis=1), 1.0, atol=1e-5)
    broken = hv.forward(hv.build(["-DBM_HP_NBR_SYNC=1", "-DEMU_NO_FLAG_WAIT=1"], tmp_path / "nowait.so"), blob, img, boxes)
    assert not np.array_equal(base[0], broken[0])
    # the persistent launch form (BlkLinkHP::n_crops, BOXMOT_HIP_REID_PERSIST; profiles/r5_hp_persist_ab.txt): ONE workgroup runs both
    # crops in a loop -- LDS is poisoned once per launch only, so whatever a crop leaves behind is what the next one finds
    import os
    os.environ["EMU_HP_PERSIST_GRID"] = "1"
    try:
        pers = hv.forward(hv.build(["-DBM_HP_PERSIST=1", "-DBM_HP_S0_RECOMP=0", "-DBM_HP_DW_STATIC_MORE=1"], tmp_path / "pers.so"), blob, img, boxes)
    finally:
        os.environ.pop("EMU_HP_PERSIST_GRID", None)
    assert np.array_equal(base[0], pers[0]) and all(np.array_equal(a, b) for a, b in zip(base[1], pers[1]))
