"""Runs the fp32-grade wide-OSNet kernels (boxmot_amd/csrc/osnet_wide_hp_kernels.hpp, device source unchanged, in the launch order of
wide_hp_forward -- the function the engine calls) on CPU threads with the emulated MFMA of tests/host_emu and compares the block
outputs and the embedding with the torch fp32 oracle, on BatchNorm-CALIBRATED random weights (the case that separates fp32-grade
arithmetic from fp16 operands) of reduced architectures whose widths are multiples of 32 like osnet_x1_0's.  Not a product path."""
import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent / "host_emu"
CLANG = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")


def _build():
    out = HERE / "libemu_wide_hp.so"
    csrc = HERE.parent.parent / "boxmot_amd" / "csrc"
    deps = [HERE / "emu_wide_hp.cpp", HERE / "hip_shim.hpp"] + [csrc / f for f in (
        "osnet_wide_hp_kernels.hpp", "osnet_wide_hp.hpp", "osnet_wide_hp_pack.hpp", "osnet_wide_kernels.hpp", "gemm_f16.hpp", "reid_hp.hpp",
        "reid_fused.hpp", "reid_hp_pack.hpp", "reid_pack.hpp", "reid_layout.hpp", "kernel_macros.hpp")]
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread",
                               "-ffp-contract=off", "-DEMU_DEFER_GLDS=1",       # global -> LDS copies land at the issuing thread's BM_WAIT_VM0 (hip_shim.hpp)
                               "-o", str(out), str(HERE / "emu_wide_hp.cpp")])
    return out


@pytest.mark.skipif(CLANG is None, reason="needs a host clang with _Float16")
@pytest.mark.parametrize("channels,weights", [((32, 128, 256, 384), "calib"), ((64, 256, 128, 128), "init"),
                                              # widths the family takes as a zero-padded copy (reid_layout.hpp: osnet_pad_weights): osnet_x0_75's
                                              # (48, 192, 288, 384) itself, and a reduced net with osnet_x0_5's 48-wide middle stage and a
                                              # projection that the padding turns into cin == cout
                                              ((48, 192, 288, 384), "calib"), ((32, 128, 192, 256), "calib")])
def test_wide_hp_kernels_emulated_vs_oracle(channels, weights):
    import torch

    from boxmot_amd.reid_weights import pack_osnet, random_osnet_state_dict, reference_init_state_dict
    from oracle.crops import get_crops
    from oracle.osnet import osnet_forward

    lib = ctypes.CDLL(str(_build()))
    lib.emu_wide_hp_forward.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    sd = random_osnet_state_dict(channels, seed=0) if weights == "calib" else reference_init_state_dict(channels, seed=0)
    blob = pack_osnet(sd)
    assert tuple(blob[:16].view(np.int32)[1:5]) == channels
    img = np.random.default_rng(5).integers(0, 255, (480, 640, 3), dtype=np.uint8)
    boxes = np.array([[30.2, 40.7, 90.1, 200.3], [300, 100, 420, 380]], dtype=np.float32)
    crops = get_crops(boxes, img)
    nhwc = np.ascontiguousarray(np.transpose(crops, (0, 2, 3, 1)))
    n = len(boxes)
    feats = np.zeros((n + 1, 512), np.float32)
    rows = np.array([2, 0], dtype=np.int32)
    shapes = [(2048, channels[1])] * 2 + [(512, channels[2])] * 2 + [(128, channels[3])] * 2
    bufs = [np.zeros((n,) + s, np.float32) for s in shapes]
    ptrs = (ctypes.c_void_p * 6)(*[b.ctypes.data for b in bufs])
    assert lib.emu_wide_hp_forward(blob.ctypes.data, blob.size, nhwc.ctypes.data, n, rows.ctypes.data, feats.ctypes.data, ptrs) == 0
    want, st = osnet_forward(sd, torch.from_numpy(crops), return_stages=True)
    for nm, buf in zip(["conv2.0", "conv2.1", "conv3.0", "conv3.1", "conv4.0", "conv4.1"], bufs):
        ref = st[nm].permute(0, 2, 3, 1).reshape(buf.shape).numpy()
        err = np.abs(buf - ref).max() / max(np.abs(ref).max(), 1e-6)
        print(f"{nm}: max|diff| / max|ref| = {err:.2e}")
        assert err < 1e-4, (nm, err)
    want = want.numpy()
    want = want / np.linalg.norm(want, axis=1, keepdims=True)
    got = feats[rows]
    err = np.abs(got - want).max()
    print(f"wide OSNet fp32-grade, emulated {channels} ({weights}): embedding max|diff| = {err:.2e}")
    assert err < 2e-5
    assert not feats[1].any()
