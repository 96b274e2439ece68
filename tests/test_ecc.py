"""ECC camera-motion estimation: the oracle (oracle/ecc.py) on known shifts and on the reference's MOT17-mini frames (golden
fixture made here from the reference's own jpg assets, tests/golden/make_ecc_golden.py), and the device kernels
(boxmot_amd/csrc/cmc_ecc.hpp) run on CPU threads against the oracle."""
import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from common import GOLDEN

HERE = Path(__file__).resolve().parent / "host_emu"
CLANG = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")


def _textured(h, w, seed=0, sigma=8):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    base = gaussian_filter(rng.integers(0, 255, (h, w, 3)).astype(np.float32), (sigma, sigma, 0))
    return ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)


def test_oracle_recovers_known_translations():
    from oracle.ecc import EccOracle
    base = _textured(700, 1100)
    for dx, dy in ((-20, 9), (6, -13), (0, 0)):
        e = EccOracle()
        assert np.array_equal(e.apply(base[50:590, 70:1030]), np.eye(2, 3, dtype=np.float32))      # first call: identity, stores the frame
        w = e.apply(base[50 + dy:590 + dy, 70 + dx:1030 + dx])
        assert w.dtype == np.float32 and w.shape == (2, 3) and w[0, 0] == 1 and w[1, 1] == 1 and w[0, 1] == 0
        # curr(x, y) = prev(x + dx, y + dy)  ->  curr(x - dx, y - dy) = prev(x, y): the warp that maps template to input is (-dx, -dy)
        assert abs(w[0, 2] + dx) < 0.35 and abs(w[1, 2] + dy) < 0.35, (dx, dy, w)


def test_oracle_identity_on_uncorrelated_frames_like_the_reference():
    """OpenCV's StsNoConv exits -> ecc.py:67-76 returns the identity and keeps tracking from the new frame."""
    from oracle.ecc import EccOracle
    e = EccOracle()
    e.apply(_textured(300, 400, seed=1))
    flat = np.full((300, 400, 3), 90, np.uint8)                   # zero variance: rho is NaN
    assert np.array_equal(e.apply(flat), np.eye(2, 3, dtype=np.float32))


def test_oracle_on_the_reference_mot17_frames_golden():
    from oracle.ecc import preprocess, find_transform_ecc_translation
    g = np.load(GOLDEN / "ecc_golden.npz")
    for seq in ("02", "04"):
        small = g[f"small_{seq}"]
        for k in range(len(small) - 1):
            rho, (tx, ty), it = find_transform_ecc_translation(small[k], small[k + 1])
            assert (tx, ty, it) == (float(g[f"warp_{seq}"][k, 0]), float(g[f"warp_{seq}"][k, 1]), int(g[f"iters_{seq}"][k]))


def _emu():
    out = HERE / "libemu_ecc.so"
    csrc = HERE.parent.parent / "boxmot_amd" / "csrc"
    deps = [HERE / "emu_ecc.cpp", HERE / "hip_shim.hpp", csrc / "cmc_ecc.hpp", csrc / "reid_kernels_v1.hpp", csrc / "kernel_macros.hpp"]
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
                               "-o", str(out), str(HERE / "emu_ecc.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.emu_ecc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


@pytest.mark.skipif(CLANG is None, reason="needs a host clang")
@pytest.mark.parametrize("shift", [(-20, 9), (3, -2), (0, 0)])
def test_device_kernels_emulated_vs_oracle(shift):
    from oracle.ecc import EccOracle, preprocess
    lib = _emu()
    dx, dy = shift
    base = _textured(420, 640, seed=3, sigma=5)
    prev = np.ascontiguousarray(base[30:390, 40:600])
    curr = np.ascontiguousarray(base[30 + dy:390 + dy, 40 + dx:600 + dx])
    warp, info = np.zeros(6), np.zeros(2, np.int32)
    h, w = int(np.rint(360 * 0.15)), int(np.rint(560 * 0.15))
    small = np.zeros((2, h, w), np.float32)
    assert lib.emu_ecc(prev.ctypes.data, curr.ctypes.data, 360, 560, 0.15, 1e-5, 100, warp.ctypes.data, info.ctypes.data, small.ctypes.data) == 0
    assert np.array_equal(small[0], preprocess(prev).astype(np.float32)) and np.array_equal(small[1], preprocess(curr).astype(np.float32))
    e = EccOracle()
    e.apply(prev)
    want = e.apply(curr)
    print(f"shift {shift}: device {warp[2]:.5f}, {warp[5]:.5f} ({info[1]} iterations), oracle {want[0, 2]:.5f}, {want[1, 2]:.5f} ({e.last_iterations})")
    assert info[0] == 1 and info[1] == e.last_iterations
    assert abs(warp[2] - want[0, 2]) < 1e-3 and abs(warp[5] - want[1, 2]) < 1e-3
    assert (warp[0], warp[1], warp[3], warp[4]) == (1.0, 0.0, 0.0, 1.0)


@pytest.mark.skipif(CLANG is None, reason="needs a host clang")
def test_device_kernels_emulated_no_convergence_is_identity():
    lib = _emu()
    prev = _textured(200, 320, seed=4)
    curr = np.full((200, 320, 3), 77, np.uint8)
    warp, info = np.ones(6), np.zeros(2, np.int32)
    assert lib.emu_ecc(prev.ctypes.data, curr.ctypes.data, 200, 320, 0.15, 1e-5, 100, warp.ctypes.data, info.ctypes.data, None) == 0
    assert info[0] == 0 and np.array_equal(warp, [1, 0, 0, 0, 1, 0])
