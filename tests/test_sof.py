"""Sparse-optical-flow camera-motion estimation: the oracle (oracle/sof.py) on known motions, against the REFERENCE's ``SOF`` class
driven with the oracle's restatements of the OpenCV calls (the control flow is the reference's own), on the cases the reference's
unit tests hold (tests/unit/test_cmcs_u.py), on the MOT17-mini frames (golden made by tests/golden/make_sof_golden.py), and the
device kernels (boxmot_amd/csrc/cmc_sof.hpp, kernel sequence included) run on CPU threads against the oracle."""
import ctypes
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from common import GOLDEN

HERE = Path(__file__).resolve().parent / "host_emu"
CLANG = shutil.which("clang++", path="/opt/rocm/lib/llvm/bin") or shutil.which("clang++")


def _textured(h, w, seed=0, sigma=5):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    base = gaussian_filter(rng.integers(0, 255, (h, w, 3)).astype(np.float32), (sigma, sigma, 0))
    return ((base - base.min()) / (base.max() - base.min()) * 255).astype(np.uint8)


def test_oracle_recovers_known_translations():
    from oracle.sof import SofOracle
    base = _textured(700, 1100, sigma=6)
    for dx, dy in ((-20, 9), (6, -13), (0, 0)):
        e = SofOracle()
        assert np.array_equal(e.apply(base[50:590, 70:1030]), np.eye(2, 3, dtype=np.float32))      # first call: identity, keypoints stored
        assert e.initialized and len(e.prev_keypoints) >= 100
        w = e.apply(base[50 + dy:590 + dy, 70 + dx:1030 + dx])
        assert w.dtype == np.float32 and w.shape == (2, 3)
        # curr(x, y) = prev(x + dx, y + dy): a point of prev at p is at p - (dx, dy) in curr
        assert abs(w[0, 2] + dx) < 0.4 and abs(w[1, 2] + dy) < 0.4, (dx, dy, w)
        assert abs(w[0, 0] - 1) < 2e-3 and abs(w[1, 1] - 1) < 2e-3 and abs(w[0, 1]) < 2e-3 and w[0, 1] == -w[1, 0] and w[0, 0] == w[1, 1]
        assert e.last["inliers"] >= 0.9 * e.last["matches"]


def test_oracle_partial_affine_fit_and_rng_sequence():
    """estimateAffinePartial2D: an exact similarity with 30 % gross outliers is recovered by RANSAC + the refinement; the index
    draws are cv::RNG's multiply-with-carry recurrence from the all-ones state."""
    from oracle.sof import CvRng, estimate_affine_partial_2d
    r = CvRng()
    s0 = 0xFFFFFFFFFFFFFFFF
    for _ in range(5):
        s0 = ((s0 & 0xFFFFFFFF) * 4164903690 + (s0 >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert r.next() == s0 & 0xFFFFFFFF
    assert 0 <= CvRng().uniform(0, 7) < 7 and CvRng().uniform(3, 3) == 3
    rng = np.random.default_rng(5)
    src = rng.uniform(0, 280, (200, 2)).astype(np.float32)
    a, b, tx, ty = 1.02 * np.cos(0.03), 1.02 * np.sin(0.03), 4.5, -2.25
    dst = np.stack([a * src[:, 0] - b * src[:, 1] + tx, b * src[:, 0] + a * src[:, 1] + ty], 1).astype(np.float32)
    bad = rng.choice(200, 60, replace=False)
    dst[bad] += rng.uniform(20, 60, (60, 2)).astype(np.float32)
    H, inl = estimate_affine_partial_2d(src, dst)
    assert int(inl.sum()) == 140 and not inl[bad].any()
    assert np.allclose(H, [[a, -b, tx], [b, a, ty]], atol=2e-4)
    H2, inl2 = estimate_affine_partial_2d(src[:1], dst[:1])
    assert H2 is None and not inl2.any()


def test_oracle_cases_of_the_reference_unit_tests():
    """tests/unit/test_cmcs_u.py:41-75: empty detections on a black frame -> identity; the detection box is masked out."""
    from oracle.sof import SofOracle, generate_mask
    assert np.array_equal(SofOracle().apply(np.zeros((100, 100, 3), np.uint8), np.array([])), np.eye(2, 3, dtype=np.float32))
    m = generate_mask(15, 15, np.array([[0, 0, 50, 50]], dtype=np.float32), 0.15)
    assert m[3, 3] == 0 and m[10, 10] == 255 and m[0, 10] == 255 and m[14, 10] == 0       # rows int(0.02 h) .. int(0.98 h) - 1


def _sequence(rows=360, cols=520, seed=2):
    """frames + detections exercising every branch of sof.py:55-129: black frames (no keypoints), initialisation, tracking with
    moving boxes, a black frame in the middle (every point fails -> re-detection), recovery."""
    base = _textured(rows + 120, cols + 160, seed=seed, sigma=4)
    black = np.zeros((rows, cols, 3), np.uint8)
    shifts = [None, (0, 0), (-9, 4), (3, -7), None, (5, 5), (12, -3), (12, -3)]
    out = []
    for t, sh in enumerate(shifts):
        fr = black if sh is None else np.ascontiguousarray(base[60 + sh[1]:60 + sh[1] + rows, 80 + sh[0]:80 + sh[0] + cols])
        dets = np.array([[40 + 6 * t, 50, 160 + 6 * t, 300, 0.9, 0], [300, 20 + 4 * t, 380, 200 + 4 * t, 0.8, 1]], dtype=np.float32)
        out.append((fr, dets if t % 3 else dets[:, :4]))
    return out


def test_oracle_equals_the_reference_class_driven_with_the_restated_opencv_calls():
    from oracle import ref_harness
    if not ref_harness.reference_available():
        pytest.skip("needs /root/reference")
    from oracle.sof import SofOracle
    ref, orc = ref_harness.load_sof()(), SofOracle()
    modes = []
    for t, (fr, dets) in enumerate(_sequence()):
        want, got = ref.apply(fr, dets), orc.apply(fr, dets)
        assert want.dtype == got.dtype == np.float32 and np.array_equal(want, got), t
        assert ref.initialized == orc.initialized, t
        rk = None if ref.prev_keypoints is None else np.asarray(ref.prev_keypoints).reshape(-1, 2)
        assert (rk is None) == (orc.prev_keypoints is None) and (rk is None or np.array_equal(rk, orc.prev_keypoints)), t
        modes.append((bool(orc.initialized), not np.array_equal(got, np.eye(2, 3, dtype=np.float32)), "status" in orc.last, "matches" in orc.last))
    # the sequence did visit: not initialised; initialised; estimated warps; tracking INTO the black frame (no new corners: the tracked
    # points are kept, sof.py:123-125); tracking OUT of it (no texture in the template: every point fails -> _reset, sof.py:95-98); recovery
    assert modes[0] == (False, False, False, False) and modes[1] == (True, False, False, False)
    assert modes[2] == modes[3] == (True, True, True, True) and modes[4][2:] == (True, True)
    assert modes[5] == (True, False, True, False) and modes[6] == (True, True, True, True)


def test_oracle_on_the_reference_mot17_frames_golden():
    from oracle.sof import SofOracle
    g = np.load(GOLDEN / "sof_golden.npz")
    for seq in ("02", "04"):
        small = np.load(GOLDEN / "ecc_golden.npz")[f"small_{seq}"]
        o = SofOracle(scale=1.0)
        for k in range(len(small)):
            w = o.apply(np.repeat(small[k][:, :, None], 3, axis=2), g[f"dets_{seq}"][k])
            assert np.array_equal(w, g[f"warp_{seq}"][k]), (seq, k)
            assert len(o.prev_keypoints) == int(g[f"nkps_{seq}"][k])
        assert np.array_equal(o.prev_keypoints, g[f"kps_last_{seq}"])


# ---------------------------------------------------------------------------------------------------------------------------
def _emu():
    out = HERE / "libemu_sof.so"
    csrc = HERE.parent.parent / "boxmot_amd" / "csrc"
    deps = [HERE / "emu_sof.cpp", HERE / "hip_shim.hpp", csrc / "cmc_sof.hpp", csrc / "cmc_ecc.hpp", csrc / "reid_kernels_v1.hpp", csrc / "kernel_macros.hpp"]
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        subprocess.check_call([CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-ffp-contract=off",
                               "-o", str(out), str(HERE / "emu_sof.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.emu_sof_create.restype = ctypes.c_void_p
    lib.emu_sof_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    lib.emu_sof_destroy.argtypes = [ctypes.c_void_p]
    lib.emu_sof_apply.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.emu_sof_points.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    return lib


class EmuSof:
    """The emulated device estimator with ``apply(img, dets)`` (one stream)."""

    def __init__(self, lib, rows, cols, scale=0.15):
        self.lib, self.h = lib, lib.emu_sof_create(rows, cols, scale, 8, 0.2, 3.0)
        self.state = np.zeros(12, np.int32)

    def apply(self, img, dets=None):
        img = np.ascontiguousarray(img)
        d = None if dets is None or not np.size(dets) else np.ascontiguousarray(np.asarray(dets)[:, :4], dtype=np.float32)
        warp = np.zeros(6)
        assert self.lib.emu_sof_apply(self.h, img.ctypes.data, None if d is None else d.ctypes.data, 0 if d is None else len(d), 4,
                                      warp.ctypes.data, self.state.ctypes.data) == 0
        return warp.reshape(2, 3)

    def points(self, which=0):
        buf = np.zeros((1000, 2), np.float32)
        n = self.lib.emu_sof_points(self.h, which, buf.ctypes.data, 1000)
        return buf[:n] if which == 0 else buf

    def status(self):
        buf = np.zeros(1000, np.float32)
        self.lib.emu_sof_points(self.h, 2, buf.ctypes.data, 1000)
        return buf.astype(np.uint8)

    def close(self):
        self.lib.emu_sof_destroy(self.h)


def check_against_oracle(est, orc, frames, tol=2e-6):
    """frame by frame: warp, the state word, the tracked points and their status, the keypoints kept for the next frame"""
    for t, (fr, dets) in enumerate(frames):
        want, got = orc.apply(fr, dets), est.apply(fr, dets)
        assert np.allclose(got, want, rtol=0, atol=tol * max(1.0, float(np.abs(want).max()))), (t, got, want)
        kp = est.points(0)
        ok = orc.prev_keypoints
        assert len(kp) == (0 if ok is None else len(ok)), t
        if ok is not None and len(ok):
            assert np.abs(kp - ok).max() <= 1e-4, t                    # corners are integers; refined / tracked ones fp32
        if "status" in orc.last:
            n = len(orc.last["status"])
            assert np.array_equal(est.status()[:n], orc.last["status"]), t
            sel = orc.last["status"] == 1
            assert np.abs(est.points(1)[:n][sel] - orc.last["next"][sel]).max() <= 1e-4 if sel.any() else True, t
        if "inliers" in orc.last:
            assert (int(est.state[3]), int(est.state[4])) == (orc.last["matches"], orc.last["inliers"]), t
        assert bool(est.state[0]) == bool(orc.initialized), t


@pytest.mark.skipif(CLANG is None, reason="needs a host clang")
def test_device_kernels_emulated_vs_oracle_every_branch():
    from oracle.sof import SofOracle
    frames = _sequence(rows=300, cols=420, seed=7)
    est = EmuSof(_emu(), 300, 420)
    check_against_oracle(est, SofOracle(), frames)
    est.close()


@pytest.mark.skipif(CLANG is None, reason="needs a host clang")
def test_device_kernels_emulated_on_mot17_frames():
    from oracle.sof import SofOracle
    g = np.load(GOLDEN / "sof_golden.npz")
    small = np.load(GOLDEN / "ecc_golden.npz")["small_02"]
    frames = [(np.repeat(small[k][:, :, None], 3, axis=2), g["dets_02"][k]) for k in range(2)]
    est = EmuSof(_emu(), small.shape[1], small.shape[2], scale=1.0)
    check_against_oracle(est, SofOracle(scale=1.0), frames)
    est.close()
