"""Runs the DEVICE kernel source (boxmot_amd/csrc/botsort_step.hpp, unchanged) on CPU threads
through tests/host_emu and compares it with the oracle frame by frame; a second pass runs it
under AddressSanitizer/UBSan.  This is test infrastructure for the kernel logic (index lists,
LAP, Kalman order) -- the shipped library has no CPU path."""
import numpy as np
import pytest

from boxmot_amd.scenario import Scenario, stress_frames
from emu_util import EmuBotSort
from oracle.botsort import DEFAULTS, BotSortOracle


def _run(frames, dim, cap, nd, sanitize=False, dense=False, warps=None, threads=64, **kw):
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    orc, emu = BotSortOracle(**kw), EmuBotSort(cfg, cap=cap, nd=nd, dim=dim, sanitize=sanitize, dense=dense, threads=threads)
    try:
        for t, (d, e) in enumerate(frames):
            w = None if warps is None else warps[t]
            want = orc.update(d.copy(), None, e.copy(), warp=w)
            got = emu.update(d, e, warp=w)
            assert got.shape == want.shape, t
            assert np.array_equal(got[:, 4:], want[:, 4:]), t
            assert np.allclose(got[:, :4], want[:, :4], rtol=0, atol=1e-4), t
        od = orc.dump()
        for which, key in ((0, "active"), (1, "lost")):
            d = emu.dump(which)
            assert np.array_equal(d["ints"][:, 0], od[key]["id"])
            assert np.array_equal(d["ints"][:, 1], od[key]["state"])
            assert np.array_equal(d["ints"][:, 3], od[key]["frame_id"])
            if d["n"]:
                ref = np.concatenate([od[key]["mean"], od[key]["cov"].reshape(-1, 64)], 1)
                assert np.allclose(d["kf"], ref, rtol=1e-9, atol=1e-12)
        assert emu.dump(0)["counters"][1] == od["id_count"]
    finally:
        emu.close()


@pytest.mark.parametrize("kw", [{}, dict(track_buffer=5, removed_stracks_buffer=3, fuse_first_associate=True),
                                dict(with_reid=False)])
def test_emulated_kernel_matches_oracle_stress(kw):
    _run(stress_frames(80, seed=7), 32, 128, 64, **kw)


def test_emulated_kernel_four_wavefronts():
    """The same parity with a 256-thread workgroup: cross-wavefront reductions, stream compaction and the assignment solver
    with more than one wavefront."""
    _run(stress_frames(50, seed=17), 32, 128, 64, threads=256)


def test_emulated_kernel_applies_camera_warp():
    """STrack.multi_gmc on the device state (botsort_track.py:117-132), warp every frame and every third frame."""
    from boxmot_amd.scenario import camera_warps
    _run(stress_frames(60, seed=7), 32, 128, 64, warps=camera_warps(60, seed=7))
    _run(stress_frames(60, seed=11), 32, 128, 64, warps=camera_warps(60, seed=3, every=3), fuse_first_associate=True)


def test_emulated_kernel_dense_cosine_path():
    """Same kernel built with the sparse-pair limit at 0: every frame takes the LDS-tiled dense contraction."""
    _run(stress_frames(60, seed=7), 32, 128, 64, dense=True)
    _run(stress_frames(40, seed=9), 32, 128, 64, dense=True, fuse_first_associate=True)


def test_emulated_kernel_matches_oracle_c2_shape():
    sc = Scenario(64, 256, emb_dim=64, random_image=False)
    _run(sc.frames(8), 64, 512, 256)


def test_emulated_kernel_clean_under_asan():
    import ctypes.util
    import os
    if not os.environ.get("LD_PRELOAD") and ctypes.util.find_library("asan") is None:
        pytest.skip("libasan not available")
    import subprocess
    import sys
    code = ("import sys; sys.path[:0]=['.', 'tests']\n"
            "from test_kernel_emu import _run\n"
            "from boxmot_amd.scenario import stress_frames\n"
            "_run(stress_frames(25, seed=7), 32, 64, 32, sanitize=True)\n"
            # the ByteTrack mode of the same kernel (XYAH filter, per-slot removed flag)
            "import numpy as np\n"
            "from common import bytetrack_device_config\n"
            "from emu_util import EmuBotSort\n"
            "from oracle.bytetrack import ByteTrackOracle\n"
            "emu, orc = EmuBotSort(bytetrack_device_config(), cap=64, nd=32, dim=1, sanitize=True), ByteTrackOracle()\n"
            "for d, _ in stress_frames(20, seed=7):\n"
            "    g, w = emu.update(d[:32], None), orc.update(d[:32].copy())\n"
            "    assert g.shape == w.shape and np.array_equal(g[:, 4:], w[:, 4:])\n"
            "emu.close()\nprint('ASAN-OK')\n")
    import glob
    libasan = sorted(glob.glob("/usr/lib/gcc/x86_64-linux-gnu/*/libasan.so"))
    if not libasan:
        pytest.skip("libasan.so not found")
    env = dict(os.environ, LD_PRELOAD=libasan[-1], ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert "ASAN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize("threads", [64, 256])
def test_block_argmin_returns_infinite_candidates_and_breaks_ties_by_index(threads):
    """block_prims.hpp: the workgroup (value, index) minimum on the DPP path.  Round-4 advisor finding: with DBL_MAX as the filler of
    threads without a candidate, a candidate whose value is +inf lost to the filler and came back as "none" (the shuffle version had
    returned it; only lap_solve's `best >= 1e300` guard hid that).  The filler is +inf now: any candidate up to +inf is returned,
    ties go to the lowest index, no candidate at all gives -1.  (NaN stays outside the contract: the callers compare with `<` before
    they offer a value.)"""
    import ctypes

    import emu_util
    lib = ctypes.CDLL(str(emu_util.build(threads=threads)))
    lib.emu_block_argmin.argtypes = [ctypes.c_void_p] * 4
    assert lib.emu_nthr() == threads

    def argmin(v, idx):
        v, idx = np.ascontiguousarray(v, np.float64), np.ascontiguousarray(idx, np.int32)
        ov, oi = ctypes.c_double(0), ctypes.c_int(0)
        lib.emu_block_argmin(v.ctypes.data, idx.ctypes.data, ctypes.byref(ov), ctypes.byref(oi))
        return ov.value, oi.value
    rng = np.random.default_rng(0)
    none = np.full(threads, -1, np.int32)
    assert argmin(np.zeros(threads), none)[1] == -1
    for _ in range(20):
        v = rng.uniform(0, 10, threads).round(1)               # many exact ties
        idx = np.where(rng.uniform(size=threads) < 0.4, rng.permutation(threads) + 5, -1).astype(np.int32)
        if (idx >= 0).any():
            cand = [(v[t], idx[t]) for t in range(threads) if idx[t] >= 0]
            assert argmin(v, idx) == (float(min(cand)[0]), int(min(cand)[1]))
    # the only candidates are +inf (one per wave and several in one wave): the lowest index among them, value +inf
    v, idx = np.full(threads, 3.0), none.copy()
    for t, i in ((5, 40), (threads - 3, 7), (9, 12)):
        v[t], idx[t] = np.inf, i
    assert argmin(v, idx) == (np.inf, 7)
    # a finite candidate beats infinite ones; values at and above DBL_MAX are ordinary candidates
    v[20], idx[20] = 1.7976931348623157e308, 99
    assert argmin(v, idx) == (1.7976931348623157e308, 99)
