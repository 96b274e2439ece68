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
