"""Runs the DeepOCSORT DEVICE step (boxmot_amd/csrc/deepocsort_step.hpp, unchanged) on CPU threads through
tests/host_emu and compares it with the oracle frame by frame (rows, ids, ages, fp64 filter state, embeddings).
Test infrastructure for the kernel logic -- the shipped library has no CPU path."""
import numpy as np
import pytest

from boxmot_amd.scenario import Scenario, stress_frames
from emu_util import EmuDeepOcSort
from oracle.deepocsort import DEFAULTS, DeepOcSortOracle


def _run(frames, dim, cap, nd, sanitize=False, warps=None, threads=64, **kw):
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    orc, emu = DeepOcSortOracle(**kw), EmuDeepOcSort(cfg, cap=cap, nd=nd, dim=dim, sanitize=sanitize, threads=threads)
    try:
        for t, (d, e) in enumerate(frames):
            w = None if warps is None else warps[t]
            want = orc.update(d.copy(), None, e.copy(), warp=w).reshape(-1, 8)
            got = emu.update(d, e, warp=w)
            assert got.shape == want.shape, t
            assert np.array_equal(got[:, 4:], want[:, 4:]), t
            assert np.allclose(got[:, :4], want[:, :4], rtol=0, atol=1e-4), t
        od, d = orc.dump(), emu.dump()
        assert np.array_equal(d["ints"][:, 0], od["id"])
        assert np.array_equal(d["ints"][:, 1], od["age"])
        assert np.array_equal(d["ints"][:, 2], od["time_since_update"])
        assert np.array_equal(d["ints"][:, 3], od["hit_streak"])
        if d["n"]:
            x = d["kf"][:, :7]
            P = d["kf"][:, 8:].reshape(-1, 8, 8)[:, :7, :7]
            assert np.allclose(x, od["x"], rtol=1e-9, atol=1e-9)
            assert np.allclose(P, od["P"], rtol=1e-8, atol=1e-9)
            if not cfg["embedding_off"]:
                for r, emb in enumerate(od["emb"]):
                    assert np.allclose(d["emb"][r], emb, rtol=0, atol=1e-6)
        assert d["counters"][1] + 1 == od["count"]
    finally:
        emu.close()


@pytest.mark.parametrize("kw,seed", [({}, 7), (dict(max_age=5, min_hits=1), 11), (dict(aw_off=True, inertia=0.4, w_association_emb=0.75), 3),
                                     (dict(embedding_off=True), 5)])
def test_emulated_deepocsort_matches_oracle_stress(kw, seed):
    _run(stress_frames(60, seed=seed), 32, 128, 64, **kw)


def _run_oc(frames, **kw):
    """OC-SORT (appearance off, optional BYTE round) on the emulated device step against OcSortOracle."""
    from oracle.deepocsort import OcSortOracle
    cfg = {**DEFAULTS, **{k: v for k, v in kw.items() if k in DEFAULTS}, "embedding_off": 1, "use_byte": int(kw.get("use_byte", False)),
           "min_conf": kw.get("min_conf", 0.1)}
    orc, emu = OcSortOracle(**kw), EmuDeepOcSort(cfg, cap=128, nd=64, dim=1)
    try:
        for t, (d, _) in enumerate(frames):
            want = np.asarray(orc.update(d.copy()), dtype=np.float32).reshape(-1, 8)
            got = emu.update(d, None)
            assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
            assert np.allclose(got[:, :4], want[:, :4], rtol=0, atol=1e-4), t
        od, dd = orc.dump(), emu.dump()
        assert np.array_equal(dd["ints"][:, 0], od["id"]) and np.array_equal(dd["ints"][:, 3], od["hit_streak"])
        if dd["n"]:
            assert np.allclose(dd["kf"][:, :7], od["x"], rtol=1e-9, atol=1e-9)
    finally:
        emu.close()


@pytest.mark.parametrize("asso_func,thr", [("giou", 0.6), ("diou", 0.6), ("ciou", 0.6), ("hmiou", 0.3), ("centroid", 0.9)])
def test_emulated_deepocsort_association_functions(asso_func, thr):
    """BaseTracker's asso_func (iou.py:118-423) on the device step: every "iou" matrix of the frame (first association, OC-SORT's
    BYTE round, the recovery round) comes from the named function; rows, ids and filter state against the oracle, whose functions
    are pinned bit for bit on the reference classes (tests/test_oracle_vs_reference.py)."""
    _run(stress_frames(60, seed=9), 32, 128, 64, asso_func=asso_func, iou_threshold=thr, frame_wh=(640, 480))
    _run_oc(stress_frames(60, seed=10), asso_func=asso_func, iou_threshold=thr, frame_wh=(640, 480), use_byte=True)


def test_emulated_deepocsort_four_wavefronts():
    """The same parity with a 256-thread workgroup: the wave-per-row cost loops, the cross-wavefront reductions and the solver
    with more than one wavefront."""
    _run(stress_frames(50, seed=13, max_objects=30), 32, 128, 64, threads=256)


@pytest.mark.parametrize("seed", [22, 24, 28])
def test_emulated_deepocsort_tie_prone_scenes(seed):
    """Crowded births with more detections than tracks: the assignment has several optima (zero-cost pairs), and
    which detections end up "never assigned" decides the id order of the new tracks.  The reference's choice comes
    from lapx (unavailable, parity unpinned); here the oracle is run with the device solver's tie rule so that
    everything else -- costs, filters, recovery round, bookkeeping -- is still compared exactly."""
    frames = stress_frames(80, seed=seed, max_objects=30)
    _run(frames, 32, 128, 64)
    _run(frames, 32, 128, 64, max_age=8, min_hits=2, iou_threshold=0.2)


def test_emulated_deepocsort_camera_motion_correction():
    """apply_affine_correction (deepocsort.py:190-209, xysr.py:311-366) incl. the doubly-transformed newest observation
    and the frozen filter copy of unobserved tracks."""
    from boxmot_amd.scenario import camera_warps
    _run(stress_frames(70, seed=7), 32, 128, 64, warps=camera_warps(70, seed=7))
    _run(stress_frames(60, seed=11), 32, 128, 64, warps=camera_warps(60, seed=3, every=2), max_age=6, min_hits=1)


def test_emulated_deepocsort_c2_shape():
    sc = Scenario(64, 256, emb_dim=64, random_image=False)
    _run(sc.frames(8), 64, 512, 256)


def test_emulated_deepocsort_c3_shape():
    """BASELINE configuration 3's shape -- 128 detections, 512 live tracks, eight wavefronts -- for ten frames: the assignment is
    a 640 x 640 extended problem per association round (fibers make this a seconds-long CPU test)."""
    sc = Scenario(128, 512, emb_dim=64, random_image=False)
    _run(sc.frames(10), 64, 1024, 512, threads=512)


def test_emulated_kernels_clean_under_asan():
    """Same device source under AddressSanitizer / UBSan (index lists, LDS carving, scratch sizing)."""
    import ctypes.util
    import glob
    import os
    import subprocess
    import sys
    libasan = sorted(glob.glob("/usr/lib/gcc/x86_64-linux-gnu/*/libasan.so"))
    if not libasan:
        pytest.skip("libasan.so not found")
    code = ("import sys; sys.path[:0]=['.', 'tests']\n"
            "from test_docs_emu import _run\n"
            "from boxmot_amd.scenario import stress_frames\n"
            "_run(stress_frames(14, seed=7), 32, 64, 32, sanitize=True)\n"
            # OC-SORT's BYTE round (second detection list appended behind the kept one)
            "import numpy as np\n"
            "from emu_util import EmuDeepOcSort\n"
            "from oracle.deepocsort import DEFAULTS, OcSortOracle\n"
            "emu = EmuDeepOcSort({**DEFAULTS, 'embedding_off': 1, 'use_byte': 1, 'min_conf': 0.1}, cap=64, nd=32, dim=1, sanitize=True)\n"
            "orc = OcSortOracle(use_byte=True)\n"
            "for d, _ in stress_frames(20, seed=7):\n"
            "    g, w = emu.update(d[:32], None), np.asarray(orc.update(d[:32].copy()), dtype=np.float32).reshape(-1, 8)\n"
            "    assert g.shape == w.shape and np.array_equal(g[:, 4:], w[:, 4:])\n"
            "emu.close()\nprint('ASAN-OK')\n")
    env = dict(os.environ, LD_PRELOAD=libasan[-1], ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "ASAN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


@pytest.mark.parametrize("kw,seed", [(dict(), 7), (dict(min_conf=0.2, det_thresh=0.6, inertia=0.1), 3)])
def test_emulated_ocsort_byte_association_matches_oracle(kw, seed):
    """OC-SORT's BYTE round (ocsort.py:456-485) in the device step vs the oracle pinned on the reference OcSort(use_byte=True):
    rows and the fp64 filter state of every track."""
    from oracle.deepocsort import OcSortOracle
    cfg = {**DEFAULTS, **{k: v for k, v in kw.items() if k in DEFAULTS}, "embedding_off": 1, "use_byte": 1, "min_conf": kw.get("min_conf", 0.1)}
    orc, emu = OcSortOracle(use_byte=True, **kw), EmuDeepOcSort(cfg, cap=128, nd=64, dim=1)
    try:
        for t, (d, _) in enumerate(stress_frames(70, seed=seed)):
            want = np.asarray(orc.update(d.copy()), dtype=np.float32).reshape(-1, 8)
            got = emu.update(d, None)
            assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
            assert np.allclose(got[:, :4], want[:, :4], rtol=0, atol=1e-4), t
        od, dd = orc.dump(), emu.dump()
        assert np.array_equal(dd["ints"][:, 0], od["id"]) and np.array_equal(dd["ints"][:, 3], od["hit_streak"])
        if dd["n"]:
            assert np.allclose(dd["kf"][:, :7], od["x"], rtol=1e-9, atol=1e-9)
    finally:
        emu.close()
