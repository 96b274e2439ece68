"""Runs the StrongSORT DEVICE kernels (boxmot_amd/csrc/strongsort_step.hpp, unchanged) on CPU threads through
tests/host_emu and compares them with the oracle frame by frame (rows, ids, states, fp64 filter state, features,
sample-bank sizes), and the device assignment solver alone against scipy.optimize.linear_sum_assignment.
Test infrastructure for the kernel logic -- the shipped library has no CPU path."""
import ctypes

import numpy as np
import pytest

from boxmot_amd.scenario import Scenario, camera_warps, stress_frames
from emu_util import EmuStrongSort, build_ss
from oracle.strongsort import DEFAULTS, StrongSortOracle


def _run(frames, dim, cap, nd, warps=None, sanitize=False, threads=64, **kw):
    cfg = dict(DEFAULTS)
    cfg.update(kw)
    orc, emu = StrongSortOracle(**kw), EmuStrongSort(cfg, cap=cap, nd=nd, dim=dim, sanitize=sanitize, threads=threads)
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
        assert np.array_equal(d["ints"][:, 1], od["state"])
        assert np.array_equal(d["ints"][:, 2], od["hits"])
        assert np.array_equal(d["ints"][:, 3], od["age"])
        assert np.array_equal(d["ints"][:, 4], od["time_since_update"])
        if d["n"]:
            ref = np.concatenate([od["mean"], od["cov"].reshape(-1, 64)], 1)
            assert np.allclose(d["kf"], ref, rtol=1e-9, atol=1e-10)
            for r, f in enumerate(od["feat"]):
                assert np.abs(d["feat"][r] - f).max() < 1e-5
            bank = [od["bank"].get(int(i), 0) for i in od["id"]]
            assert d["ints"][:, 5].tolist() == bank
        assert d["counters"][1] == od["next_id"]
    finally:
        emu.close()


@pytest.mark.parametrize("kw,seed", [({}, 7), (dict(max_age=5, n_init=1, nn_budget=3), 11),
                                     (dict(max_cos_dist=0.4, max_iou_dist=0.9, mc_lambda=0.9, ema_alpha=0.8, min_conf=0.3), 3)])
def test_emulated_strongsort_matches_oracle_stress(kw, seed):
    _run(stress_frames(45, seed=seed), 32, 128, 64, **kw)


def test_emulated_strongsort_four_wavefronts():
    """The same parity with a 256-thread workgroup (two scanning + two idle wavefronts in the assignment solver, the wave-0
    replay of the set order next to waiting waves, wave-per-row cost build): the cross-wavefront paths of the frame step."""
    frames = stress_frames(40, seed=9, max_objects=30)
    _run(frames, 32, 128, 64, warps=camera_warps(len(frames), seed=9), threads=256)


def test_emulated_strongsort_camera_update_and_set_order():
    """Crowded scene with camera warps.  Frame 5 of this sequence has several unmatched confirmed tracks whose order
    in the reference is the iteration order of a CPython set (linear_assignment.py:141) -- not ascending -- and that
    order decides which equally good IoU assignment SciPy returns, hence the ids of the tracks born in that frame."""
    frames = stress_frames(70, seed=5, max_objects=30)
    _run(frames, 32, 128, 64, warps=camera_warps(len(frames), seed=5))


def test_emulated_strongsort_c2_shape():
    sc = Scenario(64, 256, emb_dim=64, random_image=False)
    _run(sc.frames(6), 64, 512, 256, nn_budget=4)


def test_emulated_strongsort_c5_shape():
    """BASELINE configuration 5's table sizes -- 256 detections, 1024 live tracks -- for four frames, four wavefronts (embedding width
    128 instead of 1280: the width only scales the dot products)."""
    sc = Scenario(256, 1024, emb_dim=128, random_image=False)
    _run(sc.frames(4), 128, 2048, 1024, nn_budget=4, threads=256)


@pytest.mark.parametrize("threads", [64, 256])
def test_device_assignment_solver_equals_scipy_incl_ties(threads):
    """lsa_scipy restates SciPy's rectangular_lsap.cpp; tie-heavy matrices (clamped costs) must give the same columns.
    threads = 256: four wavefronts, the cross-wavefront combine and the single barrier per scan."""
    from scipy.optimize import linear_sum_assignment
    lib = ctypes.CDLL(str(build_ss(threads=threads)))
    lib.emu_lsa.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(2)
    for it in range(150):
        nr = int(rng.integers(1, 12))
        nc = int(rng.integers(nr, 14))
        c = rng.integers(0, 3, (nr, nc)).astype(float)
        if it % 3 == 0:
            c = np.where(rng.random((nr, nc)) < 0.5, 0.70001, rng.random((nr, nc)))
        c = np.ascontiguousarray(c)
        out = np.zeros(nr, np.int32)
        lib.emu_lsa(c.ctypes.data, nr, nc, out.ctypes.data)
        assert np.array_equal(linear_sum_assignment(c)[1], out), (it, c)
    # wide problems: several wavefronts scan the columns, ties within and across wavefronts, long augmenting paths
    for it in range(40):
        nr = int(rng.integers(1, 60))
        nc = int(rng.integers(max(nr, 65), 420))
        c = rng.integers(0, 3 if it % 2 else 6, (nr, nc)).astype(float)
        if it % 4 == 0:
            c = np.where(rng.random((nr, nc)) < 0.7, 0.20001, np.round(rng.random((nr, nc)), 2))
        c = np.ascontiguousarray(c)
        out = np.zeros(nr, np.int32)
        lib.emu_lsa(c.ctypes.data, nr, nc, out.ctypes.data)
        assert np.array_equal(linear_sum_assignment(c)[1], out), (it, nr, nc)
    # stage-A shape of a crowded frame (parity soak seed 159): 21 confirmed tracks x 9 detections, every gated entry
    # clamped to the same 0.20001 -- SciPy solves the transpose, so the device does too
    c = np.full((21, 9), 0.20001)
    for r, q, val in [(1, 3, 0.06822), (7, 1, 0.03715), (7, 2, 0.06602), (11, 6, 0.0644), (13, 7, 0.09929), (14, 4, 0.06037),
                      (16, 8, 0.0711), (20, 0, 0.04554)]:
        c[r, q] = val
    ct = np.ascontiguousarray(c.T)
    out = np.zeros(9, np.int32)
    lib.emu_lsa(ct.ctypes.data, 9, 21, out.ctypes.data)
    rows, cols = linear_sum_assignment(c)
    assert sorted(zip(out.tolist(), range(9))) == list(zip(rows.tolist(), cols.tolist()))


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
            "from test_ssort_emu import _run\n"
            "from boxmot_amd.scenario import stress_frames\n"
            "_run(stress_frames(14, seed=7), 32, 64, 32, sanitize=True)\nprint('ASAN-OK')\n")
    env = dict(os.environ, LD_PRELOAD=libasan[-1], ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "ASAN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


def test_device_dot_rule_reproduces_the_kernels_appearance_costs_bit_for_bit():
    """StrongSortOracle(dot_rule="device") restates the kernels' documented fp32 summation order (lane-strided fmaf + butterfly
    for the norms, one fmaf per k for the products, 1 - dot / (|a| |b|)) and their fp64 operation order in the Kalman update and
    the gating distance: every appearance distance the bank kernel produces is bit-identical to the oracle's, frame after frame,
    and so is the filter state -- the cost matrix, not just the ids, is pinned."""
    frames = stress_frames(40, seed=13, max_objects=24)
    cfg = dict(DEFAULTS)
    orc, emu = StrongSortOracle(dot_rule="device"), EmuStrongSort(cfg, cap=128, nd=64, dim=32)
    checked = 0
    try:
        prev_ids = []
        for t, (d, e) in enumerate(frames):
            want = orc.update(d.copy(), None, e.copy()).reshape(-1, 8)
            got = emu.update(d, e)
            assert got.shape == want.shape and np.array_equal(got[:, 4:], want[:, 4:]), t
            app = emu.app(len(prev_ids), len(d))
            for (tid, j), v in getattr(orc, "last_app", {}).items():
                r = prev_ids.index(tid)
                assert app[r, j].tobytes() == np.float32(v).tobytes(), (t, tid, j, float(app[r, j]), float(v))
                checked += 1
            prev_ids = emu.dump()["ints"][:, 0].tolist()
        # ... and so is the fp64 filter state: the device rule also restates the kernels' Kalman update / gating operation order
        od, d = orc.dump(), emu.dump()
        ref = np.concatenate([od["mean"], od["cov"].reshape(-1, 64)], 1)
        assert np.array_equal(d["kf"], ref)
    finally:
        emu.close()
    assert checked > 500


def test_unmatched_track_order_equals_cpython_set_iteration():
    """tracker.py / linear_assignment.py:141: `list(set(track_indices) - set(matched))` -- the frame step reproduces CPython's
    iteration order (identity-layout shortcut, the wavefront replay of the hash tables out of LDS for both set_difference paths,
    and the one-thread fallback when the tables outgrow the LDS area), checked against the interpreter itself."""
    lib = ctypes.CDLL(str(build_ss(threads=256)))
    lib.emu_set_order.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    rng = np.random.default_rng(3)
    cases = []
    for it in range(120):
        nt = int(rng.integers(1, 700))
        kind = it % 6
        if kind == 0:                      # every track confirmed (dense keys): the shortcut
            a = np.arange(nt)
        elif kind == 1:                    # persistent objects first, then a sparse tail
            a = np.sort(rng.choice(nt, size=max(1, nt // 2), replace=False))
        elif kind == 2:                    # confirmed tracks start late in the list: collisions in the small tables
            lo = int(rng.integers(0, nt))
            a = np.arange(lo, nt)
        else:
            a = np.sort(rng.choice(nt, size=int(rng.integers(1, nt + 1)), replace=False))
        frac = [0.0, 0.1, 0.24, 0.26, 0.6, 1.0][int(rng.integers(0, 6))]      # both sides of the len(a) / 4 > len(b) switch
        nb = int(round(frac * len(a)))
        if kind in (1, 4):                 # the matched ones are the first keys (persistent objects): the unmatched start late
            b = a[:nb].copy()
            rng.shuffle(b)
        else:
            b = rng.choice(a, size=nb, replace=False)
        cases.append((a.astype(np.int32), b.astype(np.int32), nt))
    for lds_big in (2048, 16):             # 16: the tables do not fit the LDS area -> the one-thread path
        for a, b, nt in cases:
            want = list(set(a.tolist()) - set(b.tolist()))
            out = np.zeros(len(a) + 1, np.int32)
            n = lib.emu_set_order(a.ctypes.data, len(a), b.ctypes.data, len(b), nt, 2048, lds_big, out.ctypes.data)
            assert n == len(want) and out[:n].tolist() == want, (lds_big, len(a), len(b), nt)
