"""The device's Jonker-Volgenant solver (boxmot_amd/csrc/lap_jv.hpp, executed unchanged on CPU threads) against the oracle's
sequential restatement of lap.lapjv (oracle/lapjv.c) -- the SAME assignment, not just the same optimum: tie-heavy matrices (small
integer costs, blocks of zeros as zero-IoU pairs produce them, clamped costs), both rectangular orientations, with and without
cost_limit, one and four wavefronts per workgroup."""
import ctypes

import numpy as np
import pytest

from emu_util import build_docs
from oracle import lap as olap


def _lib(threads):
    lib = ctypes.CDLL(str(build_docs(threads=threads)))
    lib.emu_lap_jv.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    lib.emu_lap_jv.restype = ctypes.c_int
    return lib


def _cases(rng, n_cases):
    for k in range(n_cases):
        nr, nc = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        kind = k % 6
        if kind == 0:
            c = rng.integers(0, 3, (nr, nc)).astype(np.float64)                     # heavy ties
        elif kind == 1:
            c = -np.where(rng.random((nr, nc)) < 0.15, rng.random((nr, nc)), 0.0)   # DeepOCSORT-like: mostly zero, a few negative
        elif kind == 2:
            c = rng.random((nr, nc))                                                # generic
        elif kind == 3:
            c = np.minimum(rng.random((nr, nc)) * 2.0, 1.0)                         # clamped
        elif kind == 4:
            c = np.zeros((nr, nc))                                                  # everything tied
        else:
            c = -np.round(rng.random((nr, nc)), 1)
        yield np.ascontiguousarray(c)


N_CASES = {64: 72, 256: 30}


@pytest.mark.parametrize("threads", [64, 256])
def test_device_jv_returns_the_oracles_assignment_tie_for_tie(threads):
    lib = _lib(threads)
    rng = np.random.default_rng(7)
    n = 0
    for c in _cases(rng, N_CASES[threads]):
        nr, nc = c.shape
        for limit in (None, 0.5, 1.5):
            x = np.full(nr, -9, np.int32)
            y = np.full(nc, -9, np.int32)
            ok = lib.emu_lap_jv(nr, nc, c.ctypes.data, int(limit is not None), float(limit or 0.0), x.ctypes.data, y.ctypes.data)
            assert ok == 1
            _, wx, wy = olap.lapjv(c, extend_cost=True, cost_limit=np.inf if limit is None else limit)
            assert np.array_equal(x, wx) and np.array_equal(y, wy), (c.shape, limit, c, x, wx)
            n += 1
    assert n == 3 * N_CASES[threads]


def test_device_jv_larger_problem_with_ties():
    lib = _lib(256)
    rng = np.random.default_rng(3)
    for nr, nc in ((128, 90), (70, 150)):
        c = -np.where(rng.random((nr, nc)) < 0.05, np.round(rng.random((nr, nc)), 2), 0.0)
        x = np.full(nr, -9, np.int32)
        y = np.full(nc, -9, np.int32)
        assert lib.emu_lap_jv(nr, nc, c.ctypes.data, 0, 0.0, x.ctypes.data, y.ctypes.data) == 1
        _, wx, wy = olap.lapjv(c, extend_cost=True)
        assert np.array_equal(x, wx) and np.array_equal(y, wy)


def test_device_jv_thin_recovery_round_shapes():
    """The second association round of DeepOCSORT: a handful of leftover detections against hundreds of leftover tracks, nearly
    all pairs at cost 0 -- hundreds of augmentations whose TODO lists are one long tie (the run fast path of the device's scan)."""
    lib = _lib(64)
    rng = np.random.default_rng(11)
    for nr, nc in ((1, 300), (3, 290), (280, 2)):
        c = -np.where(rng.random((nr, nc)) < 0.02, np.round(rng.random((nr, nc)), 2), 0.0)
        x = np.full(nr, -9, np.int32)
        y = np.full(nc, -9, np.int32)
        assert lib.emu_lap_jv(nr, nc, c.ctypes.data, 0, 0.0, x.ctypes.data, y.ctypes.data) == 1
        _, wx, wy = olap.lapjv(c, extend_cost=True)
        assert np.array_equal(x, wx) and np.array_equal(y, wy), (nr, nc, x, wx)


def test_list_permutation_of_a_chunk_matches_the_sequential_swaps():
    """jv_move (one step per 64-column chunk segment) against the loop it replaces: `cols[k], cols[hi] = cols[hi], cols[k]; hi += 1`
    for every joining position k in ascending order -- random join masks, queue lengths (positions between hi and the chunk),
    segment starts, ragged list ends."""
    lib = ctypes.CDLL(str(build_docs(threads=64)))
    lib.emu_jv_move.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
    lib.emu_jv_move.restype = ctypes.c_int
    rng = np.random.default_rng(5)
    for case in range(300):
        g0 = int(rng.choice([0, 0, 1, 2, 5, 40, 100]))
        l0 = int(rng.choice([0, 0, 0, 3, 17, 63]))
        hi = int(rng.integers(0, 4))
        base = hi + g0 - l0
        if base < 0:
            continue
        valid = int(rng.integers(l0 + 1, 65))                       # lanes of the chunk that hold list entries
        seg_end = int(rng.integers(l0, valid + 1))
        n = base + valid
        dens = float(rng.choice([0.05, 0.5, 0.95, 1.0]))
        q = 0
        for l in range(l0, seg_end):
            if rng.random() < dens:
                q |= 1 << l
        cols = rng.permutation(n).astype(np.int32)
        want = cols.copy()
        whi = hi
        for l in range(l0, seg_end):
            if (q >> l) & 1:
                k = base + l
                want[k], want[whi] = want[whi], want[k]
                whi += 1
        got_hi = lib.emu_jv_move(n, cols.ctypes.data, base, l0, ctypes.c_uint64(q), hi)
        assert got_hi == whi and np.array_equal(cols, want), (case, g0, l0, hi, base, valid, seg_end, bin(q))


def test_device_jv_tie_heavy_shapes_of_the_fast_paths():
    """Shapes and densities that put the solver on its three shortcuts -- the contiguous-run two-smallest reduction of the row
    reduction, the single-reduction augmentation (first ready run already holds a free column) and the recognition of repeated
    no-op pops of extension rows (95 - 98 % of the pops of these matrices, profiles/r4_jv_prof.txt) -- and off them again (denser
    negatives: real rows relax, the memo is invalidated).  Tie for tie against the sequential code, with and without cost_limit."""
    lib = _lib(64)
    rng = np.random.default_rng(1)
    for nr, nc, p in ((30, 400, 0.05), (60, 300, 0.1), (100, 512, 0.02), (20, 400, 0.3), (128, 512, 0.2), (17, 401, 0.0025), (40, 40, 0.5), (300, 20, 0.05)):
        c = -np.where(rng.random((nr, nc)) < p, np.round(rng.random((nr, nc)), 2), 0.0)
        for limit in (None, 0.5):
            x = np.full(nr, -9, np.int32)
            y = np.full(nc, -9, np.int32)
            assert lib.emu_lap_jv(nr, nc, c.ctypes.data, int(limit is not None), float(limit or 0.0), x.ctypes.data, y.ctypes.data) == 1
            _, wx, wy = olap.lapjv(c, extend_cost=True, cost_limit=np.inf if limit is None else limit)
            assert np.array_equal(x, wx) and np.array_equal(y, wy), (nr, nc, p, limit)
