"""Oracle stand-in for ``lap.lapjv`` (lapx 0.9.4) -- TEST INFRASTRUCTURE ONLY.

Call sites restated: boxmot/trackers/association/matching.py:28-43
(``lap.lapjv(cost, extend_cost=True, cost_limit=thresh)``) and
boxmot/trackers/association/association.py:20-24 (``extend_cost=True``).

Two interchangeable back ends:
  * ``lapjv``       -- the C Jonker-Volgenant restatement in ``oracle/lapjv.c``
                       (compiled by ``__graft_entry__.build()`` into
                       ``oracle/_build/liboracle.so``);
  * ``lapjv_scipy`` -- the same extended-matrix formulation solved by SciPy's
                       exact ``linear_sum_assignment`` (independent solver used
                       to cross-check the C code in tests).
PARITY UNPINNED against the real lapx package (unavailable offline).
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liboracle.so"
_lib = None

# How `extend_cost=True` WITHOUT a cost limit (the DeepOCSORT / OC-SORT call site, association.py:20-24) squares the matrix:
#   "sum_max_plus_one"  (n_rows + n_cols)^2 filled with max(cost) + 1, zero bottom-right block -- the restatement of lapx's wrapper
#                       this oracle (and the device solver, tie for tie) follows; the default;
#   "zero_pad"          max(n_rows, n_cols)^2 zero-padded -- the alternative reading SURVEY.md section 8(c) flags.
# Same optimum either way; the choice among exactly tied optima may differ.  tests/test_lap_forms.py runs every golden of that
# call site under both and asserts identical rows, so the unpinnable choice is a tested invariance, not an assumption.
NO_LIMIT_FORM = "sum_max_plus_one"


def build_oracle_lib(force: bool = False) -> Path:
    """Compile oracle/lapjv.c with gcc (used by __graft_entry__.build())."""
    src = _HERE / "lapjv.c"
    _LIB_PATH.parent.mkdir(exist_ok=True)
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", str(_LIB_PATH), str(src)]
        )
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build_oracle_lib()
        lib = ctypes.CDLL(str(_LIB_PATH))
        lib.oracle_lapjv_extended.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
            ctypes.c_void_p, ctypes.c_void_p,
        ]
        lib.oracle_lapjv_extended.restype = ctypes.c_int
        lib.oracle_lapjv_zero_padded.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_lapjv_zero_padded.restype = ctypes.c_int
        _lib = lib
    return _lib


def lapjv(cost, extend_cost: bool = False, cost_limit: float = np.inf, return_cost: bool = True):
    """``lap.lapjv`` stand-in: returns (opt, x, y) like lapx."""
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    if cost.ndim != 2:
        raise ValueError("2-dimensional array expected")
    nr, nc = cost.shape
    if nr != nc and not extend_cost and not cost_limit < np.inf:
        raise ValueError("Square cost array expected; pass extend_cost=True")
    x = np.full(nr, -1, dtype=np.int32)
    y = np.full(nc, -1, dtype=np.int32)
    if nr and nc:
        use_limit = bool(cost_limit < np.inf)
        if not use_limit and NO_LIMIT_FORM == "zero_pad":
            _load().oracle_lapjv_zero_padded(nr, nc, cost.ctypes.data, x.ctypes.data, y.ctypes.data)
        else:
            assert use_limit or NO_LIMIT_FORM == "sum_max_plus_one", NO_LIMIT_FORM
            _load().oracle_lapjv_extended(
                nr, nc, cost.ctypes.data, int(use_limit), float(cost_limit if use_limit else 0.0),
                x.ctypes.data, y.ctypes.data,
            )
    x = x.astype(np.int64)
    y = y.astype(np.int64)
    if return_cost:
        rows = np.nonzero(x >= 0)[0]
        return float(cost[rows, x[rows]].sum()), x, y
    return x, y


def lapjv_scipy(cost, extend_cost: bool = False, cost_limit: float = np.inf, return_cost: bool = True):
    """Same extended formulation, solved with scipy.optimize.linear_sum_assignment."""
    from scipy.optimize import linear_sum_assignment

    cost = np.asarray(cost, dtype=np.float64)
    nr, nc = cost.shape
    x = np.full(nr, -1, dtype=np.int64)
    y = np.full(nc, -1, dtype=np.int64)
    if nr and nc:
        n = nr + nc
        fill = cost_limit / 2.0 if cost_limit < np.inf else cost.max() + 1.0
        ext = np.full((n, n), fill, dtype=np.float64)
        ext[nr:, nc:] = 0.0
        ext[:nr, :nc] = cost
        r, c = linear_sum_assignment(ext)
        xe = np.full(n, -1, dtype=np.int64)
        ye = np.full(n, -1, dtype=np.int64)
        xe[r] = c
        ye[c] = r
        x = xe[:nr].copy()
        y = ye[:nc].copy()
        x[x >= nc] = -1
        y[y >= nr] = -1
    if return_cost:
        rows = np.nonzero(x >= 0)[0]
        return float(cost[rows, x[rows]].sum()), x, y
    return x, y
