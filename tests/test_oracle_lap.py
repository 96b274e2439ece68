"""oracle/lapjv.c (Jonker-Volgenant restatement) against SciPy's exact solver on the same
extended-matrix formulation, incl. empty / ragged / gated inputs."""
import numpy as np
import pytest

from oracle import lap
from oracle.matching import linear_assignment


@pytest.mark.parametrize("seed", range(4))
def test_lapjv_c_equals_scipy(seed):
    rng = np.random.default_rng(seed)
    for _ in range(120):
        nr, nc = rng.integers(0, 48), rng.integers(0, 48)
        c = rng.uniform(0, 1, (nr, nc))
        c[rng.uniform(size=c.shape) < 0.5] = 1.0          # gated entries exactly 1.0 (botsort.py:313-314)
        limit = rng.uniform(0.2, 0.95)
        for kw in (dict(cost_limit=limit), dict()):
            ca, xa, ya = lap.lapjv(c, extend_cost=True, **kw)
            cb, xb, yb = lap.lapjv_scipy(c, extend_cost=True, **kw)
            assert abs(ca - cb) < 1e-9
            assert (xa >= 0).sum() == (xb >= 0).sum()
            if "cost_limit" in kw and nr and nc:
                m = xa >= 0
                assert np.all(c[np.nonzero(m)[0], xa[m]] < limit)   # a pair is matched only if c < L
            # x / y are inverse maps
            for i, j in enumerate(xa):
                if j >= 0:
                    assert ya[j] == i


def test_linear_assignment_shapes():
    m, ua, ub = linear_assignment(np.zeros((0, 5)), 0.5)
    assert m.shape == (0, 2) and list(ua) == [] and list(ub) == [0, 1, 2, 3, 4]
    m, ua, ub = linear_assignment(np.array([[0.9, 0.1], [0.2, 0.95], [0.99, 0.99]]), 0.8)
    assert m.tolist() == [[0, 1], [1, 0]] and ua.tolist() == [2] and ub.tolist() == []
    # everything gated: nothing matches
    m, ua, ub = linear_assignment(np.ones((3, 2)), 0.8)
    assert len(m) == 0 and ua.tolist() == [0, 1, 2] and ub.tolist() == [0, 1]
