"""CPU: the library's host LSAP must agree with scipy's solver pair-for-pair, including on the
tie-heavy (clamped) matrices deep_sort produces (deep_sort/linear_assignment.py:57-58 upstream)."""
import numpy as np
from scipy.optimize import linear_sum_assignment as scipy_lsa

from deepdish_amd.deep_sort.linear_assignment import linear_sum_assignment


def _cases(rng, n):
    for trial in range(n):
        nr, nc = rng.integers(1, 16, 2)
        kind = trial % 5
        if kind == 0:
            c = rng.random((nr, nc))
        elif kind == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(float)
        elif kind == 2:
            c = rng.random((nr, nc)); c[c > 0.3] = 0.2 + 1e-5
        elif kind == 3:
            c = rng.random((nr, nc)) * 0.5; c[rng.random((nr, nc)) < 0.6] = 0.7 + 1e-5
        else:
            c = np.full((nr, nc), 0.2 + 1e-5)
        yield c


def test_lsap_matches_scipy_with_ties():
    rng = np.random.default_rng(7)
    for c in _cases(rng, 2500):
        r1, c1 = linear_sum_assignment(c)
        r2, c2 = scipy_lsa(c)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(c1, c2)


def test_lsap_large_and_degenerate():
    rng = np.random.default_rng(8)
    c = rng.random((256, 256)); c[c > 0.2] = 0.2 + 1e-5
    r1, c1 = linear_sum_assignment(c)
    r2, c2 = scipy_lsa(c)
    np.testing.assert_array_equal(c1, c2)
    c = rng.random((300, 40))
    r1, c1 = linear_sum_assignment(c)
    r2, c2 = scipy_lsa(c)
    np.testing.assert_array_equal(r1, r2)
    np.testing.assert_array_equal(c1, c2)
    r, cc = linear_sum_assignment(np.zeros((0, 5)))
    assert len(r) == 0 and len(cc) == 0
