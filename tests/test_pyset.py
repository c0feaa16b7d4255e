"""CPU: csrc/pyset.cpp reproduces this interpreter's list(set(a) - set(b)) order (the row order deep_sort
feeds into its IoU-stage assignment, deep_sort/linear_assignment.py:140 upstream)."""
import ctypes
import numpy as np

from deepdish_amd._lib import lib, check
from deepdish_amd.runtime import ptr


def _order(a, b):
    a = np.asarray(a, dtype=np.int32); b = np.asarray(b, dtype=np.int32)
    out = np.zeros(max(1, len(a)), dtype=np.int32)
    n = ctypes.c_int()
    check(lib().dd_pyset_difference_order_host(ptr(a), len(a), ptr(b), len(b), ptr(out), ctypes.byref(n)))
    return out[:n.value].tolist()


def test_matches_cpython_on_tracker_shaped_inputs():
    rng = np.random.default_rng(0)
    for trial in range(4000):
        t = int(rng.integers(1, 400))
        # confirmed-track indices: an increasing subset of range(t), like tracker.py:104-105
        a = sorted(rng.choice(t, size=int(rng.integers(0, t + 1)), replace=False).tolist())
        # matched subset, in match order (arbitrary)
        m = rng.permutation(a)[:int(rng.integers(0, len(a) + 1))].tolist() if a else []
        want = list(set(a) - set(k for k in m))
        assert _order(a, m) == want, (a, m)


def test_edge_cases():
    assert _order([], []) == []
    assert _order([0], []) == [0]
    assert _order([0, 8, 16, 24], []) == list(set([0, 8, 16, 24]))
    big = list(range(0, 3000, 7))
    assert _order(big, big[::3]) == list(set(big) - set(big[::3]))
    assert _order(big, big) == []
