"""GPU parity of the deep_sort math (through the C ABI) against the golden vectors made from the
reference's own modules, and against the oracle on fresh seeded inputs."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')


def _load(name):
    return np.load(os.path.join(G, name))


@pytest.fixture(scope='module')
def kf():
    from deepdish_amd.deep_sort.kalman_filter import KalmanFilter
    return KalmanFilter()


def test_native_library_loaded():
    from deepdish_amd._lib import lib
    assert lib().dd_version() >= 100
    with open('/proc/self/maps') as f:
        assert 'libdeepdish_hip.so' in f.read()


def test_kalman_golden(kf):
    g = _load('kalman.npz')
    m, c = kf.initiate(g['z0'])
    np.testing.assert_array_equal(m, g['init_mean'])
    np.testing.assert_allclose(c, g['init_cov'], rtol=1e-15, atol=0)
    pm, pc = kf.predict(g['mean'], g['cov'])
    np.testing.assert_allclose(pm, g['pred_mean'], rtol=1e-14)
    np.testing.assert_allclose(pc, g['pred_cov'], rtol=1e-13, atol=1e-18)
    jm, jc = kf.project(g['pred_mean'], g['pred_cov'])
    np.testing.assert_allclose(jm, g['proj_mean'], rtol=1e-14)
    np.testing.assert_allclose(jc, g['proj_cov'], rtol=1e-13, atol=1e-18)
    um, uc = kf.update(g['pred_mean'], g['pred_cov'], g['z'])
    np.testing.assert_allclose(um, g['upd_mean'], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(uc, g['upd_cov'], rtol=1e-10, atol=1e-13)
    for i in range(len(g['zs'])):
        d2 = kf.gating_distance(g['pred_mean'][i], g['pred_cov'][i], g['zs'][i])
        np.testing.assert_allclose(d2, g['d2'][i], rtol=1e-11)
        d2p = kf.gating_distance(g['pred_mean'][i], g['pred_cov'][i], g['zs'][i], only_position=True)
        np.testing.assert_allclose(d2p, g['d2_pos'][i], rtol=1e-11)
    # single-state call shape, as the reference's Track uses it
    m1, c1 = kf.predict(g['mean'][3], g['cov'][3])
    assert m1.shape == (8,) and c1.shape == (8, 8)
    np.testing.assert_allclose(c1, g['pred_cov'][3], rtol=1e-13, atol=1e-18)


def test_kalman_predict_is_bit_exact_on_integer_states(kf):
    """Linearity/known-answer: with dyadic inputs every sum is exact, so predict must be exact."""
    rng = np.random.default_rng(0)
    m = rng.integers(-64, 64, (300, 8)).astype(np.float64)
    m[:, 3] = 160.0
    a = rng.integers(-8, 8, (300, 8, 8)).astype(np.float64)
    c = a @ a.transpose(0, 2, 1)
    pm, pc = kf.predict(m, c)
    F = np.eye(8); F[:4, 4:] = np.eye(4)
    q = np.diag(np.square([8.0, 8.0, 1e-2, 8.0, 1.0, 1.0, 1e-5, 1.0]))
    np.testing.assert_array_equal(pm, m @ F.T)
    np.testing.assert_array_equal(pc, F @ c @ F.T + q)


def test_iou_golden():
    from deepdish_amd.deep_sort import iou_matching
    g = _load('iou_nms.npz')
    got = np.array([iou_matching.iou(x, g['iou_b']) for x in g['iou_a'][:8]])
    np.testing.assert_allclose(got, g['iou'][:8], rtol=1e-15, atol=1e-16)
    cost = iou_matching._iou_cost_arrays(g['iou_a'], None, g['iou_b'])
    np.testing.assert_allclose(1.0 - cost, g['iou'], rtol=1e-15, atol=1e-16)
    tsu = np.ones(len(g['iou_a']), dtype=np.int32); tsu[::3] = 2
    cost = iou_matching._iou_cost_arrays(g['iou_a'], tsu, g['iou_b'])
    assert np.all(cost[::3] == 1e5)
    np.testing.assert_array_equal(cost[1::3], 1.0 - g['iou'][1::3])


def test_nms_golden():
    from deepdish_amd.deep_sort.preprocessing import non_max_suppression
    g = _load('iou_nms.npz')
    for k in (1, 7, 20, 64, 256, 1000):
        b, s = g[f'nms{k}_boxes'], g[f'nms{k}_scores']
        for thr in (0.6, 0.3, 1.0):
            assert non_max_suppression(b, thr, s) == g[f'nms{k}_keep_{thr}'].tolist(), (k, thr)
        y2 = b[:, 1] + b[:, 3]
        if len(np.unique(y2)) == k:      # scores=None sorts by y2: only defined when y2 is tie-free
            assert non_max_suppression(b, 0.6, None) == g[f'nms{k}_keep_noscore'].tolist(), k
        else:                            # (the reference's unstable argsort leaves tie order open)
            keep = non_max_suppression(b, 0.6, None)
            assert len(set(keep)) == len(keep) and np.all(np.diff(y2[keep]) <= 0)
    assert non_max_suppression(np.zeros((0, 4)), 0.6, np.zeros(0)) == []


def test_nms_vs_oracle_dense():
    """Heavy overlap (long suppression chains) and the 4096-box capacity edge."""
    from deepdish_amd.deep_sort.preprocessing import non_max_suppression
    from oracle import deepsort_np as ds
    rng = np.random.default_rng(5)
    for k, span in ((300, 200), (1500, 600), (4096, 3000)):
        b = np.c_[rng.integers(0, span, (k, 2)), rng.integers(20, 80, (k, 2))].astype(np.int64)
        s = rng.permutation(k) / (k + 1.0)
        assert non_max_suppression(b, 0.6, s) == ds.non_max_suppression(b, 0.6, s)
    from deepdish_amd._lib import DeepDishHipError
    with pytest.raises(DeepDishHipError):
        non_max_suppression(np.ones((5000, 4)), 0.6, np.arange(5000.0))


def test_cosine_golden():
    from deepdish_amd.deep_sort.nn_matching import NearestNeighborDistanceMetric
    g = _load('cosine.npz')
    m = NearestNeighborDistanceMetric('cosine', 0.2, None)
    sizes = g['gallery_sizes']
    targets = np.repeat(np.arange(1, len(sizes) + 1), sizes)
    m.partial_fit(g['gallery'], targets, list(range(1, len(sizes) + 1)))
    cost = m.distance(g['query'], list(range(1, len(sizes) + 1)))
    assert cost.dtype == np.float64 and cost.shape == g['nn_cost'].shape
    # f32 arithmetic in both; only the summation order differs (stated tolerance: 2e-6 absolute)
    np.testing.assert_allclose(cost, g['nn_cost'], rtol=0, atol=2e-6)
    # un-normalised inputs: each gallery row is its own target -> full pairwise matrix
    m2 = NearestNeighborDistanceMetric('cosine', 0.2, None)
    m2.partial_fit(g['a'], np.arange(len(g['a'])), list(range(len(g['a']))))
    c2 = m2.distance(g['b'], list(range(len(g['a']))))
    np.testing.assert_allclose(c2, g['cos'], rtol=0, atol=2e-6)


def test_gate_cost_matrix_and_matching_wrappers(kf):
    from deepdish_amd.deep_sort import linear_assignment as la
    from deepdish_amd.deep_sort.detection import Detection
    from oracle import deepsort_np as ds
    g = _load('kalman.npz')

    class T:
        pass
    tracks = []
    for i in range(12):
        t = T(); t.mean, t.covariance = g['pred_mean'][i], g['pred_cov'][i]; tracks.append(t)
    dets = [Detection(np.r_[g['pred_mean'][i % 12][:2] - 20 + 3 * i, 40, 80], 'person', 0.9, np.zeros(128))
            for i in range(9)]
    cost = np.random.default_rng(1).random((12, 9))
    want = ds.gate_cost_matrix(cost.copy(), tracks, dets, list(range(12)), list(range(9)))
    got = la.gate_cost_matrix(kf, cost.copy(), tracks, dets, list(range(12)), list(range(9)))
    np.testing.assert_array_equal(got, want)
