"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the deep_sort association math.

Every function names the reference lines it follows (paths relative to the
upstream tree).  Parity of this file against the real reference is PINNED by
``tests/golden/deepsort_*.npz`` (made by ``scripts/make_golden.py`` which
imports the reference's own modules) -- see ``tests/test_oracle_golden.py``.

The per-track loop structure of the reference is kept on purpose: this file is
also what ``bench.py`` times as the CPU baseline ("port"), so it must cost what
the reference costs, not what a vectorised rewrite would.
"""
import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

# deep_sort/kalman_filter.py:11-20 (4 degrees of freedom) and linear_assignment.py:8
CHI2INV95_4 = 9.4877
CHI2INV95_2 = 5.9915
INFTY_COST = 1e5
# deep_sort/kalman_filter.py:52-53
W_POS = 1.0 / 20
W_VEL = 1.0 / 160

_F = np.eye(8)
for _i in range(4):
    _F[_i, 4 + _i] = 1.0          # kalman_filter.py:44-46, dt = 1
_H = np.eye(4, 8)                 # kalman_filter.py:47


# --------------------------------------------------------------------------- Kalman
def kf_initiate(z):
    """kalman_filter.py:55-86.  z = (cx, cy, a, h) -> mean[8], cov[8,8]."""
    z = np.asarray(z, dtype=float)
    mean = np.concatenate([z, np.zeros(4)])
    h = z[3]
    sd = np.array([2 * W_POS * h, 2 * W_POS * h, 1e-2, 2 * W_POS * h,
                   10 * W_VEL * h, 10 * W_VEL * h, 1e-5, 10 * W_VEL * h])
    return mean, np.diag(sd * sd)


def kf_predict(mean, cov):
    """kalman_filter.py:88-123.  Q uses h = mean[3] BEFORE the step."""
    h = mean[3]
    sd = np.array([W_POS * h, W_POS * h, 1e-2, W_POS * h,
                   W_VEL * h, W_VEL * h, 1e-5, W_VEL * h])
    q = np.diag(sd * sd)
    return _F @ mean, np.linalg.multi_dot((_F, cov, _F.T)) + q


def kf_project(mean, cov):
    """kalman_filter.py:125-152."""
    h = mean[3]
    sd = np.array([W_POS * h, W_POS * h, 1e-1, W_POS * h])
    return _H @ mean, np.linalg.multi_dot((_H, cov, _H.T)) + np.diag(sd * sd)


def kf_update(mean, cov, z):
    """kalman_filter.py:154-186 (Cholesky solve for the gain)."""
    pm, pc = kf_project(mean, cov)
    cf = scipy.linalg.cho_factor(pc, lower=True, check_finite=False)
    gain = scipy.linalg.cho_solve(cf, (cov @ _H.T).T, check_finite=False).T
    innov = np.asarray(z, dtype=float) - pm
    return mean + innov @ gain.T, cov - np.linalg.multi_dot((gain, pc, gain.T))


def kf_gating_distance(mean, cov, zs, only_position=False):
    """kalman_filter.py:188-229.  zs[N,4] -> squared Mahalanobis d2[N]."""
    pm, pc = kf_project(mean, cov)
    zs = np.asarray(zs, dtype=float)
    if only_position:
        pm, pc, zs = pm[:2], pc[:2, :2], zs[:, :2]
    L = np.linalg.cholesky(pc)
    y = scipy.linalg.solve_triangular(L, (zs - pm).T, lower=True, check_finite=False)
    return np.sum(y * y, axis=0)


# --------------------------------------------------------------------------- boxes
def tlwh_to_xyah(tlwh):
    """detection.py:43-50."""
    r = np.array(tlwh, dtype=float)
    r[:2] += r[2:] / 2
    r[2] /= r[3]
    return r


def mean_to_tlwh(mean):
    """track.py:84-97."""
    r = np.array(mean[:4], dtype=float)
    r[2] *= r[3]
    r[:2] -= r[2:] / 2
    return r


def mean_to_tlbr(mean):
    """track.py:99-111."""
    r = mean_to_tlwh(mean)
    r[2:] = r[:2] + r[2:]
    return r


def iou(box, cands):
    """iou_matching.py:7-39 -- plain IoU on tlwh, NO +1 pixel."""
    box = np.asarray(box, dtype=float)
    cands = np.asarray(cands, dtype=float)
    b_br = box[:2] + box[2:]
    c_br = cands[:, :2] + cands[:, 2:]
    tl = np.maximum(box[:2], cands[:, :2])
    br = np.minimum(b_br, c_br)
    wh = np.maximum(0.0, br - tl)
    inter = wh[:, 0] * wh[:, 1]
    return inter / (box[2] * box[3] + cands[:, 2] * cands[:, 3] - inter)


def non_max_suppression(boxes, max_overlap, scores=None):
    """preprocessing.py:6-73 -- greedy, +1 pixel, overlap = inter / area_other."""
    if len(boxes) == 0:
        return []
    b = np.asarray(boxes).astype(float)
    x1, y1 = b[:, 0], b[:, 1]
    x2, y2 = b[:, 2] + b[:, 0], b[:, 3] + b[:, 1]
    area = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.argsort(scores) if scores is not None else np.argsort(y2)
    keep = []
    while len(order) > 0:
        i = order[-1]
        rest = order[:-1]
        keep.append(int(i))
        w = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
        h = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
        order = rest[~((w * h) / area[rest] > max_overlap)]
    return keep


# --------------------------------------------------------------------------- appearance
def cosine_distance(a, b):
    """nn_matching.py:31-54 (data_is_normalized=False): f32 in, f32 out."""
    a = np.asarray(a) / np.linalg.norm(a, axis=1, keepdims=True)
    b = np.asarray(b) / np.linalg.norm(b, axis=1, keepdims=True)
    return 1.0 - np.dot(a, b.T)


def nn_cosine_distance(gallery, feats):
    """nn_matching.py:78-96."""
    return cosine_distance(gallery, feats).min(axis=0)


class Metric:
    """nn_matching.py:99-177, metric == "cosine"."""

    def __init__(self, matching_threshold, budget=None):
        self.matching_threshold = matching_threshold
        self.budget = budget
        self.samples = {}

    def partial_fit(self, features, targets, active_targets):
        for f, t in zip(features, targets):
            self.samples.setdefault(t, []).append(f)
            if self.budget is not None:
                self.samples[t] = self.samples[t][-self.budget:]
        self.samples = {k: self.samples[k] for k in active_targets}

    def distance(self, features, targets):
        cost = np.zeros((len(targets), len(features)))
        for r, t in enumerate(targets):
            cost[r, :] = nn_cosine_distance(self.samples[t], features)
        return cost


# --------------------------------------------------------------------------- assignment
def min_cost_matching(cost_fn, max_distance, tracks, dets, track_idx, det_idx):
    """linear_assignment.py:11-75."""
    if len(det_idx) == 0 or len(track_idx) == 0:
        return [], list(track_idx), list(det_idx)
    cost = cost_fn(tracks, dets, track_idx, det_idx)
    cost[cost > max_distance] = max_distance + 1e-5
    rows, cols = linear_sum_assignment(cost)
    matches, un_t, un_d = [], [], []
    colset, rowset = set(cols.tolist()), set(rows.tolist())
    for c, d in enumerate(det_idx):
        if c not in colset:
            un_d.append(d)
    for r, t in enumerate(track_idx):
        if r not in rowset:
            un_t.append(t)
    for r, c in zip(rows, cols):
        if cost[r, c] > max_distance:
            un_t.append(track_idx[r])
            un_d.append(det_idx[c])
        else:
            matches.append((track_idx[r], det_idx[c]))
    return matches, un_t, un_d


def matching_cascade(cost_fn, max_distance, depth, tracks, dets, track_idx, det_idx=None):
    """linear_assignment.py:78-141."""
    if det_idx is None:
        det_idx = list(range(len(dets)))
    un_d = det_idx
    matches = []
    for level in range(depth):
        if len(un_d) == 0:
            break
        lvl = [k for k in track_idx if tracks[k].time_since_update == 1 + level]
        if not lvl:
            continue
        m, _, un_d = min_cost_matching(cost_fn, max_distance, tracks, dets, lvl, un_d)
        matches += m
    un_t = list(set(track_idx) - set(k for k, _ in matches))
    return matches, un_t, un_d


def gate_cost_matrix(cost, tracks, dets, track_idx, det_idx):
    """linear_assignment.py:144-190 (only_position=False)."""
    zs = np.asarray([tlwh_to_xyah(dets[i].tlwh) for i in det_idx])
    for r, t in enumerate(track_idx):
        d2 = kf_gating_distance(tracks[t].mean, tracks[t].covariance, zs)
        cost[r, d2 > CHI2INV95_4] = INFTY_COST
    return cost


def iou_cost(tracks, dets, track_idx, det_idx):
    """iou_matching.py:42-81."""
    cost = np.zeros((len(track_idx), len(det_idx)))
    for r, t in enumerate(track_idx):
        if tracks[t].time_since_update > 1:
            cost[r, :] = INFTY_COST
            continue
        cands = np.asarray([dets[i].tlwh for i in det_idx])
        cost[r, :] = 1.0 - iou(mean_to_tlwh(tracks[t].mean), cands)
    return cost


# --------------------------------------------------------------------------- tracker
TENTATIVE, CONFIRMED, DELETED = 1, 2, 3      # track.py:5-17


class Det:
    """detection.py:29-50."""

    def __init__(self, tlwh, label, confidence, feature):
        self.tlwh = np.asarray(tlwh, dtype=float)
        self.label = label
        self.confidence = float(confidence)
        self.feature = np.asarray(feature, dtype=np.float32)

    def to_xyah(self):
        return tlwh_to_xyah(self.tlwh)

    def to_tlbr(self):
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r


class Trk:
    """track.py:67-209 minus get_label voting (host-only; see oracle/countline_np.py)."""

    def __init__(self, mean, cov, track_id, n_init, max_age, det):
        self.mean, self.covariance = mean, cov
        self.track_id = track_id
        self.hits, self.age, self.time_since_update = 1, 1, 0
        self.state = TENTATIVE
        self.features = [det.feature]
        self.labels = [det.label]
        self.dist = {det.label: [det.confidence]}
        self._n_init, self._max_age = n_init, max_age

    def to_tlwh(self):
        return mean_to_tlwh(self.mean)

    def to_tlbr(self):
        return mean_to_tlbr(self.mean)

    def predict(self):
        self.mean, self.covariance = kf_predict(self.mean, self.covariance)
        self.age += 1
        self.time_since_update += 1

    def update(self, det):
        self.mean, self.covariance = kf_update(self.mean, self.covariance, det.to_xyah())
        self.features.append(det.feature)
        self.hits += 1
        self.time_since_update = 0
        if self.state == TENTATIVE and self.hits >= self._n_init:
            self.state = CONFIRMED
        self.labels.append(det.label)
        self.dist.setdefault(det.label, []).append(det.confidence)

    def mark_missed(self):
        if self.state == TENTATIVE:
            self.state = DELETED
        elif self.time_since_update > self._max_age:
            self.state = DELETED

    def is_confirmed(self):
        return self.state == CONFIRMED

    def is_tentative(self):
        return self.state == TENTATIVE

    def is_deleted(self):
        return self.state == DELETED

    def get_label(self, return_confidence=False):
        """track.py:154-188 -- Dirichlet-multinomial vote + motorbike/bicycle rule."""
        if not self.labels:
            return (None, 0) if return_confidence else None
        stats = [(l, len(s), np.average(s)) for l, s in self.dist.items()]
        alphas = np.array([a for _, _, a in stats])
        counts = np.array([c for _, c, _ in stats])
        names = [l for l, _, _ in stats]
        ranked = list(reversed(sorted(zip((alphas + counts) / (counts.sum() + alphas.sum()), names))))
        pick = ranked[0][1]
        if len(ranked) > 1 and ranked[0][1] == 'motorbike' and ranked[1][1] == 'bicycle':
            pick = 'motorbike' if ranked[0][0] > ranked[1][0] * 4 else 'bicycle'
        return (pick, np.average(self.dist[pick])) if return_confidence else pick


class Tracker:
    """tracker.py:40-138."""

    def __init__(self, metric, max_iou_distance=0.7, max_age=30, n_init=3):
        self.metric = metric
        self.max_iou_distance = max_iou_distance
        self.max_age, self.n_init = max_age, n_init
        self.tracks, self.deleted_tracks = [], []
        self._next_id = 1

    def predict(self):
        for t in self.tracks:
            t.predict()

    def _gated_metric(self, tracks, dets, track_idx, det_idx):
        feats = np.array([dets[i].feature for i in det_idx])
        targets = np.array([tracks[i].track_id for i in track_idx])
        cost = self.metric.distance(feats, targets)
        return gate_cost_matrix(cost, tracks, dets, track_idx, det_idx)

    def _match(self, dets):
        confirmed = [i for i, t in enumerate(self.tracks) if t.is_confirmed()]
        unconfirmed = [i for i, t in enumerate(self.tracks) if not t.is_confirmed()]
        m_a, un_t_a, un_d = matching_cascade(
            self._gated_metric, self.metric.matching_threshold, self.max_age,
            self.tracks, dets, confirmed)
        iou_cands = unconfirmed + [k for k in un_t_a if self.tracks[k].time_since_update == 1]
        un_t_a = [k for k in un_t_a if self.tracks[k].time_since_update != 1]
        m_b, un_t_b, un_d = min_cost_matching(
            iou_cost, self.max_iou_distance, self.tracks, dets, iou_cands, un_d)
        return m_a + m_b, list(set(un_t_a + un_t_b)), un_d

    def update(self, dets):
        matches, un_t, un_d = self._match(dets)
        for t, d in matches:
            self.tracks[t].update(dets[d])
        for t in un_t:
            self.tracks[t].mark_missed()
        for d in un_d:
            mean, cov = kf_initiate(dets[d].to_xyah())
            self.tracks.append(Trk(mean, cov, self._next_id, self.n_init, self.max_age, dets[d]))
            self._next_id += 1
        self.deleted_tracks = [t for t in self.tracks if t.is_deleted()]
        self.tracks = [t for t in self.tracks if not t.is_deleted()]
        active = [t.track_id for t in self.tracks if t.is_confirmed()]
        feats, targets = [], []
        for t in self.tracks:
            if not t.is_confirmed():
                continue
            feats += t.features
            targets += [t.track_id for _ in t.features]
            t.features = []
        self.metric.partial_fit(np.asarray(feats), np.asarray(targets), active)
