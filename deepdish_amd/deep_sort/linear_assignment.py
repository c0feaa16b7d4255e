"""Thresholded assignment, matching cascade and chi-square gating with the reference's signatures
(deep_sort/linear_assignment.py:11-190 upstream).  The solver is the library's host LSAP."""
import numpy as np

from .._lib import lib, check
from ..runtime import ptr
from . import kalman_filter

INFTY_COST = 1e+5


def linear_sum_assignment(cost_matrix):
    c = np.ascontiguousarray(cost_matrix, dtype=np.float64)
    nr, nc = c.shape
    k = min(nr, nc)
    rows, cols = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32)
    check(lib().dd_lsap_host(ptr(c), nr, nc, ptr(rows), ptr(cols)), 'dd_lsap_host')
    return rows.astype(np.int64), cols.astype(np.int64)


def min_cost_matching(distance_metric, max_distance, tracks, detections, track_indices=None,
                      detection_indices=None):
    if track_indices is None:
        track_indices = np.arange(len(tracks))
    if detection_indices is None:
        detection_indices = np.arange(len(detections))
    if len(detection_indices) == 0 or len(track_indices) == 0:
        return [], track_indices, detection_indices
    cost = distance_metric(tracks, detections, track_indices, detection_indices)
    cost[cost > max_distance] = max_distance + 1e-5
    row_ind, col_ind = linear_sum_assignment(cost)
    used_r, used_c = set(row_ind.tolist()), set(col_ind.tolist())
    matches = []
    unmatched_detections = [d for c, d in enumerate(detection_indices) if c not in used_c]
    unmatched_tracks = [t for r, t in enumerate(track_indices) if r not in used_r]
    for r, c in zip(row_ind, col_ind):
        if cost[r, c] > max_distance:
            unmatched_tracks.append(track_indices[r])
            unmatched_detections.append(detection_indices[c])
        else:
            matches.append((track_indices[r], detection_indices[c]))
    return matches, unmatched_tracks, unmatched_detections


def matching_cascade(distance_metric, max_distance, cascade_depth, tracks, detections,
                     track_indices=None, detection_indices=None):
    if track_indices is None:
        track_indices = list(range(len(tracks)))
    if detection_indices is None:
        detection_indices = list(range(len(detections)))
    unmatched_detections = detection_indices
    matches = []
    for level in range(cascade_depth):
        if len(unmatched_detections) == 0:
            break
        level_tracks = [k for k in track_indices if tracks[k].time_since_update == 1 + level]
        if len(level_tracks) == 0:
            continue
        m, _, unmatched_detections = min_cost_matching(
            distance_metric, max_distance, tracks, detections, level_tracks, unmatched_detections)
        matches += m
    unmatched_tracks = list(set(track_indices) - set(k for k, _ in matches))
    return matches, unmatched_tracks, unmatched_detections


def gate_cost_matrix(kf, cost_matrix, tracks, detections, track_indices, detection_indices,
                     gated_cost=INFTY_COST, only_position=False):
    thr = kalman_filter.chi2inv95[2 if only_position else 4]
    zs = np.asarray([detections[i].to_xyah() for i in detection_indices])
    means = np.asarray([tracks[i].mean for i in track_indices])
    covs = np.asarray([tracks[i].covariance for i in track_indices])
    d2 = kf.gating_distance(means, covs, zs, only_position)          # one launch for all rows
    cost_matrix[d2 > thr] = gated_cost
    return cost_matrix
