"""NearestNeighborDistanceMetric (deep_sort/nn_matching.py:99-177 upstream).

Stand-alone use (partial_fit / distance) runs csrc/cost.hip's exact-f32 MFMA kernel on a gallery
uploaded per call.  When handed to `Tracker`, only matching_threshold and budget are read: the
tracker keeps its own gallery resident in HBM.
"""
import numpy as np
import torch

from .._lib import lib, check
from ..runtime import default_context, ptr


class NearestNeighborDistanceMetric(object):
    def __init__(self, metric, matching_threshold, budget=None):
        if metric != 'cosine':
            # deepdish.py:516 only ever asks for "cosine"; the euclidean variant is not on the hot path
            raise ValueError("Invalid metric; this build implements 'cosine' only")
        self.matching_threshold = matching_threshold
        self.budget = budget
        self.samples = {}

    def partial_fit(self, features, targets, active_targets):
        for feature, target in zip(features, targets):
            self.samples.setdefault(target, []).append(np.asarray(feature, dtype=np.float32))
            if self.budget is not None:
                self.samples[target] = self.samples[target][-self.budget:]
        self.samples = {k: self.samples[k] for k in active_targets}

    def distance(self, features, targets, context=None):
        ctx = context or default_context()
        feats = np.asarray(features, dtype=np.float32).reshape(-1, 128)
        nt, nd = len(targets), len(feats)
        if nt == 0 or nd == 0:
            return np.zeros((nt, nd))
        offsets = np.zeros(nt + 1, dtype=np.int32)
        rows = []
        for i, t in enumerate(targets):
            rows += list(self.samples[t])
            offsets[i + 1] = len(rows)
        gal = ctx.to_device(np.asarray(rows, dtype=np.float32).reshape(-1, 128))
        df = ctx.to_device(feats)
        out = ctx.empty((nt, nd), torch.float64)
        check(lib().dd_cosine_nn_cost(ctx.handle, ptr(gal), ptr(offsets), nt, ptr(df), nd, ptr(out), None),
              'dd_cosine_nn_cost')
        return ctx.to_host(out)
