"""non_max_suppression (deep_sort/preprocessing.py:6-73 upstream) on csrc/nms.hip."""
import numpy as np
import torch

from .._lib import lib, check
from ..runtime import default_context, ptr


def non_max_suppression(boxes, max_bbox_overlap, scores=None, context=None):
    """boxes [K,4] tlwh -> list of surviving indices, best first.  Empty input -> []."""
    if len(boxes) == 0:
        return []
    ctx = context or default_context()
    b = np.asarray(boxes).astype(np.float64).reshape(-1, 4)
    keys = np.asarray(scores, dtype=np.float64) if scores is not None else b[:, 3] + b[:, 1]
    k = len(b)
    db, dk = ctx.to_device(b), ctx.to_device(keys)
    out = ctx.empty((k,), torch.int32)
    cnt = ctx.empty((1,), torch.int32)
    check(lib().dd_nms(ctx.handle, ptr(db), ptr(dk), k, float(max_bbox_overlap), ptr(out), ptr(cnt), None), 'dd_nms')
    n = int(ctx.to_host(cnt)[0])
    return [int(i) for i in ctx.to_host(out)[:n]]
