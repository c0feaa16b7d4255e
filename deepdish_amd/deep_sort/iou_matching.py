"""IoU cost (deep_sort/iou_matching.py:7-81 upstream) on the HIP kernel of csrc/cost.hip."""
import numpy as np
import torch

from .._lib import lib, check
from ..runtime import default_context, ptr


def _iou_cost_arrays(tlwh_t, tsu, tlwh_d, ctx=None):
    ctx = ctx or default_context()
    a = np.asarray(tlwh_t, dtype=np.float64).reshape(-1, 4)
    b = np.asarray(tlwh_d, dtype=np.float64).reshape(-1, 4)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    da, db = ctx.to_device(a), ctx.to_device(b)
    dt = ctx.to_device(np.asarray(tsu, dtype=np.int32)) if tsu is not None else None
    out = ctx.empty((len(a), len(b)), torch.float64)
    check(lib().dd_iou_cost(ctx.handle, ptr(da), ptr(dt), len(a), ptr(db), len(b), ptr(out), None), 'dd_iou_cost')
    return ctx.to_host(out)


def iou(bbox, candidates):
    """IoU of one tlwh box against candidate rows (no +1 pixel)."""
    return 1.0 - _iou_cost_arrays(np.asarray(bbox)[None, :], None, candidates)[0]


def iou_cost(tracks, detections, track_indices=None, detection_indices=None):
    if track_indices is None:
        track_indices = np.arange(len(tracks))
    if detection_indices is None:
        detection_indices = np.arange(len(detections))
    boxes = [tracks[i].to_tlwh() for i in track_indices]
    tsu = [tracks[i].time_since_update for i in track_indices]
    dets = [detections[i].tlwh for i in detection_indices]
    return _iou_cost_arrays(boxes, tsu, dets)
