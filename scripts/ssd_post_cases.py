"""Second stage of TFLite_Detection_PostProcess (dd_ssd_postprocess_decoded) on crafted per-anchor arrays: candidate counts on both
sides of the 64-row chunk and of the sort's power-of-two paddings, every anchor a candidate, none, tied scores (uint8-like
levels), degenerate and NaN boxes / scores, nested boxes that chain suppressions across chunks.  `cases()` is shared by
tests/test_gpu_detectors.py (against the oracle) and tests/test_gpu_switches.py (DD_NMS_SELECT=0 against the default: same bits);
run as a script it prints a digest of the device outputs."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

A = 1917                                                # the SSD's anchors


def cases():
    out = []
    rng = np.random.default_rng(20261005)
    for m, levels, max_det, thr, iou in ((0, 0, 10, 0.5, 0.6), (1, 0, 10, 0.5, 0.6), (7, 4, 10, 0.5, 0.6), (63, 0, 10, 0.3, 0.6),
                                         (64, 16, 20, 0.3, 0.5), (65, 0, 10, 0.5, 0.6), (129, 8, 64, 0.5, 0.3), (700, 32, 10, 0.4, 0.6),
                                         (1100, 0, 64, 0.2, 0.9), (1917, 64, 33, 1e-8, 0.6), (1917, 0, 10, 1e-8, 0.05), (300, 2, 64, 0.5, 0.999), (700, 0, 64, 0.4, 0.3), (1917, 16, 64, 1e-8, 0.2), (-2500, 20, 64, 0.1, 0.4)):
        A = 3000 if m < 0 else 1917                      # one case beyond 2048 boxes (the 16-candidates-per-thread form)
        m = abs(m)
        cy, cx = rng.random(A, dtype=np.float32), rng.random(A, dtype=np.float32)
        h, w = rng.random(A, dtype=np.float32) * np.float32(0.4), rng.random(A, dtype=np.float32) * np.float32(0.4)
        boxes = np.stack([cy - h, cx - w, cy + h, cx + w], axis=1).astype(np.float32)
        if m >= 64:                                     # clusters: many near-duplicates of a few boxes (long suppression chains) ...
            src = rng.integers(0, 8, A)
            boxes = boxes[src] + (rng.random((A, 4), dtype=np.float32) * np.float32(0.02))
            if iou > 0.3:                               # ... and unordered corners (negative or zero areas: never suppressed, never suppressing)
                boxes[::7] = rng.random((len(boxes[::7]), 4), dtype=np.float32)
        boxes[5] = boxes[4]                             # identical boxes
        boxes[11, 2] = boxes[11, 0]                     # zero height
        boxes[13, 1] = np.nan
        score = rng.random(A, dtype=np.float32) * np.float32(thr * 0.99)          # below the threshold ...
        pick = rng.permutation(A)[:m]
        hi = np.float32(thr) + rng.random(m, dtype=np.float32) * np.float32(1 - thr)
        if levels:
            hi = np.float32(thr) + np.floor(rng.random(m) * levels).astype(np.float32) * np.float32((1 - thr) / levels)
        score[pick] = hi                                # ... except m of them, at `levels` distinct values when asked (ties)
        if m > 3:
            score[pick[0]] = np.float32(thr)            # exactly at the threshold: a candidate
            score[pick[1]] = np.nan if levels == 0 else score[pick[1]]
        cls = rng.integers(0, 90, A).astype(np.int32)
        keys = np.where(score >= np.float32(thr), score, np.float32(-1)).astype(np.float32)
        out.append(dict(A=A, boxes=boxes, score=score.astype(np.float32), cls=cls, keys=keys, max_det=max_det, thr=thr, iou=iou))
    return out


def run_device(cs):
    """-> per case (boxes [max_det, 4], classes, scores, count) from dd_ssd_postprocess_decoded; cases with the same options go in one batch."""
    import torch
    from deepdish_amd._lib import lib, check
    from deepdish_amd.runtime import default_context, ptr
    ctx = default_context()
    res = [None] * len(cs)
    groups = {}
    for i, c in enumerate(cs):
        groups.setdefault((c['A'], c['max_det'], c['thr'], c['iou']), []).append(i)
    for (A, md, thr, iou), idx in groups.items():
        n = len(idx)
        d = [ctx.to_device(np.stack([cs[i][k] for i in idx]), t) for k, t in (('boxes', np.float32), ('score', np.float32), ('cls', np.int32), ('keys', np.float32))]
        ob, oc, os_, on = ctx.empty((n, md, 4), torch.float32), ctx.empty((n, md), torch.float32), ctx.empty((n, md), torch.float32), ctx.empty((n,), torch.int32)
        check(lib().dd_ssd_postprocess_decoded(ctx.handle, *[ptr(t) for t in d], A, md, float(thr), float(iou), ptr(ob), ptr(oc), ptr(os_), ptr(on), n, None),
              'dd_ssd_postprocess_decoded')
        ob, oc, os_, on = (ctx.to_host(t) for t in (ob, oc, os_, on))
        for j, i in enumerate(idx):
            res[i] = (ob[j], oc[j], os_[j], int(on[j]))
    return res


if __name__ == '__main__':
    h = hashlib.sha256()
    for b, c, s, n in run_device(cases()):
        h.update(np.ascontiguousarray(b[:n]).tobytes()); h.update(np.ascontiguousarray(c[:n]).tobytes())
        h.update(np.ascontiguousarray(s[:n]).tobytes()); h.update(np.int32(n).tobytes())
    print('sha', h.hexdigest()[:16])
