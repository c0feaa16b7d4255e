#!/usr/bin/env python3
"""Soak of the kernels with hand-placed synchronisation that round 5 added or rewrote: the SSD post-process NMS (nms_greedy_f32_k: the
per-wave maxima of a round alternate between two LDS rows behind ONE barrier) on the crafted cases of scripts/ssd_post_cases.py, and the
MARS forward (mars_ws128_k / mars_pair64_k: counted vmcnt, bare barriers, LDS-DMA rings; conv3x3_pool_rows_k<STEM>: literal ring slots).
The digest of the outputs must never change from run to run.  Usage: python scripts/soak_post_mars.py [repeats=200] [crops=4096]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ssd_post_cases import cases, run_device

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_crops = int(sys.argv[2]) if len(sys.argv) > 2 else 4096


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


cs = cases()
first, bad = None, 0
for i in range(reps):
    d = digest([x for b, c, s, n in run_device(cs) for x in (b[:n], c[:n], s[:n], np.int32(n))])
    first = first or d
    bad += d != first
print('post-process: %d repeats of %d cases, %d mismatches, digest %s' % (reps, len(cs), bad, first))

from deepdish_amd import nets
from deepdish_amd.engine import Net
prog = nets.compile_mars(nets.synthetic_mars_weights(1234))
net = Net(prog, max_batch=n_crops)
rng = np.random.default_rng(5)
xs = [torch.from_numpy(rng.integers(0, 256, (n, 64, 32, 3), dtype=np.uint8)).cuda() for n in (n_crops, n_crops - 37)]
firsts, bad2 = [None, None], 0
for i in range(reps):
    k = i & 1
    net.forward(xs[k])
    d = digest([net.read()])
    firsts[k] = firsts[k] or d
    bad2 += d != firsts[k]
print('MARS forward: %d forwards of %d / %d crops, %d mismatches, digests %s' % (reps, n_crops, n_crops - 37, bad2, firsts))
sys.exit(1 if bad or bad2 else 0)
