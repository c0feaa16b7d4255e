#!/usr/bin/env python3
"""Soak of the uint8 detector: N forwards of the same 384 frames, the head tensors' digest must never change (the fused kernels wait for
their LDS-DMA pieces with counted vmcnt and bare barriers: a race would show as a digest that differs from run to run).
Usage: python scripts/soak_uint8.py [forwards=300] [frames=384]"""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepdish_amd import quantize, netsq
from deepdish_amd.engine import Net
n_fwd = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 384
prog = netsq.compile_ssd_mobilenet_quant(quantize.synthetic_ssd_quant_model())
net = Net(prog, max_batch=n)
rng = np.random.default_rng(3)
xs = [torch.from_numpy(rng.integers(0, 256, (n, 300, 300, 3), dtype=np.uint8)).cuda() for _ in range(2)]
digests = [None, None]
bad = 0
for i in range(n_fwd):
    k = i & 1
    net.forward(xs[k])
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(net.read()).tobytes())
    h.update(np.ascontiguousarray(net.read(tensor=prog.meta['box_tensor'])).tobytes())
    d = h.hexdigest()[:16]
    if digests[k] is None:
        digests[k] = d
    elif d != digests[k]:
        bad += 1
        print('forward %d (input %d): digest %s, first was %s' % (i, k, d, digests[k]))
print('%d forwards of %d frames: %d mismatches; digests %s' % (n_fwd, n, bad, digests))
sys.exit(1 if bad else 0)
