#!/usr/bin/env python3
"""Whole-forward device time of a network (HIP events on the launch stream), no per-op instrumentation.
Usage: python scripts/time_forward.py ssd|ssd_i8|mars|yolo BATCH [kernels]   (kernels: also print which special launches ran)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepdish_amd import nets
from deepdish_amd.engine import Net
from deepdish_amd._lib import lib, check

kind, batch = sys.argv[1], int(sys.argv[2])
if kind == 'ssd':
    prog = nets.compile_ssd_mobilenet(nets.synthetic_ssd_weights()); shape = (300, 300)
elif kind == 'ssd_i8':
    from deepdish_amd import quantize, netsq
    prog = netsq.compile_ssd_mobilenet_quant(quantize.synthetic_ssd_quant_model()); shape = (300, 300)
elif kind == 'mars':
    prog = nets.compile_mars(nets.synthetic_mars_weights()); shape = (64, 32)
else:
    prog = nets.compile_yolov5s(nets.synthetic_yolov5s_weights()); shape = (640, 640)
net = Net(prog, max_batch=batch)
x = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (batch,) + shape + (3,), dtype=np.uint8)).cuda()
ts = net.ctx.torch_stream
for _ in range(5):
    net.forward(x)
net.ctx.sync()
reps = 30
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record(ts)
for r in range(reps):
    net.forward(x)
    ev[r + 1].record(ts)
net.ctx.sync()
us = np.array([ev[r].elapsed_time(ev[r + 1]) * 1e3 for r in range(reps)])
ref = net.read().copy()
if kind == 'ssd_i8':                                       # the class rows are the output tensor; the box rows belong to the digest too
    ref = np.concatenate([np.asarray(ref).reshape(-1), np.asarray(net.read(tensor=prog.meta['box_tensor'])).reshape(-1)])
import hashlib
print(f'{kind} batch {batch}: mean {us.mean():.1f} us  min {us.min():.1f} us  checksum {float(np.abs(ref).sum()):.6e}  sha {hashlib.sha256(np.ascontiguousarray(ref).tobytes()).hexdigest()[:16]}')
if len(sys.argv) > 3 and sys.argv[3] == 'kernels':
    from deepdish_amd.profile import net_op_launches, OPK_NAMES
    print('launches:', ' '.join(sorted({OPK_NAMES[int(c)] for c in net_op_launches(net) if int(c) in OPK_NAMES})))
