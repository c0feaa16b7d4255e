#!/usr/bin/env python3
"""Per-op device time of a network program (HIP events on the launch stream) next to each op's
algorithmic FLOPs / bytes: which layers are far from their roofline."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepdish_amd import nets
from deepdish_amd.engine import Net
from deepdish_amd._lib import lib, check
from deepdish_amd.profile import net_op_times, net_op_launches, OPK_NAMES, OPK_FOLDED, OPK_FOLDED_PREV

kind, batch = sys.argv[1], int(sys.argv[2])
reps = 20
if kind == 'ssd':
    prog = nets.compile_ssd_mobilenet(nets.synthetic_ssd_weights()); shape = (300, 300)
elif kind in ('ssd_i8', 'ssd_i8_sym'):                                    # the uint8 model (csrc/netsq.hip); _sym: weight zero points at 128
    from deepdish_amd import quantize, netsq
    prog = netsq.compile_ssd_mobilenet_quant(quantize.synthetic_ssd_quant_model(symmetric_weights=kind.endswith('sym'))); shape = (300, 300)
elif kind == 'mars':
    prog = nets.compile_mars(nets.synthetic_mars_weights()); shape = (64, 32)
else:
    prog = nets.compile_yolov5s(nets.synthetic_yolov5s_weights()); shape = (640, 640)
net = Net(prog, max_batch=batch)
if kind.startswith('ssd') and os.environ.get('DD_SSD_DEC', '1') != '0':      # as the pipeline runs it: the heads decode in their epilogue
    net.ssd_decode(prog.meta['anchors'], 1e-8)
if kind not in ('ssd', 'ssd_i8', 'ssd_i8_sym', 'mars') and os.environ.get('DD_YOLO_DEC', '1') != '0':      # as the pipeline runs it: the Detect heads reduce their rows
    net.yolo_decode(True)
x = torch.randint(0, 256, (batch,) + shape + (3,), dtype=torch.uint8, device='cuda')
check(lib().dd_net_profile(net._h, 1))
acc = np.zeros(len(prog.ops))
for r in range(reps + 3):
    net.forward(x); net.ctx.sync()
    if r >= 3:
        acc += net_op_times(net)
acc /= reps
tot = 0
print(f'{kind} batch {batch}: op  kernel  ms  GFLOP  MB  TFLOP/s  GB/s')
codes = net_op_launches(net)
pend = None                                              # ops folded into the NEXT op's launch: that launch's row carries their FLOPs, reads the first one's source
for i, (ms, info, op) in enumerate(zip(acc, prog.info, prog.ops)):
    fl, by = info['flops'] * batch, info['bytes'] * batch + info.get('wbytes', 0)
    tot += ms
    if codes[i] == OPK_FOLDED:
        if pend is None:
            pend = [info['flops'] * batch, info.get('src_bytes', 0) * batch + info.get('wbytes', 0)]
        else:
            pend[0] += info['flops'] * batch; pend[1] += info.get('wbytes', 0)
    elif pend is not None and codes[i] != OPK_FOLDED_PREV:
        fl, by = fl + pend[0], by - info.get('src_bytes', 0) * batch + pend[1]
        pend = None
    kname = '(in the next launch)' if codes[i] == OPK_FOLDED else '(in the previous launch)' if codes[i] == OPK_FOLDED_PREV else OPK_NAMES.get(int(codes[i]), info['kernel'])
    print(f'{i:3d} {kname:26s} {ms*1e3:8.1f}us {fl/1e9:8.3f} {by/1e6:8.2f} {fl/ms/1e9 if ms else 0:8.1f} {by/ms/1e6 if ms else 0:8.1f}  k={op[5]}x{op[6]} s={op[7]} cin={op[10]} cout={op[11]} hw={op[26]}x{op[27]}')
print('total ms', tot, 'GFLOP', sum(i['flops'] for i in prog.info) * batch / 1e9)
