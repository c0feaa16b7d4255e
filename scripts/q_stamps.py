import sys, os
sys.path.insert(0, os.getcwd())
import torch
from deepdish_amd import quantize, netsq
from deepdish_amd.engine import Net
prog = netsq.compile_ssd_mobilenet_quant(quantize.synthetic_ssd_quant_model())
net = Net(prog, max_batch=384)
x = torch.randint(0, 256, (384, 300, 300, 3), dtype=torch.uint8, device='cuda')
for r in range(2):
    net.forward(x); net.ctx.sync()
    print('---', file=sys.stderr)
