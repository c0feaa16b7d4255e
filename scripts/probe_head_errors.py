"""Measure the element-wise error of the SSD / YOLOv5 heads and the MARS features against the f32 oracle
(what tests/test_gpu_nets.py asserts; run on the GPU box, prints the numbers DESIGN.md quotes)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepdish_amd import nets
from deepdish_amd.engine import Net
from oracle import nets_torch


def stats(name, got, want):
    err = np.abs(got - want)
    rms = float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))
    for rtol in (0.0, 5e-3, 1e-2, 2e-2):
        need = float((err - rtol * np.abs(want)).max())
        print('%s: rtol %.0e -> atol needed %.3e (%.2f %% of rms %.3f, max|want| %.3f)' % (name, rtol, need, 100 * need / rms, rms, float(np.abs(want).max())))


wd = nets.synthetic_ssd_weights(1234)
net = Net(nets.compile_ssd_mobilenet(wd), max_batch=2)
x = np.random.default_rng(2).integers(0, 256, (2, 300, 300, 3), dtype=np.uint8)
net.forward(x)
got = net.read()[:, :, 0, :]
stats('ssd box  vs w16', got[..., :4], nets_torch.ssd_forward(wd, x, w16=True)[..., :4])
stats('ssd cls  vs w16', got[..., 4:], nets_torch.ssd_forward(wd, x, w16=True)[..., 4:])
stats('ssd all  vs f32', got, nets_torch.ssd_forward(wd, x, w16=False))
wd = nets.synthetic_yolov5s_weights(1234)
net = Net(nets.compile_yolov5s(wd), max_batch=1)
x = np.random.default_rng(3).integers(0, 256, (1, 640, 640, 3), dtype=np.uint8)
net.forward(x)
got = net.read()[:, :, 0, :]
want = nets_torch.yolov5s_forward(wd, x, w16=True)
stats('yolo xywh vs w16', got[..., :4], want[..., :4])
stats('yolo obj/cls vs w16', got[..., 4:], want[..., 4:])
