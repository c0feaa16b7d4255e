#!/usr/bin/env python3
"""Writes tests/golden/tiny_quant_graph.tflite: a four-operator uint8 graph (CONV_2D, DEPTHWISE_CONV_2D, LOGISTIC, a custom
post-process op with FlexBuffer options) with hand-picked constants, through deepdish_amd/tools/tflite_writer.py.  The committed
file pins the byte format: tests/test_tflite_io.py parses it with deepdish_amd/tools/tflite_reader.py and checks every value
against the numbers below."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from deepdish_amd.tools import tflite_writer, flatbuf

W = tflite_writer.GraphWriter('tiny fixture')
x = W.tensor('input', [1, 8, 8, 3], np.uint8, None, np.float32(0.0078125), 128)
cw = W.tensor('conv/w', [4, 3, 3, 3], np.uint8, (np.arange(108) % 251).astype(np.uint8).reshape(4, 3, 3, 3), np.float32(0.02), 131)
cb = W.tensor('conv/b', [4], np.int32, np.array([-7, 0, 11, 123456], np.int32), np.float32(0.00015625), 0)
c = W.tensor('conv', [1, 4, 4, 4], np.uint8, None, np.float32(6.0 / 255), 0)
W.op('CONV_2D', [x, cw, cb], [c], dict(stride=2, act='relu6'))
dw = W.tensor('dw/w', [1, 3, 3, 4], np.uint8, (200 - np.arange(36)).astype(np.uint8).reshape(1, 3, 3, 4), np.float32(0.011), 97)
db = W.tensor('dw/b', [4], np.int32, np.array([5, -6, 7, -8], np.int32), np.float32(0.0002588), 0)
d = W.tensor('dw', [1, 4, 4, 4], np.uint8, None, np.float32(0.05), 3)
W.op('DEPTHWISE_CONV_2D', [c, dw, db], [d], dict(stride=1, act='none'))
s = W.tensor('scores', [1, 4, 4, 4], np.uint8, None, np.float32(1.0 / 256), 0)
W.op('LOGISTIC', [d], [s])
anc = W.tensor('anchors', [2, 4], np.float32, np.array([[0.5, 0.5, 0.1, 0.2], [0.25, 0.75, 1.0, 1.0]], np.float32))
o = W.tensor('out', [1, 10, 4], np.float32)
W.op('CUSTOM', [d, s, anc], [o], custom='TFLite_Detection_PostProcess',
     custom_options=flatbuf.flex_build_map(dict(max_detections=10, num_classes=90, y_scale=10.0, x_scale=10.0, h_scale=5.0, w_scale=5.0,
                                                nms_iou_threshold=0.6, nms_score_threshold=1e-8, use_regular_nms=False)))
W.inputs, W.outputs = [x], [o]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'tiny_quant_graph.tflite')
open(path, 'wb').write(W.tobytes())
print(path, os.path.getsize(path))
