"""Thin host wrapper around dd_net_* : compile once, run batches, read the output tensor."""
import ctypes
import numpy as np
import torch

from ._lib import lib, check, P
from .runtime import default_context, ptr
from . import nets


class Net:
    def __init__(self, program, max_batch, context=None, shared=False):
        """shared=True: activation buffers overlaid by lifetime (dd_net_create_shared): only the output tensor can be read back."""
        self.ctx = context or default_context()
        self.program = program
        self.max_batch = int(max_batch)
        words, blob = program.serialize()
        self._words = words
        blob_arr = np.frombuffer(blob, dtype=np.uint8)
        h = P()
        create = lib().dd_net_create_shared if shared else lib().dd_net_create
        check(create(self.ctx.handle, ptr(words), len(words), ptr(blob_arr), len(blob), self.max_batch, ctypes.byref(h)), 'dd_net_create')
        self._h = h
        t = program.tensors[program.out_tensor]
        self.out_shape = (t['h'], t['w'], t['cs'])
        self.out_dtype = {nets.DT_F16: torch.float16, nets.DT_F32: torch.float32, nets.DT_U8: torch.uint8}[t['dtype']]
        self.weight_bytes = len(blob)

    def __del__(self):
        try:
            if self._h:
                lib().dd_net_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def activation_bytes(self):
        v = ctypes.c_int64()
        check(lib().dd_net_activation_bytes(self._h, ctypes.byref(v)), 'dd_net_activation_bytes')
        return v.value

    def use_graph(self, enable=True):
        """Latency mode: replay each (input buffer, batch) forward as one hipGraph launch."""
        check(lib().dd_net_use_graph(self._h, int(bool(enable))), 'dd_net_use_graph')

    def ssd_decode(self, anchors, score_thr, enable=True):
        """SSD detector: first stage of TFLite_Detection_PostProcess inside the head layers' epilogues (the head matrix is then
        never written); results via ssd_decoded()."""
        if not enable:
            check(lib().dd_net_ssd_decode(self._h, None, 0, 0.0, 0), 'dd_net_ssd_decode')
            self._dec_anchors = 0
            return
        a = np.ascontiguousarray(anchors, dtype=np.float32)
        check(lib().dd_net_ssd_decode(self._h, ptr(a), len(a), float(score_thr), int(bool(enable))), 'dd_net_ssd_decode')
        self._dec_anchors = len(a) if enable else 0

    def ssd_decoded(self, n=None):
        """(boxes [n, A, 4] f32, scores [n, A] f32, classes [n, A] i32, keys [n, A] f32) of the last forward, on the host."""
        n = self._last_n if n is None else n
        A = self._dec_anchors
        out = (np.zeros((n, A, 4), np.float32), np.zeros((n, A), np.float32), np.zeros((n, A), np.int32), np.zeros((n, A), np.float32))
        check(lib().dd_net_ssd_decoded_read(self._h, n, *[ptr(o) for o in out]), 'dd_net_ssd_decoded_read')
        return out

    def yolo_decode(self, enable=True):
        """YOLOv5 Detect heads reduce their rows to (box, confidence, class) in their epilogue (dd_net_yolo_decode): `read()` of the
        head matrix then raises; `yolo_decoded()` returns what tools/yolov5.py:121-128 computes from it."""
        check(lib().dd_net_yolo_decode(self._h, int(bool(enable))), 'dd_net_yolo_decode')
        self._yolo_rows = int(self.program.meta['rows']) if enable else 0

    def yolo_decoded(self, n=None):
        """(boxes [n, R, 4] f32 xywh, conf [n, R] f32, classes [n, R] i32) of the last forward, on the host."""
        n = self._last_n if n is None else n
        R = self._yolo_rows
        out = (np.zeros((n, R, 4), np.float32), np.zeros((n, R), np.float32), np.zeros((n, R), np.int32))
        check(lib().dd_net_yolo_decoded_read(self._h, n, *[ptr(o) for o in out]), 'dd_net_yolo_decoded_read')
        return out

    def forward(self, images, stream=None):
        """images: u8 [n, in_h, in_w, 3] torch cuda tensor (or numpy, uploaded).  Enqueues only."""
        if isinstance(images, np.ndarray):
            images = self.ctx.to_device(images, np.uint8)
        n = int(images.shape[0])
        assert images.dtype == torch.uint8 and tuple(images.shape[1:]) == (self.program.in_h, self.program.in_w, 3)
        check(lib().dd_net_forward(self._h, ptr(images), n, stream), 'dd_net_forward')
        self._last_n = n
        return n

    def read(self, n=None, tensor=-1, to_host=True):
        """Output rows of the last forward: [n, h*w, cs] (host numpy or a fresh device tensor)."""
        n = self._last_n if n is None else n
        t = self.program.tensors[self.program.out_tensor if tensor < 0 else tensor]
        shape = (t['h'] + 2, t['cs'] // 16, t['w'] + 2, 16) if t.get('q16') else (t['h'], t['w'], t['cs'])   # bordered uint8 layout: as it lies (netsq.unpack_q16)
        dt = {nets.DT_F16: torch.float16, nets.DT_F32: torch.float32, nets.DT_U8: torch.uint8}[t['dtype']]
        if to_host:
            out = np.zeros((n,) + shape, dtype={torch.float16: np.float16, torch.float32: np.float32, torch.uint8: np.uint8}[dt])
            check(lib().dd_net_read(self._h, tensor, n, ptr(out), 0, None), 'dd_net_read')
            return out
        out = self.ctx.empty((n,) + shape, dt)
        check(lib().dd_net_read(self._h, tensor, n, ptr(out), 1, None), 'dd_net_read')
        return out

    def output_ptr(self, tensor=-1):
        p = P()
        check(lib().dd_net_output(self._h, tensor, ctypes.byref(p), None, None, None, None, None), 'dd_net_output')
        return p.value
