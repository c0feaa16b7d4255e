"""SSD-MobileNet detector plugin with the reference's surface (tools/ssd_mobilenet.py:30-213 upstream):
`SSD_MOBILENET(wanted_labels, model_file, label_file, num_threads, edgetpu)` and
`detect_image(img) -> (boxes tlwh, labels, scores)`.

Device side: Lanczos stretch resize (csrc/image.hip) -> MobileNet-v1 SSD forward (csrc/nets.hip, f16
MFMA) -> anchor decode / sigmoid / fast NMS / top-10 (csrc/post.hip) -> per-class NMS
(csrc/nms.hip mode 1).  Host side: the dozen-element list handling of predict / detect_image.
"""
import ctypes
import os
import numpy as np
import torch

from .._lib import lib, check
from ..runtime import default_context, ptr
from .. import nets, netsq
from ..engine import Net
from .weights_io import load_named_weights, load_ssd_model, ssd_post_options

COCO_LABELS_FALLBACK = None


class SSDMobileNet:
    MAX_DET = 10                                  # the stock export's max_detections; an instance takes its model file's

    def __init__(self, model_path, label_path, num_threads=None, edgetpu=False, libedgetpu=None,
                 score_threshold=0.5, context=None, max_batch=1):
        if edgetpu:
            raise ValueError('EdgeTPU delegates do not exist on MI355X')
        self.ctx = context or default_context()
        self.use_edgetpu = False
        self.num_threads = num_threads
        kind, wd = load_ssd_model(model_path)       # ('uint8', QModel): a quantised model as the reference's own file is (ssd_mobilenet.py:102)
        self.weights = wd
        self.quantized = kind == 'uint8'
        # the post-process op runs with the options its file states (the interpreter at ssd_mobilenet.py:100-109 does): a model that does
        # not come from a file takes the stock export's 10 / 1e-8 / 0.6
        post = ssd_post_options(wd)
        self.MAX_DET = int(post['max_detections'])
        self.nms_score_threshold, self.nms_iou_threshold = float(post['nms_score_threshold']), float(post['nms_iou_threshold'])
        prog = netsq.compile_ssd_mobilenet_quant(wd) if self.quantized else nets.compile_ssd_mobilenet(wd)
        self.net = Net(prog, max_batch=max_batch, context=self.ctx)
        if self.quantized:                           # integer heads: the post-process op's first stage reads the quantised tensors (csrc/netsq.hip)
            self.net.ssd_decode(prog.meta['anchors'], self.nms_score_threshold)
        self.height = self.width = prog.in_h
        self.anchors = prog.meta['anchors']
        self.n_classes = prog.meta['n_classes']
        self._anchors_dev = self.ctx.to_device(self.anchors)
        self.labels = self.load_labels(label_path)
        c = self.ctx
        self._resized = c.empty((1, self.height, self.width, 3), torch.uint8)
        self._boxes = c.empty((self.MAX_DET, 4), torch.float32)
        self._classes = c.empty((self.MAX_DET,), torch.float32)
        self._scores = c.empty((self.MAX_DET,), torch.float32)
        self._count = c.empty((1,), torch.int32)

    def load_labels(self, path):
        with open(path, 'r') as f:
            return {i: line.strip() for i, line in enumerate(f.readlines())}

    # ---- device stages
    def prepare_image_device(self, img_dev, H, W, src_c, swap_rb=False):
        """ssd_mobilenet.py:54-57: img.convert('RGB').resize((w,h), ANTIALIAS) -> u8 [1,h,w,3] in HBM."""
        check(lib().dd_resize_lanczos(self.ctx.handle, ptr(img_dev), H, W, src_c, int(swap_rb),
                                      ptr(self._resized), self.height, self.width, None), 'dd_resize_lanczos')
        return self._resized

    def invoke_device(self, resized_dev, read=True):
        """ssd_mobilenet.py:102-109: interpreter.invoke() and its four output tensors (read=False leaves them in HBM)."""
        self.net.forward(resized_dev)
        if self.quantized:
            P4 = [ctypes.c_void_p() for _ in range(4)]
            check(lib().dd_net_ssd_decoded(self.net._h, *[ctypes.byref(q) for q in P4]), 'dd_net_ssd_decoded')
            check(lib().dd_ssd_postprocess_decoded(self.ctx.handle, P4[0], P4[1], P4[2], P4[3], len(self.anchors), self.MAX_DET,
                                                   self.nms_score_threshold, self.nms_iou_threshold,
                                                   ptr(self._boxes), ptr(self._classes), ptr(self._scores), ptr(self._count), 1, None),
                  'dd_ssd_postprocess_decoded')
        else:
            raw = self.net.output_ptr()
            check(lib().dd_ssd_postprocess(self.ctx.handle, raw, ptr(self._anchors_dev), len(self.anchors), self.n_classes,
                                           self.MAX_DET, self.nms_score_threshold, self.nms_iou_threshold, ptr(self._boxes), ptr(self._classes),
                                           ptr(self._scores), ptr(self._count), None), 'dd_ssd_postprocess')
        if not read:
            return None
        self.ctx.sync()
        return [self._boxes.cpu().numpy(), self._classes.cpu().numpy(), self._scores.cpu().numpy(),
                float(self._count.cpu().numpy()[0])]

    def nms_boxes(self, boxes, labels, scores, iou_threshold):
        """ssd_mobilenet.py:59-98 (one dd_nms_ssd launch per class present)."""
        nboxes, nlabels, nscores = [], [], []
        for c in set(labels):
            inds = np.where(labels == c)
            b, cl, s = boxes[inds], labels[inds], scores[inds]
            k = len(b)
            db, dsc = self.ctx.to_device(b.astype(np.float64)), self.ctx.to_device(s.astype(np.float64))
            out, cnt = self.ctx.empty((k,), torch.int32), self.ctx.empty((1,), torch.int32)
            check(lib().dd_nms_ssd(self.ctx.handle, ptr(db), ptr(dsc), k, float(iou_threshold), ptr(out), ptr(cnt), None),
                  'dd_nms_ssd')
            keep = self.ctx.to_host(out)[:int(self.ctx.to_host(cnt)[0])]
            nboxes.append(b[keep]); nlabels.append(cl[keep]); nscores.append(s[keep])
        return nboxes, nlabels, nscores

    def _names(self, labels):
        names = []
        for li in labels:                                          # ssd_mobilenet.py:142-147
            if 0 <= li < len(self.labels) - 1:
                names.append(self.labels[li + 1])
            else:
                print("Invalid label index: {} in {}".format(li, labels))
        return names

    def finish_device(self, boxes_dev, classes_dev, scores_dev, n, confidence=0.5, iou_threshold=0.5,
                      original_image_size=None):
        """ssd_mobilenet.py:111-150 on the post-process op's outputs where they lie in HBM (csrc/post.hip
        ssd_finish_k): NaN scrub, confidence filter, reorder + scale, per-class nms_boxes -- one launch."""
        w, h = original_image_size if original_image_size is not None else (self.width, self.height)
        c = self.ctx
        ob, oc = c.empty((n, 4), torch.float64), c.empty((n,), torch.int32)
        osc, on = c.empty((n,), torch.float64), c.empty((1,), torch.int32)
        check(lib().dd_ssd_detections(c.handle, ptr(boxes_dev), ptr(classes_dev), ptr(scores_dev), 1, n, float(confidence),
                                      float(iou_threshold), float(w), float(h), ptr(ob), ptr(oc), ptr(osc), ptr(on), None),
              'dd_ssd_detections')
        k = int(c.to_host(on)[0])
        if k == 0:
            return [], [], []
        labels = c.to_host(oc)[:k].astype(np.uint)
        return c.to_host(ob)[:k], self._names(labels), c.to_host(osc)[:k].astype(np.float32)

    def postprocess(self, output, confidence=0.5, iou_threshold=0.5, original_image_size=None):
        """ssd_mobilenet.py:111-150 for host arrays [boxes, classes, scores, count] (the four tensors predict() reads)."""
        n = len(output[2])
        c = self.ctx
        return self.finish_device(c.to_device(output[0], np.float32), c.to_device(output[1], np.float32),
                                  c.to_device(output[2], np.float32), n, confidence, iou_threshold, original_image_size)

    def predict_array(self, rgb, confidence=0.5, iou_threshold=0.5):
        """rgb: u8 ndarray [H,W,3 or 4] in RGB(A) order, original resolution."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        H, W, C = rgb.shape
        dev = self.ctx.to_device(rgb)
        self.invoke_device(self.prepare_image_device(dev, H, W, C), read=False)
        return self.finish_device(self._boxes, self._classes, self._scores, self.MAX_DET, confidence, iou_threshold, (W, H))


class SSD_MOBILENET():
    def __init__(self, wanted_labels=None, model_file=None, label_file=None, num_threads=None, edgetpu=False,
                 libedgetpu=None, score_threshold=0.5, context=None):
        if model_file is None:
            model_file = 'ssd_mobilenet.tflite'
        if label_file is None:
            label_file = 'coco_labelmap.txt'
        self.ssdm = SSDMobileNet(model_file, label_file, num_threads=num_threads, edgetpu=edgetpu,
                                 libedgetpu=libedgetpu, score_threshold=score_threshold, context=context)
        self.wanted_labels = ['person'] if wanted_labels is None else wanted_labels
        self.score_threshold = score_threshold
        self.labels = self.ssdm.labels
        self.width, self.height = self.ssdm.width, self.ssdm.height
        self.use_edgetpu = self.ssdm.use_edgetpu
        self.num_threads = self.ssdm.num_threads

    def _filter(self, boxes, labels, scores):
        rb, rl, rs = [], [], []
        for i in range(len(boxes)):                                # ssd_mobilenet.py:204-212
            if labels[i] in self.wanted_labels and scores[i] >= self.score_threshold:
                box = boxes[i]
                rb.append([box[0], box[1], box[2] - box[0], box[3] - box[1]])
                rl.append(labels[i])
                rs.append(scores[i])
        return rb, rl, rs

    def detect_image(self, img):
        """img: PIL image (RGB/RGBA, as deepdish.py:882 builds it) or an RGB(A) ndarray."""
        arr = np.asarray(img)
        return self._filter(*self.ssdm.predict_array(arr))

    def detect_frame_device(self, frame_dev, H, W):
        """Hot path: BGR u8 frame already in HBM (the BGR->RGB swap of deepdish.py:882 is fused into
        the resize kernel)."""
        s = self.ssdm
        s.invoke_device(s.prepare_image_device(frame_dev, H, W, 3, swap_rb=True), read=False)
        return self._filter(*s.finish_device(s._boxes, s._classes, s._scores, s.MAX_DET, original_image_size=(W, H)))
