"""YOLOv5 detector plugin with the reference's surface (tools/yolov5.py:37-146 upstream).

Device side: Lanczos stretch resize -> YOLOv5s forward with the Detect decode fused into the head
convs (csrc/nets.hip) -> score/argmax/threshold/scale compaction (csrc/post.hip).  No NMS here, as in
the reference: every candidate flows on to deep_sort's non_max_suppression.
"""
import os
import numpy as np
import torch

from .._lib import lib, check
from ..runtime import default_context, ptr
from .. import nets
from ..engine import Net
from .weights_io import load_named_weights, load_yolov5_weights


class YOLOV5:
    MAX_ROWS = 4096

    def __init__(self, wanted_labels=None, model_file=None, label_file=None, num_threads=None, edgetpu=False,
                 libedgetpu=None, score_threshold=0.25, context=None):
        basedir = os.getenv('DEEPDISHHOME', '.')
        if model_file is None:
            model_file = os.path.join(basedir, 'detectors/yolov5/yolov5s-int8.tflite')
        if label_file is None:
            label_file = os.path.join(basedir, 'detectors/yolov5/coco_classes.txt')
        if edgetpu:
            raise ValueError('EdgeTPU delegates do not exist on MI355X')
        self.wanted_labels = ['person'] if wanted_labels is None else wanted_labels
        self.label_file = label_file
        self.score_threshold = score_threshold
        self.labels = self._get_labels()
        self.use_edgetpu = False
        self.num_threads = num_threads
        self.mode = 'hip'
        self.ctx = context or default_context()
        if 'int8' in str(model_file) and str(model_file).endswith('.tflite') and os.path.exists(str(model_file)):
            raise ValueError('%s: the int8 YOLOv5 file (tools/yolov5.py:61,102-104,115-118 upstream) is not built; the float / fp16-weight file is' % model_file)
        wd = load_yolov5_weights(model_file)                     # <file>.tflite (yolov5.py:68-79), .npz of named arrays, or synthetic[:seed]
        self.weights = wd
        prog = nets.compile_yolov5s(wd, int(wd.get('__in_size__', 640)))
        self.net = Net(prog, max_batch=1, context=self.ctx)
        self.height = self.width = prog.in_h
        self.anchors = nets.YOLO_ANCHORS
        self.n_rows, self.n_cls = prog.meta['rows'], prog.meta['n_classes']
        c = self.ctx
        self._resized = c.empty((1, self.height, self.width, 3), torch.uint8)
        self._boxes = c.empty((self.MAX_ROWS, 4), torch.float32)
        self._scores = c.empty((self.MAX_ROWS,), torch.float32)
        self._cls = c.empty((self.MAX_ROWS,), torch.int32)
        self._n = c.empty((1,), torch.int32)

    def _get_labels(self):
        with open(os.path.expanduser(self.label_file)) as f:
            return {i: line.strip() for i, line in enumerate(f.readlines())}

    def _run_device(self, img_dev, H, W, src_c, swap_rb):
        check(lib().dd_resize_lanczos(self.ctx.handle, ptr(img_dev), H, W, src_c, int(swap_rb), ptr(self._resized),
                                      self.height, self.width, None), 'dd_resize_lanczos')       # yolov5.py:99
        self.net.forward(self._resized)                                                          # :107-109
        check(lib().dd_yolov5_decode(self.ctx.handle, self.net.output_ptr(), self.n_rows, self.n_cls,
                                     float(self.score_threshold), float(W), float(H), ptr(self._boxes),
                                     ptr(self._scores), ptr(self._cls), self.MAX_ROWS, ptr(self._n), None),
              'dd_yolov5_decode')                                                                # :120-131
        self.ctx.sync()
        n = min(int(self._n.cpu().numpy()[0]), self.MAX_ROWS)
        return self._boxes[:n].cpu().numpy(), self._scores[:n].cpu().numpy(), self._cls[:n].cpu().numpy()

    def _collect(self, boxes, scores, cls):
        rb, rl, rs = [], [], []
        for xyxy, score, li in zip(boxes, scores, cls):              # yolov5.py:137-145
            label = self.labels[int(li)]
            if label in self.wanted_labels and score >= self.score_threshold:
                tlwh = np.copy(xyxy)
                tlwh[2] = xyxy[2] - xyxy[0]
                tlwh[3] = xyxy[3] - xyxy[1]
                rb.append(list(tlwh)); rl.append(label); rs.append(score)
        return rb, rl, rs

    def detect_image(self, img):
        arr = np.ascontiguousarray(np.asarray(img), dtype=np.uint8)
        H, W, C = arr.shape
        return self._collect(*self._run_device(self.ctx.to_device(arr), H, W, C, False))

    def detect_frame_device(self, frame_dev, H, W):
        return self._collect(*self._run_device(frame_dev, H, W, 3, True))
