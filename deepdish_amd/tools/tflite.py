"""Generic TFLite-Task-style detector plugin with the reference's surface:
`TFLITE(wanted_labels, model_file, label_file, num_threads, edgetpu)` (tools/tflite.py:9-41 upstream) over
an `ObjectDetector` (tools/tflite_object_detector.py:91-295).

Device side: cv2-style bilinear stretch (csrc/image.hip) -> (x - mean) / std fused into the network's
input op -> SSD forward (csrc/nets.hip) -> the detection post-process op (csrc/post.hip).  Host side:
the integer truncation, score sort and allow/deny filtering of `_postprocess`.
"""
import numpy as np
import torch
from typing import List, NamedTuple, Optional

from .._lib import lib, check
from ..runtime import default_context, ptr
from .. import nets
from ..engine import Net
from .weights_io import load_named_weights, load_ssd_model, ssd_post_options


class ObjectDetectorOptions(NamedTuple):
    """tflite_object_detector.py:45-66"""
    enable_edgetpu: bool = False
    label_allow_list: Optional[List[str]] = None
    label_deny_list: Optional[List[str]] = None
    max_results: int = -1
    num_threads: int = 1
    score_threshold: float = 0.0


class Rect(NamedTuple):
    left: float
    top: float
    right: float
    bottom: float


class Category(NamedTuple):
    label: str
    score: float
    index: int


class Detection(NamedTuple):
    bounding_box: Rect
    categories: List[Category]


class ObjectDetector:
    MAX_DET = 10

    def __init__(self, model_path, options=ObjectDetectorOptions(), label_file=None, context=None):
        if options.enable_edgetpu:
            raise OSError("Coral EdgeTPU delegates do not exist on MI355X")
        self.ctx = context or default_context()
        kind, wd = load_ssd_model(model_path)                 # an SSD-MobileNet .tflite file goes through tools/tflite_reader.py
        if kind != 'f32':
            raise ValueError('%s: the generic TFLite adaptor is built for float models (a uint8 SSD-MobileNet runs behind tools/ssd_mobilenet.py)' % model_path)
        self.weights = wd
        post = ssd_post_options(wd)                           # the post-process op's own options (its file's, else the stock export's)
        self.MAX_DET = int(post['max_detections'])
        self._score_thr, self._iou_thr = float(post['nms_score_threshold']), float(post['nms_iou_threshold'])
        # tflite_object_detector.py:117-137: mean / std and the label list come from the model file's own metadata.  A model WITHOUT metadata
        # (the reference's MetadataDisplayer raises on one) is taken with a label_file and the defaults: an extension for this build's
        # synthetic / .npz models, never consulted when the file carries metadata (the reference ignores its label_file argument, tflite.py:16-23).
        import os
        meta = None
        if str(model_path).endswith('.tflite') and os.path.exists(str(model_path)):
            from . import tflite_reader
            try:
                meta = tflite_reader.read_metadata(str(model_path))
            except tflite_reader.UnsupportedModel:
                if label_file is None:
                    raise
        elif label_file is None:
            raise ValueError('%s: not a .tflite file with metadata, and no label_file given' % model_path)
        self._mean, self._std = (meta['mean'], meta['std']) if meta else (127.5, 127.5)     # :124-131 (defaults without a NormalizationOptions unit)
        prog = nets.compile_ssd_mobilenet(wd, mean=self._mean, std=self._std)     # float model: (x - mean) / std in the input op (:222-224)
        self.net = Net(prog, max_batch=1, context=self.ctx)
        self._input_size = prog.in_w, prog.in_h
        self._is_quantized_input = False
        self._anchors = prog.meta['anchors']
        self._anchors_dev = self.ctx.to_device(self._anchors)
        self._n_classes = prog.meta['n_classes']
        if meta:
            self._label_list = list(meta['labels'])           # :134-137 the packed label map: class id 0 is its first line
        else:
            with open(label_file) as f:                       # (no metadata) a label file in the SSD adaptor's form: line 0 is the background entry
                lines = [l.strip() for l in f.readlines()]
            self._label_list = list(filter(len, lines[1:]))
        self._options = options
        c = self.ctx
        self._resized = c.empty((1, prog.in_h, prog.in_w, 3), torch.uint8)
        self._boxes = c.empty((self.MAX_DET, 4), torch.float32)
        self._classes = c.empty((self.MAX_DET,), torch.float32)
        self._scores = c.empty((self.MAX_DET,), torch.float32)
        self._count = c.empty((1,), torch.int32)

    def detect(self, input_image):
        """input_image: RGB u8 [H, W, 3] (tflite_object_detector.py:180-205)."""
        img = np.ascontiguousarray(input_image, dtype=np.uint8)
        image_height, image_width, _ = img.shape
        w, h = self._input_size
        check(lib().dd_resize_bilinear(self.ctx.handle, ptr(self.ctx.to_device(img)), image_height, image_width, 3,
                                       ptr(self._resized), h, w, None), 'dd_resize_bilinear')      # :211 cv2.resize
        self.net.forward(self._resized)
        check(lib().dd_ssd_postprocess(self.ctx.handle, self.net.output_ptr(), ptr(self._anchors_dev), len(self._anchors),
                                       self._n_classes, self.MAX_DET, self._score_thr, self._iou_thr, ptr(self._boxes), ptr(self._classes),
                                       ptr(self._scores), ptr(self._count), None), 'dd_ssd_postprocess')
        self.ctx.sync()
        return self._postprocess(self._boxes.cpu().numpy(), self._classes.cpu().numpy(), self._scores.cpu().numpy(),
                                 int(self._count.cpu().numpy()[0]), image_width, image_height)

    def _postprocess(self, boxes, classes, scores, count, image_width, image_height):
        """tflite_object_detector.py:234-295"""
        results = []
        for i in range(count):
            if scores[i] >= self._options.score_threshold:
                y_min, x_min, y_max, x_max = boxes[i]
                bounding_box = Rect(top=int(y_min * image_height), left=int(x_min * image_width),
                                    bottom=int(y_max * image_height), right=int(x_max * image_width))
                class_id = int(classes[i])
                category = Category(score=scores[i], label=self._label_list[class_id], index=class_id)
                results.append(Detection(bounding_box=bounding_box, categories=[category]))
        out = sorted(results, key=lambda d: d.categories[0].score, reverse=True)
        if self._options.label_deny_list is not None:
            out = [d for d in out if d.categories[0].label not in self._options.label_deny_list]
        if self._options.label_allow_list is not None:
            out = [d for d in out if d.categories[0].label in self._options.label_allow_list]
        if self._options.max_results > 0:
            out = out[:min(len(out), self._options.max_results)]
        return out


class TFLITE:
    def __init__(self, wanted_labels=None, model_file=None, label_file=None, num_threads=None, edgetpu=False,
                 libedgetpu=None, score_threshold=0.5, context=None):
        self.opts = ObjectDetectorOptions(num_threads=num_threads or 1, score_threshold=score_threshold,
                                          enable_edgetpu=edgetpu)
        self.use_edgetpu = edgetpu
        self.num_threads = num_threads or 1
        self.detector = ObjectDetector(model_path=model_file, options=self.opts, label_file=label_file, context=context)
        self.wanted_labels = ['person'] if wanted_labels is None else wanted_labels
        self.label_list = self.detector._label_list
        self.labels = {i + 1: self.label_list[i] for i in range(0, len(self.label_list))}     # tflite.py:22-23
        self.width, self.height = self.detector._input_size

    def detect_image(self, img):
        dets = self.detector.detect(np.array(img)[..., :3])
        return_boxs, return_lbls, return_scrs = [], [], []
        for det in dets:                                                                      # tflite.py:30-40
            lblscrs = [(w, c.score) for c in det.categories for w in self.wanted_labels if c.label == w]
            if lblscrs:
                b = det.bounding_box
                return_boxs.append([b.left, b.top, b.right - b.left, b.bottom - b.top])
                return_lbls.append(lblscrs[0][0])
                return_scrs.append(lblscrs[0][1])
        return (return_boxs, return_lbls, return_scrs)

    def detect_frame_device(self, frame_dev, H, W):
        """BGR frame in HBM -> RGB host array -> detect (the generic adaptor is not on the batched path)."""
        return self.detect_image(frame_dev.cpu().numpy()[..., ::-1])
