"""Minimal per-frame harness around the four hot-path calls the reference's Pipeline makes
(deepdish.py upstream): run_object_detector :880-885 -> box hygiene :940-960 ->
non_max_suppression :995 -> encoder :1008 -> Detection :1014 -> tracker.predict/update :1028-1029 ->
count-line logic :1035-1114.  Everything else of deepdish.py (capture, MQTT, rendering, web UI) is
out of scope.  Stage timings use the reference's short labels (objd, feat, trak, e2e).

Flag names follow deepdish.py:1355-1506 for the parameters that touch the hot path.
"""
import os
from time import time
import numpy as np
import torch

from .background import createBackgroundSubtractorMOG2, motion_filter
from .deep_sort import nn_matching, preprocessing
from .deep_sort.detection import Detection
from .deep_sort.tracker import Tracker
from .tools import generate_detections as gdet
from .tools.countline import CountLine
from .runtime import default_context
from .wire import ResultSink
from .framerecords import FrameRecords, load_cvat_annotations

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LABELS = os.path.join(HERE, 'assets', 'coco_labels_ssd.txt')
DEFAULT_YOLO_LABELS = os.path.join(HERE, 'assets', 'coco_classes.txt')


def make_detector(model, labels=None, wanted_labels=('person',), num_threads=4, context=None):
    """Plugin selection by substring of --model, as deepdish.py:482-502."""
    wanted = list(wanted_labels)
    if 'yolov5' in model:
        from .tools.yolov5 import YOLOV5
        return YOLOV5(wanted_labels=wanted, model_file=model, label_file=labels or DEFAULT_YOLO_LABELS,
                      num_threads=num_threads, context=context)
    if 'yolo' in model or 'saved_model' in model:
        raise ValueError('%s: the Keras YOLOv3 / SavedModel detectors are outside the hot path of this build' % model)
    if 'mobilenet' in model:
        from .tools.ssd_mobilenet import SSD_MOBILENET
        return SSD_MOBILENET(wanted_labels=wanted, model_file=model, label_file=labels or DEFAULT_LABELS,
                             num_threads=num_threads, context=context)
    if 'tflite' in model:
        from .tools.tflite import TFLITE
        return TFLITE(wanted_labels=wanted, model_file=model, label_file=labels or DEFAULT_LABELS,
                      num_threads=num_threads, context=context)
    raise ValueError('Unsure what to do with model file {}'.format(model))


def clean_boxes(boxes0, labels0, scores0, max_x, max_y, motion=None):
    """deepdish.py:940-960.  motion = None (--disable-background-subtraction) or (subtractor, ratio): a
    background.BackgroundSubtractorMOG2 that has seen this frame, and --background-subtraction-ratio (:957)."""
    boxes, labels, scores = [], [], []
    if len(boxes0) and np.any(np.isnan(np.asarray(boxes0, dtype=np.float64))):
        return boxes, labels, scores                      # :947 drops every box of the frame
    for (x, y, w, h), lbl, scr in zip(boxes0, labels0, scores0):
        x, y = int(np.clip(x, 0, max_x)), int(np.clip(y, 0, max_y))
        w, h = int(np.clip(w, 0, max_x - x)), int(np.clip(h, 0, max_y - y))
        if w * h > 0.9 * max_x * max_y:
            continue
        boxes.append((x, y, w, h)); labels.append(lbl); scores.append(scr)
    if motion is not None and boxes:
        sub, ratio = motion
        ok = motion_filter(sub.box_counts(boxes), boxes, ratio)
        boxes, labels, scores = ([v for v, k in zip(seq, ok) if k] for seq in (boxes, labels, scores))
    return boxes, labels, scores


class HotPath:
    """One video stream: detector + encoder + tracker + counters, all device state owned here."""

    def __init__(self, model='synthetic-ssd_mobilenet_v1', encoder_model='synthetic-mars-64x32x3', labels=None,
                 wanted_labels=('person',), input_size=(640, 480), line=None, max_cosine_distance=0.2,
                 nms_max_overlap=0.6, max_iou_distance=0.7, max_age=60, encoder_batch_size=32, num_threads=4,
                 context=None, run_detector=True, disable_background_subtraction=True, background_subtraction_ratio=0.25,
                 enable_background_masking=False, log=None, restore_from_log=False, mqtt_publish=None, mqtt_topic='default/topic',
                 mqtt_acp_id=None, mqtt_verbosity=1, cpu_temp=None, annotations=None):
        self.ctx = context or default_context()
        self.input_size = tuple(input_size)
        # deepdish.py:512,889: the reference defaults to background subtraction ON; its benchmarks (and this class)
        # run with --disable-background-subtraction unless asked otherwise
        self.background_subtraction = not disable_background_subtraction
        self.background_subtraction_ratio = background_subtraction_ratio
        self.enable_background_masking = enable_background_masking
        self.backSub = createBackgroundSubtractorMOG2(context=self.ctx) if self.background_subtraction else None
        self.wanted_labels = list(wanted_labels)
        self.nms_max_overlap = nms_max_overlap
        self.object_detector = make_detector(model, labels, wanted_labels, num_threads, self.ctx) if run_detector else None
        self.encoder = gdet.create_box_encoder(encoder_model, batch_size=encoder_batch_size, num_threads=num_threads,
                                               context=self.ctx)
        metric = nn_matching.NearestNeighborDistanceMetric("cosine", max_cosine_distance, None)   # deepdish.py:515-516
        self.tracker = Tracker(metric, max_iou_distance=max_iou_distance, max_age=max_age, context=self.ctx)
        w, h = self.input_size
        if line is None:                                                                          # :739-741
            line = np.array([[w / 2, 0], [w / 2, h]], dtype=int)
        self.counter = CountLine(np.asarray(line, dtype=float), self.wanted_labels)
        self.frame_count = 0
        self.timings = {}
        # frame records (deepdish.py:613-641): always present upstream, a pass-through unless CVAT annotations are given
        # (--input-cvat-dir; `annotations` = the parsed annotations.xml)
        det_labels = getattr(self.object_detector, 'labels', None) or {i: l for i, l in enumerate(self.wanted_labels)}
        self.framerec = FrameRecords(det_labels)
        self.xmltree = annotations
        if annotations is not None:
            load_cvat_annotations(self.framerec, annotations, det_labels)
        # result packaging (deepdish.py:545-561,1147-1185): MQTT payloads through the host's publish(topic, json) and
        # the JSON-lines log the counters can be restored from
        self.sink = None
        if log is not None or mqtt_publish is not None:
            self.sink = ResultSink(self.counter, acp_id=mqtt_acp_id, topic=mqtt_topic, publish=mqtt_publish,
                                   mqtt_verbosity=mqtt_verbosity, log=log, restore_from_log=restore_from_log, temp=cpu_temp)
            self.frame_count = self.sink.frame_count

    def step(self, frame_dev, injected=None, t_frame=None):
        """frame_dev: u8 [H, W, 3] BGR torch tensor in HBM.  injected = (boxes tlwh, labels, scores)
        replaces the detector's OUTPUT (the detector still runs) -- how bench.py feeds synthetic
        detections, since random weights detect nothing meaningful."""
        H, W = int(frame_dev.shape[0]), int(frame_dev.shape[1])
        t0 = time()
        if self.background_subtraction:                                                           # :920-924
            masked = torch.empty_like(frame_dev) if self.enable_background_masking else None
            self.backSub.apply_device(frame_dev[None], masked_out=masked[None] if masked is not None else None)
            if masked is not None:
                frame_dev = masked
        if self.object_detector is not None:
            boxes0, labels0, scores0 = self.object_detector.detect_frame_device(frame_dev, H, W)   # :883
        else:
            boxes0, labels0, scores0 = [], [], []
        if injected is not None:
            boxes0, labels0, scores0 = injected
        boxes, labels, scores = clean_boxes(boxes0, labels0, scores0, self.input_size[0], self.input_size[1],
                                            (self.backSub, self.background_subtraction_ratio) if self.background_subtraction else None)
        t1 = time()
        boxesA0, scoresA0 = np.array(boxes), np.array(scores)
        indices = preprocessing.non_max_suppression(boxesA0, self.nms_max_overlap, scoresA0, context=self.ctx)   # :995
        boxesA1 = boxesA0[indices] if len(indices) else np.zeros((0, 4), dtype=np.int64)
        scoresA1 = scoresA0[indices] if len(indices) else np.zeros(0)
        labels1 = [labels[i] for i in indices]
        framenum = self.frame_count + 1
        if self.xmltree is not None:                                                               # :1001
            boxes2, labels1, scores2 = self.framerec.process_boxes(framenum, boxesA1, labels1, scoresA1)
            boxesA1 = np.array(boxes2, dtype=np.float64).reshape(-1, 4)
            scoresA1 = np.array(scores2, dtype=np.float64)
            indices = list(range(len(labels1)))
        if len(indices):
            feats_dev, _valid = self.encoder.encode_device(frame_dev, H, W, boxesA1)                    # :1008
        else:
            feats_dev = None
        t2 = time()
        detections = [Detection(b, l, s, _NOFEAT) for b, l, s in zip(boxesA1, labels1, scoresA1)]        # :1014
        self.tracker.predict()                                                                          # :1028
        if self.xmltree is not None:
            detections = self.framerec.process_detections(framenum, detections)                         # :1017
        self.tracker.update_arrays(boxesA1.astype(np.float64), feats_dev, detections)                    # :1029
        if self.xmltree is not None:
            self._attach_features(detections, feats_dev)
            self.tracker.tracks = self.framerec.process_tracking(framenum, self.tracker)                # :1047
        t3 = time()
        events = self.counter.step(self.tracker)                                                         # :1035-1114
        t4 = time()
        self.frame_count += 1
        if self.sink is not None:
            self.sink.crossings(events, t0 if t_frame is None else t_frame, self.frame_count)            # :1116-1123
        self.timings = dict(objd=t1 - t0, feat=t2 - t1, trak=t3 - t2, proc=t4 - t3, e2e=t4 - t0)
        return events

    def _attach_features(self, detections, feats_dev):
        """With annotations a track may be extended from a detection later (framerecords.process_tracking): give the
        detections their feature rows (one small device-to-host copy per frame, only in this mode)."""
        if feats_dev is not None and len(detections):
            host = self.ctx.to_host(feats_dev)[:len(detections)]
            for d, f in zip(detections, host):
                d.feature = f

    def cvat_xml(self):
        """deepdish.py:795-805: the annotations.xml of --output-cvat-dir as an ElementTree."""
        meta = self.xmltree.getroot().find('./meta') if self.xmltree is not None else None
        return self.framerec.xml_output(meta=meta)

    def counts(self):
        return self.counter.vector()


_NOFEAT = np.zeros(0, dtype=np.float32)
