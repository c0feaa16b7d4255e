"""Multi-stream hot path on one GPU (csrc/pipeline.hip): S independent DeepDish streams advanced one
frame each per step, device work batched across streams.  Mirrors `pipeline.HotPath` (one stream,
reference-shaped Python objects) but keeps the per-frame orchestration in C++."""
import ctypes
import os
import numpy as np

from ._lib import lib, check, P
from .runtime import default_context, ptr
from . import nets, netsq
from .engine import Net
from .pipeline import DEFAULT_LABELS, DEFAULT_YOLO_LABELS
from .tools.weights_io import load_named_weights, load_ssd_model, load_mars_weights, load_yolov5_weights, ssd_post_options


class _TrackerView:
    """Read-only view of one stream's C++ tracker (for parity tests)."""

    def __init__(self, handle):
        self._h = handle

    def table(self):
        n = ctypes.c_int()
        check(lib().dd_tracker_count(self._h, 0, ctypes.byref(n)))
        ints = np.zeros((n.value, 6), dtype=np.int64)
        means = np.zeros((n.value, 8), dtype=np.float64)
        if n.value:
            check(lib().dd_tracker_read(self._h, 0, ptr(ints), ptr(means), None))
        return ints, means

    @property
    def next_id(self):
        v = ctypes.c_int64()
        check(lib().dd_tracker_next_id(self._h, ctypes.byref(v)))
        return v.value


class MultiStreamPipeline:
    def __init__(self, n_streams, model='synthetic-ssd_mobilenet_v1', encoder_model='synthetic-mars-64x32x3',
                 labels=None, wanted_labels=('person',), input_size=(640, 480), line=None, max_cosine_distance=0.2,
                 nms_max_overlap=0.6, max_iou_distance=0.7, max_age=60, n_init=3, context=None, run_detector=True,
                 encoder_max_batch=None, track_capacity=512, gallery_capacity=256, background_subtraction_ratio=None,
                 background_masking=False, graph=None):
        self.ctx = context or default_context()
        self.S = int(n_streams)
        self.W, self.H = input_size
        self.wanted = list(wanted_labels)
        self.model_name = str(model)
        # the reference picks the detector plugin by a substring of --model (deepdish.py:482-502)
        self.kind = ('yolov5' if 'yolov5' in model else None if ('yolo' in model or 'saved_model' in model) else
                     'ssd_mobilenet' if 'mobilenet' in model else 'tflite' if 'tflite' in model else None)
        if self.kind is None:
            raise ValueError('the multi-stream pipeline batches the SSD-MobileNet, YOLOv5 and generic TFLite detectors (got %s)' % model)
        with open(labels or (DEFAULT_YOLO_LABELS if self.kind == 'yolov5' else DEFAULT_LABELS)) as f:
            self.label_lines = [l.strip() for l in f.readlines()]
        meta = None
        if self.kind == 'tflite' and str(model).endswith('.tflite') and os.path.exists(str(model)):
            # the generic adaptor takes its label list and input normalisation from the model file's metadata (tools/tflite_object_detector.py:117-137
            # upstream); a file without metadata keeps the label file and the 127.5 / 127.5 defaults (see tools/tflite.py here)
            from .tools import tflite_reader
            try:
                meta = tflite_reader.read_metadata(str(model))
                self.label_lines = ['???'] + list(meta['labels'])      # (line 0 = the background entry the adaptor's label map skips)
            except tflite_reader.UnsupportedModel:
                if labels is None:
                    raise
        self.det = None
        anchors, n_anchors, n_classes = None, 0, 0
        if run_detector and self.kind == 'yolov5':
            wd = load_yolov5_weights(model)                        # a yolov5s .tflite file on disk goes through tools/tflite_reader.load_yolov5s
            prog = nets.compile_yolov5s(wd, int(wd.get('__in_size__', 640)))
            self.det = Net(prog, max_batch=self.S, context=self.ctx)
            n_anchors, n_classes = prog.meta['rows'], prog.meta['n_classes']
        elif run_detector:
            kind, wd = load_ssd_model(model)                        # ('uint8', QModel): the reference's own arithmetic (csrc/netsq.hip)
            prog = (netsq.compile_ssd_mobilenet_quant(wd) if kind == 'uint8' else
                    nets.compile_ssd_mobilenet(wd, mean=meta['mean'], std=meta['std']) if meta else nets.compile_ssd_mobilenet(wd))
            self.det_dtype = 'u8' if kind == 'uint8' else 'f16'
            self.det = Net(prog, max_batch=self.S, context=self.ctx)
            anchors = np.ascontiguousarray(prog.meta['anchors'], dtype=np.float32)
            n_anchors, n_classes = len(anchors), prog.meta['n_classes']
            ssd_post = ssd_post_options(wd)                         # the post-process op's options as the model file states them
        wd = load_mars_weights(encoder_model)                       # a mars .tflite file on disk goes through tools/tflite_reader.load_mars
        self.enc_weights = wd
        if tuple(wd.get('__in_hw__', (64, 32))) != (64, 32):
            raise ValueError('%s takes %s crops: the batched pipeline is built for the 64 x 32 encoder (mars-64x32x3, deepdish.py:505-510)' % (encoder_model, wd['__in_hw__']))
        # the pipeline never reads an intermediate encoder tensor: its activation buffers share memory by lifetime (6.2 -> ~2 GB at 12 288 crops)
        self.enc = Net(nets.compile_mars(wd), max_batch=encoder_max_batch or max(64, 32 * self.S), context=self.ctx,
                       shared=os.environ.get('DD_NET_SHARED', '1') != '0')
        # latency mode: with a handful of streams a forward is a train of 20-75 kernels of a few microseconds each;
        # replaying it as one hipGraph takes the per-launch host cost out of the frame latency (DD_GRAPH=0/1 overrides)
        if graph is None:
            graph = self.S <= 4 if os.environ.get('DD_GRAPH') is None else os.environ['DD_GRAPH'] == '1'
        self.graph = bool(graph)
        if self.graph:
            self.enc.use_graph(True)
            if self.det is not None:
                self.det.use_graph(True)
        if line is None:
            line = np.array([[self.W / 2, 0], [self.W / 2, self.H]], dtype=int)        # deepdish.py:739-741
        self.line = np.ascontiguousarray(np.asarray(line, dtype=np.float64).reshape(4))
        h = P()
        check(lib().dd_pipeline_create(self.ctx.handle, self.S, self.H, self.W, self.det._h if self.det else None,
                                       ptr(anchors), n_anchors, n_classes, self.enc._h,
                                       '\n'.join(self.label_lines).encode(), '\n'.join(self.wanted).encode(),
                                       float(max_cosine_distance), float(nms_max_overlap), float(max_iou_distance),
                                       int(max_age), int(n_init), ptr(self.line), int(track_capacity),
                                       int(gallery_capacity), ctypes.byref(h)), 'dd_pipeline_create')
        self._h = h
        if self.det is not None and self.kind != 'yolov5':
            check(lib().dd_pipeline_ssd_options(self._h, int(ssd_post['max_detections']), float(ssd_post['nms_score_threshold']),
                                                float(ssd_post['nms_iou_threshold'])), 'dd_pipeline_ssd_options')
        if self.kind == 'tflite' and self.det is not None:          # tools/tflite.py's adaptor instead of tools/ssd_mobilenet.py's
            check(lib().dd_pipeline_detector_adaptor(self._h, 2), 'dd_pipeline_detector_adaptor')
        off = 0 if self.kind == 'yolov5' else 1                     # yolov5.py:134 labels[idx]; ssd_mobilenet.py:142-147 labels[idx + 1]
        self._class_id = {name: i - off for i, name in enumerate(self.label_lines) if i >= off}
        if background_subtraction_ratio is not None:
            self.background_subtraction(background_subtraction_ratio, background_masking)

    def background_subtraction(self, ratio, masking=False):
        """deepdish.py:512,889,957: ratio = --background-subtraction-ratio (reference default 0.25); None or a negative
        value = --disable-background-subtraction (how a pipeline starts, and how the reference's benchmarks run)."""
        check(lib().dd_pipeline_background_subtraction(self._h, -1.0 if ratio is None else float(ratio), int(bool(masking))),
              'dd_pipeline_background_subtraction')

    def motion_mask(self, read=True):
        """-> (fgMask of the last step as ndarray u8 [S, H, W], boxes rejected by the motion test so far)."""
        n = ctypes.c_longlong()
        out = np.zeros((self.S, self.H, self.W), dtype=np.uint8) if read else None
        check(lib().dd_pipeline_motion_mask(self._h, ptr(out), 0, ctypes.byref(n)), 'dd_pipeline_motion_mask')
        return out, n.value

    def __del__(self):
        try:
            if self._h:
                lib().dd_pipeline_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def pack_injected(self, per_stream):
        """per_stream: list of (boxes tlwh, labels, scores) per stream -> arrays for step()."""
        off = np.zeros(self.S + 1, dtype=np.int32)
        b, sc, cl = [], [], []
        for z, (boxes, labels, scores) in enumerate(per_stream):
            b += [tuple(float(v) for v in bb) for bb in boxes]
            sc += [float(v) for v in scores]
            cl += [self._class_id[l] for l in labels]
            off[z + 1] = len(sc)
        return (np.ascontiguousarray(np.array(b, dtype=np.float64).reshape(-1, 4)), np.array(sc, dtype=np.float64),
                np.array(cl, dtype=np.int32), off)

    def step(self, frames_dev, injected=None, frames_next=None):
        """frames_dev: u8 [S, H, W, 3] BGR torch tensor in HBM; injected: pack_injected(...) or None.
        frames_next: the frames of the following step (same shape): their detector run is queued on the
        detector stream and overlaps this step's NMS / encoder / tracker; the next step() must receive them."""
        assert tuple(frames_dev.shape) == (self.S, self.H, self.W, 3)
        assert frames_next is None or tuple(frames_next.shape) == (self.S, self.H, self.W, 3)
        b, sc, cl, off = injected if injected is not None else (None, None, None, None)
        check(lib().dd_pipeline_step2(self._h, ptr(frames_dev), ptr(frames_next), ptr(b), ptr(sc), ptr(cl), ptr(off)),
              'dd_pipeline_step')

    def detector_stream(self):
        """The stream the look-ahead detector run is queued on (None without a detector): the consumer to name when acquiring the NEXT
        step's frames from an ingest ring (`ring.frames(slot, stream=pipe.detector_stream())`), so that their upload is waited for by
        their own detector run and not by this step's kernels."""
        h = P()
        check(lib().dd_pipeline_detector_stream(self._h, ctypes.byref(h)), 'dd_pipeline_detector_stream')
        return h if h.value else None

    def counts(self):
        out = np.zeros((self.S, len(self.wanted), 4), dtype=np.int64)
        check(lib().dd_pipeline_counts(self._h, ptr(out)), 'dd_pipeline_counts')
        return out

    def tracker(self, stream):
        h = P()
        check(lib().dd_pipeline_tracker(self._h, stream, ctypes.byref(h)), 'dd_pipeline_tracker')
        return _TrackerView(h)

    def stage_ms(self):
        """Per step: the stages' GPU milliseconds under the reference's timer names (deepdish.py:975-981 objd, :1018-1021 feat, :1031-1032
        trak; nms = the deep_sort NMS of :995), measured with HIP events on the streams the kernels run on, `host` = the step's wall time
        outside its waits for the GPU, `wall` = the step.  objd runs on the detector stream, one frame ahead beside the other stages, so
        the stages do not add up to `wall`.  `host_wall` = the old host-side stopwatch per program section (NOT per-stage GPU time: its
        `trak` contains the wait for the encoder kernels)."""
        g = np.zeros(6, dtype=np.float64)
        n = ctypes.c_longlong()
        check(lib().dd_pipeline_stage_gpu_ms(self._h, ptr(g), ctypes.byref(n)), 'dd_pipeline_stage_gpu_ms')
        t = np.zeros(4, dtype=np.float64)
        check(lib().dd_pipeline_stage_seconds(self._h, ptr(t), ctypes.byref(n)), 'dd_pipeline_stage_seconds')
        k = max(1, n.value)
        return dict(objd=g[0] / k, nms=g[1] / k, feat=g[2] / k, trak=g[3] / k, host=g[4] / k, wall=g[5] / k, steps=n.value,
                    host_wall=dict(objd=1e3 * t[0] / k, nms=1e3 * t[1] / k, feat=1e3 * t[2] / k, trak=1e3 * t[3] / k))

    def detections(self, stream):
        """The detector adaptor's output for one stream in the last step: what the reference's detect_image(...) returns
        (tools/ssd_mobilenet.py:198-213) -- (boxes tlwh f64 [n, 4], label names, scores f64 [n])."""
        n = ctypes.c_int()
        check(lib().dd_pipeline_detections(self._h, int(stream), None, None, None, 0, ctypes.byref(n)), 'dd_pipeline_detections')
        b, sc, cl = np.zeros((n.value, 4), np.float64), np.zeros(n.value, np.float64), np.zeros(n.value, np.int32)
        if n.value:
            check(lib().dd_pipeline_detections(self._h, int(stream), ptr(b), ptr(sc), ptr(cl), n.value, ctypes.byref(n)), 'dd_pipeline_detections')
        off = 0 if self.kind == 'yolov5' else 1
        return b, [self.label_lines[int(c) + off] for c in cl], sc
