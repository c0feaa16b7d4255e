#!/usr/bin/env python3
"""Generate tests/golden/{ssd_tail,yolov5_tail}.npz by running the REFERENCE's own detector adaptors.

tools/ssd_mobilenet.py and tools/yolov5.py are the reference's glue between ``interpreter.invoke()`` and
the tracker: NaN scrub, confidence filter, box reorder/scale, per-class NMS with its own overlap formula
(ssd_mobilenet.py:59-98, :111-150, :198-213) and the YOLOv5 row decode (yolov5.py:120-146).  Those lines are
plain numpy and run here; only the interpreter itself (tflite_runtime, absent, and the weight blobs) cannot.
So this script drives the reference classes with CANNED interpreter outputs:

  * a stub ``tflite_runtime.interpreter`` module whose ``Interpreter`` holds no model and no arithmetic -- it
    reports an input shape and hands back whatever tensors the script stored in it (a data feeder);
  * ``Image.ANTIALIAS = Image.LANCZOS`` (the constant was removed in Pillow 10; same filter);
  * ``DEEPDISHHOME`` pointing at the reference tree so YOLOV5 finds detectors/yolov5/yolov5s.yaml.

Nothing in the reference tree is touched; the fixtures hold inputs and the reference's outputs only.
Runs only in the build container (needs /root/reference).
"""
import os
import sys
import types
import numpy as np
from PIL import Image

REF = os.environ.get('DEEPDISH_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tests', 'golden')
SSD_LABELS = os.path.join(ROOT, 'deepdish_amd', 'assets', 'coco_labels_ssd.txt')
YOLO_LABELS = os.path.join(REF, 'detectors', 'yolov5', 'coco_classes.txt')


class CannedInterpreter:
    """Stands where tflite_runtime's Interpreter would: no model, no maths, returns stored tensors."""
    canned = {}                 # tensor index -> ndarray
    input_hw = (300, 300)

    def __init__(self, model_path=None, num_threads=None, experimental_delegates=None):
        pass

    def allocate_tensors(self):
        pass

    def get_input_details(self):
        h, w = CannedInterpreter.input_hw
        return [dict(index=100, shape=np.array([1, h, w, 3]), quantization=(1.0, 0))]

    def get_output_details(self):
        return [dict(index=i, quantization=(1.0, 0)) for i in sorted(CannedInterpreter.canned)]

    def set_tensor(self, index, value):
        pass

    def invoke(self):
        pass

    def get_tensor(self, index):
        return np.array(CannedInterpreter.canned[index], copy=True)


def install_shims():
    pkg = types.ModuleType('tflite_runtime')
    mod = types.ModuleType('tflite_runtime.interpreter')
    mod.Interpreter = CannedInterpreter
    mod.load_delegate = lambda *a, **k: None
    pkg.interpreter = mod
    sys.modules['tflite_runtime'] = pkg
    sys.modules['tflite_runtime.interpreter'] = mod
    if not hasattr(Image, 'ANTIALIAS'):
        Image.ANTIALIAS = Image.LANCZOS
    os.environ['DEEPDISHHOME'] = REF
    sys.path.insert(0, REF)


def ssd_case(rng, kind):
    """Ten rows as the post-process op returns them: boxes (ymin, xmin, ymax, xmax) normalised, class ids, scores."""
    n = 10
    boxes = np.zeros((n, 4), np.float32)
    cls = np.zeros(n, np.float32)
    scores = (rng.uniform(0.3, 1.0, n) + 1e-4 * np.arange(n)).astype(np.float32)      # tie-free
    n_cls = int(rng.integers(1, 5))
    ids = rng.choice([0, 1, 2, 3, 16, 27, 43, 61, 89], n_cls, replace=False)
    k = 0
    while k < n:                                   # clusters of near-duplicates of one class: IoU straddles 0.5
        m = int(min(n - k, rng.integers(1, 5)))
        cy, cx = rng.uniform(0.2, 0.8, 2)
        h, w = rng.uniform(0.08, 0.4), rng.uniform(0.05, 0.3)
        c = float(rng.choice(ids))
        for _ in range(m):
            if kind == 'tiny':                     # a few pixels wide: the +1 on the intersection decides
                h, w = rng.uniform(0.004, 0.02), rng.uniform(0.003, 0.015)
            dy, dx = rng.normal(0, 0.25 * h), rng.normal(0, 0.25 * w)
            sh, sw = h * rng.uniform(0.8, 1.25), w * rng.uniform(0.8, 1.25)
            boxes[k] = [cy + dy - sh / 2, cx + dx - sw / 2, cy + dy + sh / 2, cx + dx + sw / 2]
            cls[k] = c if rng.random() < 0.85 else float(rng.choice(ids))
            k += 1
    if kind == 'nan':                              # the detector's "soft failure" encoding (ssd_mobilenet.py:111-116)
        r = int(rng.integers(0, n))
        boxes[r, int(rng.integers(0, 4))] = np.nan
        if rng.random() < 0.5:
            scores[int(rng.integers(0, n))] = np.nan
    if kind == 'lowconf':
        scores *= np.float32(0.6)
    return boxes, cls, scores


def make_ssd(n_cases=240, seed=20260401):
    from tools.ssd_mobilenet import SSD_MOBILENET
    CannedInterpreter.input_hw = (300, 300)
    CannedInterpreter.canned = {0: np.zeros((1, 10, 4), np.float32), 1: np.zeros((1, 10), np.float32),
                                2: np.zeros((1, 10), np.float32), 3: np.array([10.0], np.float32)}
    wanted = [l.strip() for l in open(SSD_LABELS)][1:]
    det_all = SSD_MOBILENET(wanted_labels=wanted, model_file='canned.tflite', label_file=SSD_LABELS)
    det_person = SSD_MOBILENET(wanted_labels=['person', 'car'], model_file='canned.tflite', label_file=SSD_LABELS)
    rng = np.random.default_rng(seed)
    rec = dict(boxes=[], cls=[], scores=[], size=[], kind=[], off=[0], pred_boxes=[], pred_cls=[], pred_scores=[],
               doff=[0], det_boxes=[], det_cls=[], det_scores=[], poff=[0], pdet_boxes=[], pdet_cls=[], pdet_scores=[])
    name_to_id = {v: k - 1 for k, v in det_all.labels.items() if k > 0}
    kinds = ['plain', 'tiny', 'nan', 'lowconf']
    sizes = [(640, 480), (1280, 720), (300, 300), (97, 61)]
    for i in range(n_cases):
        kind = kinds[i % len(kinds)] if i % 3 else 'plain'
        size = sizes[(i // 2) % len(sizes)]
        b, c, s = ssd_case(rng, kind)
        CannedInterpreter.canned = {0: b[None], 1: c[None], 2: s[None], 3: np.array([10.0], np.float32)}
        img = Image.new('RGBA', size)
        inp = det_all.ssdm.prepare_image(img)
        pb, pl, ps = det_all.ssdm.predict(inp, original_image_size=img.size)           # reference :100-150
        db, dl, dsc = det_all.detect_image(img)                                         # reference :198-213
        qb, ql, qs = det_person.detect_image(img)
        rec['boxes'].append(b); rec['cls'].append(c); rec['scores'].append(s); rec['size'].append(size)
        rec['kind'].append(kinds.index(kind))
        rec['pred_boxes'] += [np.asarray(x, np.float64) for x in pb]
        rec['pred_cls'] += [name_to_id[x] for x in pl]
        rec['pred_scores'] += [np.float32(x) for x in ps]
        rec['off'].append(len(rec['pred_scores']))
        rec['det_boxes'] += [np.asarray(x, np.float64) for x in db]
        rec['det_cls'] += [name_to_id[x] for x in dl]
        rec['det_scores'] += [np.float32(x) for x in dsc]
        rec['doff'].append(len(rec['det_scores']))
        rec['pdet_boxes'] += [np.asarray(x, np.float64) for x in qb]
        rec['pdet_cls'] += [name_to_id[x] for x in ql]
        rec['pdet_scores'] += [np.float32(x) for x in qs]
        rec['poff'].append(len(rec['pdet_scores']))
    # nms_boxes alone, larger K, several thresholds (reference :59-98)
    nb = dict(n_boxes=[], n_cls=[], n_scores=[], n_thr=[], n_off=[0], k_boxes=[], k_cls=[], k_scores=[], k_off=[0])
    for i in range(60):
        K = int(rng.integers(1, 40))
        ids = rng.choice(90, int(rng.integers(1, 4)), replace=False).astype(np.float32)
        ctr = rng.uniform(40, 600, (max(1, K // 4), 2))
        which = rng.integers(0, len(ctr), K)
        wh = rng.uniform(2, 120, (K, 2)) if i % 2 else rng.uniform(1, 6, (K, 2))
        xy = ctr[which] + rng.normal(0, 0.2, (K, 2)) * wh
        boxes = np.concatenate([xy - wh / 2, xy + wh / 2], axis=1)                      # f64 xyxy like predict() passes
        labels = rng.choice(ids, K).astype(np.float32)
        scores = (rng.uniform(0.5, 1, K) + 1e-5 * np.arange(K)).astype(np.float32)
        thr = float(rng.choice([0.3, 0.5, 0.7]))
        ob, ol, osc = det_all.ssdm.nms_boxes(boxes, labels, scores, thr)
        nb['n_boxes'].append(boxes); nb['n_cls'].append(labels); nb['n_scores'].append(scores); nb['n_thr'].append(thr)
        nb['n_off'].append(nb['n_off'][-1] + K)
        kb = np.concatenate(ob) if ob else np.zeros((0, 4))
        nb['k_boxes'].append(kb); nb['k_cls'].append(np.concatenate(ol)); nb['k_scores'].append(np.concatenate(osc))
        nb['k_off'].append(nb['k_off'][-1] + len(kb))
    out = dict(boxes=np.stack(rec['boxes']), cls=np.stack(rec['cls']), scores=np.stack(rec['scores']),
               size=np.array(rec['size'], np.int64), kind=np.array(rec['kind'], np.int64),
               off=np.array(rec['off'], np.int64), pred_boxes=np.array(rec['pred_boxes'], np.float64).reshape(-1, 4),
               pred_cls=np.array(rec['pred_cls'], np.int64), pred_scores=np.array(rec['pred_scores'], np.float32),
               doff=np.array(rec['doff'], np.int64), det_boxes=np.array(rec['det_boxes'], np.float64).reshape(-1, 4),
               det_cls=np.array(rec['det_cls'], np.int64), det_scores=np.array(rec['det_scores'], np.float32),
               poff=np.array(rec['poff'], np.int64), pdet_boxes=np.array(rec['pdet_boxes'], np.float64).reshape(-1, 4),
               pdet_cls=np.array(rec['pdet_cls'], np.int64), pdet_scores=np.array(rec['pdet_scores'], np.float32),
               n_boxes=np.concatenate(nb['n_boxes']), n_cls=np.concatenate(nb['n_cls']), n_scores=np.concatenate(nb['n_scores']),
               n_thr=np.array(nb['n_thr']), n_off=np.array(nb['n_off'], np.int64), k_boxes=np.concatenate(nb['k_boxes']),
               k_cls=np.concatenate(nb['k_cls']), k_scores=np.concatenate(nb['k_scores']), k_off=np.array(nb['k_off'], np.int64),
               wanted_person=np.array(['person', 'car']))
    np.savez_compressed(os.path.join(OUT, 'ssd_tail.npz'), **out)
    kept = out['off'][-1]
    print('ssd_tail.npz: %d cases, %d rows after predict(), %d after detect_image(all), %d after detect_image(person, car); '
          'nms_boxes: %d cases, %d -> %d boxes' % (n_cases, kept, out['doff'][-1], out['poff'][-1], 60, out['n_off'][-1], out['k_off'][-1]))


def yolo_rows(rng, n, n_cls=80):
    """xywh U(0,1), obj / class scores Beta(0.5, 4) so that a few percent of the rows pass 0.25;
    values rounded to f16 so the fixture stores them in half the bytes without changing what either side reads."""
    x = np.empty((n, 5 + n_cls), np.float32)
    x[:, :4] = rng.uniform(0, 1, (n, 4))
    x[:, 2:4] *= 0.4
    x[:, 4] = rng.beta(2.0, 2.0, n)
    x[:, 5:] = rng.beta(0.5, 4.0, (n, n_cls))
    hot = rng.random(n) < 0.3                         # rows that look like a detection: one class near 1
    x[hot, 5 + rng.integers(0, 8, hot.sum())] = rng.uniform(0.4, 1.0, hot.sum())
    return x.astype(np.float16).astype(np.float32)


def make_yolo(seed=20260402):
    from tools.yolov5 import YOLOV5
    CannedInterpreter.input_hw = (640, 640)
    CannedInterpreter.canned = {0: np.zeros((1, 8, 85), np.float32)}
    rng = np.random.default_rng(seed)
    cases = [(700, (640, 480), ['person', 'car', 'bicycle'], 0.25), (300, (1280, 720), ['person'], 0.25),
             (500, (640, 640), ['person', 'car', 'bus', 'truck', 'motorbike'], 0.4), (64, (97, 61), ['car'], 0.1)]
    rec = dict(raw=[], roff=[0], size=[], thr=[], wanted=[], boxes=[], labels=[], scores=[], off=[0])
    for n, size, wanted, thr in cases:
        det = YOLOV5(wanted_labels=wanted, model_file='canned-fp16.tflite', label_file=YOLO_LABELS, score_threshold=thr)
        raw = yolo_rows(rng, n)
        CannedInterpreter.canned = {0: raw[None]}
        b, l, s = det.detect_image(Image.new('RGBA', size))                       # reference :97-146
        name_to_id = {v: k for k, v in det.labels.items()}
        rec['raw'].append(raw.astype(np.float16)); rec['roff'].append(rec['roff'][-1] + n)
        rec['size'].append(size); rec['thr'].append(thr); rec['wanted'].append(','.join(wanted))
        rec['boxes'] += [np.asarray(x, np.float32) for x in b]
        rec['labels'] += [name_to_id[x] for x in l]
        rec['scores'] += [np.float32(x) for x in s]
        rec['off'].append(len(rec['scores']))
    np.savez_compressed(os.path.join(OUT, 'yolov5_tail.npz'), raw_f16=np.concatenate(rec['raw']),
                        roff=np.array(rec['roff'], np.int64), size=np.array(rec['size'], np.int64), thr=np.array(rec['thr']),
                        wanted=np.array(rec['wanted']), boxes=np.array(rec['boxes'], np.float32).reshape(-1, 4),
                        labels=np.array(rec['labels'], np.int64), scores=np.array(rec['scores'], np.float32),
                        off=np.array(rec['off'], np.int64))
    print('yolov5_tail.npz: %d cases, rows %s -> detections %s' % (len(cases), np.diff(rec['roff']).tolist(), np.diff(rec['off']).tolist()))


if __name__ == '__main__':
    install_shims()
    os.makedirs(OUT, exist_ok=True)
    make_ssd()
    make_yolo()
