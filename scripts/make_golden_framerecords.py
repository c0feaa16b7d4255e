#!/usr/bin/env python3
"""Generate tests/golden/framerecords.json by running the REFERENCE's deepdish/framerecords.py and deep_sort
tracker (build container only; same harness shims as scripts/make_golden.py).  The scenario: three annotated CVAT
tracks (one with a label the detector does not know), detections that sometimes overlap them, sometimes miss them
and sometimes belong to an object nobody annotated; per frame the driver restates deepdish.py:1001,1008-1017,
1028-1029,1047 (process_boxes -> Detection -> process_detections -> predict/update -> process_tracking).  The
fixture holds inputs and the reference's outputs only."""
import io
import json
import os
import sys
import types
import numpy as np

REF = os.environ.get('DEEPDISH_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
np.float = float
np.int = int
sys.modules.setdefault('cv2', types.ModuleType('cv2'))
sys.path.insert(0, REF)

from deepdish.framerecords import FrameRecords  # noqa: E402
from deep_sort import nn_matching  # noqa: E402
from deep_sort.tracker import Tracker  # noqa: E402
from deep_sort.detection import Detection  # noqa: E402

LABELS = {0: 'person', 1: 'bicycle', 2: 'car'}
FEATS = []            # features of every detection fed to the tracker, in order; saved as f32 rows next to the JSON
N_FRAMES = 18


def POS(f):
    """Object boxes (x, y, w, h) in frame f.  B jumps 160 px at frame 9 (the annotation follows it): the tracker's gate
    rejects the jump, a second track starts on the annotation-fed detection, and process_tracking has to extend the
    old track from the annotation and drop the duplicate."""
    bx = 400 - 5 * f - (160 if f >= 9 else 0)
    return [(40 + 6 * f, 60, 30, 80), (bx, 200, 90, 50), (300, 40 + 4 * f, 28, 70), (500, 300, 40, 40)]


def scenario():
    rng = np.random.default_rng(77)
    ident = rng.standard_normal((4, 128)).astype(np.float32)
    # objects: A person (annotated track 1), B car (annotated track 2, detector often misses it),
    #          C person nobody annotated, D 'scooter' annotation (label unknown to the detector), never detected
    def pos(o, f):
        return POS(f)['ABCD'.index(o)]
    ann = []
    for f in range(0, 15):
        x, y, w, h = pos('A', f)
        ann.append(dict(frame=f, track=1, label='person', pts=[x + 0.5, y - 0.25, x + w + 0.5, y + h - 0.25], outside=False,
                        occluded=f in (6, 7), keyframe=f % 5 == 0, z_order=0))
    for f in range(3, 13):
        x, y, w, h = pos('B', f)
        ann.append(dict(frame=f, track=2, label='car', pts=[float(x), float(y), float(x + w), float(y + h)], outside=f == 12,
                        occluded=False, keyframe=True, z_order=1))
    for f in range(5, 9):
        x, y, w, h = pos('D', f)
        ann.append(dict(frame=f, track=5, label='scooter', pts=[float(x), float(y), float(x + w), float(y + h)], outside=False,
                        occluded=False, keyframe=True, z_order=0))
    frames = []
    for f in range(N_FRAMES):
        boxes, labels, scores, who = [], [], [], []
        def add(o, lbl, jit, sc):
            x, y, w, h = pos(o, f)
            boxes.append([x + jit[0], y + jit[1], w + jit[2], h + jit[3]]); labels.append(lbl); scores.append(sc); who.append('ABCD'.index(o))
        if f not in (4, 9, 10):
            add('A', 'person', (1, 0, -1, 1) if f % 2 else (0, 0, 0, 0), 0.9 - 0.01 * f)
        if f in (2, 3, 6, 14, 15):
            add('B', 'car', (0, 1, 0, -1), 0.7)
        if f == 7:
            add('B', 'bicycle', (0, 0, 0, 0), 0.55)          # right place, wrong label: no match with the annotation
        if f >= 1 and f != 11:
            add('C', 'person', (0, 0, 1, 0), 0.8)
        frames.append(dict(boxes=boxes, labels=labels, scores=scores, who=who))
    return ident, ann, frames


def feature_for(ident, box, f, frames_pos):
    """Identity of an output box = the nearest object centre in this frame; deterministic small perturbation."""
    cx, cy = box[0] + box[2] / 2.0, box[1] + box[3] / 2.0
    k = int(np.argmin([(cx - (p[0] + p[2] / 2.0)) ** 2 + (cy - (p[1] + p[3] / 2.0)) ** 2 for p in frames_pos]))
    v = ident[k] + 0.03 * np.sin(np.arange(128) * (f + 1) * 0.37 + k).astype(np.float32)
    return (v / np.linalg.norm(v)).astype(np.float32), k


def run(with_unmapped):
    ident, ann, frames = scenario()
    if not with_unmapped:
        ann = [a for a in ann if a['label'] != 'scooter']
    fr = FrameRecords(LABELS)
    fr.add_annotation_label_info('person', 0, '#ff0000')
    fr.add_annotation_label_info('car', 2, '#00ff00')
    fr.add_annotation_label_info('scooter', None, '#0000ff')
    for a in ann:
        fr.add_annotated_track(a['frame'], a['track'], a['label'], np.array(a['pts'], dtype=float), a['outside'], a['occluded'],
                               a['keyframe'], a['z_order'])
    tracker = Tracker(nn_matching.NearestNeighborDistanceMetric('cosine', 0.2, None), max_iou_distance=0.7, max_age=5, n_init=3)
    out_frames = []
    events = dict(dropped=[], extended=[])
    for f, fin in enumerate(frames):
        pos_now = POS(f)
        boxes = np.array(fin['boxes'], dtype=np.int64).reshape(-1, 4)
        b2, l2, s2 = fr.process_boxes(f, boxes, fin['labels'], np.array(fin['scores']))
        feats = [feature_for(ident, b, f, pos_now)[0] for b in b2]
        dets = [Detection(b, l, s, ft) for b, l, s, ft in zip(b2, l2, s2, feats)]
        feats_now = list(feats)
        dets = fr.process_detections(f, dets)
        tracker.predict()
        tracker.update(dets)
        before = [(t.track_id, t.time_since_update) for t in tracker.tracks]
        tracker.tracks = fr.process_tracking(f, tracker)
        after = {t.track_id: t.time_since_update for t in tracker.tracks}
        events['dropped'] += [[f, i] for i, _ in before if i not in after]
        events['extended'] += [[f, i] for i, tsu in before if tsu > 0 and after.get(i) == 0]
        out_frames.append(dict(
            boxes_in=fin['boxes'], labels_in=fin['labels'], scores_in=fin['scores'],
            boxes_out=[[float(v) for v in b] for b in b2], labels_out=list(l2), scores_out=[float(s) for s in s2],
            feature_rows=[len(FEATS), len(feats)],
            record_kinds=[type(r).__name__ for r in fr.frames[f]],
            tracks=[[int(t.track_id), int(t.state), int(t.time_since_update), int(t.hits), int(t.age)] for t in tracker.tracks],
            means=[[float(v) for v in t.mean] for t in tracker.tracks]))
        FEATS.extend(feats_now)
    xml, xml_error = None, None
    try:
        buf = io.BytesIO()
        fr.xml_output(meta=None).write(buf, xml_declaration=True, encoding='utf-8', short_empty_elements=False)
        xml = buf.getvalue().decode('utf-8')
    except Exception as e:                      # an annotation label the detector does not know has no name to export
        xml_error = type(e).__name__
    kinds = sorted({k for fo in out_frames for k in fo['record_kinds']})
    print('unmapped label' if with_unmapped else 'mapped labels', '; record kinds', kinds, '; final tracks', out_frames[-1]['tracks'],
          '; xml', xml_error or len(xml), '; events', events)
    return dict(annotations=ann, frames=out_frames, xml=xml, xml_error=xml_error, events=events)


def main():
    fixture = dict(labels={str(k): v for k, v in LABELS.items()},
                   annotation_labels=[['person', 0, '#ff0000'], ['car', 2, '#00ff00'], ['scooter', None, '#0000ff']],
                   tracker=dict(max_cosine_distance=0.2, max_iou_distance=0.7, max_age=5, n_init=3),
                   scenarios=[run(False), run(True)])
    path = os.path.join(ROOT, 'tests', 'golden', 'framerecords.json')
    with open(path, 'w') as fh:
        json.dump(fixture, fh)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'framerecords_features.npz'), features=np.array(FEATS, dtype=np.float32))
    print('wrote', path, os.path.getsize(path), 'bytes +', len(FEATS), 'feature rows')
    print(fixture['scenarios'][0]['xml'][:900])


if __name__ == '__main__':
    main()
