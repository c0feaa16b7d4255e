"""deepdish_amd/framerecords.py against tests/golden/framerecords.json, which scripts/make_golden_framerecords.py made by
running the reference's own deepdish/framerecords.py + deep_sort tracker.  CPU: the box merge (tracker-independent) and
the pass-through case.  GPU: the whole per-frame flow on the device tracker, track table after every frame and the CVAT
XML byte for byte."""
import io
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), 'golden')


def load():
    fx = json.load(open(os.path.join(G, 'framerecords.json')))
    feats = np.load(os.path.join(G, 'framerecords_features.npz'))['features']
    return fx, feats


def build(fx, scenario):
    from deepdish_amd.framerecords import FrameRecords
    fr = FrameRecords({int(k): v for k, v in fx['labels'].items()})
    for name, det_id, color in fx['annotation_labels']:
        fr.add_annotation_label_info(name, det_id, color)
    for a in scenario['annotations']:
        fr.add_annotated_track(a['frame'], a['track'], a['label'], np.array(a['pts'], dtype=float), a['outside'], a['occluded'],
                               a['keyframe'], a['z_order'])
    return fr


@pytest.mark.parametrize('which', [0, 1])
def test_box_merge_matches_the_reference(which):
    fx, _ = load()
    sc = fx['scenarios'][which]
    fr = build(fx, sc)
    for f, fo in enumerate(sc['frames']):
        boxes = np.array(fo['boxes_in'], dtype=np.int64).reshape(-1, 4)
        b, l, s = fr.process_boxes(f, boxes, fo['labels_in'], np.array(fo['scores_in']))
        assert [[float(v) for v in x] for x in b] == fo['boxes_out'], f
        assert l == fo['labels_out'] and [float(v) for v in s] == fo['scores_out'], f
        assert [type(r).__name__ for r in fr.frames[f]] == fo['record_kinds'], f
    if sc['xml_error']:
        with pytest.raises(KeyError):                     # an annotation label without a detector name cannot be exported
            fr.xml_output()


def test_without_annotations_everything_passes_through():
    from deepdish_amd.framerecords import FrameRecords
    fr = FrameRecords({0: 'person', 1: 'car'})
    boxes = np.array([[10, 20, 30, 40], [5, 6, 7, 8]])
    b, l, s = fr.process_boxes(3, boxes, ['person', 'car'], np.array([0.9, 0.4]))
    np.testing.assert_array_equal(np.array(b), boxes)
    assert l == ['person', 'car'] and s == [0.9, 0.4]

    class D:
        pass
    dets = [D(), D()]
    assert fr.process_detections(3, dets) is dets and dets[1].record.order == 1

    class T:
        def __init__(self, i, d):
            self.track_id, self.detections, self.time_since_update = i, d, 0

    class K:
        tracks = [T(1, [dets[0]]), T(2, [dets[1]])]
        kf = None
    assert fr.process_tracking(3, K) == K.tracks and dets[0].record.track is K.tracks[0]
    xml = io.BytesIO()
    fr.xml_output().write(xml, xml_declaration=True, encoding='utf-8', short_empty_elements=False)
    assert b'<track' not in xml.getvalue()                # one-frame tracks are below minimum_track_frames


@pytest.mark.gpu
def test_whole_flow_on_the_device_tracker_matches_the_reference():
    from deepdish_amd.deep_sort import nn_matching
    from deepdish_amd.deep_sort.tracker import Tracker
    from deepdish_amd.deep_sort.detection import Detection
    fx, feats = load()
    sc = fx['scenarios'][0]
    fr = build(fx, sc)
    tk = fx['tracker']
    tracker = Tracker(nn_matching.NearestNeighborDistanceMetric('cosine', tk['max_cosine_distance'], None),
                      max_iou_distance=tk['max_iou_distance'], max_age=tk['max_age'], n_init=tk['n_init'])
    seen = dict(dropped=[], extended=[])
    for f, fo in enumerate(sc['frames']):
        boxes = np.array(fo['boxes_in'], dtype=np.int64).reshape(-1, 4)
        b2, l2, s2 = fr.process_boxes(f, boxes, fo['labels_in'], np.array(fo['scores_in']))               # deepdish.py:1001
        r0, n = fo['feature_rows']
        dets = [Detection(b, l, s, ft) for b, l, s, ft in zip(b2, l2, s2, feats[r0:r0 + n])]               # :1014
        dets = fr.process_detections(f, dets)                                                             # :1017
        tracker.predict()
        tracker.update(dets)
        before = [(t.track_id, t.time_since_update) for t in tracker.tracks]
        tracker.tracks = fr.process_tracking(f, tracker)                                                  # :1047
        after = {t.track_id: t.time_since_update for t in tracker.tracks}
        seen['dropped'] += [[f, i] for i, _ in before if i not in after]
        seen['extended'] += [[f, i] for i, tsu in before if tsu > 0 and after.get(i) == 0]
        got = [[t.track_id, t.state, t.time_since_update, t.hits, t.age] for t in tracker.tracks]
        assert got == fo['tracks'], f
        if got:
            np.testing.assert_allclose(np.array([t.mean for t in tracker.tracks]), np.array(fo['means']), rtol=1e-8, atol=1e-8, err_msg=str(f))
    assert seen == sc['events'] and seen['dropped'] and seen['extended']
    out = io.BytesIO()
    fr.xml_output(meta=None).write(out, xml_declaration=True, encoding='utf-8', short_empty_elements=False)
    assert out.getvalue().decode('utf-8') == sc['xml']


@pytest.mark.gpu
def test_host_path_with_cvat_annotations():
    """HotPath(annotations=...) = --input-cvat-dir: an annotated object the detector never reports is still tracked (its
    annotation boxes go through the encoder and the tracker), and the exported CVAT file holds the manual track plus the
    tracks the tracker found by itself."""
    import torch
    import xml.etree.ElementTree as ET
    from deepdish_amd.pipeline import HotPath
    from deepdish_amd.synth import Scene
    sc = Scene(seed=41, n_obj=4, n_frames=16, churn=False, p_miss=0.0, n_dup=0.0)
    root = ET.Element('annotations')
    labels = ET.SubElement(ET.SubElement(ET.SubElement(root, 'meta'), 'task'), 'labels')
    lab = ET.SubElement(labels, 'label'); ET.SubElement(lab, 'name').text = 'person'; ET.SubElement(lab, 'color').text = '#33ddff'
    tr = ET.SubElement(root, 'track', attrib={'id': '0', 'label': 'person', 'source': 'manual'})
    hidden = 0                                                        # object 0 is annotated; the detector never sees it
    for f in range(1, 17):                                            # HotPath numbers frames from 1
        x, y = sc.xy[f - 1, hidden]
        ET.SubElement(tr, 'box', attrib={'frame': str(f), 'outside': '0', 'occluded': '0', 'keyframe': '1', 'z_order': '0',
                                         'xtl': '%.2f' % x, 'ytl': '%.2f' % y, 'xbr': '%.2f' % (x + sc.w[hidden]), 'ybr': '%.2f' % (y + sc.h[hidden])})
    hp = HotPath(run_detector=False, wanted_labels=('person',), annotations=ET.ElementTree(root))
    hp.object_detector = None
    for f in range(16):
        boxes, scores, who, _ = sc.detections(f)
        keep = who != hidden
        inj = ([tuple(int(v) for v in b) for b in boxes[keep]], ['person'] * int(keep.sum()), [float(s) for s in scores[keep]])
        hp.step(torch.from_numpy(sc.frame(f)).cuda(), injected=inj)
    assert len(hp.tracker.tracks) == 4 and all(t.is_confirmed() for t in hp.tracker.tracks)
    fed = [d for t in hp.tracker.tracks for d in t.detections if type(getattr(d, 'record', None)).__name__ == 'AnnotationRecord']
    assert len(fed) == 16 and {d.record.track.track_id for d in fed} == {fed[0].record.track.track_id}
    out = hp.cvat_xml().getroot()
    assert out.find('./meta/task/labels/label/name').text == 'person'
    tracks = out.findall('./track')
    assert [t.get('source') for t in tracks] == ['manual', 'automatic', 'automatic', 'automatic']
    assert [t.get('id') for t in tracks] == ['0', '1', '2', '3'] and len(tracks[0].findall('box')) == 16
    assert tracks[1].findall('box')[-1].get('outside') == '1'
