"""CVAT annotation merge and export around the hot path (SURVEY.md section 8 f, n4).

Behaviour of deepdish/framerecords.py:12-257 upstream, which the reference's Pipeline calls on every frame
(deepdish.py:1001 process_boxes after NMS, :1017 process_detections, :1047 process_tracking -> tracker.tracks) and at
shutdown (:795-805 xml_output).  Without annotations all three are pass-throughs.  With annotations (--input-cvat-dir):
detector boxes that overlap an annotated box are replaced by it, annotated boxes the detector missed are fed to the
encoder and tracker as detections with score 1.0, a track that lost its annotated object is extended from the
annotation (Track.update + Confirmed, on the device through dd_tracker_track_update / dd_tracker_track_set), and
duplicate tracks of one annotated object are dropped (dd_tracker_remove via the tracker.tracks assignment).
Pinned by tests/golden/framerecords.json, produced by the reference's own module (scripts/make_golden_framerecords.py).
"""
import xml.etree.ElementTree as ET
from copy import copy

from .deep_sort.track import TrackState


class FrameRecord:
    """A box (tlbr) with the detector's label id; `order` = its row among the frame's detections."""

    def __init__(self, tlbr, label_id, order=None):
        self.tlbr, self.label_id, self.order = tlbr, label_id, order


class AnnotationRecord(FrameRecord):
    def __init__(self, annot_track_id, lbl, det_lbl_id, tlbr, outside, occluded, keyframe, z_order, order=None):
        super().__init__(tlbr, det_lbl_id, order=order)
        self.annotation_track_id = annot_track_id
        self.annotation_label = lbl
        self.is_outside, self.is_occluded, self.is_keyframe = outside, occluded, keyframe
        self.z_order = z_order
        self.tentative_matches = {}
        self.score = 1.0


class TentativeRecord(FrameRecord):
    def __init__(self, tlbr, label_id, score, order=None):
        super().__init__(tlbr, label_id, order=order)
        self.score = score


def overlap(a, b):
    """Intersection area over the smaller of the two box areas (framerecords.py:35-40)."""
    ax1, ay1, ax2, ay2 = list(a.tlbr)
    bx1, by1, bx2, by2 = list(b.tlbr)
    inter = max(0, min(ax2, bx2) - max(ax1, bx1)) * max(0, min(ay2, by2) - max(ay1, by1))
    return inter / min(abs(ax2 - ax1) * abs(ay2 - ay1), abs(bx2 - bx1) * abs(by2 - by1))


def _tlwh_to_tlbr(box):
    out = copy(box)
    out[2:] = out[:2] + out[2:]
    return out


def _tlbr_to_tlwh(box):
    out = copy(box)
    out[2:] = out[2:] - out[:2]
    return out


class FrameRecords:
    def __init__(self, detector_id_to_labelname, overlap_threshold=0.9, override_tentative_detections=True, minimum_track_frames=3):
        self.frames = {}
        self.labels = {}
        self.detector_id_to_labelname = detector_id_to_labelname
        self.detector_labelname_to_id = {name: i for i, name in detector_id_to_labelname.items()}
        self.overlap_threshold = overlap_threshold
        self.override_tentative_detections = override_tentative_detections
        self.minimum_track_frames = minimum_track_frames

    # ---- loading the annotation file (deepdish.py:617-641 feeds these)
    def add_annotation_label_info(self, annotlabelname, detectorlabelid, annotlabelcolor):
        self.labels[annotlabelname] = {'detector_id': detectorlabelid, 'color': annotlabelcolor}

    def add_annotated_track(self, frame, annot_track_id, lbl, pts, outside, occluded, keyframe, z_order):
        rec = AnnotationRecord(annot_track_id, lbl, self.labels[lbl]['detector_id'], pts, outside, occluded, keyframe, z_order)
        self.frames.setdefault(frame, []).append(rec)

    # ---- per frame, after NMS (deepdish.py:1001)
    def _claim(self, rec, pool, frame):
        """First unclaimed detection that covers the annotation and agrees on the label (an annotation whose label the
        detector does not know agrees with anything); it leaves the pool."""
        for k, cand in enumerate(pool):
            if overlap(rec, cand) >= self.overlap_threshold and (rec.label_id == cand.label_id or rec.label_id is None):
                rec.tentative_matches[frame] = copy(cand)
                del pool[k]
                return True
        return False

    def process_boxes(self, frame, boxes_in, labelnames_in, scores_in):
        pool = [TentativeRecord(_tlwh_to_tlbr(tlwh), self.detector_labelname_to_id[name], score, order=i)
                for i, (tlwh, name, score) in enumerate(zip(boxes_in, labelnames_in, scores_in))]
        confirmed, missed, unknown = [], [], []
        for rec in self.frames.setdefault(frame, []):
            if not isinstance(rec, AnnotationRecord):
                continue
            if self._claim(rec, pool, frame):
                confirmed.append(rec)
            elif rec.label_id is not None:
                missed.append(rec)                     # the detector did not see it: the annotation becomes a detection
            else:
                unknown.append(rec)                    # nothing the encoder / tracker could be told about
        feed = confirmed + pool + missed
        boxes_out, labels_out, scores_out = [], [], []
        for i, rec in enumerate(feed):
            rec.order = i
            boxes_out.append(_tlbr_to_tlwh(rec.tlbr))
            labels_out.append(self.detector_id_to_labelname[rec.label_id])
            scores_out.append(rec.score)
        self.frames[frame] = feed + unknown
        return boxes_out, labels_out, scores_out

    # ---- per frame, after the encoder (deepdish.py:1017)
    def process_detections(self, frame, detections):
        for det, rec in zip(detections, self.frames[frame]):
            rec.detection, det.record = det, rec
        return detections

    # ---- per frame, after tracker.update (deepdish.py:1047: tracker.tracks = process_tracking(...))
    def process_tracking(self, frame, tracker, tracks=None):
        tracks = tracker.tracks if tracks is None else tracks
        followers = {}                                  # annotated track id -> [(tracker id, detections with a record)]
        for t in tracks:
            recorded = [d for d in t.detections if hasattr(d, 'record')]
            annotated = {d.record.annotation_track_id for d in recorded if isinstance(d.record, AnnotationRecord)}
            if len(annotated) == 1:
                (i,) = annotated
                here = next((r for r in self.frames[frame] if isinstance(r, AnnotationRecord) and r.annotation_track_id == i), None)
                if here is not None:
                    followers.setdefault(i, []).append((t.track_id, len(recorded)))
                    if t.time_since_update > 0:         # the tracker lost it this frame, the annotation has not
                        t.update(tracker.kf, here.detection)
                        t.state = TrackState.Confirmed
                        t.time_since_update = 0
            for d in recorded:
                d.record.track = t
        doomed = set()
        for entries in followers.values():
            best = max(n for _, n in entries)
            doomed.update(tid for tid, n in entries if n < best)
        return [t for t in tracks if t.track_id not in doomed]

    # ---- at shutdown (deepdish.py:795-805)
    @staticmethod
    def _box(parent, frame, rec, occluded, outside, keyframe, z_order):
        return ET.SubElement(parent, 'box', attrib={'frame': str(frame), 'occluded': occluded, 'outside': outside, 'keyframe': keyframe,
                                                    'z_order': z_order, 'xtl': str(rec.tlbr[0]), 'ytl': str(rec.tlbr[1]),
                                                    'xbr': str(rec.tlbr[2]), 'ybr': str(rec.tlbr[3])})

    def xml_output(self, meta=None):
        root = ET.Element('annotations')
        ET.SubElement(root, 'version').text = '1.1'
        if meta is not None:
            root.append(meta)
        manual, automatic = {}, {}
        for frame, recs in self.frames.items():
            for rec in recs:
                if hasattr(rec, 'annotation_track_id'):
                    manual.setdefault(rec.annotation_track_id, {})[frame] = rec
                elif hasattr(rec, 'track'):
                    automatic.setdefault(rec.track.track_id, {})[frame] = rec
        next_id = 0
        for i, by_frame in sorted(manual.items()):                # the annotated tracks keep their ids
            next_id = max(next_id, i)
            el = ET.SubElement(root, 'track', attrib={'id': str(i), 'source': 'manual'})
            for frame, rec in sorted(by_frame.items()):
                self._box(el, frame, rec, '1' if rec.is_occluded else '0', '1' if rec.is_outside else '0',
                          '1' if rec.is_keyframe else '0', str(rec.z_order))
                name = self.detector_id_to_labelname[rec.label_id]
            el.set('label', name)
        next_id += 1
        for _, by_frame in sorted(automatic.items()):             # tracks the tracker found on its own, if long enough
            if len(by_frame) < self.minimum_track_frames:
                continue
            el = ET.SubElement(root, 'track', attrib={'id': str(next_id), 'source': 'automatic'})
            next_id += 1
            votes = {}
            for frame, rec in sorted(by_frame.items()):
                votes[rec.label_id] = votes.get(rec.label_id, 0) + 1
                last = self._box(el, frame, rec, '0', '0', '1', '0')
            last.set('outside', '1')                              # the track ends here
            el.set('label', self.detector_id_to_labelname[max(votes, key=votes.get)])
        tree = ET.ElementTree(root)
        ET.indent(tree)
        return tree


def load_cvat_annotations(framerec, xmltree, detector_labels):
    """deepdish.py:617-641: register the label table and every annotated box of a CVAT 1.1 annotations.xml.
    detector_labels: the detector plugin's id -> name dict; annotation labels it does not know map to None."""
    name_to_id = {v: k for k, v in detector_labels.items()}
    root = xmltree.getroot()
    for l in root.findall('./meta/task/labels/label'):
        name = l.find('name').text
        framerec.add_annotation_label_info(name, name_to_id.get(name, None), l.find('color').text)
    import numpy as np
    for t in root.findall('./track'):
        for b in t.findall('box'):
            pts = np.array([b.get('xtl'), b.get('ytl'), b.get('xbr'), b.get('ybr')], dtype=float)
            framerec.add_annotated_track(int(b.get('frame')), int(t.get('id')), t.get('label'), pts, b.get('outside') == '1',
                                         b.get('occluded') == '1', b.get('keyframe') == '1', int(b.get('z_order')))
    return framerec
