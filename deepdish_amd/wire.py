"""Result packaging: what leaves the hot path (SURVEY.md section 8 f, n3).

The reference turns the counters into MQTT messages, a JSON-lines log it can resume from, a per-frame text line and
a per-frame JSON message (deepdish.py upstream: update_payload_with_state :1141-1145, publish_crossing_event
:1147-1166, periodic_heartbeat :1168-1185, on_mqtt_connect :643-665, log / --restore-from-log :545-561,
FrameInfo.do_text :244-253, text_output :1224-1238, the do_json methods :255-329).  The broker connection, the
asyncio plumbing and the CPU-temperature file are the host's; here are the payloads, key for key and in the
reference's key order (json.dumps keeps insertion order), built from this package's counters.

`counter` is anything with poscount / negcount / intcount / delcount dicts keyed by label (tools.countline.CountLine)
or a `Counts` made from a [n_labels, 4] (pos, neg, int, del) array (MultiStreamPipeline.counts()[stream])."""
import json
import os
from collections import deque
from time import asctime, localtime, time

import numpy as np


class Counts:
    """Adapter: labels + int array [n_labels, 4] = (pos, neg, int, del) -> the four dicts the reference keeps."""

    def __init__(self, labels, table=None):
        self.labels = list(labels)
        t = np.zeros((len(self.labels), 4), dtype=np.int64) if table is None else np.asarray(table).reshape(len(self.labels), 4)
        self.poscount = {l: int(t[i, 0]) for i, l in enumerate(self.labels)}
        self.negcount = {l: int(t[i, 1]) for i, l in enumerate(self.labels)}
        self.intcount = {l: int(t[i, 2]) for i, l in enumerate(self.labels)}
        self.delcount = {l: int(t[i, 3]) for i, l in enumerate(self.labels)}


def _labels(counter):
    return list(getattr(counter, 'labels', None) or counter.poscount.keys())


def update_payload_with_state(payload, counter):
    """deepdish.py:1141-1145: five fields per wanted label, in this order."""
    for lbl in _labels(counter):
        payload['poscount_' + lbl] = counter.poscount[lbl]
        payload['negcount_' + lbl] = counter.negcount[lbl]
        payload['diff_' + lbl] = counter.poscount[lbl] - counter.negcount[lbl]
        payload['intcount_' + lbl] = counter.intcount[lbl]
        payload['delcount_' + lbl] = counter.delcount[lbl]
    return payload


def crossing_type(cp):
    """deepdish.py:1118-1121: the sign of the cross product names the direction."""
    return 'pos' if cp >= 0 else 'neg'


def crossing_mqtt_payload(t_frame, acp_id, crossing, temp, counter):
    """:1156-1158."""
    return update_payload_with_state({'acp_ts': str(t_frame), 'acp_id': acp_id, 'acp_event': 'crossing',
                                      'acp_event_value': crossing, 'temp': temp}, counter)


def heartbeat_mqtt_payload(now, acp_id, temp, counter):
    """:1172-1175."""
    return update_payload_with_state({'acp_ts': str(now), 'acp_id': acp_id, 'acp_event': 'heartbeat', 'temp': temp}, counter)


def crossing_log_record(t_frame, frame_count, temp, counter):
    """:1162-1164."""
    return update_payload_with_state({'timestamp': str(t_frame), 'asctime': asctime(localtime(t_frame)),
                                      'frame_count': frame_count, 'temp': temp}, counter)


def heartbeat_log_record(now, frame_count, temp, counter):
    """:1178-1182 (frame_count is set before the counters here)."""
    payload = {'timestamp': str(now), 'asctime': asctime(localtime(now)), 'temp': temp}
    payload['frame_count'] = frame_count
    return update_payload_with_state(payload, counter)


def initialisation_payload(now, acp_id, hot_path, model, encoder_model, input_name, args=None):
    """:646-664, for a pipeline.HotPath; `args` may carry the host-only settings (powersaving, skip frames, ...)."""
    a = args or {}
    det, enc = hot_path.object_detector, hot_path.encoder
    bs = None if not hot_path.background_subtraction else hot_path.background_subtraction_ratio
    return {'acp_ts': str(now), 'acp_event': 'initialisation', 'acp_id': acp_id,
            'model': model, 'model_class': type(det).__name__,
            'encoder_model': encoder_model, 'encoder_model_class': type(enc.image_encoder).__name__,
            'input': input_name,
            'use_edgetpu': getattr(det, 'use_edgetpu', False),
            'input_shape': [det.width, det.height],
            'encoder_input_shape': [enc.width, enc.height],
            'num_threads': det.num_threads,
            'max_age': a.get('max_age', hot_path.tracker.max_age),
            'max_iou_distance': a.get('max_iou_distance', hot_path.tracker.max_iou_distance),
            'nms_max_overlap': hot_path.nms_max_overlap,
            'max_cosine_distance': a.get('max_cosine_distance', hot_path.tracker.metric.matching_threshold),
            'background_subtraction': bs,
            'powersaving': a.get('powersaving'),
            'cpu_governor': a.get('cpu_governor'),
            'object_detector_skip_frames': a.get('object_detector_skip_frames'),
            'interframe_interval': a.get('interframe_interval'),
            'simulate_camera': a.get('simulate_camera')}


def frame_text_line(framenum, timings, temp=None, pipe=None):
    """:244-253.  timings: [(short_label, seconds)] in display order; pipe: (frames in flight, cpu percent)."""
    s = 'Frame {}:'.format(framenum)
    for label, dt in timings:
        s += ' {}={:.0f}ms'.format(label, dt * 1000)
    if temp is not None:
        s += ' temp={:.0f}C'.format(temp)
    if pipe is not None:
        s += ' pipe={} cpup={:.0f}%'.format(pipe[0], pipe[1])
    return s + '\n'


def frame_mqtt_payload(acp_id, framenum, t_frame, timings=(), temp=None, pipe=None, detections=(), tracks=(), ratios=(1.0, 1.0)):
    """:1232-1238 with the do_json methods in priority order (:255-329).  detections: tlbr arrays; tracks:
    (tlbr, label, confidence, track_id); pipe: (count, qsizes, cpup, freq); ratios = trackdata_ratios (:734-735)."""
    p = {'acp_event': 'frame', 'acp_id': acp_id, 'framenum': framenum, 'acp_ts': str(t_frame)}
    for label, dt in timings:
        p.setdefault('timing', {})[label] = round(dt * 1000)
    if temp is not None:
        p['temp'] = temp
    if pipe is not None:
        p['pipe'], p['qsizes'], p['cpup'], p['freq'] = pipe
    for bbox in detections:
        p.setdefault('detections', []).append({'bbox': np.asarray(bbox).astype(np.int32).tolist()})
    wr, hr = ratios
    for bbox, label, conf, tid in tracks:
        b = np.asarray(bbox).astype(np.float32) * [wr, hr, wr, hr]
        p.setdefault('tracks', []).append({'bbox': b.astype(np.int32).tolist(), 'label': label, 'confidence': conf, 'track_id': tid})
    return p


class EventLog:
    """--log FILE / --restore-from-log (:545-561): JSON lines, one per crossing or heartbeat; on start either resume
    the counters and the frame count from the last line or truncate the file."""

    def __init__(self, path, counter, restore=False):
        self.path = path
        self.frame_count = 0
        if restore and os.path.exists(path):
            with open(path, mode='r') as f:
                last = deque(f, 1)
            if len(last) > 0:
                data = json.loads(last.pop())
                for lbl in _labels(counter):
                    counter.poscount[lbl] = data.get('poscount_' + lbl, 0)
                    counter.negcount[lbl] = data.get('negcount_' + lbl, 0)
                    counter.delcount[lbl] = data.get('delcount_' + lbl, 0)
                    counter.intcount[lbl] = data.get('intcount_' + lbl, 0)
                self.frame_count = data.get('frame_count', 0)
        else:
            with open(path, mode='w+') as f:
                f.truncate()

    def write(self, record):
        with open(self.path, mode='a+') as f:
            f.write(json.dumps(record) + '\n')


class ResultSink:
    """What pipeline.HotPath calls after the count line: one MQTT message and one log line per crossing
    (:1116-1123,1147-1166), heartbeats on demand (:1168-1185).  publish(topic, json_string) is the host's MQTT client
    method (gmqtt Client.publish upstream); temp() its CPU-temperature reader (None when there is none, :818)."""

    def __init__(self, counter, acp_id=None, topic='default/topic', publish=None, mqtt_verbosity=1, log=None, restore_from_log=False,
                 temp=None):
        self.counter, self.acp_id, self.topic = counter, acp_id, topic
        self.publish, self.verbosity = publish, mqtt_verbosity
        self.temp = temp or (lambda: None)
        self.log = EventLog(log, counter, restore_from_log) if log is not None else None
        self.frame_count = self.log.frame_count if self.log else 0

    def crossings(self, events, t_frame, framenum):
        """events: [(label, cross product, track id)] from CountLine.step, counters already updated."""
        self.frame_count = framenum
        for lbl, cp, _tid in events:
            if lbl not in self.counter.poscount:
                continue
            kind, temp = crossing_type(cp), self.temp()
            if self.publish is not None and self.verbosity > 0:
                self.publish(self.topic, json.dumps(crossing_mqtt_payload(t_frame, self.acp_id, kind, temp, self.counter)))
            if self.log is not None:
                self.log.write(crossing_log_record(t_frame, framenum, temp, self.counter))

    def heartbeat(self, now=None):
        now = time() if now is None else now
        temp = self.temp()
        if self.publish is not None and self.verbosity > 0:
            self.publish(self.topic, json.dumps(heartbeat_mqtt_payload(now, self.acp_id, temp, self.counter)))
        if self.log is not None:
            self.log.write(heartbeat_log_record(now, self.frame_count, temp, self.counter))
