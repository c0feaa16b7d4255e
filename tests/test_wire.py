"""Result packaging (deepdish_amd/wire.py) against the reference's published messages (README.md "MQTT output
examples" upstream, quoted below as data) and against the key order of deepdish.py:1141-1185,545-561,244-253."""
import json
import os

import numpy as np

from deepdish_amd import wire

# The two crossing examples of the reference's README, verbatim.
README_CROSSING_NEG = ('{"acp_ts": "1606480244.4554827", "acp_id": "deepdish-dd01", "acp_event": "crossing", "acp_event_value": "neg", '
                       '"temp": 61.835, "poscount_person": 5, "negcount_person": 7, "diff_person": -2, "intcount_person": 12, '
                       '"delcount_person": 1}')
README_CROSSING_POS = ('{"acp_ts": "1606480245.8179724", "acp_id": "deepdish-dd01", "acp_event": "crossing", "acp_event_value": "pos", '
                       '"temp": 62.322, "poscount_person": 6, "negcount_person": 7, "diff_person": -1, "intcount_person": 13, '
                       '"delcount_person": 1}')


def test_crossing_messages_reproduce_the_readme_examples():
    c = wire.Counts(['person'], [[5, 7, 12, 1]])
    assert json.dumps(wire.crossing_mqtt_payload(1606480244.4554827, 'deepdish-dd01', wire.crossing_type(-3.5), 61.835, c)) == README_CROSSING_NEG
    c = wire.Counts(['person'], [[6, 7, 13, 1]])
    assert json.dumps(wire.crossing_mqtt_payload(1606480245.8179724, 'deepdish-dd01', wire.crossing_type(0.0), 62.322, c)) == README_CROSSING_POS


def test_heartbeat_and_log_records_have_the_reference_keys_in_order():
    c = wire.Counts(['person', 'car'], [[6, 7, 13, 2], [1, 0, 1, 0]])
    hb = wire.heartbeat_mqtt_payload(1606480354.9866521, 'deepdish-dd01', 58.426, c)
    assert list(hb) == ['acp_ts', 'acp_id', 'acp_event', 'temp',
                        'poscount_person', 'negcount_person', 'diff_person', 'intcount_person', 'delcount_person',
                        'poscount_car', 'negcount_car', 'diff_car', 'intcount_car', 'delcount_car']
    assert hb['acp_event'] == 'heartbeat' and hb['acp_ts'] == '1606480354.9866521' and hb['diff_person'] == -1 and hb['diff_car'] == 1
    cr = wire.crossing_log_record(1606480245.5, 321, None, c)
    assert list(cr)[:4] == ['timestamp', 'asctime', 'frame_count', 'temp'] and cr['frame_count'] == 321 and cr['temp'] is None
    hl = wire.heartbeat_log_record(1606480245.5, 99, 40.0, c)
    assert list(hl)[:4] == ['timestamp', 'asctime', 'temp', 'frame_count'] and hl['asctime'] == cr['asctime']


def test_log_restore_round_trip(tmp_path):
    path = str(tmp_path / 'events.log')
    c = wire.Counts(['person', 'bicycle'])
    sent = []
    sink = wire.ResultSink(c, acp_id='dd', publish=lambda topic, msg: sent.append((topic, json.loads(msg))), log=path, temp=lambda: 50.5)
    c.poscount['person'] += 1; c.intcount['person'] += 1
    sink.crossings([('person', 2.0, 7), ('dog', -1.0, 8)], 1000.25, 41)           # labels nobody counts are not reported
    c.negcount['bicycle'] += 1; c.intcount['bicycle'] += 1; c.delcount['person'] += 3
    sink.crossings([('bicycle', -0.5, 9)], 1001.0, 57)
    sink.heartbeat(now=1002.0)
    assert [m['acp_event'] for _, m in sent] == ['crossing', 'crossing', 'heartbeat']
    assert [m.get('acp_event_value') for _, m in sent] == ['pos', 'neg', None] and sent[0][0] == 'default/topic'
    lines = [json.loads(l) for l in open(path)]
    assert len(lines) == 3 and [l['frame_count'] for l in lines] == [41, 57, 57]
    # a new process resumes from the last line (--restore-from-log) ...
    c2 = wire.Counts(['person', 'bicycle', 'car'])
    log2 = wire.EventLog(path, c2, restore=True)
    assert log2.frame_count == 57
    assert (c2.poscount, c2.negcount) == ({'person': 1, 'bicycle': 0, 'car': 0}, {'person': 0, 'bicycle': 1, 'car': 0})
    assert c2.delcount['person'] == 3 and c2.intcount == {'person': 1, 'bicycle': 1, 'car': 0}
    assert len(open(path).readlines()) == 3
    # ... and one started without it truncates the file (:559-561)
    wire.EventLog(path, wire.Counts(['person']), restore=False)
    assert os.path.getsize(path) == 0
    # mqtt_verbosity 0 silences MQTT but not the log
    quiet = wire.ResultSink(c, publish=lambda *a: sent.append(a), mqtt_verbosity=0, log=path)
    quiet.heartbeat(now=5.0)
    assert len(sent) == 3 and len(open(path).readlines()) == 1


def test_frame_text_and_frame_json():
    line = wire.frame_text_line(12, [('objd', 0.0314), ('feat', 0.0096), ('e2e', 0.25)], temp=61.6, pipe=(3, 87.4))
    assert line == 'Frame 12: objd=31ms feat=10ms e2e=250ms temp=62C pipe=3 cpup=87%\n'
    assert wire.frame_text_line(1, []) == 'Frame 1:\n'
    p = wire.frame_mqtt_payload('dd', 12, 1000.5, [('objd', 0.0314)], temp=61.6, pipe=(3, [1, 0, 2], 87.4, 1500000),
                                detections=[np.array([1.9, 2.2, 30.7, 40.1])],
                                tracks=[(np.array([10.6, 20.2, 50.9, 90.0]), 'person', 0.75, 4)], ratios=(0.5, 2.0))
    assert list(p) == ['acp_event', 'acp_id', 'framenum', 'acp_ts', 'timing', 'temp', 'pipe', 'qsizes', 'cpup', 'freq', 'detections', 'tracks']
    assert p['timing'] == {'objd': 31} and p['detections'] == [{'bbox': [1, 2, 30, 40]}]
    assert p['tracks'] == [{'bbox': [5, 40, 25, 180], 'label': 'person', 'confidence': 0.75, 'track_id': 4}]
    json.dumps(p)
