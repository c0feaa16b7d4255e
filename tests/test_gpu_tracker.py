"""GPU: whole-sequence parity of the HBM-resident tracker (track ids, states, hits, ages, means,
NMS keep lists, crossing counts) against the golden scenes produced by the reference Tracker."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')


def _scene_for(g, name):
    from deepdish_amd.synth import Scene, tracker_scene
    if name == 'n256':
        return tracker_scene(seed=int(g['seed']), n_obj=int(g['n_obj']), n_frames=int(g['n_frames']))
    kw = dict(n5={}, n20={}, n20_age5=dict(p_miss=0.1))[name]
    return Scene(seed=int(g['seed']), n_obj=int(g['n_obj']), n_frames=int(g['n_frames']), **kw)


@pytest.mark.parametrize('name', ['n5', 'n20', 'n20_age5', 'n256'])
def test_scene_golden(name):
    from deepdish_amd.deep_sort import nn_matching, preprocessing
    from deepdish_amd.deep_sort.tracker import Tracker
    from deepdish_amd.deep_sort.detection import Detection
    from deepdish_amd.tools.countline import CountLine
    g = np.load(os.path.join(G, f'scene_{name}.npz'))
    scene = _scene_for(g, name)
    metric = nn_matching.NearestNeighborDistanceMetric('cosine', 0.2, None)
    trk = Tracker(metric, max_iou_distance=0.7, max_age=int(g['max_age']))
    counter = CountLine(scene.countline())
    fp, kp = g['frame_ptr'], g['keep_ptr']
    for f in range(int(g['n_frames'])):
        boxes, scores, who, feats = scene.detections(f)
        keep = preprocessing.non_max_suppression(boxes, 0.6, scores)
        assert keep == g['nms_keep'][kp[f]:kp[f + 1]].tolist(), f'nms frame {f}'
        dets = [Detection(boxes[i], 'person', scores[i], feats[i]) for i in keep]
        trk.predict()
        trk.update(dets)
        counter.step(trk)
        want_i = g['track_int'][fp[f]:fp[f + 1]]
        want_m = g['track_mean'][fp[f]:fp[f + 1]]
        got_i = np.array([[t.track_id, t.state, t.time_since_update, t.hits, t.age] for t in trk.tracks],
                         dtype=np.int64).reshape(-1, 5)
        np.testing.assert_array_equal(got_i, want_i, err_msg=f'frame {f}')
        if len(trk.tracks):
            np.testing.assert_allclose(np.array([t.mean for t in trk.tracks]), want_m, rtol=1e-8, atol=1e-8,
                                       err_msg=f'frame {f}')
    assert trk._next_id == int(g['next_id'])
    np.testing.assert_array_equal(counter.vector()[0], g['counts'])
    # covariance fetch path + symmetry / positive-definiteness property
    c = trk.tracks[0].covariance
    assert c.shape == (8, 8)
    np.testing.assert_allclose(c, c.T, rtol=1e-9, atol=1e-12)
    assert np.all(np.linalg.eigvalsh((c + c.T) / 2) > 0)


def test_long_scene_unbounded_gallery():
    """nn_budget=None keeps every sample (deepdish.py:515, nn_matching.py:137-154): 420 frames, eight tracks alive
    throughout, galleries of up to 413 samples.  Track table identical to the reference's every frame, and the
    appearance costs the device associated with at frames 200 / 300 / 360 / 419 equal the reference metric's
    (tests/test_oracle_golden.py shows a 256-sample ring misses them by > 1e-4).  gallery_capacity=64 starts the
    chunk table at two chunks per track, so it is re-built at 4, 8 and 16 on the way."""
    from deepdish_amd.deep_sort import nn_matching, preprocessing
    from deepdish_amd.deep_sort.tracker import Tracker
    from deepdish_amd.deep_sort.detection import Detection
    from deepdish_amd.tools.countline import CountLine
    from deepdish_amd.synth import Scene
    g = np.load(os.path.join(G, 'scene_long_n8.npz'))
    scene = Scene(seed=int(g['seed']), n_obj=int(g['n_obj']), n_frames=int(g['n_frames']), churn=False)
    trk = Tracker(nn_matching.NearestNeighborDistanceMetric('cosine', 0.2, None), max_iou_distance=0.7,
                  max_age=int(g['max_age']), gallery_capacity=64)
    counter = CountLine(scene.countline())
    fp, kp = g['frame_ptr'], g['keep_ptr']
    compared, worst = 0, 0.0
    for f in range(int(g['n_frames'])):
        boxes, scores, who, feats = scene.detections(f)
        keep = preprocessing.non_max_suppression(boxes, 0.6, scores)
        assert keep == g['nms_keep'][kp[f]:kp[f + 1]].tolist(), f'nms frame {f}'
        trk.predict()
        before = [(t.track_id, t.is_confirmed()) for t in trk.tracks]
        trk.update([Detection(boxes[i], 'person', scores[i], feats[i]) for i in keep])
        counter.step(trk)
        got_i = np.array([[t.track_id, t.state, t.time_since_update, t.hits, t.age] for t in trk.tracks],
                         dtype=np.int64).reshape(-1, 5)
        np.testing.assert_array_equal(got_i, g['track_int'][fp[f]:fp[f + 1]], err_msg=f'frame {f}')
        np.testing.assert_allclose(np.array([t.mean for t in trk.tracks]), g['track_mean'][fp[f]:fp[f + 1]], rtol=1e-8, atol=1e-8)
        if f'cost_{f}' in g:
            app, _ = trk.last_cost()
            rows = [r for r, (_, conf) in enumerate(before) if conf]
            assert [before[r][0] for r in rows] == g[f'cost_ids_{f}'].tolist()
            got, want = app[rows], g[f'cost_{f}']
            live = got < 1e4                                        # gated entries are 1e5 (linear_assignment.py:181-189)
            assert live.sum() >= min(got.shape) - 2, f                # (nearly) every track's own detection passes the gate
            worst = max(worst, float(np.abs(got[live] - want[live]).max()))
            compared += int(live.sum())
    assert compared >= 30 and worst <= 2e-6, (compared, worst)      # stated cosine tolerance (f32 both sides)
    np.testing.assert_array_equal(counter.vector()[0], g['counts'])
    assert trk._next_id == int(g['next_id'])


@pytest.mark.parametrize('budget', [40, 100])
def test_budget_ring_across_chunks_matches_oracle(budget):
    """nn_budget = B keeps the last B samples of a track (nn_matching.py:150-153).  On the device that is a ring over
    ceil(B / 32) gallery chunks: B = 40 wraps inside the second chunk, B = 100 inside the fourth.  A scene with births,
    deaths and misses (slots and chunks are recycled): track table every frame and the appearance costs the device
    associated with equal the oracle's metric with the same budget."""
    from deepdish_amd.deep_sort import nn_matching, preprocessing
    from deepdish_amd.deep_sort.tracker import Tracker
    from deepdish_amd.deep_sort.detection import Detection
    from deepdish_amd.synth import Scene
    from oracle import deepsort_np as ds
    scene = Scene(seed=17, n_obj=10, n_frames=150, p_miss=0.08)
    trk = Tracker(nn_matching.NearestNeighborDistanceMetric('cosine', 0.2, budget), max_iou_distance=0.7, max_age=8,
                  track_capacity=24, gallery_capacity=32)
    otrk = ds.Tracker(ds.Metric(0.2, budget), max_iou_distance=0.7, max_age=8)
    compared, worst = 0, 0.0
    for f in range(150):
        boxes, scores, who, feats = scene.detections(f)
        keep = preprocessing.non_max_suppression(boxes, 0.6, scores)
        trk.predict(); otrk.predict()
        ids = [t.track_id for t in otrk.tracks if t.state == 2]
        odets = [ds.Det(boxes[i], 'person', scores[i], feats[i]) for i in keep]
        want = otrk.metric.distance(np.array([d.feature for d in odets]), ids) if ids and odets else None
        before = [(t.track_id, t.is_confirmed()) for t in trk.tracks]
        trk.update([Detection(boxes[i], 'person', scores[i], feats[i]) for i in keep])
        otrk.update(odets)
        got_i = [(t.track_id, t.state, t.time_since_update, t.hits, t.age) for t in trk.tracks]
        assert got_i == [(t.track_id, t.state, t.time_since_update, t.hits, t.age) for t in otrk.tracks], f
        if want is not None:
            app, _ = trk.last_cost()
            rows = [r for r, (_, conf) in enumerate(before) if conf]
            assert [before[r][0] for r in rows] == ids
            got = app[rows]
            live = got < 1e4
            if live.any():
                worst = max(worst, float(np.abs(got[live] - want[live]).max()))
                compared += int(live.sum())
    assert compared > 300 and worst <= 2e-6, (compared, worst)
    assert trk._next_id > 12                                   # tracks died and were born: slots and chunks were recycled
    assert max(len(v) for v in otrk.metric.samples.values()) == budget


def test_tracker_edge_cases():
    from deepdish_amd.deep_sort import nn_matching
    from deepdish_amd.deep_sort.tracker import Tracker
    from deepdish_amd.deep_sort.detection import Detection
    metric = nn_matching.NearestNeighborDistanceMetric('cosine', 0.2, 4)      # budget ring
    trk = Tracker(metric, max_age=3, track_capacity=8, gallery_capacity=8)
    trk.predict(); trk.update([])                                              # empty frame, no tracks
    assert trk.tracks == [] and trk.deleted_tracks == []
    f = np.zeros(128, np.float32); f[0] = 1
    for _ in range(12):                                                        # overflow budget ring
        trk.predict(); trk.update([Detection([10, 10, 20, 40], 'person', 0.9, f)])
    assert len(trk.tracks) == 1 and trk.tracks[0].is_confirmed() and trk.tracks[0].hits == 12
    assert trk.tracks[0].features == [] and len(trk.tracks[0].detections) == 12      # tracker.py:91 / track.py:152
    assert trk.tracks[0].get_label(True)[0] == 'person'
    for _ in range(4):                                                         # starve -> deleted after max_age
        trk.predict(); trk.update([])
    assert len(trk.tracks) == 0 and len(trk.deleted_tracks) == 1 and trk.deleted_tracks[0].is_deleted()
    from deepdish_amd._lib import DeepDishHipError
    dets = [Detection([50 * i, 10, 20, 40], 'person', 0.9, f) for i in range(9)]
    trk.predict()
    with pytest.raises(DeepDishHipError):                                      # capacity is a loud error
        trk.update(dets)


def test_host_interventions_between_updates_match_the_oracle():
    """the host may call track.update(kf, det), assign track.state / time_since_update and reassign
    tracker.tracks between two updates (deepdish/framerecords.py:133-165 via deepdish.py:1047), and call
    mark_missed().  Same interventions on the oracle tracker; ids, states, counters and means must keep agreeing
    on every later frame (the forced feature lands in the gallery, the dropped track's slot is recycled)."""
    from deepdish_amd.deep_sort import nn_matching
    from deepdish_amd.deep_sort.tracker import Tracker
    from deepdish_amd.deep_sort.detection import Detection
    from deepdish_amd.deep_sort.track import TrackState
    from deepdish_amd.synth import Scene
    from oracle import deepsort_np as ds
    scene = Scene(seed=31, n_obj=9, n_frames=40, p_miss=0.25)
    trk = Tracker(nn_matching.NearestNeighborDistanceMetric('cosine', 0.2, None), max_iou_distance=0.7, max_age=8)
    ora = ds.Tracker(ds.Metric(0.2), max_iou_distance=0.7, max_age=8)
    done = dict(forced=0, removed=0, missed=0)
    for f in range(40):
        boxes, scores, who, feats = scene.detections(f)
        keep = ds.non_max_suppression(boxes, 0.6, scores)
        trk.predict(); ora.predict()
        trk.update([Detection(boxes[i], 'person', scores[i], feats[i]) for i in keep])
        ora.update([ds.Det(boxes[i], 'person', scores[i], feats[i]) for i in keep])
        if f >= 6 and f % 3 == 0:
            # framerecords.py:156-161: a track that missed this frame is extended from an annotation and confirmed
            for t, o in zip(trk.tracks, ora.tracks):
                if t.time_since_update > 0:
                    box = np.round(t.to_tlwh()).astype(np.int64) + np.array([1, -1, 0, 2])
                    feat = feats[0] if len(feats) else np.ones(128, np.float32)
                    t.update(trk.kf, Detection(box, 'person', 1.0, feat))
                    t.state = TrackState.Confirmed
                    t.time_since_update = 0
                    o.update(ds.Det(box, 'person', 1.0, feat)); o.state = ds.CONFIRMED; o.time_since_update = 0
                    done['forced'] += 1
                    break
        if f in (10, 20) and len(trk.tracks) > 2:
            # framerecords.py:176-188: a duplicate track is dropped by reassigning tracker.tracks
            victim = trk.tracks[1].track_id
            trk.tracks = [t for t in trk.tracks if t.track_id != victim]
            ora.tracks = [t for t in ora.tracks if t.track_id != victim]
            done['removed'] += 1
        if f == 17 and trk.tracks:                                   # track.py:113-125 on one track
            trk.tracks[0].predict(trk.kf); ora.tracks[0].predict()
            np.testing.assert_allclose(trk.tracks[0].covariance, ora.tracks[0].covariance, rtol=1e-9, atol=1e-9)
        if f in (13, 26):
            for t, o in zip(list(trk.tracks), list(ora.tracks)):
                if t.is_tentative():
                    t.mark_missed(); o.mark_missed()
                    ora.tracks = [x for x in ora.tracks if not x.is_deleted()]
                    assert t.is_deleted() and t not in trk.tracks
                    done['missed'] += 1
                    break
        got = [(t.track_id, t.state, t.time_since_update, t.hits, t.age) for t in trk.tracks]
        want = [(t.track_id, t.state, t.time_since_update, t.hits, t.age) for t in ora.tracks]
        assert got == want, f
        if want:
            np.testing.assert_allclose(np.array([t.mean for t in trk.tracks]), np.array([t.mean for t in ora.tracks]),
                                       rtol=1e-8, atol=1e-8, err_msg=f'frame {f}')
    assert done['forced'] >= 3 and done['removed'] == 2, done
    assert trk._next_id == ora._next_id
    from deepdish_amd.deep_sort.track import Track
    stranger = Track(np.zeros(8), None, 9999, 3, 8, Detection([0, 0, 5, 5], 'person', 0.5, np.zeros(128, np.float32)))
    before = list(trk.tracks)
    with pytest.raises(ValueError):
        trk.tracks = trk.tracks + [stranger]                     # tracks can be dropped or reordered, not invented
    assert trk.tracks == before
