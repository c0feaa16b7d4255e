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
    assert trk.tracks[0].get_label(True)[0] == 'person'
    for _ in range(4):                                                         # starve -> deleted after max_age
        trk.predict(); trk.update([])
    assert len(trk.tracks) == 0 and len(trk.deleted_tracks) == 1 and trk.deleted_tracks[0].is_deleted()
    from deepdish_amd._lib import DeepDishHipError
    dets = [Detection([50 * i, 10, 20, 40], 'person', 0.9, f) for i in range(9)]
    trk.predict()
    with pytest.raises(DeepDishHipError):                                      # capacity is a loud error
        trk.update(dets)
