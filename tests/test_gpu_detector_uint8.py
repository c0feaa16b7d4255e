"""GPU: the headline's detector -- the uint8 SSD-MobileNet-v1 (csrc/netsq.hip) -- held to the oracle at the level the reference
exposes it: SSD_MOBILENET(...).detect_image(img) -> (boxes tlwh, labels, scores) (tools/ssd_mobilenet.py:100-150,198-213 upstream).
tests/test_gpu_quant.py pins the head tensors bit for bit; these tests pin the chain the default configuration runs behind them --
Lanczos stretch -> uint8 forward -> the post-process op's two stages -> predict()'s tail -> detect_image()'s filter -- through the
plugin, through the batched C++ pipeline WITHOUT injected detections, and with the post-process options a model file states."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MODEL = 'synthetic-ssd_mobilenet_v1-uint8.tflite'


def _labels():
    from deepdish_amd.pipeline import DEFAULT_LABELS
    return {i: l.strip() for i, l in enumerate(open(DEFAULT_LABELS))}


def _frames():
    """Four scene frames (two scenes) and two noise frames, BGR u8 [480, 640, 3]."""
    from deepdish_amd.synth import Scene
    a, b = Scene(seed=3, n_obj=8, n_frames=4), Scene(seed=11, n_obj=20, n_frames=4)
    rng = np.random.default_rng(5)
    return [a.frame(0), a.frame(3), b.frame(1), b.frame(2), rng.integers(0, 256, (480, 640, 3), dtype=np.uint8),
            rng.integers(0, 64, (480, 640, 3), dtype=np.uint8)]


def _oracle_detect(qm, frame_bgr, wanted, score_threshold=0.5, max_det=10, score_thr=1e-8, iou_thr=0.6):
    """The reference's detect_image on the oracle's arithmetic: deepdish.py:882 (BGR frame -> RGBA PIL image), ssd_mobilenet.py:54-57
    (convert + ANTIALIAS stretch: Pillow itself), :102-109 (the interpreter: oracle/nets_quant.py + the op's second stage in
    oracle/nets_torch.py), :111-150 and :198-213 (oracle/detectors_np.py).  -> (boxes, labels, scores, the op's four outputs)."""
    from PIL import Image
    from oracle import nets_quant, nets_torch, detectors_np
    h, w = frame_bgr.shape[:2]
    rgba = np.dstack([frame_bgr[..., ::-1], np.full((h, w, 1), 255, np.uint8)])
    resized = np.asarray(Image.fromarray(rgba, 'RGBA').convert('RGB').resize((300, 300), Image.LANCZOS))
    box_q, cls_q, _ = nets_quant.ssd_quant_forward(qm, resized[None])
    anchors = nets_quant.ssd_anchors(300)
    b, s, c, _ = nets_quant.ssd_quant_decode(qm, box_q[0], cls_q[0], anchors, score_thr)
    op = nets_torch.ssd_postprocess_decoded(b, s, c, max_det, score_thr, iou_thr)
    boxes, names, scores = detectors_np.ssd_predict_tail(list(op), _labels(), original_image_size=(w, h))
    return detectors_np.ssd_detect_filter(boxes, names, scores, wanted, score_threshold) + (op,)


def _by_class(boxes, labels, scores):
    """Rows grouped by label, pick order kept inside a label.  (The reference emits the classes in the iteration order of a Python set,
    ssd_finish_k in ascending id -- DESIGN.md section 2, known deviations; nothing downstream sees the order across classes.)"""
    out = {}
    for b, l, s in zip(boxes, labels, scores):
        out.setdefault(l, []).append((np.asarray(b, dtype=np.float64), float(s)))
    return out


def _same_detections(got, want, where):
    g, w = _by_class(*got), _by_class(*want)
    assert sorted(g) == sorted(w), (where, sorted(g), sorted(w))
    for label in w:
        assert len(g[label]) == len(w[label]), (where, label)
        for (gb, gs), (wb, ws) in zip(g[label], w[label]):
            assert gs == ws, (where, label, gs, ws)                    # a table entry / 256: no arithmetic between the byte and the score
            # decoded corners differ by expf vs numpy's exp (<= 2e-6 of the normalised box), scaled to pixels
            np.testing.assert_allclose(gb, wb, rtol=0, atol=2e-6 * 640, err_msg=str((where, label)))


@pytest.fixture(scope='module')
def plugin():
    from deepdish_amd.pipeline import make_detector
    wanted = [l for l in _labels().values() if l and l != '???']
    det = make_detector(MODEL, wanted_labels=wanted)
    assert det.ssdm.quantized and (det.width, det.height) == (300, 300)
    return det, wanted


def test_plugin_detections_equal_the_oracle_chain(plugin):
    from PIL import Image
    det, wanted = plugin
    qm = det.ssdm.weights
    n_rows = 0
    for k, frame in enumerate(_frames()):
        rgba = np.dstack([frame[..., ::-1], np.full(frame.shape[:2] + (1,), 255, np.uint8)])
        got = det.detect_image(Image.fromarray(rgba, 'RGBA'))
        got_dev = det.detect_frame_device(torch.from_numpy(frame).cuda(), 480, 640)
        wb, wl, ws, op = _oracle_detect(qm, frame, wanted)
        _same_detections(got, (wb, wl, ws), 'frame %d (detect_image)' % k)
        _same_detections(got_dev, (wb, wl, ws), 'frame %d (detect_frame_device)' % k)
        # the op's own four outputs, row by row: same anchors picked in the same order
        out = det.ssdm.invoke_device(det.ssdm.prepare_image_device(torch.from_numpy(rgba).cuda(), 480, 640, 4))
        assert int(out[3]) == op[3]
        np.testing.assert_array_equal(out[1][:op[3]], op[1][:op[3]])
        np.testing.assert_array_equal(out[2][:op[3]], op[2][:op[3]])
        np.testing.assert_allclose(out[0][:op[3]], op[0][:op[3]], rtol=0, atol=2e-6)
        n_rows += len(wl)
    assert n_rows > 0                                                  # the comparison saw detections, not six empty lists


def test_batched_pipeline_consumes_its_own_uint8_detector():
    """No injection: the C++ pipeline's Lanczos -> dd_net_ssd_decoded on the uint8 program -> ssd_postprocess_decoded -> ssd_finish ->
    adaptor filter -> hygiene -> NMS -> crops -> MARS -> tracker must leave every stream with the track table of a single-stream
    pipeline built from the plugin (whose detections the test above ties to the oracle)."""
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.pipeline import HotPath
    from deepdish_amd.synth import Scene
    wanted = sorted({l for l in _labels().values() if l and l != '???'})
    S, F = 8, 6
    scenes = [Scene(seed=20 + z, n_obj=4 + 2 * z, n_frames=F) for z in range(S)]
    mp = MultiStreamPipeline(S, model=MODEL, wanted_labels=wanted)
    assert mp.det_dtype == 'u8'
    hps = [HotPath(model=MODEL, wanted_labels=wanted) for _ in range(S)]
    seen = 0
    for f in range(F):
        frames = torch.from_numpy(np.stack([sc.frame(f) for sc in scenes])).cuda()
        mp.step(frames)
        for z in range(S):
            hps[z].step(frames[z])
            ints, means = mp.tracker(z).table()
            want = np.array([[t.track_id, t.state, t.time_since_update, t.hits, t.age] for t in hps[z].tracker.tracks], dtype=np.int64).reshape(-1, 5)
            np.testing.assert_array_equal(ints[:, :5], want, err_msg='frame %d stream %d' % (f, z))
            if len(want):
                np.testing.assert_allclose(means, np.array([t.mean for t in hps[z].tracker.tracks]), rtol=1e-9, atol=1e-9)
            seen = max(seen, len(want))
    assert seen > 0


def test_a_model_files_post_process_options_are_the_ones_that_run(tmp_path):
    """A file with max_detections = 20, nms_iou_threshold = 0.5, nms_score_threshold = 0.3 runs with those values (plugin and batched
    pipeline), i.e. gives the oracle's detections for those values -- and not the stock export's 10 / 0.6 / 1e-8."""
    from PIL import Image
    from deepdish_amd import quantize
    from deepdish_amd.tools import tflite_writer
    from deepdish_amd.tools.ssd_mobilenet import SSD_MOBILENET
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.pipeline import DEFAULT_LABELS
    qm = quantize.synthetic_ssd_quant_model(1234)
    path = str(tmp_path / 'ssd_mobilenet_v1_opts.tflite')
    tflite_writer.write_ssd_mobilenet(qm, path, post=dict(max_detections=20, nms_iou_threshold=0.5, nms_score_threshold=0.3))
    wanted = [l for l in _labels().values() if l and l != '???']
    det = SSD_MOBILENET(wanted_labels=wanted, model_file=path, label_file=DEFAULT_LABELS)
    assert det.ssdm.MAX_DET == 20 and det.ssdm.nms_iou_threshold == 0.5
    frames = _frames()[:3]
    differs = 0
    for k, frame in enumerate(frames):
        rgba = np.dstack([frame[..., ::-1], np.full(frame.shape[:2] + (1,), 255, np.uint8)])
        got = det.detect_image(Image.fromarray(rgba, 'RGBA'))
        want = _oracle_detect(qm, frame, wanted, max_det=20, score_thr=0.3, iou_thr=0.5)
        _same_detections(got, want[:3], 'frame %d' % k)
        stock = _oracle_detect(qm, frame, wanted)
        differs += int(stock[3][3] != want[3][3] or len(stock[1]) != len(want[1]))       # rows the op returns / detections that survive
    assert differs > 0                                                 # the options matter on these frames
    # the batched pipeline takes them from the same file: one stream against the plugin-built single pipeline
    from deepdish_amd.pipeline import HotPath
    mp = MultiStreamPipeline(2, model=path, wanted_labels=wanted)
    hp = HotPath(model=path, wanted_labels=wanted)
    for f, frame in enumerate(frames):
        fr = torch.from_numpy(frame).cuda()
        mp.step(torch.stack([fr, fr]))
        hp.step(fr)
        want_t = np.array([[t.track_id, t.state, t.time_since_update, t.hits, t.age] for t in hp.tracker.tracks], dtype=np.int64).reshape(-1, 5)
        for z in range(2):
            np.testing.assert_array_equal(mp.tracker(z).table()[0][:, :5], want_t, err_msg='frame %d stream %d' % (f, z))


def test_pipeline_loads_a_mars_tflite_file(tmp_path):
    """--encoder-model <file>.tflite in the batched pipeline (deepdish.py:505-510): the written encoder file gives the tracks of the named weights."""
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_writer
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.synth import Scene
    wd = nets.synthetic_mars_weights(1234)
    path = str(tmp_path / 'mars-64x32x3.tflite')
    tflite_writer.write_mars(wd, path)
    sc = Scene(seed=4, n_obj=6, n_frames=5)
    a = MultiStreamPipeline(1, encoder_model=path, run_detector=False)
    b = MultiStreamPipeline(1, run_detector=False)
    for f in range(5):
        boxes, scores, _, _ = sc.detections(f)
        one = ([tuple(int(v) for v in bb) for bb in boxes], ['person'] * len(boxes), [float(x) for x in scores])
        fr = torch.from_numpy(sc.frame(f)[None]).cuda()
        a.step(fr, a.pack_injected([one])); b.step(fr, b.pack_injected([one]))
        np.testing.assert_array_equal(a.tracker(0).table()[0], b.tracker(0).table()[0])
    assert len(a.tracker(0).table()[0]) > 0


def test_stage_gpu_ms_and_detections_getter():
    """dd_pipeline_stage_gpu_ms: every stage has GPU time, the host share and the wall time are consistent; dd_pipeline_detections returns
    the injected rows when detections are injected and the adaptor's own rows otherwise."""
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.synth import Scene
    sc = Scene(seed=4, n_obj=6, n_frames=5)
    wanted = sorted({l for l in _labels().values() if l and l != '???'})
    mp = MultiStreamPipeline(2, model=MODEL, wanted_labels=wanted)
    for f in range(4):
        boxes, scores, _, _ = sc.detections(f)
        one = ([tuple(int(v) for v in bb) for bb in boxes], ['person'] * len(boxes), [float(x) for x in scores])
        fr = torch.from_numpy(sc.frame(f)).cuda()
        mp.step(torch.stack([fr, fr]), mp.pack_injected([one, one]))
        b, l, s_ = mp.detections(1)
        assert l == ['person'] * len(boxes) and np.allclose(b, np.array(one[0], dtype=np.float64)) and np.allclose(s_, one[2])
    t = mp.stage_ms()
    assert t['steps'] == 4 and all(t[k] > 0 for k in ('objd', 'nms', 'feat', 'trak', 'host', 'wall'))
    assert t['host'] < t['wall'] and t['feat'] < t['wall'] and t['objd'] < 4 * t['wall']
    mp.step(torch.stack([fr, fr]))                                      # no injection: the detector's own rows, the same for both slots
    a, b = mp.detections(0), mp.detections(1)
    assert a[1] == b[1] and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
