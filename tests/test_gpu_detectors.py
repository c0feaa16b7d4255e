"""GPU: the detector adaptors' tails against fixtures the REFERENCE's own classes produced
(tests/golden/ssd_tail.npz, yolov5_tail.npz; scripts/make_golden_detectors.py) -- dd_nms_ssd, dd_ssd_detections
(csrc/post.hip ssd_finish_k, the kernel the batched C++ pipeline runs) and dd_yolov5_decode, all through the C ABI.
Everything here is bit-exact: the kernels do the reference's f64 / f32 operations in the reference's order."""
import ctypes
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), 'golden')
ASSETS = os.path.join(os.path.dirname(__file__), '..', 'deepdish_amd', 'assets')


def _lines(name):
    with open(os.path.join(ASSETS, name)) as f:
        return {i: line.strip() for i, line in enumerate(f.readlines())}


def _by_class(boxes, cls, scores):
    """rows grouped per class id, pick order kept inside a class (the class order itself is a Python-set artefact)."""
    out = {}
    for b, c, s in zip(boxes, cls, scores):
        out.setdefault(int(c), []).append((tuple(float(v) for v in b), float(s)))
    return out


def test_dd_nms_ssd_matches_reference_nms_boxes():
    """tools/ssd_mobilenet.py:59-98 -- 60 reference cases (1..39 boxes, 1..3 classes, thresholds 0.3/0.5/0.7, half of
    them with boxes a few pixels wide where the +1 on the intersection decides): kept rows identical, in order."""
    from deepdish_amd._lib import lib, check
    from deepdish_amd.runtime import default_context, ptr
    ctx = default_context()
    g = np.load(os.path.join(G, 'ssd_tail.npz'))
    for i in range(len(g['n_thr'])):
        a, b = g['n_off'][i], g['n_off'][i + 1]
        boxes, labels, scores = g['n_boxes'][a:b], g['n_cls'][a:b], g['n_scores'][a:b]
        got_b, got_c, got_s = [], [], []
        for c in set(labels):                                   # same scalars, same interpreter => the reference's class order
            idx = np.flatnonzero(labels == c)
            k = len(idx)
            db, ds_ = ctx.to_device(boxes[idx], np.float64), ctx.to_device(scores[idx], np.float64)
            out, cnt = ctx.empty((k,), torch.int32), ctx.empty((1,), torch.int32)
            check(lib().dd_nms_ssd(ctx.handle, ptr(db), ptr(ds_), k, float(g['n_thr'][i]), ptr(out), ptr(cnt), None), 'dd_nms_ssd')
            keep = ctx.to_host(out)[:int(ctx.to_host(cnt)[0])]
            got_b.append(boxes[idx][keep]); got_c.append(labels[idx][keep]); got_s.append(scores[idx][keep])
        ka, kb = g['k_off'][i], g['k_off'][i + 1]
        np.testing.assert_array_equal(np.concatenate(got_b), g['k_boxes'][ka:kb], err_msg=f'case {i}')
        np.testing.assert_array_equal(np.concatenate(got_c), g['k_cls'][ka:kb])
        np.testing.assert_array_equal(np.concatenate(got_s), g['k_scores'][ka:kb])


def test_dd_ssd_detections_matches_reference_predict_tail():
    """tools/ssd_mobilenet.py:111-150 -- 240 canned interpreter outputs (NaN boxes / scores, tiny boxes, low
    confidence, four image sizes), all images of one size in ONE launch: boxes (f64), class ids and scores identical
    to SSDMobileNet.predict's, class by class in pick order."""
    from deepdish_amd._lib import lib, check
    from deepdish_amd.runtime import default_context, ptr
    ctx = default_context()
    g = np.load(os.path.join(G, 'ssd_tail.npz'))
    sizes = sorted({tuple(int(v) for v in s) for s in g['size']})
    assert len(sizes) == 4
    checked = 0
    for size in sizes:
        sel = np.flatnonzero((g['size'] == np.array(size)).all(axis=1))
        n = len(sel)
        db, dc, ds_ = (ctx.to_device(g[k][sel], np.float32) for k in ('boxes', 'cls', 'scores'))
        ob, oc = ctx.empty((n, 10, 4), torch.float64), ctx.empty((n, 10), torch.int32)
        osc, on = ctx.empty((n, 10), torch.float64), ctx.empty((n,), torch.int32)
        check(lib().dd_ssd_detections(ctx.handle, ptr(db), ptr(dc), ptr(ds_), n, 10, 0.5, 0.5, float(size[0]), float(size[1]),
                                      ptr(ob), ptr(oc), ptr(osc), ptr(on), None), 'dd_ssd_detections')
        ob, oc, osc, on = (ctx.to_host(t) for t in (ob, oc, osc, on))
        for j, i in enumerate(sel):
            a, b = g['off'][i], g['off'][i + 1]
            assert on[j] == b - a, (i, on[j], b - a)
            want = _by_class(g['pred_boxes'][a:b], g['pred_cls'][a:b], g['pred_scores'][a:b])
            got = _by_class(ob[j, :on[j]], oc[j, :on[j]], osc[j, :on[j]].astype(np.float32))
            assert got == want, i
            assert list(oc[j, :on[j]]) == sorted(oc[j, :on[j]])         # ascending class id, as documented
            checked += int(on[j])
    assert checked == g['off'][-1] > 1000


def test_ssd_plugin_tail_matches_reference_detect_image():
    """SSDMobileNet.postprocess + SSD_MOBILENET._filter (the Python plugin, routed through dd_ssd_detections) ==
    the reference's detect_image on the same interpreter outputs: tlwh boxes, label names, scores."""
    from deepdish_amd.pipeline import make_detector
    g = np.load(os.path.join(G, 'ssd_tail.npz'))
    lines = _lines('coco_labels_ssd.txt')
    wanted = [str(x) for x in g['wanted_person']]
    det = make_detector('synthetic-ssd_mobilenet_v1.tflite', wanted_labels=wanted)
    name_to_id = {v: k - 1 for k, v in lines.items() if k > 0}
    for i in range(0, len(g['boxes']), 3):
        out = [g['boxes'][i].copy(), g['cls'][i].copy(), g['scores'][i].copy(), 10.0]
        rb, rl, rs = det._filter(*det.ssdm.postprocess(out, original_image_size=tuple(int(v) for v in g['size'][i])))
        a, b = g['poff'][i], g['poff'][i + 1]
        assert len(rs) == b - a, i
        got = _by_class(rb, [name_to_id[x] for x in rl], np.asarray(rs, np.float32))
        want = _by_class(g['pdet_boxes'][a:b], g['pdet_cls'][a:b], g['pdet_scores'][a:b])
        assert got == want, i


def test_dd_yolov5_decode_matches_reference_detect_image():
    """tools/yolov5.py:120-146 -- dd_yolov5_decode (xywh -> xyxy, cls *= obj, argmax, >= thr, scale) + the plugin's
    label filter / tlwh conversion on the reference's canned head tensors: identical boxes (f32), labels, scores."""
    from deepdish_amd.pipeline import make_detector
    g = np.load(os.path.join(G, 'yolov5_tail.npz'))
    lines = _lines('coco_classes.txt')
    name_to_id = {v: k for k, v in lines.items()}
    from deepdish_amd._lib import lib, check
    from deepdish_amd.runtime import ptr
    for i in range(len(g['thr'])):
        wanted = str(g['wanted'][i]).split(',')
        det = make_detector('synthetic-yolov5s-fp16.tflite', wanted_labels=wanted) if i == 0 else det
        det.wanted_labels, det.score_threshold = wanted, float(g['thr'][i])
        raw = g['raw_f16'][g['roff'][i]:g['roff'][i + 1]].astype(np.float32)
        W, H = (int(v) for v in g['size'][i])
        d_raw = det.ctx.to_device(raw)
        check(lib().dd_yolov5_decode(det.ctx.handle, ptr(d_raw), len(raw), 80, float(g['thr'][i]), float(W), float(H),
                                     ptr(det._boxes), ptr(det._scores), ptr(det._cls), det.MAX_ROWS, ptr(det._n), None),
              'dd_yolov5_decode')
        det.ctx.sync()
        n = int(det._n.cpu().numpy()[0])
        b, l, s = det._collect(det._boxes[:n].cpu().numpy(), det._scores[:n].cpu().numpy(), det._cls[:n].cpu().numpy())
        a, e = g['off'][i], g['off'][i + 1]
        assert len(s) == e - a, (i, len(s), e - a)
        np.testing.assert_array_equal(np.asarray(b, np.float32), g['boxes'][a:e])
        np.testing.assert_array_equal([name_to_id[x] for x in l], g['labels'][a:e])
        np.testing.assert_array_equal(np.asarray(s, np.float32), g['scores'][a:e])


@pytest.mark.parametrize('n,thr', [(3, 1e-8), (40, 0.3)])
def test_ssd_head_layers_decode_in_their_epilogue_with_the_same_bits(n, thr):
    """dd_net_ssd_decode: the six head GEMMs of the SSD run one anchor per 96-channel tile and do the first stage of
    TFLite_Detection_PostProcess on their accumulators (best class with the lowest index on ties, anchor decode, sigmoid,
    threshold) -- per anchor the bits of dd_ssd_decode on the head matrix of an engine without the switch, then the same
    top-10 from dd_ssd_postprocess_decoded as from dd_ssd_postprocess.  The head matrix of the decoding engine is never written:
    reading it is an error.  Frames 1 and 2 are flat (every class logit of an anchor column ties across pixels)."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    from deepdish_amd._lib import lib, check, DeepDishHipError
    from deepdish_amd.runtime import default_context, ptr
    ctx = default_context()
    wd = nets.synthetic_ssd_weights(1234)
    prog = nets.compile_ssd_mobilenet(wd)
    anchors = np.ascontiguousarray(prog.meta['anchors'], dtype=np.float32)
    A, C = len(anchors), prog.meta['n_classes']
    rng = np.random.default_rng(n)
    x = rng.integers(0, 256, (n, 300, 300, 3), dtype=np.uint8)
    if n > 2:
        x[1] = 0; x[2] = 255
    plain = Net(prog, max_batch=n, context=ctx)
    plain.forward(x)
    d_anchors = ctx.to_device(anchors, np.float32)
    want = [ctx.empty((n, A, 4), torch.float32), ctx.empty((n, A), torch.float32), ctx.empty((n, A), torch.int32), ctx.empty((n, A), torch.float32)]
    check(lib().dd_ssd_decode(ctx.handle, plain.output_ptr(), ptr(d_anchors), A, C, float(thr), *[ptr(t) for t in want], n, None), 'dd_ssd_decode')
    want = [ctx.to_host(t) for t in want]
    fused = Net(nets.compile_ssd_mobilenet(wd), max_batch=n, context=ctx)
    fused.ssd_decode(anchors, thr)
    fused.forward(x)
    got = fused.ssd_decoded()
    for name, g, w in zip(('boxes', 'scores', 'classes', 'keys'), got, want):
        np.testing.assert_array_equal(g, w, err_msg=name)
    assert np.isfinite(got[0]).all() and (got[3] >= 0).any() and ((got[3] < 0).any() or thr < 1e-6)
    with pytest.raises(DeepDishHipError, match='head matrix'):
        fused.read()
    # second stage from the decoded arrays == the whole op from the head matrix, frame by frame
    ps = [ctypes.c_void_p() for _ in range(4)]
    check(lib().dd_net_ssd_decoded(fused._h, *[ctypes.byref(p) for p in ps]), 'dd_net_ssd_decoded')
    ob, oc, os_, on = ctx.empty((n, 10, 4), torch.float32), ctx.empty((n, 10), torch.float32), ctx.empty((n, 10), torch.float32), ctx.empty((n,), torch.int32)
    check(lib().dd_ssd_postprocess_decoded(ctx.handle, *ps, A, 10, float(thr), 0.6, ptr(ob), ptr(oc), ptr(os_), ptr(on), n, None),
          'dd_ssd_postprocess_decoded')
    ob, oc, os_, on = (ctx.to_host(t) for t in (ob, oc, os_, on))
    raw = plain.output_ptr()
    stride = A * (4 + C) * 4
    for z in range(n):
        b1, c1, s1, n1 = ctx.empty((10, 4), torch.float32), ctx.empty((10,), torch.float32), ctx.empty((10,), torch.float32), ctx.empty((1,), torch.int32)
        check(lib().dd_ssd_postprocess(ctx.handle, ctypes.c_void_p(raw + z * stride), ptr(d_anchors), A, C, 10, float(thr), 0.6, ptr(b1), ptr(c1),
                                       ptr(s1), ptr(n1), None), 'dd_ssd_postprocess')
        k = int(ctx.to_host(n1)[0])
        assert k == on[z]
        np.testing.assert_array_equal(ctx.to_host(b1)[:k], ob[z, :k])
        np.testing.assert_array_equal(ctx.to_host(c1)[:k], oc[z, :k])
        np.testing.assert_array_equal(ctx.to_host(s1)[:k], os_[z, :k])
    # switching it off brings the head matrix back
    fused.ssd_decode(anchors, thr, enable=False)
    fused.forward(x)
    np.testing.assert_array_equal(fused.read(), plain.read())


@pytest.mark.parametrize('n', [2, 5])
def test_yolo_heads_reduce_rows_in_their_epilogue(n):
    """dd_net_yolo_decode: the Detect layers write (box, confidence, class) per row instead of the [rows][85] matrix.  Against the
    matrix of the same forward with the switch off: boxes are its first four columns bit for bit, confidence and class what
    tools/yolov5.py:126-128 computes from it (numpy, f32) -- incl. rows whose class products tie, and a NaN row."""
    import torch
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    wd = nets.synthetic_yolov5s_weights()
    prog = nets.compile_yolov5s(wd)
    net = Net(prog, max_batch=n)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (n, 640, 640, 3), dtype=np.uint8)
    x[1] = 128                                                  # a flat frame: many equal rows
    xd = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    net.forward(xd)
    raw = net.read()[:, :, 0, :] if net.read().ndim == 4 else net.read()
    raw = np.asarray(raw, dtype=np.float32).reshape(n, -1, 85)
    net.yolo_decode(True)
    net.forward(xd)
    boxes, conf, cls = net.yolo_decoded()
    with pytest.raises(Exception):
        net.read()
    np.testing.assert_array_equal(boxes, raw[:, :, :4])
    prod = raw[:, :, 5:] * raw[:, :, 4:5]                       # x[..., 5:] *= x[..., 4:5]
    want_cls = np.argmax(prod, axis=-1)
    want_conf = np.take_along_axis(prod, want_cls[..., None], axis=-1)[..., 0]
    np.testing.assert_array_equal(cls, want_cls.astype(np.int32))
    np.testing.assert_array_equal(conf, want_conf)
    net.yolo_decode(False)
    net.forward(xd)
    np.testing.assert_array_equal(np.asarray(net.read(), dtype=np.float32).reshape(n, -1, 85), raw)


def test_yolov5_plugin_loads_a_tflite_file(tmp_path):
    """YOLOV5(model_file='.../yolov5s-fp16.tflite') as deepdish.py:482-491 constructs it: the detector written to disk in the interchange
    format (export-shaped graph, float16 filters behind DEQUANTIZE) gives, bit for bit, the detections of the same weights handed over as
    named arrays (tools/tflite_reader.load_yolov5s), also through the batched pipeline."""
    import torch
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_writer
    from deepdish_amd.tools.yolov5 import YOLOV5
    from deepdish_amd.pipeline import DEFAULT_YOLO_LABELS
    from deepdish_amd.synth import Scene
    wd = nets.synthetic_yolov5s_weights(1234)
    path = str(tmp_path / 'yolov5s-fp16.tflite')
    tflite_writer.write_yolov5s(wd, path, half_weights=True)
    wanted = sorted({l.strip() for l in open(DEFAULT_YOLO_LABELS)})
    a = YOLOV5(wanted_labels=wanted, model_file=path, label_file=DEFAULT_YOLO_LABELS)
    b = YOLOV5(wanted_labels=wanted, model_file='synthetic-yolov5s', label_file=DEFAULT_YOLO_LABELS)
    frame = Scene(seed=3, n_obj=8).frame(0)
    ra = a.detect_frame_device(torch.from_numpy(frame).cuda(), 480, 640)
    rb = b.detect_frame_device(torch.from_numpy(frame).cuda(), 480, 640)
    assert ra[1] == rb[1] and len(ra[1]) > 0
    np.testing.assert_array_equal(np.asarray(ra[0]), np.asarray(rb[0]))
    np.testing.assert_array_equal(np.asarray(ra[2]), np.asarray(rb[2]))


def test_ssd_select_nms_matches_the_oracle_on_crafted_candidates():
    """csrc/nms.hip nms_greedy_f32_k (no sort: max_detections rounds of a workgroup-wide arg-max over the live candidates + one IoU each) behind
    dd_ssd_postprocess_decoded against oracle/nets_torch.ssd_postprocess_decoded (NonMaxSuppressionMultiClassFastHelper restated):
    0 .. 1917 candidates, ties, NaN / degenerate boxes, max_detections 10 .. 64, chains that cross many 64-row chunks.  Bit-exact."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'scripts'))
    from ssd_post_cases import cases, run_device
    from oracle import nets_torch
    cs = cases()
    got = run_device(cs)
    crossed = 0
    for c, (gb, gc, gs, gn) in zip(cs, got):
        ob, oc, os_, on = nets_torch.ssd_postprocess_decoded(c['boxes'], c['score'], c['cls'], c['max_det'], c['thr'], c['iou'])
        assert gn == on, (gn, on, c['max_det'], c['thr'])
        np.testing.assert_array_equal(gb[:on], ob[:on])
        np.testing.assert_array_equal(gc[:on], oc[:on])
        np.testing.assert_array_equal(gs[:on], os_[:on])
        assert not gb[on:].any() and not gs[on:].any()
        crossed += on < c['max_det'] and (c['score'] >= np.float32(c['thr'])).sum() > 128
    assert crossed >= 2                                  # some cases end by exhausting several chunks, not by reaching max_detections
