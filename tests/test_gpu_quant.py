"""GPU: the uint8 SSD-MobileNet-v1 program (csrc/netsq.hip) against the integer restatement of TFLite's reference
kernels (oracle/nets_quant.py).  Integer work: the bar is bit-exact, tensor by tensor, borders included.
(Parity against the real ssdmobilenetv1.tflite is unpinned: blob and tflite_runtime absent.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHECK = ['conv0', 'dw1', 'pw1', 'dw2', 'pw2', 'pw3', 'dw4', 'pw4', 'pw5', 'pw6', 'dw7', 'pw7', 'pw11', 'dw12', 'pw12', 'dw13', 'pw13',
         'extra1_1', 'extra1_2', 'extra2_2', 'extra3_2', 'extra4_1', 'extra4_2']


def _frames(n, seed):
    from deepdish_amd import quantize
    fr = quantize.calibration_frames(n, seed=seed)
    rng = np.random.default_rng(seed)
    if n > 1:
        fr[1] = rng.integers(0, 256, fr[1].shape, dtype=np.uint8)      # white noise: every clamp and both signs of every rounding
    return fr


@pytest.fixture(scope='module', params=['asymmetric', 'symmetric'])
def qnet(request):
    from deepdish_amd import quantize, netsq
    from deepdish_amd.engine import Net
    qm = quantize.synthetic_ssd_quant_model(1234, symmetric_weights=request.param == 'symmetric')
    prog = netsq.compile_ssd_mobilenet_quant(qm)
    return qm, prog, Net(prog, max_batch=8)


def _tensor_of(prog, name):
    """Program tensor written by layer `name` (ops are emitted in the model's order; a fused block writes only its pointwise output)."""
    from deepdish_amd import netsq
    names = ['conv0']
    for i in range(1, 14):
        names += [f'dw{i}', f'pw{i}']
    for j in range(1, 5):
        names += [f'extra{j}_1', f'extra{j}_2']
    it = iter(names)
    for o in prog.ops:
        if o[0] in (netsq.OP_QCONV0, netsq.OP_QCONV, netsq.OP_QDW):
            if next(it) == name:
                return int(o[2])
        elif o[0] == netsq.OP_QDWPW:
            d, p = next(it), next(it)
            if p == name:
                return int(o[2])
            if d == name:
                return None
    return None


def test_every_layer_is_bit_exact(qnet):
    from deepdish_amd import netsq
    from oracle import nets_quant
    qm, prog, net = qnet
    fr = _frames(3, 11)
    fr[2] = 0
    net.forward(fr)
    box_w, cls_w, kept = nets_quant.ssd_quant_forward(qm, fr, keep=CHECK)
    for name in CHECK:
        t = _tensor_of(prog, name)
        if t is None:
            continue                                                       # the depthwise output of a fused block never leaves the CU
        d = prog.tensors[t]
        raw = net.read(tensor=t)
        got = netsq.unpack_q16(raw, d['h'], d['w'], d['c'])
        assert got.shape == kept[name].shape, name
        bad = int((got != kept[name]).sum())
        assert bad == 0, '%s: %d of %d bytes differ (max |diff| %d)' % (name, bad, got.size, int(np.abs(got.astype(int) - kept[name]).max()))
        assert (netsq.borders_q16(raw, d['h'], d['w'], d['c']) == d['zp']).all(), name + ': border overwritten'
    box = net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :]
    cls = net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']]
    np.testing.assert_array_equal(box, box_w)
    np.testing.assert_array_equal(cls, cls_w)
    assert len(np.unique(cls)) > 100 and len(np.unique(box)) > 50          # not a degenerate comparison


def test_front_row_pipeline_is_bit_exact(qnet, monkeypatch):
    """First layer + blocks 1 and 2 as ONE launch, blocks 3 and 4 as another (csrc/netsq_front.hip, csrc/netsq_mid.hip: the default from 24
    frames per forward; forced here): block 2's
    tensor and everything behind it against the oracle, borders included; the two tensors the launch keeps in LDS are not readable; the
    three-launch form gives the same bytes.  Frame counts that end a workgroup's row range inside a frame and at a frame's edge."""
    from deepdish_amd import netsq
    from deepdish_amd.profile import net_op_launches
    from oracle import nets_quant
    qm, prog, net = qnet
    fr = _frames(7, 17)
    fr[2] = 0
    fr[5] = 255
    want_box, want_cls, kept = nets_quant.ssd_quant_forward(qm, fr, keep=['pw2', 'pw4', 'pw13'])
    monkeypatch.setenv('DD_Q_FRONT_MIN', '1')
    for n in (7, 1, 2, 3):
        net.forward(fr[:n])
        codes = net_op_launches(net)
        assert list(codes[:5]) == [1, 1, 19, 1, 20], codes[:5]          # ... and blocks 3 + 4 as one launch behind it (csrc/netsq_mid.hip)
        for name in ('pw2', 'pw4', 'pw13'):
            t = _tensor_of(prog, name)
            d = prog.tensors[t]
            raw = net.read(tensor=t)
            got = netsq.unpack_q16(raw, d['h'], d['w'], d['c'])
            bad = int((got != kept[name][:n]).sum())
            assert bad == 0, '%d frames, %s: %d of %d bytes differ' % (n, name, bad, got.size)
            assert (netsq.borders_q16(raw, d['h'], d['w'], d['c']) == d['zp']).all(), name + ': border overwritten'
        np.testing.assert_array_equal(net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :], want_box[:n])
        np.testing.assert_array_equal(net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']], want_cls[:n])
    with pytest.raises(Exception):
        net.read(tensor=_tensor_of(prog, 'conv0'))                         # never written by this forward
    with pytest.raises(Exception):
        net.read(tensor=_tensor_of(prog, 'pw1'))
    with pytest.raises(Exception):
        net.read(tensor=_tensor_of(prog, 'pw3'))
    monkeypatch.setenv('DD_Q_MID', '0')                                    # the front end as one launch, blocks 3 and 4 as two
    net.forward(fr)
    assert list(net_op_launches(net)[:5]) == [1, 1, 19, 0, 0]
    np.testing.assert_array_equal(net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :], want_box)
    d3 = prog.tensors[_tensor_of(prog, 'pw3')]
    assert netsq.unpack_q16(net.read(tensor=_tensor_of(prog, 'pw3')), d3['h'], d3['w'], d3['c']).shape[0] == 7
    monkeypatch.setenv('DD_Q_FRONT', '0')
    monkeypatch.delenv('DD_Q_MID')                                         # ... and the other way round
    net.forward(fr)
    assert list(net_op_launches(net)[:5]) == [0, 0, 0, 1, 20]
    np.testing.assert_array_equal(net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :], want_box)
    np.testing.assert_array_equal(net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']], want_cls)
    monkeypatch.setenv('DD_Q_MID', '0')
    net.forward(fr)
    assert list(net_op_launches(net)[:5]) == [0, 0, 0, 0, 0]
    np.testing.assert_array_equal(net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :], want_box)
    np.testing.assert_array_equal(net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']], want_cls)
    t = _tensor_of(prog, 'pw1')
    assert net.read(tensor=t).shape[0] == 7


@pytest.mark.parametrize('variant', ['own_clamp', 'long_shift'])
def test_row_pipelines_with_other_requantisation_parameters(variant, monkeypatch):
    """The front end and blocks 3 + 4 (csrc/netsq_front.hip, netsq_mid.hip) pick their requantise-and-pack form from the layers' parameters:
    byte-range clamps with shifts <= 8 (the synthetic model as it comes: every other test), byte-range clamps with longer shifts
    ('long_shift': the weight scales of the first nine layers divided by 16, their biases multiplied by 16), and clamps of their own ('own_clamp': zero point 6 and a
    coarser scale on the tensors between those layers, so lo = 6 and hi = 206).  Same integers as the oracle in each, fused and unfused."""
    import copy
    from deepdish_amd import quantize, netsq
    from deepdish_amd.engine import Net
    from deepdish_amd.profile import net_op_launches
    from oracle import nets_quant
    qm = copy.deepcopy(quantize.synthetic_ssd_quant_model(1234))
    chain = ['conv0', 'dw1', 'pw1', 'dw2', 'pw2', 'dw3', 'pw3', 'dw4', 'pw4', 'dw5']
    for a, b in zip(chain[:-1], chain[1:]):
        La, Lb = qm['layers'][a], qm['layers'][b]
        if variant == 'own_clamp':
            La['out_zp'], La['out_scale'] = 6, np.float32(6.0 / 200.0)
            Lb['in_zp'], Lb['in_scale'] = 6, np.float32(6.0 / 200.0)
        else:                                                               # (biases scaled up with it, so that the tensors stay lively)
            La['w_scale'] = np.float32(La['w_scale'] / 16.0)
            La['bias'] = (La['bias'].astype(np.int64) * 16).astype(np.int32)
    lo, hi = quantize.activation_range(qm['layers']['pw2'])
    m, shift = quantize.conv_multiplier(qm['layers']['dw2'])
    assert ((lo, hi) == (6, 206)) if variant == 'own_clamp' else ((lo, hi) == (0, 255) and -shift > 8)
    prog = netsq.compile_ssd_mobilenet_quant(qm)
    net = Net(prog, max_batch=4)
    fr = _frames(3, 29)
    want_box, want_cls, kept = nets_quant.ssd_quant_forward(qm, fr, keep=['pw2', 'pw4'])
    monkeypatch.setenv('DD_Q_FRONT_MIN', '1')
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv('DD_Q_FRONT', '0')
            monkeypatch.setenv('DD_Q_MID', '0')
        net.forward(fr)
        assert list(net_op_launches(net)[:5]) == ([1, 1, 19, 1, 20] if fused else [0, 0, 0, 0, 0])
        for name in ('pw2', 'pw4'):
            t = _tensor_of(prog, name)
            d = prog.tensors[t]
            raw = net.read(tensor=t)
            got = netsq.unpack_q16(raw, d['h'], d['w'], d['c'])
            assert int((got != kept[name]).sum()) == 0, (variant, fused, name)
            assert (netsq.borders_q16(raw, d['h'], d['w'], d['c']) == d['zp']).all(), name
        np.testing.assert_array_equal(net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :], want_box)
        np.testing.assert_array_equal(net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']], want_cls)
    assert len(np.unique(kept['pw4'])) > 8                                  # not a degenerate comparison


def test_results_do_not_depend_on_the_launch(qnet):
    qm, prog, net = qnet
    fr = _frames(5, 3)
    net.forward(fr)
    cls = net.read(tensor=prog.meta['cls_tensor']).copy()
    box = net.read(tensor=prog.meta['box_tensor']).copy()
    net.forward(fr[3:4])
    np.testing.assert_array_equal(net.read(tensor=prog.meta['cls_tensor']), cls[3:4])
    np.testing.assert_array_equal(net.read(tensor=prog.meta['box_tensor']), box[3:4])


def test_decoded_arrays(qnet):
    """dd_net_ssd_decode on a uint8 program: the post-process op's first stage from the quantised head tensors."""
    from oracle import nets_quant
    qm, prog, net = qnet
    fr = _frames(2, 5)
    net.ssd_decode(nets_quant.ssd_anchors(), 0.3)
    try:
        net.forward(fr)
        boxes, score, cls, keys = net.ssd_decoded()
        box_q = net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :]
        cls_q = net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']]
        for z in range(2):
            wb, ws, wc, wk = nets_quant.ssd_quant_decode(qm, box_q[z], cls_q[z], nets_quant.ssd_anchors(), 0.3)
            np.testing.assert_array_equal(score[z], ws)
            np.testing.assert_array_equal(cls[z], wc)
            np.testing.assert_array_equal(keys[z], wk)
            np.testing.assert_allclose(boxes[z], wb, rtol=0, atol=2e-6)        # expf: device vs numpy, a few ulp of values <= 2
        assert (keys >= 0).sum() > 10
    finally:
        net.ssd_decode(None, 0.0, enable=False)


@pytest.mark.parametrize('n_frames, symmetric', [(1536, False), (768, False), (384, False), (91, False), (91, True), (23, False)])
def test_launch_of_many_frames_is_bit_exact(n_frames, symmetric):
    """The bench's launch shape (one worker group = 1 536 frames per forward since round 6, 768 in round 5, 384 before) and odd ones: picked slots against the integer oracle, every
    other slot against the slot that holds the same frame (persistent blocks, tiles that straddle frames, the last tile of a frame).
    91 frames: the register-filter pointwise kernel (from 8 192 pixels per launch) with a partly filled last tile on every map it takes,
    merged predictors included; 23 frames: the same layers on the generic kernel, the 19x19 predictors still merged (8 303 pixels)."""
    from deepdish_amd import quantize, netsq
    from deepdish_amd.engine import Net
    from oracle import nets_quant
    qm = quantize.synthetic_ssd_quant_model(1234, symmetric_weights=symmetric)
    prog = netsq.compile_ssd_mobilenet_quant(qm)
    net = Net(prog, max_batch=n_frames)
    base = _frames(4, 21)
    idx = np.arange(n_frames) % 4
    idx[[0, 1, n_frames // 2, n_frames - 1]] = [3, 2, 1, 0]
    net.forward(base[idx])
    from deepdish_amd.profile import net_op_launches
    assert list(net_op_launches(net)[:5]) == ([1, 1, 19, 1, 20] if n_frames >= 24 else [0, 0, 0, 0, 0])     # the front end and blocks 3 + 4 as one launch each from 24 frames (csrc/netsq_front.hip, netsq_mid.hip)
    box = net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :]
    cls = net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']]
    box_w, cls_w, _ = nets_quant.ssd_quant_forward(qm, base)
    for z in range(n_frames):
        assert (box[z] == box_w[idx[z]]).all() and (cls[z] == cls_w[idx[z]]).all(), 'slot %d (frame %d)' % (z, idx[z])


@pytest.mark.parametrize('zero_point_of', ['box', 'cls'])
def test_merged_predictors_with_mixed_weight_zero_points(zero_point_of):
    """The class and box predictors of a feature map run as ONE launch -- of q_pws_k from 8 192 pixels per launch, of q_conv_k below (round 6;
    csrc/netsq.hip) -- and each fragment picks its predictor's requantisation, destination rows and weight zero point.  Here only ONE of the
    two has a weight zero point other than 128 (so exactly one of them needs the activation row sums): 23 frames put the 19x19 map on q_pws_k
    and the five smaller maps on q_conv_k; four frames put all six on q_conv_k.  Same integers as the oracle, slot by slot."""
    import copy
    from deepdish_amd import quantize, netsq
    from deepdish_amd.engine import Net
    from oracle import nets_quant
    qm = copy.deepcopy(quantize.synthetic_ssd_quant_model(1234))
    changed = 0
    for name, L in qm['layers'].items():
        if name.startswith(zero_point_of) and name[len(zero_point_of):].isdigit():
            L['w_zp'] = 128
            changed += 1
    assert changed == 6
    prog = netsq.compile_ssd_mobilenet_quant(qm)
    base = _frames(4, 33)
    box_w, cls_w, _ = nets_quant.ssd_quant_forward(qm, base)
    for n_frames in (23, 4):
        net = Net(prog, max_batch=n_frames)
        idx = np.arange(n_frames) % 4
        net.forward(base[idx])
        box = net.read(tensor=prog.meta['box_tensor'])[:, :, 0, :]
        cls = net.read(tensor=prog.meta['cls_tensor'])[:, :, 0, :prog.meta['n_classes']]
        for z in range(n_frames):
            assert (box[z] == box_w[idx[z]]).all() and (cls[z] == cls_w[idx[z]]).all(), '%d frames, slot %d (frame %d)' % (n_frames, z, idx[z])
    assert len(np.unique(cls_w)) > 8 and len(np.unique(box_w)) > 8


def test_plugin_loads_a_tflite_file(tmp_path):
    """SSD_MOBILENET(model_file='....tflite') as deepdish.py:491-495 constructs it: a uint8 model written to disk in the interchange format
    gives the detections of the same model handed over in memory."""
    import os
    from deepdish_amd import quantize
    from deepdish_amd.tools import tflite_writer
    from deepdish_amd.tools.ssd_mobilenet import SSD_MOBILENET
    from deepdish_amd.pipeline import DEFAULT_LABELS
    path = str(tmp_path / 'ssdmobilenetv1.tflite')
    tflite_writer.write_ssd_mobilenet(quantize.synthetic_ssd_quant_model(1234), path)
    a = SSD_MOBILENET(wanted_labels=['person', 'car', 'bicycle'], model_file=path, label_file=DEFAULT_LABELS, num_threads=4, edgetpu=False, score_threshold=0.0)
    b = SSD_MOBILENET(wanted_labels=['person', 'car', 'bicycle'], model_file='synthetic-ssd_mobilenet_v1-uint8', label_file=DEFAULT_LABELS, score_threshold=0.0)
    assert a.ssdm.quantized and b.ssdm.quantized and (a.width, a.height) == (300, 300)
    rng = np.random.default_rng(3)
    for _ in range(3):
        img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
        ra, rb = a.detect_image(img), b.detect_image(img)
        assert ra[1] == rb[1] and len(ra[0]) == len(rb[0])
        np.testing.assert_array_equal(np.asarray(ra[0], np.float64), np.asarray(rb[0], np.float64))
        np.testing.assert_array_equal(np.asarray(ra[2], np.float64), np.asarray(rb[2], np.float64))
