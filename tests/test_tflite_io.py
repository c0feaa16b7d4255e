"""CPU: the .tflite reader (deepdish_amd/tools/tflite_reader.py) -- what makes `--model <file>.tflite` a drop-in
(tools/ssd_mobilenet.py:31-52 upstream).  None of the reference's blobs is in the tree (.MISSING_LARGE_BLOBS): the reader is held
to a committed fixture whose values are written down in scripts/make_tflite_fixture.py, and to round trips through the writer."""
import os
import numpy as np
import pytest


def test_committed_fixture_parses_to_the_written_values(golden_dir):
    from deepdish_amd.tools import tflite_reader as R
    g = R.read(os.path.join(golden_dir, 'tiny_quant_graph.tflite'))
    assert g.description == 'tiny fixture' and g.inputs == [0] and [op.kind for op in g.ops] == ['CONV_2D', 'DEPTHWISE_CONV_2D', 'LOGISTIC', 'CUSTOM']
    t = {x.name: x for x in g.tensors}
    assert t['input'].shape == [1, 8, 8, 3] and t['input'].dtype == np.uint8 and t['input'].scale[0] == np.float32(0.0078125) and t['input'].zero_point[0] == 128
    np.testing.assert_array_equal(t['conv/w'].data, (np.arange(108) % 251).astype(np.uint8).reshape(4, 3, 3, 3))
    np.testing.assert_array_equal(t['conv/b'].data, np.array([-7, 0, 11, 123456], np.int32))
    assert t['conv/w'].zero_point[0] == 131 and t['conv/w'].scale[0] == np.float32(0.02) and t['conv'].data is None
    np.testing.assert_array_equal(t['dw/w'].data, (200 - np.arange(36)).astype(np.uint8).reshape(1, 3, 3, 4))
    assert g.ops[0].options == dict(padding='SAME', stride_w=2, stride_h=2, act='relu6', dilation_w=1, dilation_h=1)
    assert g.ops[1].options['depth_multiplier'] == 1 and g.ops[1].options['act'] == 'none' and g.ops[1].options['stride_w'] == 1
    assert g.ops[3].custom == 'TFLite_Detection_PostProcess'
    o = g.ops[3].options
    assert o['max_detections'] == 10 and o['num_classes'] == 90 and o['use_regular_nms'] is False
    assert o['y_scale'] == 10.0 and o['h_scale'] == 5.0 and abs(o['nms_iou_threshold'] - 0.6) < 1e-7
    np.testing.assert_array_equal(t['anchors'].data, np.array([[0.5, 0.5, 0.1, 0.2], [0.25, 0.75, 1.0, 1.0]], np.float32))
    assert g.made_by(t['scores'].index).kind == 'LOGISTIC'


def test_a_graph_that_is_not_the_detector_is_refused_by_name(golden_dir):
    from deepdish_amd.tools import tflite_reader as R
    with pytest.raises(R.UnsupportedModel) as e:
        R.load_ssd_mobilenet(os.path.join(golden_dir, 'tiny_quant_graph.tflite'))
    assert 'CONCATENATION' in str(e.value) or 'LOGISTIC' in str(e.value)
    with pytest.raises(ValueError):
        R.read(b'\x10\x00\x00\x00XXXX' + bytes(64))             # wrong file identifier


@pytest.fixture(scope='module')
def qmodel():
    from deepdish_amd import quantize
    return quantize.synthetic_ssd_quant_model(1234)


def test_uint8_detector_round_trip(tmp_path, qmodel):
    """QModel -> .tflite -> QModel: every array and parameter, the anchors, and the compiled program are the same."""
    from deepdish_amd import netsq, nets
    from deepdish_amd.tools import tflite_writer, tflite_reader
    from deepdish_amd.tools.weights_io import load_ssd_model
    path = str(tmp_path / 'ssdmobilenetv1.tflite')
    tflite_writer.write_ssd_mobilenet(qmodel, path)
    kind, qm2 = load_ssd_model(path)                                # what SSD_MOBILENET(model_file=...) calls
    assert kind == 'uint8' and qm2['input'] == qmodel['input'] and qm2['logistic'] == qmodel['logistic']
    for name in qmodel['order']:
        a, b = qmodel['layers'][name], qm2['layers'][name]
        for k in ('w', 'bias'):
            np.testing.assert_array_equal(a[k], b[k], err_msg=name)
        for k in ('w_scale', 'w_zp', 'in_scale', 'in_zp', 'out_scale', 'out_zp', 'stride', 'act', 'kind'):
            assert a[k] == b[k], (name, k)
    np.testing.assert_array_equal(qm2['anchors'], nets.ssd_anchors(300)[0])
    p1, p2 = netsq.compile_ssd_mobilenet_quant(qmodel), netsq.compile_ssd_mobilenet_quant(qm2)
    assert bytes(p1.blob) == bytes(p2.blob) and (p1.serialize()[0] == p2.serialize()[0]).all()


def test_float_detector_round_trip(tmp_path):
    from deepdish_amd import nets, quantize
    from deepdish_amd.tools import tflite_writer
    from deepdish_amd.tools.weights_io import load_ssd_model
    wd = nets.synthetic_ssd_weights(7)
    folded = {}
    for name, kind, w, b, stride, act in quantize.folded_ssd_layers(wd):
        folded[name + '/weights'] = w if kind == 'conv' else w[:, :, :, None]
        folded[name + '/biases'] = b
    path = str(tmp_path / 'ssd_float.tflite')
    tflite_writer.write_ssd_mobilenet(folded, path)
    kind, wd2 = load_ssd_model(path)
    assert kind == 'f32'
    assert bytes(nets.compile_ssd_mobilenet(wd2).blob) == bytes(nets.compile_ssd_mobilenet(wd).blob)


def test_unsupported_graphs_name_the_operator(tmp_path, qmodel):
    """Per-channel filters and a missing predictor are errors that say where."""
    import copy
    from deepdish_amd.tools import tflite_writer, tflite_reader as R
    W = tflite_writer.ssd_mobilenet_graph(qmodel)
    t = next(x for x in W.tensors if x['name'] == 'pw3/weights')
    t['scale'] = np.full(128, 0.01, np.float32); t['zero_point'] = np.zeros(128, np.int64)
    p = str(tmp_path / 'perchannel.tflite')
    open(p, 'wb').write(W.tobytes())
    with pytest.raises(R.UnsupportedModel) as e:
        R.load_ssd_mobilenet(p)
    assert 'per-channel' in str(e.value) and 'CONV_2D' in str(e.value)
    W = tflite_writer.ssd_mobilenet_graph(qmodel)
    W.ops[-1]['custom_options'] = __import__('deepdish_amd.tools.flatbuf', fromlist=['x']).flex_build_map(dict(y_scale=8.0, x_scale=10.0, h_scale=5.0, w_scale=5.0))
    p = str(tmp_path / 'scales.tflite')
    open(p, 'wb').write(W.tobytes())
    with pytest.raises(R.UnsupportedModel) as e:
        R.load_ssd_mobilenet(p)
    assert 'y_scale' in str(e.value)


def test_post_process_options_are_read_or_refused(tmp_path, qmodel):
    """TFLite_Detection_PostProcess runs with what the file says (tools/ssd_mobilenet.py:100-109 upstream: the interpreter does): the
    options that are built are handed on, every other one is refused by name -- never ignored."""
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_writer, tflite_reader
    from deepdish_amd.tools.weights_io import load_ssd_model, ssd_post_options
    path = str(tmp_path / 'ssd.tflite')
    tflite_writer.write_ssd_mobilenet(qmodel, path, post=dict(max_detections=20, nms_iou_threshold=0.5, nms_score_threshold=0.25))
    kind, qm = load_ssd_model(path)
    post = ssd_post_options(qm)
    assert kind == 'uint8' and post['max_detections'] == 20 and abs(post['nms_iou_threshold'] - 0.5) < 1e-7 and abs(post['nms_score_threshold'] - 0.25) < 1e-7
    assert ssd_post_options(qmodel) == dict(max_detections=10, nms_score_threshold=1e-8, nms_iou_threshold=0.6)     # not from a file: the stock export's
    fpath = str(tmp_path / 'ssd_f32.tflite')
    wd = nets.synthetic_ssd_weights(7)
    from deepdish_amd import quantize
    folded = {}
    for name, kind, w, b, stride, act in quantize.folded_ssd_layers(wd):
        folded[name + '/weights'] = w if kind == 'conv' else w[:, :, :, None]
        folded[name + '/biases'] = b
    tflite_writer.write_ssd_mobilenet(folded, fpath, post=dict(max_detections=5))
    kind, wf = load_ssd_model(fpath)
    assert kind == 'f32' and ssd_post_options(wf)['max_detections'] == 5
    nets.compile_ssd_mobilenet(wf)                                  # the options ride along without disturbing the compiler
    for bad, word in ((dict(use_regular_nms=True), 'use_regular_nms'), (dict(y_scale=None), 'y_scale'), (dict(w_scale=4.0), 'w_scale'),
                      (dict(max_classes_per_detection=3), 'max_classes_per_detection'), (dict(num_classes=80), 'num_classes'),
                      (dict(max_detections=100), 'max_detections'), (dict(nms_iou_threshold=None), 'nms_iou_threshold'),
                      (dict(nms_iou_threshold=0.0), 'nms_iou_threshold')):
        tflite_writer.write_ssd_mobilenet(qmodel, path, post=bad)
        with pytest.raises(tflite_reader.UnsupportedModel) as e:
            load_ssd_model(path)
        assert word in str(e.value), (bad, str(e.value))


def test_model_properties_raise_unsupported_model_not_assert(qmodel):
    import copy
    from deepdish_amd import netsq
    from deepdish_amd.tools.tflite_reader import UnsupportedModel
    qm = copy.deepcopy(qmodel)
    qm['layers']['cls3']['out_zp'] += 1                            # the six class tensors are concatenated: one (scale, zero point)
    with pytest.raises(UnsupportedModel):
        netsq.compile_ssd_mobilenet_quant(qm)
    qm = copy.deepcopy(qmodel)
    qm['anchors'] = np.zeros((100, 4), np.float32)
    with pytest.raises(UnsupportedModel):
        netsq.compile_ssd_mobilenet_quant(qm)


def test_other_model_files_fail_loudly(tmp_path):
    from deepdish_amd.tools.weights_io import load_named_weights
    from deepdish_amd import nets
    p = tmp_path / 'mars-64x32x3.pb'
    p.write_bytes(b'\0' * 16)
    with pytest.raises(FileNotFoundError) as e:
        load_named_weights(str(p), nets.synthetic_mars_weights)
    assert 'frozen' in str(e.value) or '.pb' in str(e.value)


def test_mars_encoder_round_trip(tmp_path):
    """create_box_encoder('.../mars-64x32x3.tflite') (generate_detections.py:151-162 upstream): named weights -> file -> named weights
    gives the same program bytes as the weights handed over in memory -- batch norms folded where the converter folds them, the block
    batch norms as scale / shift, the fully connected layer transposed back, the channel reversal noticed."""
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_reader as R, tflite_writer as Wr
    from deepdish_amd.tools.weights_io import load_mars_weights
    wd = nets.synthetic_mars_weights(11)
    p = str(tmp_path / 'mars-64x32x3.tflite')
    Wr.write_mars(wd, p)
    g = R.read(p)
    kinds = [op.kind for op in g.ops]
    assert kinds[0] == 'REVERSE_V2' and kinds.count('CONV_2D') == 16 and kinds.count('ELU') == 2 + 6 + 5 + 1 and kinds[-1] == 'DIV'
    got = load_mars_weights(p)
    assert got['__swap_rb__'] is True and got['fc1/weights'].shape == (4096, 128) and 'conv3_1/projection/biases' not in got
    a, b = nets.compile_mars(wd), nets.compile_mars(got)
    assert bytes(a.blob) == bytes(b.blob) and all((x == y).all() for x, y in zip(a.ops, b.ops))
    # a file without the reversal: the first layer takes the channels as they come
    Wr.write_mars(wd, p, reverse_channels=False)
    got = R.load_mars(p)
    assert got['__swap_rb__'] is False
    c = nets.compile_mars(got)
    assert bytes(c.blob) != bytes(a.blob)
    # a float16 weight file (filters behind DEQUANTIZE): the filters come back rounded to half precision, everything else as written
    Wr.write_mars(wd, p, half_weights=True)
    got = R.load_mars(p)
    w, _ = nets.fold_conv_bn(wd, 'conv4_1/1')
    np.testing.assert_array_equal(got['conv4_1/1/weights'], w.astype(np.float16).astype(np.float32))
    np.testing.assert_array_equal(got['ball/scale'], nets.bn_affine(wd, 'ball')[0])


def test_mars_reader_refuses_other_graphs_by_name(tmp_path, golden_dir):
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_reader as R, tflite_writer as Wr
    with pytest.raises(R.UnsupportedModel) as e:              # the detector fixture is no encoder
        R.load_mars(os.path.join(golden_dir, 'tiny_quant_graph.tflite'))
    assert '64, 32, 3' in str(e.value) or 'input' in str(e.value)
    wd = nets.synthetic_mars_weights(3)
    W = Wr.mars_graph(wd)
    # an operator the encoder does not have, in the middle of the graph
    i = next(k for k, o in enumerate(W.ops) if o['kind'] == 'MAX_POOL_2D')
    t = W.ops[i]['outputs'][0]
    extra = W.tensor('squashed', W.tensors[t]['shape'], np.float32)
    W.ops.insert(i + 1, dict(code=W._code('LOGISTIC'), kind='LOGISTIC', inputs=[t], outputs=[extra], options={}, custom_options=None))
    p = str(tmp_path / 'odd.tflite')
    open(p, 'wb').write(W.tobytes())
    with pytest.raises(R.UnsupportedModel) as e:
        R.load_mars(p)
    assert 'LOGISTIC' in str(e.value)
    # a block short: the counts are named
    W = Wr.mars_graph(wd)
    W.ops = [o for o in W.ops if not (o['kind'] == 'CONV_2D' and W.tensors[o['outputs'][0]]['name'] == 'conv4_3/2')]
    open(p, 'wb').write(W.tobytes())
    with pytest.raises(R.UnsupportedModel) as e:
        R.load_mars(p)
    assert '15 convolutions' in str(e.value)


def test_yolov5s_round_trip_and_refusals(tmp_path):
    """nets weights -> export-shaped .tflite (float32 and float16 filters) -> load_yolov5s -> the same compiled program; graphs that
    compute something else are refused by operator: Hardswish, other anchors in the Detect tail, another Focus order, an int8 input."""
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_writer, tflite_reader
    from deepdish_amd.tools.weights_io import load_yolov5_weights
    wd = nets.synthetic_yolov5s_weights(3)
    want = bytes(nets.compile_yolov5s(wd).blob)
    path = str(tmp_path / 'yolov5s-fp32.tflite')
    tflite_writer.write_yolov5s(wd, path)
    got = load_yolov5_weights(path)                                # what YOLOV5(model_file=...) calls
    assert got['__in_size__'] == 640 and bytes(nets.compile_yolov5s(got).blob) == want
    # an fp16 file: the filters are float16 constants behind DEQUANTIZE; the program keeps f16 weights anyway, so the blob is the same
    path16 = str(tmp_path / 'yolov5s-fp16.tflite')
    tflite_writer.write_yolov5s(wd, path16, half_weights=True)
    got16 = load_yolov5_weights(path16)
    for name, k, cin, cout in nets.yolov5s_convs():
        np.testing.assert_array_equal(got16[name + '/weights'], got[name + '/weights'].astype(np.float16).astype(np.float32), err_msg=name)
    assert bytes(nets.compile_yolov5s(got16).blob) == want
    tflite_writer.write_yolov5s(wd, path, activation='hardswish')
    with pytest.raises(tflite_reader.UnsupportedModel) as e:
        tflite_reader.load_yolov5s(path)
    assert 'HARD_SWISH' in str(e.value) or 'Hardswish' in str(e.value)
    old = nets.YOLO_ANCHORS
    try:                                                            # a file whose Detect tail carries other anchors: the numeric check catches it
        nets.YOLO_ANCHORS = [[12, 13, 16, 30, 33, 23]] + [list(a) for a in old[1:]]
        tflite_writer.write_yolov5s(wd, path)
    finally:
        nets.YOLO_ANCHORS = old
    with pytest.raises(tflite_reader.UnsupportedModel) as e:
        tflite_reader.load_yolov5s(path)
    assert 'detect0' in str(e.value) and 'box decode' in str(e.value)
    # another slicing order in front of the first convolution
    g = tflite_writer.yolov5s_graph(wd)
    cat = next(o for o in g.ops if o['kind'] == 'CONCATENATION')
    cat['inputs'][1], cat['inputs'][2] = cat['inputs'][2], cat['inputs'][1]
    open(path, 'wb').write(g.tobytes())
    with pytest.raises(tflite_reader.UnsupportedModel) as e:
        tflite_reader.load_yolov5s(path)
    assert 'Focus' in str(e.value)
    # a narrower model (the filters of m1 cut): named
    wd2 = dict(wd)
    with pytest.raises(Exception):
        wd2['m1/weights'] = wd['m1/weights'][:, :, :, :32]
        tflite_writer.write_yolov5s(wd2, path)
        tflite_reader.load_yolov5s(path)


def test_mars_reader_accepts_explicit_same_padding(tmp_path):
    """A stride-2 layer's SAME padding spelt PAD + VALID CONV_2D gives the same weights; other zeros in front of a VALID convolution are refused."""
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_writer, tflite_reader
    wd = nets.synthetic_mars_weights(11)
    a, b = str(tmp_path / 'a.tflite'), str(tmp_path / 'b.tflite')
    tflite_writer.write_mars(wd, a)
    tflite_writer.write_mars(wd, b, explicit_pad=True)
    assert sum(o.kind == 'PAD' for o in tflite_reader.read(b).ops) == 4          # conv3_1/1, its projection, conv4_1/1, its projection
    wa, wb = tflite_reader.load_mars(a), tflite_reader.load_mars(b)
    assert set(wa) == set(wb)
    for k in wa:
        np.testing.assert_array_equal(np.asarray(wa[k]), np.asarray(wb[k]), err_msg=k)
    # conv4_1/1 reads a 16 x 8 map: SAME adds a row below and a column to the right only; symmetric zeros there are another convolution
    g = tflite_writer.mars_graph(wd, explicit_pad=True)
    hit = 0
    for o in g.ops:
        if o['kind'] == 'PAD':
            t = g.tensors[o['inputs'][1]]
            if np.frombuffer(g.buffers[t['buffer']], np.int32).tolist() == [0, 0, 0, 1, 0, 1, 0, 0]:
                g.buffers[t['buffer']] = np.array([[0, 0], [1, 1], [1, 1], [0, 0]], np.int32).tobytes()
                hit += 1
                break
    assert hit == 1
    open(b, 'wb').write(g.tobytes())
    with pytest.raises(tflite_reader.UnsupportedModel) as e:
        tflite_reader.load_mars(b)
    assert 'explicit padding' in str(e.value)


def _float_ssd(seed=7):
    from deepdish_amd import nets, quantize
    wd = nets.synthetic_ssd_weights(seed)
    folded = {}
    for name, kind, w, b, stride, act in quantize.folded_ssd_layers(wd):
        folded[name + '/weights'] = w if kind == 'conv' else w[:, :, :, None]
        folded[name + '/biases'] = b
    return wd, folded


def test_model_metadata_round_trip_and_refusals(tmp_path):
    """TFLite Model Metadata as tools/tflite_object_detector.py:117-137 upstream reads it: NormalizationOptions mean / std of the input tensor
    and the label list = the first associated file packed (as a ZIP archive) behind the flatbuffer.  The file stays a model the reader maps;
    a file without metadata, or with metadata but no packed file, is refused where the reference's MetadataDisplayer calls raise."""
    import zipfile
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_writer, tflite_reader as R
    from deepdish_amd.tools.weights_io import load_ssd_model
    wd, folded = _float_ssd()
    labels = ['person', 'bicycle', 'car', '', 'motorcycle', 'traffic light']
    path = str(tmp_path / 'efficientdet-style.tflite')
    tflite_writer.write_ssd_mobilenet(folded, path, metadata=dict(mean=[127.0], std=[128.0], labels=labels, label_file='labelmap.txt'))
    m = R.read_metadata(path)
    assert (m['mean'], m['std'], m['label_file']) == (127.0, 128.0, 'labelmap.txt')
    assert m['labels'] == ['person', 'bicycle', 'car', 'motorcycle', 'traffic light']      # list(filter(len, ...)), :136
    assert zipfile.ZipFile(path).namelist() == ['labelmap.txt']                            # the archive a stock zip reader sees
    kind, wd2 = load_ssd_model(path)                                                       # ... and still the same detector
    assert kind == 'f32' and bytes(nets.compile_ssd_mobilenet(wd2).blob) == bytes(nets.compile_ssd_mobilenet(wd).blob)
    assert not (nets.compile_ssd_mobilenet(wd2, mean=127.0, std=128.0).serialize()[0] == nets.compile_ssd_mobilenet(wd2).serialize()[0]).all()   # (the first layer's op words carry them)
    # no NormalizationOptions unit: the reference's defaults (:124-125)
    p2 = str(tmp_path / 'no-normalisation.tflite')
    tflite_writer.write_ssd_mobilenet(folded, p2, metadata=dict(mean=None, std=None, labels=labels))
    m2 = R.read_metadata(p2)
    assert (m2['mean'], m2['std']) == (127.5, 127.5) and len(m2['labels']) == 5
    # no metadata at all / metadata without a packed file
    p3 = str(tmp_path / 'plain.tflite')
    tflite_writer.write_ssd_mobilenet(folded, p3)
    with pytest.raises(R.UnsupportedModel) as e:
        R.read_metadata(p3)
    assert 'TFLITE_METADATA' in str(e.value)
    p4 = str(tmp_path / 'no-labels.tflite')
    tflite_writer.write_ssd_mobilenet(folded, p4, metadata=dict(mean=[127.5], std=[127.5], labels=None))
    with pytest.raises(R.UnsupportedModel) as e:
        R.read_metadata(p4)
    assert 'packed' in str(e.value)
