"""Reads the model files the reference loads -- `.tflite` flatbuffers (tools/ssd_mobilenet.py:31-52 `Interpreter(model_path=...)`,
tools/yolov5.py:71-79, tools/generate_detections.py:151-162) -- without TensorFlow: tensors, buffers, quantisation parameters and
the operator list, from the public schema (tensorflow/lite/schema/schema.fbs; table slots below are its field order).

`read(path)` gives the graph as plain Python objects.  `load_ssd_mobilenet(path)` recognises the SSD-MobileNet-v1 detector the
reference ships (`detectors/mobilenet/ssdmobilenetv1.tflite`, absent from the tree: .MISSING_LARGE_BLOBS) by structure -- the
`TFLite_Detection_PostProcess` custom op, its two CONCATENATION / RESHAPE / CONV_2D fans, the CONV_2D / DEPTHWISE_CONV_2D chain
behind them -- and returns the model in the form deepdish_amd/netsq.py (uint8) or deepdish_amd/nets.py (float) compiles; the first
operator that does not fit is named in the error.  None of the reference's blobs is here to try it on: the parser is exercised on
files deepdish_amd/tools/tflite_writer.py produces from the same schema (tests/test_tflite_io.py).
"""
import numpy as np

from . import flatbuf

TENSOR_TYPES = {0: np.float32, 1: np.float16, 2: np.int32, 3: np.uint8, 4: np.int64, 6: np.bool_, 7: np.int16, 9: np.int8}
# BuiltinOperator values (schema.fbs)
OPS = {0: 'ADD', 1: 'AVERAGE_POOL_2D', 2: 'CONCATENATION', 3: 'CONV_2D', 4: 'DEPTHWISE_CONV_2D', 6: 'DEQUANTIZE', 9: 'FULLY_CONNECTED',
       14: 'LOGISTIC', 17: 'MAX_POOL_2D', 18: 'MUL', 19: 'RELU', 21: 'RELU6', 22: 'RESHAPE', 23: 'RESIZE_BILINEAR', 25: 'SOFTMAX',
       28: 'TANH', 32: 'CUSTOM', 34: 'PAD', 39: 'TRANSPOSE', 40: 'MEAN', 41: 'SUB', 42: 'DIV', 43: 'SQUEEZE', 45: 'STRIDED_SLICE',
       47: 'EXP', 49: 'SPLIT', 53: 'CAST', 55: 'MAXIMUM', 74: 'SUM', 75: 'SQRT', 76: 'RSQRT', 83: 'PACK', 92: 'SQUARE', 97: 'RESIZE_NEAREST_NEIGHBOR',
       105: 'REVERSE_V2', 11: 'L2_NORMALIZATION', 111: 'ELU', 114: 'QUANTIZE', 117: 'HARD_SWISH'}
PADDING = {0: 'SAME', 1: 'VALID'}
ACTIVATION = {0: 'none', 1: 'relu', 2: 'relu_n1_to_1', 3: 'relu6', 4: 'tanh'}


class Tensor:
    def __init__(self, index, name, shape, dtype, data, scale, zero_point, quant_dim):
        self.index, self.name, self.shape, self.dtype, self.data = index, name, shape, dtype, data
        self.scale, self.zero_point, self.quant_dim = scale, zero_point, quant_dim

    @property
    def per_tensor(self):
        return len(self.scale) == 1

    def __repr__(self):
        return 'Tensor(%d %r %s %s%s)' % (self.index, self.name, tuple(self.shape), np.dtype(self.dtype).name if self.dtype else '?',
                                          ' const' if self.data is not None else '')


class Op:
    def __init__(self, index, kind, inputs, outputs, options, custom):
        self.index, self.kind, self.inputs, self.outputs, self.options, self.custom = index, kind, inputs, outputs, options, custom

    def __repr__(self):
        return 'Op(%d %s %s -> %s %s)' % (self.index, self.kind, self.inputs, self.outputs, self.options)


class Graph:
    def __init__(self, tensors, ops, inputs, outputs, description):
        self.tensors, self.ops, self.inputs, self.outputs, self.description = tensors, ops, inputs, outputs, description
        self.producer = {o: op for op in ops for o in op.outputs}

    def made_by(self, t):
        return self.producer.get(t)


def _options(kind, op_table):
    t = op_table.table(4)                       # Operator.builtin_options (slot 3 is its union type)
    if t is None:
        return {}
    if kind == 'CONV_2D':                       # Conv2DOptions: padding, stride_w, stride_h, fused_activation_function, dilation_w, dilation_h
        return dict(padding=PADDING.get(t.scalar(0, 'i8'), '?'), stride_w=t.scalar(1, 'i32'), stride_h=t.scalar(2, 'i32'),
                    act=ACTIVATION.get(t.scalar(3, 'i8'), '?'), dilation_w=t.scalar(4, 'i32', 1), dilation_h=t.scalar(5, 'i32', 1))
    if kind == 'DEPTHWISE_CONV_2D':             # DepthwiseConv2DOptions: padding, stride_w, stride_h, depth_multiplier, fused_activation_function, dilations
        return dict(padding=PADDING.get(t.scalar(0, 'i8'), '?'), stride_w=t.scalar(1, 'i32'), stride_h=t.scalar(2, 'i32'),
                    depth_multiplier=t.scalar(3, 'i32'), act=ACTIVATION.get(t.scalar(4, 'i8'), '?'),
                    dilation_w=t.scalar(5, 'i32', 1), dilation_h=t.scalar(6, 'i32', 1))
    if kind == 'CONCATENATION':                 # ConcatenationOptions: axis, fused_activation_function
        return dict(axis=t.scalar(0, 'i32'), act=ACTIVATION.get(t.scalar(1, 'i8'), '?'))
    if kind == 'RESHAPE':                       # ReshapeOptions: new_shape
        return dict(new_shape=t.scalars(0, 'i32'))
    if kind in ('MAX_POOL_2D', 'AVERAGE_POOL_2D'):   # Pool2DOptions: padding, stride_w, stride_h, filter_width, filter_height, fused_activation_function
        return dict(padding=PADDING.get(t.scalar(0, 'i8'), '?'), stride_w=t.scalar(1, 'i32'), stride_h=t.scalar(2, 'i32'),
                    filter_w=t.scalar(3, 'i32'), filter_h=t.scalar(4, 'i32'), act=ACTIVATION.get(t.scalar(5, 'i8'), '?'))
    if kind in ('ADD', 'MUL', 'SUB', 'DIV'):
        return dict(act=ACTIVATION.get(t.scalar(0, 'i8'), '?'))
    if kind == 'FULLY_CONNECTED':
        return dict(act=ACTIVATION.get(t.scalar(0, 'i8'), '?'))
    if kind == 'STRIDED_SLICE':                 # StridedSliceOptions: begin_mask, end_mask, ellipsis_mask, new_axis_mask, shrink_axis_mask
        return dict(begin_mask=t.scalar(0, 'i32'), end_mask=t.scalar(1, 'i32'))
    return {}


def read(path_or_bytes):
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, 'rb').read()
    m = flatbuf.root(buf, b'TFL3')              # Model: version, operator_codes, subgraphs, description, buffers
    if m.scalar(0, 'u32') != 3:
        raise ValueError('TFLite schema version %d (expected 3)' % m.scalar(0, 'u32'))
    codes = []
    for c in m.tables(1):                       # OperatorCode: deprecated_builtin_code (i8), custom_code, version, builtin_code (i32)
        b = max(c.scalar(3, 'i32'), c.scalar(0, 'i8'))
        codes.append((OPS.get(b, 'BUILTIN_%d' % b), c.string(1)))
    buffers = [b.scalars(0, 'u8') for b in m.tables(4)]         # Buffer.data
    subs = m.tables(2)
    if len(subs) != 1:
        raise ValueError('%d subgraphs: control-flow models are not supported' % len(subs))
    g = subs[0]                                 # SubGraph: tensors, inputs, outputs, operators, name
    tensors = []
    for i, t in enumerate(g.tables(0)):         # Tensor: shape, type, buffer, name, quantization
        dt = TENSOR_TYPES.get(t.scalar(1, 'i8'))
        shape = t.scalars(0, 'i32')
        raw = buffers[t.scalar(2, 'u32')] if t.scalar(2, 'u32') < len(buffers) else b''
        data = None
        if len(raw) and dt is not None:
            data = np.frombuffer(bytes(raw), dtype=dt).reshape(shape)
        q = t.table(4)                          # QuantizationParameters: min, max, scale, zero_point, details_type, details, quantized_dimension
        scale = np.array(q.scalars(2, 'f32'), np.float32) if q else np.zeros(0, np.float32)
        zp = np.array(q.scalars(3, 'i64'), np.int64) if q else np.zeros(0, np.int64)
        tensors.append(Tensor(i, t.string(3), shape, dt, data, scale, zp, q.scalar(6, 'i32') if q else 0))
    ops = []
    for i, o in enumerate(g.tables(3)):         # Operator: opcode_index, inputs, outputs, builtin_options_type, builtin_options, custom_options
        kind, custom = codes[o.scalar(0, 'u32')]
        opts = _options(kind, o)
        if kind == 'CUSTOM':
            raw = bytes(o.scalars(5, 'u8'))
            opts = flatbuf.flex_map(raw) if raw else {}
        ops.append(Op(i, kind, o.scalars(1, 'i32'), o.scalars(2, 'i32'), opts, custom))
    return Graph(tensors, ops, g.scalars(1, 'i32'), g.scalars(2, 'i32'), m.string(3))


# ------------------------------------------------------------------------------------------- SSD-MobileNet-v1
class UnsupportedModel(ValueError):
    pass


def read_metadata(path_or_bytes):
    """TFLite Model Metadata of a model file, the part tools/tflite_object_detector.py:117-137 upstream takes from
    `metadata.MetadataDisplayer` (tflite_support, absent here): -> dict(mean, std, labels, label_file).
      * mean / std: the first values of the input tensor's NormalizationOptions process unit, 127.5 / 127.5 without one (:124-131);
      * labels: the FIRST packed associated file (the ZIP archive behind the flatbuffer), decoded, empty lines dropped (:134-137).
    UnsupportedModel where the reference's calls raise: no TFLITE_METADATA buffer, no packed file."""
    import io
    import zipfile
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, 'rb').read()
    name = path_or_bytes if isinstance(path_or_bytes, str) else '<bytes>'
    m = flatbuf.root(buf, b'TFL3')
    meta_buf = None
    buffers = m.tables(4)
    for e in m.tables(6):                       # Model.metadata: [Metadata {0 name, 1 buffer}]
        if e.string(0) == 'TFLITE_METADATA' and e.scalar(1, 'u32') < len(buffers):
            meta_buf = bytes(buffers[e.scalar(1, 'u32')].scalars(0, 'u8'))
    if not meta_buf:
        raise UnsupportedModel('%s: the model file carries no TFLITE_METADATA buffer (the generic adaptor takes mean / std and the label list from it, '
                               'tools/tflite_object_detector.py:117-137 upstream)' % name)
    root = flatbuf.root(meta_buf, b'M001')
    mean = std = 127.5
    subs = root.tables(3)                       # ModelMetadata.subgraph_metadata
    if subs and subs[0].tables(2):              # SubGraphMetadata.input_tensor_metadata[0].process_units
        for unit in subs[0].tables(2)[0].tables(4):
            if unit.scalar(0, 'u8') == 1:       # ProcessUnitOptions.NormalizationOptions {0 mean, 1 std}
                o = unit.table(1)
                mean, std = float(o.scalars(0, 'f32')[0]), float(o.scalars(1, 'f32')[0])
    try:
        z = zipfile.ZipFile(io.BytesIO(bytes(buf)))
        names = z.namelist()
    except zipfile.BadZipFile:
        names = []
    if not names:
        raise UnsupportedModel('%s: no associated file is packed into the model file (the label list is its first packed file)' % name)
    text = z.read(names[0]).decode()
    return dict(mean=mean, std=std, labels=list(filter(len, text.splitlines())), label_file=names[0])


def _need(cond, op, what):
    if not cond:
        raise UnsupportedModel('operator %s: %s' % (op, what))


def _conv_layer(g, op):
    """CONV_2D / DEPTHWISE_CONV_2D -> QModel layer dict (weights in HWIO / HWC as deepdish_amd/quantize.py has them)."""
    _need(op.kind in ('CONV_2D', 'DEPTHWISE_CONV_2D'), op, 'expected a convolution here')
    x, w, b = (g.tensors[i] for i in op.inputs[:3])
    y = g.tensors[op.outputs[0]]
    o = op.options
    _need(o['padding'] == 'SAME' and o['stride_w'] == o['stride_h'] and o['dilation_w'] == 1 and o['dilation_h'] == 1, op, 'SAME padding, square stride, no dilation')
    _need(o['act'] in ('none', 'relu6'), op, 'fused activation %s (NONE and RELU6 are built)' % o['act'])
    _need(w.data is not None and b.data is not None, op, 'filter and bias must be constants')
    if op.kind == 'CONV_2D':
        wd = np.transpose(w.data, (1, 2, 3, 0))                     # OHWI -> HWIO
        kind = 'conv'
    else:
        _need(o['depth_multiplier'] == 1 and w.data.shape[0] == 1, op, 'depth multiplier 1')
        wd = w.data[0]                                               # [1, H, W, C] -> HWC
        kind = 'dw'
    L = dict(kind=kind, w=np.ascontiguousarray(wd), stride=int(o['stride_w']), act=o['act'])
    if x.dtype == np.uint8:
        _need(w.dtype == np.uint8 and b.dtype == np.int32 and y.dtype == np.uint8, op, 'uint8 activations need uint8 filters and int32 biases')
        _need(x.per_tensor and w.per_tensor and y.per_tensor, op, 'per-channel quantisation is not built (per-tensor parameters only)')
        L.update(w_scale=np.float32(w.scale[0]), w_zp=int(w.zero_point[0]), bias=b.data.astype(np.int32).reshape(-1),
                 in_scale=np.float32(x.scale[0]), in_zp=int(x.zero_point[0]), out_scale=np.float32(y.scale[0]), out_zp=int(y.zero_point[0]))
    else:
        _need(x.dtype == np.float32, op, 'activations of type %s' % x.dtype)
        L.update(w=L['w'].astype(np.float32), bias=b.data.astype(np.float32).reshape(-1))
    return L, op.inputs[0]


# TFLite_Detection_PostProcess options this build runs (tools/ssd_mobilenet.py:100-109 upstream: the interpreter applies whatever the file
# says).  Passed on: max_detections, nms_score_threshold, nms_iou_threshold (SSDMobileNet, dd_pipeline_ssd_options).  Checked: the four
# box-coder scales the decode is built for must be PRESENT and equal 10 / 10 / 5 / 5; max_classes_per_detection 1; the fast
# (class-agnostic) NMS -- `use_regular_nms` true selects a per-class NMS with `detections_per_class`, which is not built: refused by name.
SSD_POST_DEFAULTS = dict(max_detections=10, nms_score_threshold=1e-8, nms_iou_threshold=0.6)
MAX_DETECTIONS_BUILT = 64                                          # csrc/post.hip: one wave lane per row of the op's output


def ssd_post_options(post):
    o = post.options
    for k, v in dict(y_scale=10.0, x_scale=10.0, h_scale=5.0, w_scale=5.0).items():
        _need(k in o, post, 'option %s is missing (the box decode needs the coder scales stated)' % k)
        _need(abs(float(o[k]) - v) < 1e-6, post, '%s = %s (the decode is built for %s)' % (k, o[k], v))
    for k in ('max_detections', 'nms_score_threshold', 'nms_iou_threshold', 'num_classes'):
        _need(k in o, post, 'option %s is missing' % k)
    _need(not bool(o.get('use_regular_nms', False)), post, 'use_regular_nms = true (per-class NMS with detections_per_class = %s): only the fast '
          'class-agnostic NMS is built' % o.get('detections_per_class', 100))
    _need(int(o.get('max_classes_per_detection', 1)) == 1, post, 'max_classes_per_detection = %s (one class per detection is built)' % o.get('max_classes_per_detection'))
    md = int(o['max_detections'])
    _need(1 <= md <= MAX_DETECTIONS_BUILT, post, 'max_detections = %d (1 .. %d are built)' % (md, MAX_DETECTIONS_BUILT))
    iou, thr = float(o['nms_iou_threshold']), float(o['nms_score_threshold'])
    _need(0.0 < iou <= 1.0, post, 'nms_iou_threshold = %s' % iou)            # the op itself rejects values outside (0, 1]
    return dict(max_detections=md, nms_score_threshold=thr, nms_iou_threshold=iou, num_classes=int(o['num_classes']))


def load_ssd_mobilenet(path):
    """-> ('uint8', QModel) or ('f32', named folded weights).  Structure walked backwards from the post-process op."""
    g = read(path)
    post = [op for op in g.ops if op.kind == 'CUSTOM']
    if len(post) != 1 or post[0].custom != 'TFLite_Detection_PostProcess':
        raise UnsupportedModel('%s: no TFLite_Detection_PostProcess op (operators: %s)' % (path, sorted({o.custom or o.kind for o in g.ops})))
    post = post[0]
    popts = ssd_post_options(post)
    anchors = g.tensors[post.inputs[2]]
    _need(anchors.data is not None and anchors.data.ndim == 2 and anchors.data.shape[1] == 4, post, 'anchors must be a constant [n, 4]')
    anc = anchors.data.astype(np.float32) if anchors.dtype != np.uint8 else (np.float32(anchors.scale[0]) * (anchors.data.astype(np.float32) - np.float32(anchors.zero_point[0])))

    def fan(t, logistic):
        """tensor -> the six head convolutions feeding it (through [LOGISTIC] CONCATENATION RESHAPE)."""
        op = g.made_by(t)
        logi = None
        if logistic:
            _need(op is not None and op.kind == 'LOGISTIC', op, 'class predictions must come from LOGISTIC')
            logi = op
            op = g.made_by(op.inputs[0])
        _need(op is not None and op.kind == 'CONCATENATION', op, 'expected the CONCATENATION of the per-map predictions')
        heads = []
        for ti in op.inputs:
            r = g.made_by(ti)
            _need(r is not None and r.kind == 'RESHAPE', r, 'expected RESHAPE of a predictor output')
            c = g.made_by(r.inputs[0])
            _need(c is not None and c.kind == 'CONV_2D', c, 'expected the predictor CONV_2D')
            heads.append(c)
        return heads, logi

    box_heads, _ = fan(post.inputs[0], False)
    cls_heads, logi = fan(post.inputs[1], True)
    _need(len(box_heads) == 6 and len(cls_heads) == 6, post, '%d / %d predictor maps (SSD-MobileNet-v1 has 6)' % (len(box_heads), len(cls_heads)))
    layers, feats = {}, []
    for k, (bh, ch) in enumerate(zip(box_heads, cls_heads)):
        _need(bh.inputs[0] == ch.inputs[0], ch, 'box and class predictor %d read different feature maps' % k)
        layers[f'box{k}'], _ = _conv_layer(g, bh)
        layers[f'cls{k}'], _ = _conv_layer(g, ch)
        feats.append(bh.inputs[0])
    # the backbone: from the last feature map back to the graph input
    chain, t = [], feats[-1]
    while g.made_by(t) is not None:
        op = g.made_by(t)
        chain.append(op)
        t = op.inputs[0]
    _need(t in g.inputs, chain[-1] if chain else post, 'the backbone does not start at a graph input')
    chain.reverse()
    names = ['conv0'] + [n for i in range(1, 14) for n in (f'dw{i}', f'pw{i}')] + [f'extra{j}_{h}' for j in range(1, 5) for h in (1, 2)]
    _need(len(chain) == len(names), chain[min(len(chain), len(names)) - 1], '%d backbone convolutions (SSD-MobileNet-v1 has %d)' % (len(chain), len(names)))
    out_of = {}
    for name, op in zip(names, chain):
        layers[name], _ = _conv_layer(g, op)
        _need(layers[name]['kind'] == ('dw' if name.startswith('dw') else 'conv'), op, 'expected %s for %s' % ('DEPTHWISE_CONV_2D' if name.startswith('dw') else 'CONV_2D', name))
        out_of[op.outputs[0]] = name
    want_feats = ['pw11', 'pw13', 'extra1_2', 'extra2_2', 'extra3_2', 'extra4_2']
    _need([out_of.get(f) for f in feats] == want_feats, post, 'feature maps come from %s (expected %s)' % ([out_of.get(f) for f in feats], want_feats))
    x = g.tensors[t]
    _need(len(x.shape) == 4 and x.shape[1] == x.shape[2] and x.shape[3] == 3, chain[0], 'input %s (expected [1, s, s, 3])' % (x.shape,))
    order = names + [n for k in range(6) for n in (f'box{k}', f'cls{k}')]
    per_anchor = layers['cls0']['w'].shape[3] // 3                      # the lowest map has three anchors per cell
    _need(popts['num_classes'] + 1 == per_anchor, post, 'num_classes = %d, but the class predictors emit %d values per anchor (classes + background)'
          % (popts['num_classes'], per_anchor))
    if x.dtype == np.uint8:
        lt = g.tensors[logi.outputs[0]]
        qm = dict(kind='ssd_mobilenet_v1_uint8', input=dict(scale=np.float32(x.scale[0]), zp=int(x.zero_point[0]), size=int(x.shape[1])),
                  layers=layers, logistic=dict(out_scale=np.float32(lt.scale[0]), out_zp=int(lt.zero_point[0])), order=order, anchors=anc,
                  post=popts, source=str(path))
        return 'uint8', qm
    wd = {}
    for name in order:
        L = layers[name]
        w = L['w'] if L['kind'] == 'conv' else L['w'][:, :, :, None]
        wd[name + '/weights'] = w.astype(np.float32)
        wd[name + '/biases'] = L['bias']
    wd['anchors'] = anc
    wd['__post__'] = popts                                            # not an array: nets.compile_ssd_mobilenet ignores it, the plugins read it
    return 'f32', wd


# ------------------------------------------------------------------------------------------- MARS encoder
MARS_BLOCKS = [('conv2_1', 32, False, True), ('conv2_3', 32, False, False), ('conv3_1', 64, True, False),
               ('conv3_3', 64, False, False), ('conv4_1', 128, True, False), ('conv4_3', 128, False, False)]


def load_mars(path):
    """`mars-64x32x3.tflite` (tools/generate_detections.py:151-177 upstream: float32 patches in, [n, 128] features out) -> the named float
    weights deepdish_amd/nets.compile_mars takes, batch norms folded as the converter leaves them.

    The file is absent from the reference tree, so the operator pattern is the one a TFLite conversion of tools/freeze_model.py's graph
    gives as far as that can be told without it: every slim.conv2d / fully_connected with a normalizer arrives as CONV_2D /
    FULLY_CONNECTED (with or without keep_num_dims: only its weights are read) with the batch norm in filter and bias, followed by ELU; a
    stride-2 layer's SAME padding may be spelt PAD + VALID CONV_2D (accepted when the zeros are exactly SAME's); a block's leading batch norm (create_link, :17-21) as MUL +
    ADD with constant operands; the skip as ADD of two activations; pool1 as a 3x3 stride-2 VALID MAX_POOL_2D; the projection as a 1x1
    stride-2 CONV_2D; "ball" as MUL + ADD behind fc1's ELU; the unit-length tail (:153-156) in any spelling out of SQUARE / MUL / SUM / ADD /
    SQRT / RSQRT / MAXIMUM / DIV / L2_NORMALIZATION.  A channel reversal in front (REVERSE_V2, or STRIDED_SLICE with a negative stride: the
    frozen graph's BGR -> RGB, :175-177) sets `__swap_rb__`; without one the first layer takes the channels as they come.  Layers are
    recognised by order, kernel size and channel counts; anything else raises UnsupportedModel naming the operator."""
    g = read(path)
    _need(len(g.inputs) == 1, g.ops[0], '%d graph inputs' % len(g.inputs))
    x = g.tensors[g.inputs[0]]
    _need(len(x.shape) == 4 and tuple(x.shape[1:]) == (64, 32, 3), g.ops[0], 'input %s (the encoder takes [n, 64, 32, 3] patches)' % (tuple(x.shape),))
    def const(i):
        """Constant data of tensor i, also through a DEQUANTIZE of a constant (float16 weight files keep their filters that way)."""
        if i < 0:
            return None
        t = g.tensors[i]
        if t.data is not None:
            return t.data
        op = g.made_by(i)
        if op is not None and op.kind == 'DEQUANTIZE' and g.tensors[op.inputs[0]].data is not None and g.tensors[op.inputs[0]].dtype == np.float16:
            return g.tensors[op.inputs[0]].data.astype(np.float32)
        return None
    convs, affines, pools, fc, flip, pending, n_skip = [], [], 0, None, False, None, 0
    pads = {}                                                          # output tensor of a PAD -> (its input, [[top, bottom], [left, right]])
    for op in g.ops:
        k = op.kind
        if k in ('CAST', 'DEQUANTIZE', 'RESHAPE', 'SQUEEZE', 'ELU'):
            continue
        if k == 'PAD':                                                 # explicit zero padding in front of a VALID convolution (some converters spell SAME this way)
            pv = const(op.inputs[1])
            _need(pv is not None and np.asarray(pv).shape == (4, 2) and not np.any(np.asarray(pv)[[0, 3]]), op, 'only constant spatial zero padding is understood')
            pads[op.outputs[0]] = (op.inputs[0], [[int(v) for v in np.asarray(pv)[1]], [int(v) for v in np.asarray(pv)[2]]])
            continue
        if k == 'REVERSE_V2':
            ax = const(op.inputs[1])
            _need(fc is None and not convs and ax is not None and [int(v) % 4 for v in np.asarray(ax).reshape(-1)] == [3], op, 'only a reversal of the channel axis of the input is understood')
            flip = True
            continue
        if k == 'STRIDED_SLICE':
            st = const(op.inputs[3])
            _need(fc is None and not convs and st is not None and list(np.asarray(st).reshape(-1)[:-1]) == [1] * (len(np.asarray(st).reshape(-1)) - 1) and int(np.asarray(st).reshape(-1)[-1]) == -1, op,
                  'only a stride -1 slice of the channel axis of the input (BGR -> RGB) is understood')
            flip = True
            continue
        if k == 'CONV_2D':
            _need(fc is None, op, 'convolution behind the fully connected layer')
            xi, wdata = g.tensors[op.inputs[0]], const(op.inputs[1])
            b = const(op.inputs[2]) if len(op.inputs) > 2 else None
            o = op.options
            _need(wdata is not None and xi.dtype == np.float32 and wdata.dtype == np.float32, op, 'float32 activations and constant float32 (or float16 behind DEQUANTIZE) filters: a quantised encoder is not built')
            w = type('W', (), dict(data=wdata))
            _need(o['stride_w'] == o['stride_h'] and o['act'] == 'none' and o['dilation_w'] == 1 and o['dilation_h'] == 1, op, 'square stride, no fused activation')
            if o['padding'] == 'VALID' and op.inputs[0] in pads:        # PAD + VALID: accepted when the padding is exactly what SAME would add
                src, pv = pads[op.inputs[0]]
                xs, kk, st = g.tensors[src].shape, wdata.shape[1], int(o['stride_w'])
                want = []
                for d in (1, 2):
                    tot = max((-(-int(xs[d]) // st) - 1) * st + kk - int(xs[d]), 0)
                    want.append([tot // 2, tot - tot // 2])
                _need(pv == want, op, 'explicit padding %s in front of a VALID convolution (TensorFlow SAME for this layer is %s)' % (pv, want))
            else:
                _need(o['padding'] == 'SAME', op, 'SAME padding (or PAD + VALID with the same zeros)')
            wt = np.ascontiguousarray(np.transpose(w.data, (1, 2, 3, 0))).astype(np.float32)     # OHWI -> HWIO
            convs.append(dict(op=op, w=wt, b=np.zeros(wt.shape[3], np.float32) if b is None else np.asarray(b, np.float32).reshape(-1), stride=int(o['stride_w'])))
            continue
        if k == 'MAX_POOL_2D':
            o = op.options
            _need((o['filter_w'], o['filter_h'], o['stride_w'], o['stride_h'], o['padding']) == (3, 3, 2, 2, 'VALID') and len(convs) == 2, op, 'pool1 is 3x3 stride 2 VALID behind conv1_2')
            pools += 1
            continue
        if k == 'FULLY_CONNECTED':
            w, b = const(op.inputs[1]), const(op.inputs[2]) if len(op.inputs) > 2 else None
            _need(fc is None and w is not None and w.ndim == 2 and w.dtype == np.float32 and op.options.get('act', 'none') == 'none', op, 'one float32 fully connected layer with constant weights')
            fc = dict(op=op, w=np.ascontiguousarray(w.T).astype(np.float32), b=np.zeros(w.shape[0], np.float32) if b is None else np.asarray(b, np.float32).reshape(-1))
            continue
        if k in ('MUL', 'ADD'):
            cs = [const(i) for i in op.inputs[:2]]
            vec = [c for c in cs if c is not None and c.size > 1]
            if vec and len([c for c in cs if c is None]) == 1:                      # activation (*|+) per-channel constant: one half of a batch norm
                act_in = op.inputs[0] if cs[0] is None else op.inputs[1]
                v = np.asarray(vec[0], np.float32).reshape(-1)
                if k == 'MUL':
                    _need(pending is None, op, 'two scalings in a row')
                    pending = (v, op.outputs[0])
                else:
                    _need(pending is not None and pending[1] == act_in and len(pending[0]) == len(v), op, 'a per-channel ADD that does not follow its MUL (batch norm as scale, then shift)')
                    affines.append((pending[0], v))
                    pending = None
                continue
            if k == 'ADD' and cs[0] is None and cs[1] is None and fc is None:       # the skip connection
                n_skip += 1
                continue
            _need(fc is not None, op, 'an element-wise %s that is neither a batch norm nor a skip connection' % k)
            continue                                                                 # (unit-length tail: x * x, + 1e-8, x * rsqrt)
        if k in ('SQUARE', 'SUM', 'SQRT', 'RSQRT', 'DIV', 'MAXIMUM', 'L2_NORMALIZATION'):
            _need(fc is not None, op, '%s in front of the fully connected layer' % k)
            continue
        _need(False, op, 'not part of the MARS encoder graph as this reader knows it')
    last = g.ops[-1]
    _need(pending is None, last, 'a per-channel MUL without its ADD')
    _need(len(convs) == 16 and len(affines) == 6 and pools == 1 and fc is not None and n_skip == 6, last,
          '%d convolutions, %d stand-alone batch norms, %d pools, %d skip additions, %s fully connected layer (MARS: 16, 6, 1, 6, one)' % (len(convs), len(affines), pools, n_skip, 'a' if fc else 'no'))
    wd = {'__swap_rb__': flip}

    def take(name, c, kk, cin, cout, stride, what):
        _need(c['w'].shape == (kk, kk, cin, cout) and c['stride'] == stride, c['op'], '%s: filter %s stride %d (expected %dx%d %d -> %d stride %d)' % (what, c['w'].shape, c['stride'], kk, kk, cin, cout, stride))
        wd[name + '/weights'], wd[name + '/biases'] = c['w'], c['b']

    it = iter(convs)
    take('conv1_1', next(it), 3, 3, 32, 1, 'conv1_1')
    take('conv1_2', next(it), 3, 32, 32, 1, 'conv1_2')
    cin = 32
    for name, c, inc, first in MARS_BLOCKS:
        blk = [next(it) for _ in range(3 if inc else 2)]
        proj = [b for b in blk if b['w'].shape[0] == 1]
        main = [b for b in blk if b['w'].shape[0] == 3]
        _need(len(proj) == (1 if inc else 0) and len(main) == 2, blk[0]['op'], 'block %s: %d 3x3 and %d 1x1 convolutions' % (name, len(main), len(proj)))
        take(name + '/1', main[0], 3, cin, c, 2 if inc else 1, name + '/1')
        take(name + '/2', main[1], 3, c, c, 1, name + '/2')
        if inc:
            take(name + '/projection', proj[0], 1, cin, c, 2, name + '/projection')
            _need(not np.any(proj[0]['b']), proj[0]['op'], 'the projection has no bias (freeze_model.py:30-36)')
            del wd[name + '/projection/biases']
        cin = c
    order = [b[0] + '/bn' for b in MARS_BLOCKS if not b[3]] + ['ball']
    for name, (sc, sh) in zip(order, affines):
        want = 128 if name == 'ball' else dict((b[0] + '/bn', p[1]) for b, p in zip(MARS_BLOCKS[1:], MARS_BLOCKS[:-1]))[name]
        _need(len(sc) == want, last, 'batch norm %s has %d channels (expected %d)' % (name, len(sc), want))
        wd[name + '/scale'], wd[name + '/shift'] = sc, sh
    _need(fc['w'].shape == (4096, 128), fc['op'], 'fc1 weights %s (expected 4096 -> 128)' % (fc['w'].shape,))
    wd['fc1/weights'], wd['fc1/biases'] = fc['w'], fc['b']
    return wd


# ------------------------------------------------------------------------------------------- YOLOv5s
YOLO_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]     # detectors/yolov5/yolov5s.yaml:7-10


class _Eval:
    """Evaluates the shape-shuffling / element-wise corner of the operator set in numpy: used to CHECK what a model file's Focus slicing and
    Detect tail compute (whatever spelling the exporter chose) against the arithmetic csrc/nets.hip runs in their place, on probe tensors."""
    KINDS = ('STRIDED_SLICE', 'CONCATENATION', 'RESHAPE', 'TRANSPOSE', 'LOGISTIC', 'MUL', 'ADD', 'SUB', 'DIV', 'SQUARE', 'DEQUANTIZE')

    def __init__(self, g):
        self.g, self.vals = g, {}

    def value(self, t):
        if t in self.vals:
            return self.vals[t]
        ten = self.g.tensors[t]
        if ten.data is not None:
            return np.asarray(ten.data)
        op = self.g.made_by(t)
        _need(op is not None, self.g.ops[0], 'tensor %s has no producer and no data' % ten.name)
        _need(op.kind in self.KINDS, op, 'not an operator the Focus / Detect check evaluates (%s)' % ', '.join(self.KINDS))
        a = [self.value(i) for i in op.inputs]
        k = op.kind
        if k == 'DEQUANTIZE':
            v = a[0].astype(np.float32)
        elif k == 'LOGISTIC':
            v = (1.0 / (1.0 + np.exp(-a[0].astype(np.float64)))).astype(np.float32)
        elif k in ('MUL', 'ADD', 'SUB', 'DIV'):
            _need(op.options.get('act', 'none') == 'none', op, 'fused activation')
            x, y = a[0].astype(np.float32), a[1].astype(np.float32)
            v = x * y if k == 'MUL' else x + y if k == 'ADD' else x - y if k == 'SUB' else x / y
        elif k == 'SQUARE':
            v = a[0] * a[0]
        elif k == 'CONCATENATION':
            v = np.concatenate(a, axis=op.options['axis'])
        elif k == 'RESHAPE':
            shape = [int(s) for s in (a[1].reshape(-1) if len(a) > 1 else op.options['new_shape'])]
            v = a[0].reshape(shape)
        elif k == 'TRANSPOSE':
            v = np.transpose(a[0], [int(p) for p in a[1].reshape(-1)])
        else:                                                       # STRIDED_SLICE (no ellipsis / new-axis / shrink masks: the reader reads none)
            begin, end, strides = ([int(p) for p in q.reshape(-1)] for q in a[1:4])
            bm, em = op.options.get('begin_mask', 0), op.options.get('end_mask', 0)
            idx = tuple(slice(None if bm >> d & 1 else begin[d], None if em >> d & 1 else end[d], strides[d]) for d in range(a[0].ndim))
            v = a[0][idx]
        self.vals[t] = v
        return v


def load_yolov5s(path):
    """`detectors/yolov5/yolov5s-fp16.tflite` (tools/yolov5.py:68-79 upstream: float32 [1, s, s, 3] pixels 0..255 in, [1, rows, 85] out) -> the
    named weights deepdish_amd/nets.compile_yolov5s takes (batch norms folded as the converter leaves them: `<layer>/weights` + `<layer>/biases`).

    The graph is walked BACKWARDS from its output along detectors/yolov5/yolov5s.yaml:12-48 (depth 0.33, width 0.50): Detect's three branches,
    then C3 / Conv / Concat / Upsample / SPP / Focus, every tensor reached by two routes (the backbone maps the head concatenates) checked to be
    the same one.  A Conv is [PAD +] CONV_2D (float, or float16 filters behind DEQUANTIZE) + SiLU spelt LOGISTIC x MUL; a Bottleneck's skip an ADD;
    SPP three SAME stride-1 MAX_POOL_2D of 5 / 9 / 13; Upsample a 2x RESIZE_NEAREST_NEIGHBOR.  What the Focus slicing and each Detect tail
    COMPUTE is not matched by spelling but evaluated in numpy on probe tensors and held to the arithmetic the kernels implement (slice order
    (0,0) (1,0) (0,1) (1,1); sigmoid, (2 s - 0.5 + grid) stride / size, (2 s)^2 anchor / size, rows anchor-major).  The file is absent from the
    reference tree (.MISSING_LARGE_BLOBS): exercised on files tflite_writer.write_yolov5s lays out like the YOLOv5 TensorFlow export; anything
    else raises UnsupportedModel naming the operator -- a HARD_SWISH (older exports), an int8 file, a per-class NMS tail, another width."""
    g = read(path)
    _need(len(g.inputs) == 1 and len(g.outputs) == 1, g.ops[0], '%d inputs / %d outputs' % (len(g.inputs), len(g.outputs)))
    x_in = g.tensors[g.inputs[0]]
    _need(x_in.dtype == np.float32, g.ops[0], 'input of type %s (the int8 file, tools/yolov5.py:61,102-104, is not built: float / fp16-weight files are)'
          % (np.dtype(x_in.dtype).name if x_in.dtype else '?'))
    _need(len(x_in.shape) == 4 and x_in.shape[1] == x_in.shape[2] and x_in.shape[3] == 3 and x_in.shape[1] % 32 == 0, g.ops[0], 'input %s (expected [1, s, s, 3], s a multiple of 32)' % (x_in.shape,))
    size = int(x_in.shape[1])
    out_t = g.tensors[g.outputs[0]]
    wd, seen = {}, {}

    def const(i):
        if i < 0:
            return None
        t = g.tensors[i]
        if t.data is not None:
            return np.asarray(t.data)
        op = g.made_by(i)
        if op is not None and op.kind == 'DEQUANTIZE' and g.tensors[op.inputs[0]].data is not None:
            return np.asarray(g.tensors[op.inputs[0]].data).astype(np.float32)
        return None

    def raw_conv(name, op, k, s):
        """CONV_2D `op` -> (weights HWIO, bias, the tensor it reads through an optional PAD); k, s as the yaml says."""
        _need(op is not None and op.kind == 'CONV_2D', op or g.ops[0], 'expected the CONV_2D of %s' % name)
        w, b = const(op.inputs[1]), const(op.inputs[2]) if len(op.inputs) > 2 else None
        _need(w is not None and b is not None and w.dtype == np.float32, op, '%s: float filter and bias constants' % name)
        o = op.options
        _need(w.shape[1] == w.shape[2] == k and o['stride_w'] == o['stride_h'] == s and o['dilation_w'] == 1 and o['act'] == 'none', op,
              '%s: %dx%d stride %d filter (yolov5s.yaml wants %dx%d stride %d, no fused activation)' % (name, w.shape[1], w.shape[2], o['stride_w'], k, k, s))
        src = op.inputs[0]
        if s == 1:
            _need(o['padding'] == 'SAME', op, '%s: stride-1 convolutions are SAME' % name)
        else:                                                       # the kernels pad k // 2 on every side (autopad): ZeroPadding2D + VALID in the export
            pad = g.made_by(src)
            _need(o['padding'] == 'VALID' and pad is not None and pad.kind == 'PAD', op, '%s: a stride-2 layer must be PAD + VALID CONV_2D (SAME pads one side only)' % name)
            pv = const(pad.inputs[1])
            _need(pv is not None and [int(v) for v in pv.reshape(-1)] == [0, 0, k // 2, k // 2, k // 2, k // 2, 0, 0], pad, '%s: paddings %s' % (name, None if pv is None else pv.tolist()))
            src = pad.inputs[0]
        return np.transpose(w, (1, 2, 3, 0)).astype(np.float32), b.astype(np.float32).reshape(-1), src

    def conv(name, t, k=1, s=1):
        """t = output of an activated Conv (yaml `Conv`: conv + bn + SiLU) -> the tensor it reads."""
        if name in seen:
            _need(seen[name][0] == t, g.made_by(t) or g.ops[0], '%s is reached by two routes that disagree' % name)
            return seen[name][1]
        op = g.made_by(t)
        _need(op is not None, g.ops[0], '%s: no producer' % name)
        _need(op.kind != 'HARD_SWISH', op, '%s: Hardswish activation (an older export): the kernels implement SiLU' % name)
        _need(op.kind == 'MUL' and len(op.inputs) == 2, op, '%s: expected SiLU spelt x * LOGISTIC(x)' % name)
        a, b = op.inputs
        la, lb = g.made_by(a), g.made_by(b)
        if lb is not None and lb.kind == 'LOGISTIC' and lb.inputs[0] == a:
            pre = a
        elif la is not None and la.kind == 'LOGISTIC' and la.inputs[0] == b:
            pre = b
        else:
            _need(False, op, '%s: expected SiLU spelt x * LOGISTIC(x)' % name)
        w, bias, src = raw_conv(name, g.made_by(pre), k, s)
        wd[name + '/weights'], wd[name + '/biases'] = w, bias
        seen[name] = (t, src)
        return src

    def cat_inputs(t, n, what):
        op = g.made_by(t)
        _need(op is not None and op.kind == 'CONCATENATION' and len(op.inputs) == n and op.options['axis'] in (3, -1), op or g.ops[0], 'expected the channel CONCATENATION of %s' % what)
        return list(op.inputs)

    def c3(name, t, n, shortcut):
        cat = conv(name + '.cv3', t)
        y, z = cat_inputs(cat, 2, name)
        src2 = conv(name + '.cv2', z)
        for i in reversed(range(n)):
            if shortcut:
                add = g.made_by(y)
                _need(add is not None and add.kind == 'ADD' and len(add.inputs) == 2, add or g.ops[0], '%s.m%d: expected the shortcut ADD' % (name, i))
                # one operand is the bottleneck's input, the other its cv2 output (which leads back to that input)
                ok = False
                for skip, branch in (add.inputs, add.inputs[::-1]):
                    m = g.made_by(branch)
                    if m is None or m.kind != 'MUL':
                        continue
                    mark = (dict(wd), dict(seen))
                    try:
                        h = conv('%s.m%d.cv2' % (name, i), branch, 3)
                        if conv('%s.m%d.cv1' % (name, i), h) == skip:
                            ok = True
                            break
                    except UnsupportedModel:
                        pass
                    wd.clear(); wd.update(mark[0]); seen.clear(); seen.update(mark[1])
                _need(ok, add, '%s.m%d: the ADD does not join a bottleneck with its input' % (name, i))
                y = skip
            else:
                y = conv('%s.m%d.cv1' % (name, i), conv('%s.m%d.cv2' % (name, i), y, 3))
        src1 = conv(name + '.cv1', y)
        _need(src1 == src2, g.made_by(t), '%s: cv1 and cv2 read different tensors' % name)
        return src1

    def upsample(t, what):
        op = g.made_by(t)
        _need(op is not None and op.kind == 'RESIZE_NEAREST_NEIGHBOR', op or g.ops[0], 'expected the 2x nearest-neighbour upsample of %s' % what)
        sz = const(op.inputs[1])
        src = g.tensors[op.inputs[0]]
        _need(sz is not None and [int(v) for v in sz.reshape(-1)] == [2 * src.shape[1], 2 * src.shape[2]], op, 'upsample of %s to %s (2x is built)' % (src.shape, None if sz is None else sz.tolist()))
        return op.inputs[0]

    # ---- Detect: three branches concatenated along the rows
    _need(len(out_t.shape) == 3 and out_t.shape[2] > 5, g.ops[-1], 'output %s (expected [1, rows, 5 + classes])' % (out_t.shape,))
    no = int(out_t.shape[2])
    top = g.made_by(out_t.index)
    _need(top is not None and top.kind == 'CONCATENATION' and len(top.inputs) == 3 and top.options['axis'] == 1, top or g.ops[-1],
          'expected the CONCATENATION of the three Detect layers along the rows (a file with NMS inside, or another head, is not built)')
    feats = []
    for i, z in enumerate(top.inputs):
        # back along the first operands to the 1x1 CONV_2D with 3 * (5 + classes) outputs
        t, hops, head = z, 0, None
        while hops < 64:
            op = g.made_by(t)
            _need(op is not None, top, 'Detect branch %d does not start at a convolution' % i)
            if op.kind == 'CONV_2D':
                head = op
                break
            t = op.inputs[0]
            if op.kind == 'CONCATENATION':                         # [xy, wh, rest]: any of them leads back
                t = op.inputs[0]
            hops += 1
        _need(head is not None, top, 'Detect branch %d: no convolution within 64 operators' % i)
        w, b, src = raw_conv('detect%d' % i, head, 1, 1)
        _need(w.shape[3] == 3 * no, head, 'detect%d: %d output channels (3 anchors x %d)' % (i, w.shape[3], no))
        hw = int(g.tensors[head.outputs[0]].shape[1])
        _need(size % hw == 0 and g.tensors[head.outputs[0]].shape[2] == hw, head, 'detect%d: %s map' % (i, g.tensors[head.outputs[0]].shape))
        stride = size // hw
        # what the tail computes, on a probe: csrc/nets.hip's Detect epilogue restated (conv_epilogue, EPI_YOLO)
        rng = np.random.default_rng(100 + i)
        probe = rng.standard_normal((1, hw, hw, 3 * no)).astype(np.float32) * 2.0
        ev = _Eval(g)
        ev.vals[head.outputs[0]] = probe
        got = ev.value(z)
        s = 1.0 / (1.0 + np.exp(-probe.astype(np.float64).reshape(hw * hw, 3, no)))
        gy, gx = np.meshgrid(np.arange(hw), np.arange(hw), indexing='ij')
        grid = np.stack([gx, gy], axis=-1).reshape(hw * hw, 1, 2)
        anc = np.array(YOLO_ANCHORS[i], np.float64).reshape(1, 3, 2)
        want = s.copy()
        want[..., 0:2] = (s[..., 0:2] * 2 - 0.5 + grid) * stride / size
        want[..., 2:4] = (s[..., 2:4] * 2) ** 2 * anc / size
        want = np.transpose(want, (1, 0, 2)).reshape(1, 3 * hw * hw, no)           # rows anchor-major inside a layer
        _need(got.shape == want.shape and np.allclose(got, want, rtol=1e-4, atol=1e-5), head,
              'detect%d: the operators behind this convolution do not compute the YOLOv5 box decode with anchors %s at stride %d' % (i, YOLO_ANCHORS[i], stride))
        wd['detect%d/weights' % i], wd['detect%d/biases' % i] = w, b
        feats.append(src)
    p3, p4, p5 = feats
    # ---- head and backbone, backwards (yolov5s.yaml:27-47, then :14-24)
    m21, m10 = cat_inputs(c3('m23', p5, 1, False), 2, 'm22')
    _need(conv('m21', m21, 3, 2) == p4, g.made_by(m21), 'm21 does not read the P4 output')
    m18, m14 = cat_inputs(c3('m20', p4, 1, False), 2, 'm19')
    _need(conv('m18', m18, 3, 2) == p3, g.made_by(m18), 'm18 does not read the P3 output')
    u14, x4 = cat_inputs(c3('m17', p3, 1, False), 2, 'm16')
    _need(upsample(u14, 'm14') == m14, g.made_by(u14), 'the P3 concatenation does not take the upsampled m14')
    u10, x6 = cat_inputs(c3('m13', conv('m14', m14), 1, False), 2, 'm12')
    _need(upsample(u10, 'm10') == m10, g.made_by(u10), 'the P4 concatenation does not take the upsampled m10')
    spp_out = c3('m9', conv('m10', m10), 1, False)
    pools = cat_inputs(conv('m8.cv2', spp_out), 4, 'SPP')
    for j, k in enumerate((5, 9, 13)):
        mp = g.made_by(pools[j + 1])
        _need(mp is not None and mp.kind == 'MAX_POOL_2D' and mp.inputs[0] == pools[0] and mp.options['filter_w'] == mp.options['filter_h'] == k and
              mp.options['stride_w'] == 1 and mp.options['padding'] == 'SAME', mp or g.ops[0], 'SPP: expected a SAME stride-1 %dx%d max pool of the cv1 output' % (k, k))
    _need(conv('m7', conv('m8.cv1', pools[0]), 3, 2) == x6, g.made_by(pools[0]), 'm7 does not read the map the P4 concatenation takes (m6)')
    _need(conv('m5', c3('m6', x6, 3, True), 3, 2) == x4, g.made_by(x6), 'm5 does not read the map the P3 concatenation takes (m4)')
    focus_out = conv('m0.focus', conv('m1', c3('m2', conv('m3', c3('m4', x4, 3, True), 3, 2), 1, True), 3, 2), 3)
    # ---- Focus: evaluated on an index image
    probe = np.arange(8 * 8 * 3, dtype=np.float32).reshape(1, 8, 8, 3)
    ev = _Eval(g)
    ev.vals[g.inputs[0]] = probe
    fop = g.made_by(focus_out)
    _need(fop is not None, g.ops[0], 'the first convolution reads the graph input directly (no Focus slicing)')
    got = ev.value(focus_out)
    want = np.concatenate([probe[:, 0::2, 0::2], probe[:, 1::2, 0::2], probe[:, 0::2, 1::2], probe[:, 1::2, 1::2]], axis=3)
    _need(got.shape == want.shape and np.array_equal(got, want), fop, 'the operators in front of the first convolution are not the Focus slicing (0,0) (1,0) (0,1) (1,1)')
    # ---- the widths are yolov5s' (the kernels' program is compiled for them)
    from .. import nets
    for name, k, cin, cout in nets.yolov5s_convs():
        _need(name + '/weights' in wd and wd[name + '/weights'].shape == (k, k, cin, cout), g.ops[0],
              '%s: filter %s (YOLOv5s, width multiple 0.50, has %s)' % (name, wd.get(name + '/weights', np.zeros(0)).shape, (k, k, cin, cout)))
    _need(no == 5 + nets.YOLO_NC, top, '%d values per row (80 classes are built)' % no)
    wd['__in_size__'] = size
    return wd
