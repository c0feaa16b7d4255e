"""Writes `.tflite` flatbuffers (schema version 3) -- the export side of deepdish_amd/tools/tflite_reader.py, from the same public
schema: a QModel (deepdish_amd/quantize.py) or named float weights become a file `SSD_MOBILENET(model_file=...)` loads like the
reference's own detector file.  Used by the tests to produce fixtures (the reference's blobs are absent) and by anyone who wants
the quantiser's output in the interchange format.
"""
import numpy as np

from . import flatbuf
from .. import nets

TYPE_CODE = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.int32): 2, np.dtype(np.uint8): 3, np.dtype(np.int64): 4, np.dtype(np.int8): 9}
OP_CODE = {'DEQUANTIZE': 6, 'CONCATENATION': 2, 'CONV_2D': 3, 'DEPTHWISE_CONV_2D': 4, 'LOGISTIC': 14, 'RESHAPE': 22, 'CUSTOM': 32, 'MAX_POOL_2D': 17, 'ADD': 0,
           'MUL': 18, 'FULLY_CONNECTED': 9, 'ELU': 111, 'REVERSE_V2': 105, 'SUM': 74, 'SQRT': 75, 'DIV': 42, 'PAD': 34, 'TRANSPOSE': 39, 'SUB': 41,
           'STRIDED_SLICE': 45, 'RESIZE_NEAREST_NEIGHBOR': 97, 'HARD_SWISH': 117}
OPTIONS_TYPE = {'CONV_2D': 1, 'DEPTHWISE_CONV_2D': 2, 'CONCATENATION': 10, 'RESHAPE': 17, 'MAX_POOL_2D': 5, 'FULLY_CONNECTED': 8, 'ADD': 11, 'MUL': 21,
                'DIV': 29, 'SUM': 27, 'SUB': 28, 'STRIDED_SLICE': 32, 'RESIZE_NEAREST_NEIGHBOR': 74}
ACT = {'none': 0, 'relu': 1, 'relu6': 3}


def metadata_bytes(mean, std, label_file):
    """The metadata flatbuffer (schema tflite_metadata, file identifier M001), the fields the reference reads: ModelMetadata.subgraph_metadata[0]
    .input_tensor_metadata[0].process_units[] with a NormalizationOptions unit (mean, std: [f32]) and, on the output tensor, the associated
    label file's name.  Slots: ModelMetadata {0 name, 1 description, 2 version, 3 subgraph_metadata}; SubGraphMetadata {0 name, 2 input_tensor_metadata,
    3 output_tensor_metadata}; TensorMetadata {0 name, 4 process_units, 6 associated_files}; ProcessUnit {0 options_type (1 = NormalizationOptions),
    1 options}; NormalizationOptions {0 mean, 1 std}; AssociatedFile {0 name, 2 type (2 = TENSOR_AXIS_LABELS... 3 = TENSOR_VALUE_LABELS)}."""
    b = flatbuf.Builder()
    units = []
    if mean is not None:
        norm = b.table({0: ('offset', b.scalars([float(v) for v in np.atleast_1d(mean)], 'f32')), 1: ('offset', b.scalars([float(v) for v in np.atleast_1d(std)], 'f32'))})
        units.append(b.table({0: ('u8', 1), 1: ('offset', norm)}))
    tin = b.table({0: ('offset', b.string('image')), 4: ('offset', b.offsets(units) if units else 0)})
    outs = []
    for k, name in enumerate(('location', 'category', 'score', 'number of detections')):
        files = 0
        if k == 1 and label_file is not None:
            files = b.offsets([b.table({0: ('offset', b.string(label_file)), 2: ('i8', 3)})])
        outs.append(b.table({0: ('offset', b.string(name)), 6: ('offset', files)}))
    sub = b.table({0: ('offset', b.string('main')), 2: ('offset', b.offsets([tin])), 3: ('offset', b.offsets(outs))})
    root = b.table({0: ('offset', b.string('ObjectDetector')), 2: ('offset', b.string('v1')), 3: ('offset', b.offsets([sub]))})
    return b.finish(root, b'M001')


def packed_files(files):
    """{name: text} -> the ZIP archive the metadata tooling appends to the model file (stored, no compression)."""
    import io
    import zipfile
    bio = io.BytesIO()
    with zipfile.ZipFile(bio, 'w', zipfile.ZIP_STORED) as z:
        for name, text in files.items():
            z.writestr(name, text)
    return bio.getvalue()


class GraphWriter:
    def __init__(self, description='deepdish_amd'):
        self.tensors, self.ops, self.buffers, self.codes = [], [], [b''], []
        self.inputs, self.outputs, self.description = [], [], description
        self.metadata = None              # dict(mean=[..], std=[..] (None: no NormalizationOptions unit), labels=[..], label_file='labelmap.txt'): see metadata_bytes

    def tensor(self, name, shape, dtype, data=None, scale=None, zero_point=None):
        buf = 0
        if data is not None:
            self.buffers.append(np.ascontiguousarray(data, dtype=dtype).tobytes())
            buf = len(self.buffers) - 1
        self.tensors.append(dict(name=name, shape=[int(v) for v in shape], dtype=np.dtype(dtype), buffer=buf, scale=scale, zero_point=zero_point))
        return len(self.tensors) - 1

    def _code(self, kind, custom=''):
        key = (kind, custom)
        if key not in self.codes:
            self.codes.append(key)
        return self.codes.index(key)

    def op(self, kind, inputs, outputs, options=None, custom='', custom_options=None):
        self.ops.append(dict(code=self._code(kind, custom), kind=kind, inputs=list(inputs), outputs=list(outputs), options=options or {},
                             custom_options=custom_options))

    def _options(self, b, kind, o):
        if kind == 'CONV_2D':
            return b.table({0: ('i8', 1 if o.get('padding', 'SAME') == 'VALID' else 0), 1: ('i32', o['stride']), 2: ('i32', o['stride']), 3: ('i8', ACT[o['act']]), 4: ('i32', 1), 5: ('i32', 1)})
        if kind == 'DEPTHWISE_CONV_2D':
            return b.table({0: ('i8', 0), 1: ('i32', o['stride']), 2: ('i32', o['stride']), 3: ('i32', 1), 4: ('i8', ACT[o['act']]), 5: ('i32', 1), 6: ('i32', 1)})
        if kind == 'CONCATENATION':
            return b.table({0: ('i32', o['axis']), 1: ('i8', 0)})
        if kind == 'RESHAPE':
            return b.table({0: ('offset', b.scalars(o['new_shape'], 'i32'))})
        if kind == 'MAX_POOL_2D':                 # Pool2DOptions: padding (1 = VALID), strides, filter, activation
            return b.table({0: ('i8', 1 if o.get('padding', 'VALID') == 'VALID' else 0), 1: ('i32', o['stride']), 2: ('i32', o['stride']), 3: ('i32', o['k']), 4: ('i32', o['k']), 5: ('i8', 0)})
        if kind == 'STRIDED_SLICE':               # StridedSliceOptions: begin_mask, end_mask, ellipsis_mask, new_axis_mask, shrink_axis_mask
            return b.table({0: ('i32', o.get('begin_mask', 0)), 1: ('i32', o.get('end_mask', 0)), 2: ('i32', 0), 3: ('i32', 0), 4: ('i32', 0)})
        if kind == 'RESIZE_NEAREST_NEIGHBOR':     # ResizeNearestNeighborOptions: align_corners, half_pixel_centers
            return b.table({0: ('bool', False), 1: ('bool', False), 2: ('i8', 0)})
        if kind in ('ADD', 'MUL', 'DIV', 'SUB', 'FULLY_CONNECTED'):
            return b.table({0: ('i8', 0), 1: ('i8', 0)})      # (a second field so that the table is not empty: fused activation NONE)
        if kind == 'SUM':                         # ReducerOptions: keep_dims
            return b.table({0: ('bool', True), 1: ('i8', 0)})
        return 0

    def tobytes(self):
        b = flatbuf.Builder()
        buf_pos = []
        for data in self.buffers:
            buf_pos.append(b.table({0: ('offset', b.vector(data, 1, align=16) if len(data) else 0)}))
        buffers = b.offsets(buf_pos)
        t_pos = []
        for t in self.tensors:
            q = 0
            if t['scale'] is not None:
                sc = b.scalars([float(v) for v in np.atleast_1d(t['scale'])], 'f32')
                zp = b.scalars([int(v) for v in np.atleast_1d(t['zero_point'])], 'i64')
                q = b.table({2: ('offset', sc), 3: ('offset', zp)})
            name, shape = b.string(t['name']), b.scalars(t['shape'], 'i32')
            t_pos.append(b.table({0: ('offset', shape), 1: ('i8', TYPE_CODE[t['dtype']]), 2: ('u32', t['buffer']), 3: ('offset', name), 4: ('offset', q)}))
        o_pos = []
        for o in self.ops:
            opt = self._options(b, o['kind'], o['options'])
            cust = b.vector(o['custom_options'], 1) if o['custom_options'] else 0
            ins, outs = b.scalars(o['inputs'], 'i32'), b.scalars(o['outputs'], 'i32')
            o_pos.append(b.table({0: ('u32', o['code']), 1: ('offset', ins), 2: ('offset', outs), 3: ('u8', OPTIONS_TYPE.get(o['kind'], 0) if opt else 0),
                                  4: ('offset', opt), 5: ('offset', cust)}))
        tensors, operators = b.offsets(t_pos), b.offsets(o_pos)
        ins, outs, name = b.scalars(self.inputs, 'i32'), b.scalars(self.outputs, 'i32'), b.string('main')
        sub = b.table({0: ('offset', tensors), 1: ('offset', ins), 2: ('offset', outs), 3: ('offset', operators), 4: ('offset', name)})
        subs = b.offsets([sub])
        c_pos = []
        for kind, custom in self.codes:
            cs = b.string(custom) if custom else 0
            code = OP_CODE[kind]
            c_pos.append(b.table({0: ('i8', min(code, 127)), 1: ('offset', cs), 2: ('i32', 1), 3: ('i32', code)}))
        codes = b.offsets(c_pos)
        desc = b.string(self.description)
        fields = {0: ('u32', 3), 1: ('offset', codes), 2: ('offset', subs), 3: ('offset', desc), 4: ('offset', buffers)}
        if self.metadata is not None:             # Model.metadata: [Metadata{name, buffer}] -- the buffer (appended last) holds the metadata flatbuffer
            entry = b.table({0: ('offset', b.string('TFLITE_METADATA')), 1: ('u32', len(self.buffers) - 1)})
            fields[6] = ('offset', b.offsets([entry]))
        model = b.table(fields)
        out = b.finish(model, b'TFL3')
        if self.metadata is not None and self.metadata.get('labels') is not None:
            out += packed_files({self.metadata.get('label_file', 'labelmap.txt'): '\n'.join(self.metadata['labels']) + '\n'})
        return out

    def set_metadata(self, mean=None, std=None, labels=None, label_file='labelmap.txt'):
        """TFLite Model Metadata as the Task library's tooling writes it (what tools/tflite_object_detector.py:117-137 upstream reads): a
        `TFLITE_METADATA` buffer -- NormalizationOptions of the input tensor -- and the label file packed behind the flatbuffer as a ZIP archive."""
        self.metadata = dict(mean=mean, std=std, labels=labels, label_file=label_file)
        self.buffers.append(metadata_bytes(mean, std, label_file if labels is not None else None))


def ssd_mobilenet_graph(model, anchors=None, post=None):
    """QModel (uint8) or named float weights (folded: name/weights, name/biases) -> GraphWriter of the SSD-MobileNet-v1 graph as the
    TF Object Detection API exports it: backbone, predictor convolutions, RESHAPE / CONCATENATION fans, LOGISTIC, post-process op."""
    quant = isinstance(model, dict) and model.get('kind') == 'ssd_mobilenet_v1_uint8'
    W = GraphWriter('SSD-MobileNet-v1 (%s), written by deepdish_amd' % ('uint8' if quant else 'float'))
    size = int(model['input']['size']) if quant else 300
    if anchors is None:
        anchors = model.get('anchors') if quant and model.get('anchors') is not None else nets.ssd_anchors(size)[0]
    adt = np.uint8 if quant else np.float32

    def qp(scale, zp):
        return (scale, zp) if quant else (None, None)

    x = W.tensor('normalized_input_image_tensor', [1, size, size, 3], adt, None, *qp(model['input']['scale'], model['input']['zp']) if quant else (None, None))
    W.inputs = [x]
    names = ['conv0'] + [n for i in range(1, 14) for n in (f'dw{i}', f'pw{i}')] + [f'extra{j}_{h}' for j in range(1, 5) for h in (1, 2)]

    def layer(name):
        if quant:
            return model['layers'][name]
        w = model[name + '/weights']
        kind = 'dw' if name.startswith('dw') else 'conv'
        stride = 2 if name in ('conv0', 'dw2', 'dw4', 'dw6', 'dw12') or (name.startswith('extra') and name.endswith('_2')) else 1
        return dict(kind=kind, w=w[:, :, :, 0] if kind == 'dw' else w, bias=model[name + '/biases'], stride=stride,
                    act='none' if name.startswith(('box', 'cls')) else 'relu6')

    def conv(name, src, hw):
        L = layer(name)
        if L['kind'] == 'conv':
            wt, cout = np.transpose(L['w'], (3, 0, 1, 2)), L['w'].shape[3]
        else:
            wt, cout = L['w'][None], L['w'].shape[2]
        ho = -(-hw // L['stride'])
        fw = W.tensor(name + '/weights', wt.shape, adt, wt, *(qp(L['w_scale'], L['w_zp']) if quant else (None, None)))
        bs = W.tensor(name + '/bias', [cout], np.int32 if quant else np.float32, L['bias'],
                      *((np.float32(L['in_scale']) * np.float32(L['w_scale']), 0) if quant else (None, None)))
        out = W.tensor(name, [1, ho, ho, cout], adt, None, *(qp(L['out_scale'], L['out_zp']) if quant else (None, None)))
        W.op('CONV_2D' if L['kind'] == 'conv' else 'DEPTHWISE_CONV_2D', [src, fw, bs], [out], dict(stride=L['stride'], act=L['act']))
        return out, ho

    t, hw = x, size
    out_of = {}
    for name in names:
        t, hw = conv(name, t, hw)
        out_of[name] = (t, hw)
    boxes, classes = [], []
    n_cls = (model['layers']['cls0']['w'].shape[3] if quant else model['cls0/weights'].shape[3]) // nets.SSD_ANCHORS_PER_MAP[0]
    for k, (f, a) in enumerate(zip(['pw11', 'pw13', 'extra1_2', 'extra2_2', 'extra3_2', 'extra4_2'], nets.SSD_ANCHORS_PER_MAP)):
        ft, fm = out_of[f]
        for kind, per, acc in (('box', 4, boxes), ('cls', n_cls, classes)):
            c, _ = conv(f'{kind}{k}', ft, fm)
            L = layer(f'{kind}{k}')
            r = W.tensor(f'{kind}{k}/reshape', [1, fm * fm * a, per], adt, None, *(qp(L['out_scale'], L['out_zp']) if quant else (None, None)))
            W.op('RESHAPE', [c], [r], dict(new_shape=[1, fm * fm * a, per]))
            acc.append(r)
    n_anchors = len(anchors)
    Lb, Lc = layer('box0'), layer('cls0')
    bcat = W.tensor('concat', [1, n_anchors, 4], adt, None, *(qp(Lb['out_scale'], Lb['out_zp']) if quant else (None, None)))
    W.op('CONCATENATION', boxes, [bcat], dict(axis=1))
    ccat = W.tensor('concat_1', [1, n_anchors, n_cls], adt, None, *(qp(Lc['out_scale'], Lc['out_zp']) if quant else (None, None)))
    W.op('CONCATENATION', classes, [ccat], dict(axis=1))
    sig = W.tensor('convert_scores', [1, n_anchors, n_cls], adt, None,
                   *(qp(model['logistic']['out_scale'], model['logistic']['out_zp']) if quant else (None, None)))
    W.op('LOGISTIC', [ccat], [sig])
    anc = W.tensor('anchors', [n_anchors, 4], np.float32, np.asarray(anchors, np.float32))
    opts = dict(max_detections=10, max_classes_per_detection=1, nms_score_threshold=1e-8, nms_iou_threshold=0.6,
                num_classes=n_cls - 1, y_scale=10.0, x_scale=10.0, h_scale=5.0, w_scale=5.0, use_regular_nms=False)
    for k, v in (post or {}).items():                      # overrides; None drops the key (a file that does not state it)
        if v is None:
            opts.pop(k, None)
        else:
            opts[k] = v
    nd = int(opts.get('max_detections', 10))
    outs = [W.tensor('TFLite_Detection_PostProcess' + (':%d' % i if i else ''), s, np.float32) for i, s in enumerate(([1, nd, 4], [1, nd], [1, nd], [1]))]
    W.op('CUSTOM', [bcat, sig, anc], outs, custom='TFLite_Detection_PostProcess', custom_options=flatbuf.flex_build_map(opts))
    W.outputs = outs
    return W


def mars_graph(wd, reverse_channels=True, half_weights=False, explicit_pad=False):
    """Named float weights of the MARS encoder (deepdish_amd/nets.synthetic_mars_weights or the arrays of an .npz; batch norms raw or folded)
    -> GraphWriter of the graph as tools/tflite_reader.load_mars documents it: channel reversal, CONV_2D with the batch norm in filter and
    bias + ELU, block batch norms as MUL + ADD, skip ADDs, pool1, FULLY_CONNECTED, "ball", the unit-length tail as MUL / SUM / ADD / SQRT / DIV."""
    W = GraphWriter('MARS 64x32x3 encoder (float), written by deepdish_amd')
    f32 = np.float32
    x = W.tensor('images', [1, 64, 32, 3], f32)
    W.inputs = [x]
    cnt = [0]

    def T(shape, name=None):
        cnt[0] += 1
        return W.tensor(name or 't%d' % cnt[0], shape, f32)

    def const(name, arr, dtype=f32):
        a = np.ascontiguousarray(arr, dtype=dtype)
        return W.tensor(name, list(a.shape), dtype, a)

    def conv(name, src, hw, w_hwio, bias, stride):
        h, w_ = hw
        ho, wo = -(-h // stride), -(-w_ // stride)
        cout = w_hwio.shape[3]
        if half_weights:                                                # a float16 weight file: the filter behind a DEQUANTIZE
            hw_ = const(name + '/weights_f16', np.transpose(w_hwio, (3, 0, 1, 2)), np.float16)
            fw = T(list(np.transpose(w_hwio, (3, 0, 1, 2)).shape), name + '/weights')
            W.op('DEQUANTIZE', [hw_], [fw])
        else:
            fw = const(name + '/weights', np.transpose(w_hwio, (3, 0, 1, 2)))
        padding = 'SAME'
        if explicit_pad and stride == 2:                                # the same zeros SAME adds, spelt PAD + VALID (another converter's habit)
            kk = w_hwio.shape[0]
            pv = []
            for d in (h, w_):
                tot = max((-(-d // stride) - 1) * stride + kk - d, 0)
                pv.append([tot // 2, tot - tot // 2])
            padded = T([1, h + sum(pv[0]), w_ + sum(pv[1]), w_hwio.shape[2]])
            W.op('PAD', [src, const(name + '/paddings', [[0, 0], pv[0], pv[1], [0, 0]], np.int32)], [padded])
            src, padding = padded, 'VALID'
        ins = [src, fw] + ([const(name + '/bias', bias)] if bias is not None else [-1])
        out = T([1, ho, wo, cout], name)
        W.op('CONV_2D', ins, [out], dict(stride=stride, act='none', padding=padding))
        return out, (ho, wo)

    def elu(src, shape):
        out = T(shape)
        W.op('ELU', [src], [out])
        return out

    def affine(name, src, shape, sc, sh):
        m = T(shape)
        W.op('MUL', [src, const(name + '/scale', sc)], [m])
        a = T(shape)
        W.op('ADD', [m, const(name + '/shift', sh)], [a])
        return a

    rev = x
    if reverse_channels:                                               # the frozen graph's BGR -> RGB (tools/freeze_model.py:175-177)
        rev = T([1, 64, 32, 3])
        W.op('REVERSE_V2', [x, const('axis', [3], np.int32)], [rev])
    hw = (64, 32)
    w, b = nets.fold_conv_bn(wd, 'conv1_1'); t, hw = conv('conv1_1', rev, hw, w, b, 1); t = elu(t, [1, hw[0], hw[1], 32])
    w, b = nets.fold_conv_bn(wd, 'conv1_2'); t, hw = conv('conv1_2', t, hw, w, b, 1); t = elu(t, [1, hw[0], hw[1], 32])
    hw = ((hw[0] - 3) // 2 + 1, (hw[1] - 3) // 2 + 1)
    p = T([1, hw[0], hw[1], 32], 'pool1')
    W.op('MAX_POOL_2D', [t], [p], dict(k=3, stride=2, padding='VALID'))
    raw, cin = p, 32
    for name, c, inc, first in nets.MARS_BLOCKS:
        pre = raw
        if not first:
            sc, sh = nets.bn_affine(wd, name + '/bn')
            pre = elu(affine(name + '/bn', raw, [1, hw[0], hw[1], cin], sc, sh), [1, hw[0], hw[1], cin])
        w, b = nets.fold_conv_bn(wd, name + '/1')
        h1, hw2 = conv(name + '/1', pre, hw, w, b, 2 if inc else 1)
        h1 = elu(h1, [1, hw2[0], hw2[1], c])
        h2, _ = conv(name + '/2', h1, hw2, wd[name + '/2/weights'], wd[name + '/2/biases'], 1)
        skip = raw
        if inc:
            skip, _ = conv(name + '/projection', raw, hw, wd[name + '/projection/weights'], None, 2)
        out = T([1, hw2[0], hw2[1], c], name + '/add')
        W.op('ADD', [skip, h2], [out])
        raw, hw, cin = out, hw2, c
    flat = T([1, hw[0] * hw[1] * cin])
    W.op('RESHAPE', [raw], [flat], dict(new_shape=[1, hw[0] * hw[1] * cin]))
    w, b = nets.fold_conv_bn(wd, 'fc1', 'fc1/bn')                       # [4096, 128]
    fc = T([1, 128], 'fc1')
    W.op('FULLY_CONNECTED', [flat, const('fc1/weights', np.ascontiguousarray(w.T)), const('fc1/bias', b)], [fc])
    f = elu(fc, [1, 128])
    sc, sh = nets.bn_affine(wd, 'ball')
    f = affine('ball', f, [1, 128], sc, sh)
    sq = T([1, 128]); W.op('MUL', [f, f], [sq])
    sm = T([1, 1]); W.op('SUM', [sq, const('sum_axis', [1], np.int32)], [sm])
    ep = T([1, 1]); W.op('ADD', [sm, const('eps', [1e-8])], [ep])
    nr = T([1, 1]); W.op('SQRT', [ep], [nr])
    out = T([1, 128], 'features'); W.op('DIV', [f, nr], [out])
    W.outputs = [out]
    return W


def write_mars(wd, path, reverse_channels=True, half_weights=False, explicit_pad=False):
    data = mars_graph(wd, reverse_channels, half_weights, explicit_pad).tobytes()
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)


def write_ssd_mobilenet(model, path, anchors=None, post=None, metadata=None):
    """post: overrides of the TFLite_Detection_PostProcess options ({name: value}; None drops the key).  metadata: dict(mean, std, labels
    [, label_file]) -> TFLite Model Metadata + the packed label file (GraphWriter.set_metadata), as the generic adaptor's models carry them."""
    g = ssd_mobilenet_graph(model, anchors, post=post)
    if metadata is not None:
        g.set_metadata(**metadata)
    data = g.tobytes()
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)


# ------------------------------------------------------------------------------------------- YOLOv5s
def yolov5s_graph(wd, in_size=640, half_weights=False, activation='silu'):
    """Named YOLOv5s weights (deepdish_amd/nets.synthetic_yolov5s_weights; batch norms raw or folded) -> GraphWriter of the graph as the
    YOLOv5 TensorFlow export (models/tf.py of the YOLOv5 repository, which produced the reference's `detectors/yolov5/yolov5s-fp16.tflite`,
    tools/yolov5.py:59-79 upstream) lays it out: Focus as four STRIDED_SLICEs + CONCATENATION, every Conv as [PAD +] CONV_2D with the batch
    norm folded + SiLU spelt LOGISTIC + MUL, C3 / Bottleneck / SPP (MAX_POOL_2D 5 / 9 / 13, SAME) / RESIZE_NEAREST_NEIGHBOR / CONCATENATION
    per detectors/yolov5/yolov5s.yaml:12-48, and Detect as RESHAPE + TRANSPOSE + LOGISTIC + the box arithmetic on slices, normalised by the
    image size, the three layers concatenated to [1, 25200, 85].  half_weights: filters as float16 constants behind DEQUANTIZE (an fp16 file).
    activation='hardswish' writes the older export's activation (a file the reader must refuse by name)."""
    W = GraphWriter('YOLOv5s (float%s), written by deepdish_amd' % ('16 weights' if half_weights else '32'))
    f32 = np.float32
    x = W.tensor('input_1', [1, in_size, in_size, 3], f32)
    W.inputs = [x]
    cnt = [0]

    def T(shape, name=None):
        cnt[0] += 1
        return W.tensor(name or 't%d' % cnt[0], shape, f32)

    def const(name, arr, dtype=f32):
        a = np.ascontiguousarray(arr, dtype=dtype)
        return W.tensor(name, list(a.shape), dtype, a)

    def filt(name, w_ohwi):
        if half_weights:
            h = const(name + '/weights_f16', w_ohwi, np.float16)
            f = T(list(w_ohwi.shape), name + '/weights')
            W.op('DEQUANTIZE', [h], [f])
            return f
        return const(name + '/weights', w_ohwi)

    def act(t, shape):
        out = T(shape)
        if activation == 'hardswish':
            W.op('HARD_SWISH', [t], [out])
            return out
        sg = T(shape)
        W.op('LOGISTIC', [t], [sg])
        W.op('MUL', [t, sg], [out])
        return out

    def conv_raw(name, src, hw, w_hwio, bias, k, s):
        """Conv2D as models/tf.py TFConv: stride 1 -> SAME; stride 2 -> ZeroPadding2D(k // 2) + VALID."""
        h = hw
        cout = w_hwio.shape[3]
        pad = 'SAME'
        if s == 2:
            padded = T([1, h + 2 * (k // 2), h + 2 * (k // 2), w_hwio.shape[2]])
            W.op('PAD', [src, const(name + '/paddings', [[0, 0], [k // 2, k // 2], [k // 2, k // 2], [0, 0]], np.int32)], [padded])
            src, pad = padded, 'VALID'
            ho = (h + 2 * (k // 2) - k) // 2 + 1
        else:
            ho = h
        out = T([1, ho, ho, cout], name)
        W.op('CONV_2D', [src, filt(name, np.transpose(w_hwio, (3, 0, 1, 2))), const(name + '/bias', bias)], [out], dict(stride=s, act='none', padding=pad))
        return out, ho

    def cv(name, src, hw, k=1, s=1):
        w, b = nets.fold_conv_bn(wd, name)
        out, ho = conv_raw(name, src, hw, w, b, k, s)
        return act(out, [1, ho, ho, w.shape[3]]), ho, w.shape[3]

    def c3(name, src, hw, n, shortcut):
        y, _, c_ = cv(name + '.cv1', src, hw)
        for i in range(n):
            h, _, _ = cv('%s.m%d.cv1' % (name, i), y, hw)
            h, _, _ = cv('%s.m%d.cv2' % (name, i), h, hw, 3)
            if shortcut:
                a = T([1, hw, hw, c_])
                W.op('ADD', [y, h], [a])
                y = a
            else:
                y = h
        z, _, _ = cv(name + '.cv2', src, hw)
        cat = T([1, hw, hw, 2 * c_])
        W.op('CONCATENATION', [y, z], [cat], dict(axis=3))
        return cv(name + '.cv3', cat, hw)

    def up(src, hw, c):
        out = T([1, 2 * hw, 2 * hw, c])
        W.op('RESIZE_NEAREST_NEIGHBOR', [src, const('size%d' % cnt[0], [2 * hw, 2 * hw], np.int32)], [out])
        return out

    def cat(a, b, hw, c):
        out = T([1, hw, hw, c])
        W.op('CONCATENATION', [a, b], [out], dict(axis=3))
        return out

    # Focus (models/tf.py TFFocus): x[:, ::2, ::2], x[:, 1::2, ::2], x[:, ::2, 1::2], x[:, 1::2, 1::2] along the channels
    h2 = in_size // 2
    sl = []
    for (r0, c0) in ((0, 0), (1, 0), (0, 1), (1, 1)):
        o = T([1, h2, h2, 3])
        W.op('STRIDED_SLICE', [x, const('begin%d%d' % (r0, c0), [0, r0, c0, 0], np.int32), const('end%d%d' % (r0, c0), [0, 0, 0, 0], np.int32),
                               const('strides%d%d' % (r0, c0), [1, 2, 2, 1], np.int32)], [o], dict(begin_mask=0b1001, end_mask=0b1111))
        sl.append(o)
    f = T([1, h2, h2, 12])
    W.op('CONCATENATION', sl, [f], dict(axis=3))
    t, hw, _ = cv('m0.focus', f, h2, 3)
    t, hw, _ = cv('m1', t, hw, 3, 2); t, _, _ = c3('m2', t, hw, 1, True)
    t, hw, _ = cv('m3', t, hw, 3, 2); x4, _, _ = c3('m4', t, hw, 3, True); hw4 = hw
    t, hw, _ = cv('m5', x4, hw, 3, 2); x6, _, _ = c3('m6', t, hw, 3, True); hw6 = hw
    t, hw, _ = cv('m7', x6, hw, 3, 2)
    s1, _, c_ = cv('m8.cv1', t, hw)
    pools = [s1]
    for k in (5, 9, 13):
        o = T([1, hw, hw, c_])
        W.op('MAX_POOL_2D', [s1], [o], dict(k=k, stride=1, padding='SAME'))
        pools.append(o)
    sc = T([1, hw, hw, 4 * c_])
    W.op('CONCATENATION', pools, [sc], dict(axis=3))
    t, _, _ = cv('m8.cv2', sc, hw)
    t, _, _ = c3('m9', t, hw, 1, False)
    x10, _, c10 = cv('m10', t, hw); hw10 = hw
    t = cat(up(x10, hw, c10), x6, hw6, c10 + 256)
    t, _, _ = c3('m13', t, hw6, 1, False)
    x14, _, c14 = cv('m14', t, hw6)
    t = cat(up(x14, hw6, c14), x4, hw4, c14 + 128)
    p3, _, _ = c3('m17', t, hw4, 1, False)
    t, hw, _ = cv('m18', p3, hw4, 3, 2)
    p4, _, _ = c3('m20', cat(t, x14, hw6, 256), hw6, 1, False)
    t, hw, _ = cv('m21', p4, hw6, 3, 2)
    p5, _, _ = c3('m23', cat(t, x10, hw10, 512), hw10, 1, False)
    # Detect (models/tf.py TFDetect.call, inference branch)
    no, na = 5 + nets.YOLO_NC, 3
    zs, rows = [], 0
    for i, (p, hwp) in enumerate(((p3, hw4), (p4, hw6), (p5, hw10))):
        stride = in_size // hwp
        o, _ = conv_raw('detect%d' % i, p, hwp, wd['detect%d/weights' % i], wd['detect%d/biases' % i], 1, 1)
        r = T([1, hwp * hwp, na, no])
        W.op('RESHAPE', [o, const('shape_a%d' % i, [1, hwp * hwp, na, no], np.int32)], [r], dict(new_shape=[1, hwp * hwp, na, no]))
        tr = T([1, na, hwp * hwp, no])
        W.op('TRANSPOSE', [r, const('perm%d' % i, [0, 2, 1, 3], np.int32)], [tr])
        y = T([1, na, hwp * hwp, no])
        W.op('LOGISTIC', [tr], [y])

        def piece(lo, hi, width):
            o_ = T([1, na, hwp * hwp, width])
            W.op('STRIDED_SLICE', [y, const('b%d_%d' % (i, lo), [0, 0, 0, lo], np.int32), const('e%d_%d' % (i, lo), [0, 0, 0, hi], np.int32),
                                   const('s%d_%d' % (i, lo), [1, 1, 1, 1], np.int32)], [o_], dict(begin_mask=0b0111, end_mask=0b0111 if hi else 0b1111))
            return o_

        def binop(kind, a, bconst, width, name):
            o_ = T([1, na, hwp * hwp, width])
            W.op(kind, [a, bconst], [o_])
            return o_

        gy, gx = np.meshgrid(np.arange(hwp), np.arange(hwp), indexing='ij')
        grid = np.stack([gx, gy], axis=-1).reshape(1, 1, hwp * hwp, 2).astype(np.float32)
        anchors = np.array(nets.YOLO_ANCHORS[i], np.float32).reshape(1, na, 1, 2)
        xy = piece(0, 2, 2)
        xy = binop('MUL', xy, const('two%d' % i, [2.0]), 2, 'xy2')
        xy = binop('SUB', xy, const('half%d' % i, [0.5]), 2, 'xyh')
        xy = binop('ADD', xy, const('grid%d' % i, grid), 2, 'xyg')
        xy = binop('MUL', xy, const('stride%d' % i, [float(stride)]), 2, 'xys')
        xy = binop('DIV', xy, const('imgsz_xy%d' % i, [[float(in_size), float(in_size)]]), 2, 'xyn')
        wh = piece(2, 4, 2)
        wh = binop('MUL', wh, const('two_b%d' % i, [2.0]), 2, 'wh2')
        whs = T([1, na, hwp * hwp, 2])
        W.op('MUL', [wh, wh], [whs])
        wh = binop('MUL', whs, const('anchor_grid%d' % i, anchors), 2, 'wha')
        wh = binop('DIV', wh, const('imgsz_wh%d' % i, [[float(in_size), float(in_size)]]), 2, 'whn')
        rest = piece(4, 0, no - 4)
        yc = T([1, na, hwp * hwp, no])
        W.op('CONCATENATION', [xy, wh, rest], [yc], dict(axis=3))
        z = T([1, na * hwp * hwp, no])
        W.op('RESHAPE', [yc, const('shape_b%d' % i, [1, na * hwp * hwp, no], np.int32)], [z], dict(new_shape=[1, na * hwp * hwp, no]))
        zs.append(z)
        rows += na * hwp * hwp
    out = T([1, rows, no], 'Identity')
    W.op('CONCATENATION', zs, [out], dict(axis=1))
    W.outputs = [out]
    return W


def write_yolov5s(wd, path, in_size=640, half_weights=False, activation='silu'):
    data = yolov5s_graph(wd, in_size, half_weights, activation).tobytes()
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)
