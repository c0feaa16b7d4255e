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
           'MUL': 18, 'FULLY_CONNECTED': 9, 'ELU': 111, 'REVERSE_V2': 105, 'SUM': 74, 'SQRT': 75, 'DIV': 42}
OPTIONS_TYPE = {'CONV_2D': 1, 'DEPTHWISE_CONV_2D': 2, 'CONCATENATION': 10, 'RESHAPE': 17, 'MAX_POOL_2D': 5, 'FULLY_CONNECTED': 8, 'ADD': 11, 'MUL': 21,
                'DIV': 29, 'SUM': 27}
ACT = {'none': 0, 'relu': 1, 'relu6': 3}


class GraphWriter:
    def __init__(self, description='deepdish_amd'):
        self.tensors, self.ops, self.buffers, self.codes = [], [], [b''], []
        self.inputs, self.outputs, self.description = [], [], description

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
            return b.table({0: ('i8', 0), 1: ('i32', o['stride']), 2: ('i32', o['stride']), 3: ('i8', ACT[o['act']]), 4: ('i32', 1), 5: ('i32', 1)})
        if kind == 'DEPTHWISE_CONV_2D':
            return b.table({0: ('i8', 0), 1: ('i32', o['stride']), 2: ('i32', o['stride']), 3: ('i32', 1), 4: ('i8', ACT[o['act']]), 5: ('i32', 1), 6: ('i32', 1)})
        if kind == 'CONCATENATION':
            return b.table({0: ('i32', o['axis']), 1: ('i8', 0)})
        if kind == 'RESHAPE':
            return b.table({0: ('offset', b.scalars(o['new_shape'], 'i32'))})
        if kind == 'MAX_POOL_2D':                 # Pool2DOptions: padding (1 = VALID), strides, filter, activation
            return b.table({0: ('i8', 1 if o.get('padding', 'VALID') == 'VALID' else 0), 1: ('i32', o['stride']), 2: ('i32', o['stride']), 3: ('i32', o['k']), 4: ('i32', o['k']), 5: ('i8', 0)})
        if kind in ('ADD', 'MUL', 'DIV', 'FULLY_CONNECTED'):
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
        model = b.table({0: ('u32', 3), 1: ('offset', codes), 2: ('offset', subs), 3: ('offset', desc), 4: ('offset', buffers)})
        return b.finish(model, b'TFL3')


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


def mars_graph(wd, reverse_channels=True, half_weights=False):
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
        ins = [src, fw] + ([const(name + '/bias', bias)] if bias is not None else [-1])
        out = T([1, ho, wo, cout], name)
        W.op('CONV_2D', ins, [out], dict(stride=stride, act='none'))
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


def write_mars(wd, path, reverse_channels=True, half_weights=False):
    data = mars_graph(wd, reverse_channels, half_weights).tobytes()
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)


def write_ssd_mobilenet(model, path, anchors=None, post=None):
    """post: overrides of the TFLite_Detection_PostProcess options ({name: value}; None drops the key)."""
    data = ssd_mobilenet_graph(model, anchors, post=post).tobytes()
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)
