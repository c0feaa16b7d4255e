"""Frozen TensorFlow graphs (`.pb`): what the reference's `ImageEncoder` loads when the encoder model is not a `.tflite` file
(tools/generate_detections.py:118-148,187-189 upstream: `graph_def.ParseFromString(open(checkpoint_filename,'rb').read())`, input
`images`, output `features`) -- read without TensorFlow or the protobuf package.

A GraphDef is plain protobuf wire format (varints, fixed 32/64-bit values, length-delimited fields); the messages needed here and
their field numbers, from tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,types}.proto:

    GraphDef          1 node (NodeDef, repeated)
    NodeDef           1 name, 2 op, 3 input (repeated), 5 attr (map entries: 1 key, 2 AttrValue)
    AttrValue         2 s, 3 i, 4 f, 5 b, 6 type, 7 shape (TensorShapeProto), 8 tensor (TensorProto)
    TensorShapeProto  2 dim (Dim: 1 size), 3 unknown_rank
    TensorProto       1 dtype, 2 tensor_shape, 4 tensor_content, 5 float_val, 6 double_val, 7 int_val, 13 half_val
    DataType          1 float, 2 double, 3 int32, 4 uint8, 9 int64, 19 half

`load_mars(path)` recovers the MARS encoder's variables from the Const nodes `convert_variables_to_constants` leaves behind
(tools/freeze_model.py:213-215), under the names freeze_model.py:88-157 gives them, and hands them to deepdish_amd/nets.compile_mars
under its own names; the input size comes from the `images` placeholder (freeze_model.py:200-201: mars-small128 takes 128 x 64 crops).
The reference's `.pb` blobs are absent from its tree (.MISSING_LARGE_BLOBS): the reader is exercised on files `write_mars` below
produces from the same field numbers (tests/test_graphdef.py) -- a stated limit, as for the .tflite reader.
"""
import struct
import numpy as np

DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 9: np.int64, 19: np.float16}
DTYPE_CODES = {np.dtype(v): k for k, v in DTYPES.items()}


class UnsupportedGraph(ValueError):
    pass


# ------------------------------------------------------------------------------------------- wire format
def _varint(buf, pos):
    shift, val = 0, 0
    while True:
        if pos >= len(buf):
            raise UnsupportedGraph('truncated varint')
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if not b & 0x80:
            return val, pos
        shift += 7
        if shift > 70:
            raise UnsupportedGraph('varint longer than 10 bytes')


def fields(buf):
    """Yield (field number, wire type, value) of one message: value = int (varint, fixed), memoryview (length-delimited)."""
    buf = memoryview(buf)
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 5:
            val = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise UnsupportedGraph('length-delimited field of %d bytes runs past the message' % ln)
            val = buf[pos:pos + ln]; pos += ln
        else:
            raise UnsupportedGraph('wire type %d (groups are not used by GraphDef)' % wt)
        yield num, wt, val


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _shape(buf):
    dims, unknown = [], False
    for num, wt, val in fields(buf):
        if num == 2:
            size = 0
            for n2, _, v2 in fields(val):
                if n2 == 1:
                    size = _signed(v2)
            dims.append(size)
        elif num == 3:
            unknown = bool(val)
    return None if unknown else dims


def _tensor(buf):
    dtype, shape, content, vals = None, [], None, []
    for num, wt, val in fields(buf):
        if num == 1:
            dtype = val
        elif num == 2:
            shape = _shape(val) or []
        elif num == 4:
            content = bytes(val)
        elif num in (5, 6, 7, 13):                          # float_val / double_val / int_val / half_val: packed or one by one
            if wt == 2:
                raw = bytes(val)
                if num == 5:
                    vals += list(struct.unpack('<%df' % (len(raw) // 4), raw))
                elif num == 6:
                    vals += list(struct.unpack('<%dd' % (len(raw) // 8), raw))
                else:
                    p = 0
                    while p < len(raw):
                        v, p = _varint(raw, p)
                        vals.append(_signed(v))
            elif num == 5:
                vals.append(struct.unpack('<f', struct.pack('<I', val))[0])
            elif num == 6:
                vals.append(struct.unpack('<d', struct.pack('<Q', val))[0])
            else:
                vals.append(_signed(val))
    if dtype not in DTYPES:
        return None
    dt = np.dtype(DTYPES[dtype])
    n = int(np.prod(shape)) if shape else 1
    if content is not None and len(content):
        arr = np.frombuffer(content, dtype=dt)
    elif dtype == 19:                                       # half_val carries the bit patterns as ints
        arr = np.array(vals, dtype=np.uint16).view(np.float16)
    else:
        arr = np.array(vals, dtype=dt)
    if arr.size == 1 and n > 1:
        arr = np.full(n, arr.reshape(-1)[0], dtype=dt)      # a splat constant
    if arr.size != n:
        raise UnsupportedGraph('tensor of shape %s carries %d values' % (shape, arr.size))
    return arr.reshape(shape)


class Node:
    def __init__(self, name, op, inputs, attr):
        self.name, self.op, self.inputs, self.attr = name, op, inputs, attr

    def __repr__(self):
        return 'Node(%r %s <- %s)' % (self.name, self.op, self.inputs)


def read(path_or_bytes):
    """-> [Node]; attr values: 'tensor' -> ndarray, 'shape' -> dims (or None), 'type' -> DataType code, 's' / 'i' / 'f' / 'b'."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, 'rb').read()
    nodes = []
    for num, wt, val in fields(buf):
        if num != 1 or wt != 2:
            continue
        name, op, inputs, attr = '', '', [], {}
        for n2, w2, v2 in fields(val):
            if n2 == 1:
                name = bytes(v2).decode()
            elif n2 == 2:
                op = bytes(v2).decode()
            elif n2 == 3:
                inputs.append(bytes(v2).decode())
            elif n2 == 5:
                key, value = None, None
                for n3, w3, v3 in fields(v2):
                    if n3 == 1:
                        key = bytes(v3).decode()
                    elif n3 == 2:
                        value = {}
                        for n4, w4, v4 in fields(v3):
                            if n4 == 8:
                                value['tensor'] = _tensor(v4)
                            elif n4 == 7:
                                value['shape'] = _shape(v4)
                            elif n4 == 6:
                                value['type'] = v4
                            elif n4 == 2:
                                value['s'] = bytes(v4)
                            elif n4 == 3:
                                value['i'] = _signed(v4)
                            elif n4 == 4:
                                value['f'] = struct.unpack('<f', struct.pack('<I', v4))[0]
                            elif n4 == 5:
                                value['b'] = bool(v4)
                if key is not None:
                    attr[key] = value or {}
        nodes.append(Node(name, op, inputs, attr))
    if not nodes:
        raise UnsupportedGraph('no NodeDef in the file: not a GraphDef')
    return nodes


# ------------------------------------------------------------------------------------------- MARS encoder
MARS_BLOCKS = [('conv2_1', 32, False, True), ('conv2_3', 32, False, False), ('conv3_1', 64, True, False),
               ('conv3_3', 64, False, False), ('conv4_1', 128, True, False), ('conv4_3', 128, False, False)]


def _mars_variable_names():
    """{name under freeze_model.py: (name deepdish_amd/nets.compile_mars reads, required)}.  slim puts a layer's normaliser variables
    under '<layer scope>/<scope handed to batch_norm>' and freeze_model.py:6-9,100-101 hands over the full scope name again, hence the
    doubled prefixes ('conv1_1/conv1_1/bn/beta'); slim.batch_norm's default has no gamma (scale=False)."""
    out = {}

    def bn(tf_scope, scope):
        for leaf in ('beta', 'moving_mean', 'moving_variance'):
            out[tf_scope + '/' + leaf] = (scope + '/' + leaf, True)
        out[tf_scope + '/gamma'] = (scope + '/gamma', False)

    for layer in ('conv1_1', 'conv1_2'):
        out[layer + '/weights'] = (layer + '/weights', True)
        bn('%s/%s/bn' % (layer, layer), layer + '/bn')
    for name, c, inc, first in MARS_BLOCKS:
        if not first:
            bn(name + '/bn', name + '/bn')                                     # create_link's leading batch norm (:17-18)
        out[name + '/1/weights'] = (name + '/1/weights', True)
        bn('%s/1/%s/1/bn' % (name, name), name + '/1/bn')
        out[name + '/2/weights'] = (name + '/2/weights', True)
        out[name + '/2/biases'] = (name + '/2/biases', True)
        if inc:
            out[name + '/projection/weights'] = (name + '/projection/weights', True)
    out['fc1/weights'] = ('fc1/weights', True)
    bn('fc1/fc1/bn', 'fc1/bn')
    bn('ball', 'ball')
    return out


def _reverses_channels(path, nodes, consts):
    """Does the graph reverse the channel axis of its input (tools/freeze_model.py:175-177: `image[:, :, ::-1]`)?  That statement freezes to a
    StridedSlice whose strides constant is [1, .., 1, -1] (or, written with tf.reverse, a ReverseV2 over the last axis).  Every TF1 frozen graph
    also holds StridedSlice nodes that pick dimensions out of a Shape (strides [1]): the op's presence alone says nothing.  A slice with a
    negative stride that is NOT this form, or a reversal whose axis / strides are not constants, is refused rather than guessed at."""
    def const_of(ref):
        name = ref.split(':')[0].lstrip('^')
        t = consts.get(name)
        if t is None and name.endswith('/read'):
            t = consts.get(name[:-5])
        return None if t is None else np.asarray(t).reshape(-1)
    found = False
    for n in nodes:
        if n.op == 'StridedSlice' and len(n.inputs) >= 4:
            st = const_of(n.inputs[3])
            if st is None:
                continue                                     # (strides computed at run time: a shape manipulation, not the image slice)
            if (st < 0).any():
                if len(st) >= 1 and int(st[-1]) == -1 and all(int(v) == 1 for v in st[:-1]):
                    found = True
                else:
                    raise UnsupportedGraph('%s: StridedSlice %r has strides %s: a reversal other than the channel axis of the crops' % (path, n.name, st.tolist()))
        elif n.op == 'ReverseV2':
            ax = const_of(n.inputs[1]) if len(n.inputs) >= 2 else None
            if ax is None:
                raise UnsupportedGraph('%s: ReverseV2 %r with an axis that is not a constant' % (path, n.name))
            if set(int(v) for v in ax) <= {-1, 2, 3}:
                found = True
            else:
                raise UnsupportedGraph('%s: ReverseV2 %r over axis %s: a reversal other than the channel axis of the crops' % (path, n.name, ax.tolist()))
    return found


def _check_arithmetic(path, nodes):
    """The constants are mapped by name onto nets.compile_mars' fixed arithmetic (tools/freeze_model.py:13-157: scale-less batch norm with
    epsilon 1e-3, ELU); where the graph states its own, it must state the same."""
    for n in nodes:
        if n.op.startswith('FusedBatchNorm'):
            eps = n.attr.get('epsilon', {}).get('f')
            if eps is not None and abs(float(eps) - 1e-3) > 1e-7:
                raise UnsupportedGraph('%s: %s %r has epsilon %g (the encoder is built for slim\'s 1e-3)' % (path, n.op, n.name, eps))
        elif n.op in ('Relu', 'Relu6', 'LeakyRelu', 'Selu', 'Tanh', 'Sigmoid', 'Swish'):
            raise UnsupportedGraph('%s: activation %s at %r (the encoder\'s activation is ELU, tools/freeze_model.py:27,57)' % (path, n.op, n.name))


def load_mars(path, input_name='images', output_name='features'):
    """-> (named f32 weights for nets.compile_mars, (height, width) of the crops the graph takes)."""
    nodes = read(path)
    by_name = {n.name: n for n in nodes}
    consts = {n.name: n.attr['value']['tensor'] for n in nodes if n.op == 'Const' and n.attr.get('value', {}).get('tensor') is not None}
    if input_name not in by_name or by_name[input_name].op != 'Placeholder':
        raise UnsupportedGraph('%s: no Placeholder named %r (nodes: %s ...)' % (path, input_name, [n.name for n in nodes[:6]]))
    if output_name not in by_name:
        raise UnsupportedGraph('%s: no node named %r' % (path, output_name))
    shape = by_name[input_name].attr.get('shape', {}).get('shape')
    if not shape or len(shape) != 4 or shape[3] != 3 or shape[1] <= 0 or shape[2] <= 0:
        raise UnsupportedGraph('%s: placeholder %r has shape %s (expected [-1, height, width, 3])' % (path, input_name, shape))
    h, w = int(shape[1]), int(shape[2])
    if h % 8 or w % 8:
        raise UnsupportedGraph('%s: %d x %d crops (the encoder halves the map three times: multiples of 8)' % (path, h, w))
    wd = {}
    for tf_name, (name, required) in _mars_variable_names().items():
        arr = consts.get(tf_name)
        if arr is None:
            arr = consts.get(name)                           # the same variable without the doubled scope (a differently scoped checkpoint)
        if arr is None:
            if required:
                raise UnsupportedGraph('%s: no constant %r (tools/freeze_model.py:88-157 names it); constants present: %s ...'
                                       % (path, tf_name, sorted(consts)[:8]))
            continue
        wd[name] = np.asarray(arr, dtype=np.float32)
    k = (h // 8) * (w // 8) * 128
    if wd['fc1/weights'].shape != (k, 128):
        raise UnsupportedGraph('%s: fc1/weights %s, the %d x %d input gives [%d, 128]' % (path, wd['fc1/weights'].shape, h, w, k))
    for name, want in (('conv1_1/weights', (3, 3, 3, 32)), ('conv3_1/projection/weights', (1, 1, 32, 64)), ('conv4_3/2/weights', (3, 3, 128, 128))):
        if wd[name].shape != want:
            raise UnsupportedGraph('%s: %s has shape %s (expected %s)' % (path, name, wd[name].shape, want))
    # freeze_model.py:175-177,203-205: the graph reverses the channel axis of the BGR crops itself (`image[:, :, ::-1]` inside the map_fn)
    wd['__swap_rb__'] = _reverses_channels(path, nodes, consts)
    _check_arithmetic(path, nodes)
    wd['__in_hw__'] = (h, w)
    return wd, (h, w)


# ------------------------------------------------------------------------------------------- writer (tests, interchange)
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(num, payload):
    return _enc_varint(num << 3 | 2) + _enc_varint(len(payload)) + bytes(payload)


def _vi(num, v):
    return _enc_varint(num << 3) + _enc_varint(v)


def _shape_msg(dims):
    return b''.join(_ld(2, _vi(1, d)) for d in dims)


def _attr(key, value_msg):
    return _ld(5, _ld(1, key.encode()) + _ld(2, value_msg))


def node(name, op, inputs=(), attrs=()):
    return _ld(1, _ld(1, name.encode()) + _ld(2, op.encode()) + b''.join(_ld(3, i.encode()) for i in inputs) + b''.join(attrs))


def const_node(name, arr):
    arr = np.ascontiguousarray(arr)
    t = _vi(1, DTYPE_CODES[arr.dtype]) + _ld(2, _shape_msg(arr.shape)) + _ld(4, arr.tobytes())
    return node(name, 'Const', (), [_attr('dtype', _vi(6, DTYPE_CODES[arr.dtype])), _attr('value', _ld(8, t))])


def write_mars(wd, path, in_hw=(128, 64), reverse_channels=True):
    """The nodes of a frozen tools/freeze_model.py graph that carry information: the `images` placeholder with its shape, every variable as
    the Const + Identity (`/read`) pair convert_variables_to_constants leaves, the channel reversal, the `features` output.  (The
    arithmetic nodes between them -- Conv2D, FusedBatchNorm, Elu ... -- restate the architecture, which tools/freeze_model.py:88-157
    fixes and deepdish_amd/nets.compile_mars implements; a reader needs them for nothing.)  wd: raw (unfolded) named weights."""
    h, w = in_hw
    out = [node('images', 'Placeholder', (), [_attr('dtype', _vi(6, 4)), _attr('shape', _ld(7, _shape_msg([-1, h, w, 3])))])]
    out.append(node('Cast', 'Cast', ('images',), [_attr('SrcT', _vi(6, 4)), _attr('DstT', _vi(6, 1))]))
    if reverse_channels:                       # image[:, :, ::-1] of one crop inside the map_fn: begin / end / strides constants, strides [1, 1, -1]
        for suffix, vals in (('stack', [0, 0, 0]), ('stack_1', [0, 0, 0]), ('stack_2', [1, 1, -1])):
            out.append(const_node('map/while/strided_slice/' + suffix, np.asarray(vals, dtype=np.int32)))
        out.append(node('map/while/strided_slice', 'StridedSlice', ('Cast', 'map/while/strided_slice/stack', 'map/while/strided_slice/stack_1',
                                                                   'map/while/strided_slice/stack_2'), [_attr('T', _vi(6, 1))]))
    # (what every TF1 frozen graph also holds: a StridedSlice that picks a dimension out of a Shape -- not a reversal)
    out.append(node('Shape', 'Shape', ('images',), [_attr('T', _vi(6, 4))]))
    for suffix, vals in (('stack', [0]), ('stack_1', [1]), ('stack_2', [1])):
        out.append(const_node('strided_slice/' + suffix, np.asarray(vals, dtype=np.int32)))
    out.append(node('strided_slice', 'StridedSlice', ('Shape', 'strided_slice/stack', 'strided_slice/stack_1', 'strided_slice/stack_2'), [_attr('T', _vi(6, 3))]))
    inv = {v[0]: k for k, v in _mars_variable_names().items()}
    last = 'images'
    for name in sorted(k for k in wd if not k.startswith('__')):
        if name not in inv:
            raise ValueError('%s is not a variable of the MARS encoder' % name)
        tf_name = inv[name]
        out.append(const_node(tf_name, np.asarray(wd[name], dtype=np.float32)))
        out.append(node(tf_name + '/read', 'Identity', (tf_name,), [_attr('T', _vi(6, 1))]))
        last = tf_name + '/read'
    out.append(node('truediv', 'RealDiv', (last,), [_attr('T', _vi(6, 1))]))
    out.append(node('features', 'Identity', ('truediv',), [_attr('T', _vi(6, 1))]))
    data = b''.join(out)
    with open(path, 'wb') as f:
        f.write(data)
    return len(data)
