"""CPU: the frozen-graph (.pb) reader (deepdish_amd/tools/graphdef.py) -- what makes `--encoder-model <file>.pb` a drop-in
(tools/generate_detections.py:118-148,187-189 upstream).  The reference's blobs are absent (.MISSING_LARGE_BLOBS): round trips through
the writer next to it, hand-encoded protobuf messages, and refusals by name."""
import numpy as np
import pytest


def test_wire_format_primitives():
    from deepdish_amd.tools import graphdef as G
    # field 1 varint 300, field 2 "abc", field 3 fixed32 1.5f, field 4 fixed64: the encodings written out by hand
    import struct
    msg = bytes([0x08, 0xAC, 0x02, 0x12, 0x03]) + b'abc' + bytes([0x1D]) + struct.pack('<f', 1.5) + bytes([0x21]) + struct.pack('<Q', 7)
    got = [(n, w, bytes(v) if w == 2 else v) for n, w, v in G.fields(msg)]
    assert got == [(1, 0, 300), (2, 2, b'abc'), (3, 5, struct.unpack('<I', struct.pack('<f', 1.5))[0]), (4, 1, 7)]
    assert G._shape(G._shape_msg([-1, 128, 64, 3])) == [-1, 128, 64, 3]                  # -1 is a ten-byte varint
    with pytest.raises(G.UnsupportedGraph):
        list(G.fields(bytes([0x12, 0x05, 0x01])))                                        # length runs past the message
    # a TensorProto with float_val one by one and with a splat
    t = G._vi(1, 1) + G._ld(2, G._shape_msg([3])) + b''.join(bytes([0x2D]) + struct.pack('<f', v) for v in (1.0, 2.0, 3.0))
    np.testing.assert_array_equal(G._tensor(t), np.array([1, 2, 3], np.float32))
    t = G._vi(1, 1) + G._ld(2, G._shape_msg([2, 2])) + bytes([0x2D]) + struct.pack('<f', 0.25)
    np.testing.assert_array_equal(G._tensor(t), np.full((2, 2), 0.25, np.float32))


@pytest.mark.parametrize('hw', [(128, 64), (64, 32)])
def test_mars_frozen_graph_round_trip(tmp_path, hw):
    from deepdish_amd import nets
    from deepdish_amd.tools import graphdef
    from deepdish_amd.tools.weights_io import load_mars_weights
    wd = nets.synthetic_mars_weights(99)
    if hw != (64, 32):
        wd['fc1/weights'] = np.random.default_rng(1).standard_normal((hw[0] // 8 * hw[1] // 8 * 128, 128)).astype(np.float32)
    path = str(tmp_path / 'mars-small128.pb')
    graphdef.write_mars(wd, path, in_hw=hw)
    nodes = graphdef.read(path)
    assert nodes[0].name == 'images' and nodes[0].op == 'Placeholder' and nodes[0].attr['shape']['shape'] == [-1, hw[0], hw[1], 3]
    names = {n.name for n in nodes}
    assert {'conv1_1/conv1_1/bn/beta', 'conv2_3/bn/moving_variance', 'conv3_1/1/conv3_1/1/bn/moving_mean', 'conv4_1/projection/weights',
            'fc1/fc1/bn/beta', 'ball/moving_mean', 'features'} <= names                  # freeze_model.py's names, doubled scopes included
    got = load_mars_weights(path)                                  # what create_box_encoder calls
    assert got['__in_hw__'] == hw and got['__swap_rb__'] is True
    for k, v in wd.items():
        np.testing.assert_array_equal(got[k], v, err_msg=k)
    assert set(k for k in got if not k.startswith('__')) == set(wd)
    p1, p2 = nets.compile_mars(got, *hw), nets.compile_mars(wd, *hw)
    assert bytes(p1.blob) == bytes(p2.blob)


def test_other_graphs_are_refused_by_name(tmp_path):
    from deepdish_amd import nets
    from deepdish_amd.tools import graphdef as G
    wd = nets.synthetic_mars_weights(3)
    path = str(tmp_path / 'x.pb')
    del wd['conv3_3/2/biases']
    G.write_mars(wd, path)
    with pytest.raises(G.UnsupportedGraph) as e:
        G.load_mars(path)
    assert 'conv3_3/2/biases' in str(e.value)
    wd = nets.synthetic_mars_weights(3)
    G.write_mars(wd, path, in_hw=(128, 64))                        # fc1 of the 64 x 32 encoder behind a 128 x 64 placeholder
    with pytest.raises(G.UnsupportedGraph) as e:
        G.load_mars(path)
    assert 'fc1/weights' in str(e.value)
    open(path, 'wb').write(b'\x00' * 16)
    with pytest.raises(G.UnsupportedGraph):
        G.load_mars(path)
    G.write_mars(nets.synthetic_mars_weights(3), path, in_hw=(64, 32), reverse_channels=False)
    assert G.load_mars(path)[0]['__swap_rb__'] is False


def test_channel_reversal_is_detected_specifically(tmp_path):
    """`__swap_rb__` (the graph reverses the BGR crops itself, tools/freeze_model.py:175-177) follows from a StridedSlice with strides
    [1, 1, -1] (or a ReverseV2 over the last axis) -- not from the shape-picking StridedSlice every TF1 frozen graph holds; a reversal of
    another axis and a graph that states other arithmetic (batch-norm epsilon, activation) are refused."""
    import struct
    from deepdish_amd import nets
    from deepdish_amd.tools import graphdef as G
    wd = nets.synthetic_mars_weights(5)
    p = str(tmp_path / 'a.pb')
    G.write_mars(wd, p, in_hw=(64, 32), reverse_channels=True)
    assert any(n.op == 'StridedSlice' and n.name == 'strided_slice' for n in G.read(p))      # the shape slice is there in both files
    assert G.load_mars(p)[0]['__swap_rb__'] is True
    G.write_mars(wd, p, in_hw=(64, 32), reverse_channels=False)
    assert any(n.op == 'StridedSlice' for n in G.read(p)) and G.load_mars(p)[0]['__swap_rb__'] is False
    base = open(p, 'rb').read()
    # tf.reverse(image, [-1]) instead of the slice
    extra = G.const_node('ReverseV2/axis', np.asarray([-1], np.int32)) + G.node('ReverseV2', 'ReverseV2', ('Cast', 'ReverseV2/axis'), [G._attr('T', G._vi(6, 1))])
    open(p, 'wb').write(base + extra)
    assert G.load_mars(p)[0]['__swap_rb__'] is True
    # a flip of the rows is not the channel reversal
    extra = G.const_node('flip/stack_2', np.asarray([-1, 1, 1], np.int32)) + G.const_node('flip/stack', np.asarray([0, 0, 0], np.int32)) + \
        G.node('flip', 'StridedSlice', ('Cast', 'flip/stack', 'flip/stack', 'flip/stack_2'), [G._attr('T', G._vi(6, 1))])
    open(p, 'wb').write(base + extra)
    with pytest.raises(G.UnsupportedGraph) as e:
        G.load_mars(p)
    assert 'flip' in str(e.value)
    # other arithmetic than the encoder's
    eps = G._ld(5, G._ld(1, b'epsilon') + G._ld(2, bytes([0x25]) + struct.pack('<f', 1e-5)))
    open(p, 'wb').write(base + G.node('conv1_1/conv1_1/bn/FusedBatchNorm', 'FusedBatchNorm', ('Cast',), [eps]))
    with pytest.raises(G.UnsupportedGraph) as e:
        G.load_mars(p)
    assert 'epsilon' in str(e.value)
    eps = G._ld(5, G._ld(1, b'epsilon') + G._ld(2, bytes([0x25]) + struct.pack('<f', 1e-3)))
    open(p, 'wb').write(base + G.node('conv1_1/conv1_1/bn/FusedBatchNorm', 'FusedBatchNorm', ('Cast',), [eps]))
    assert G.load_mars(p)[0]['__swap_rb__'] is False
    open(p, 'wb').write(base + G.node('conv1_1/Relu', 'Relu', ('Cast',), [G._attr('T', G._vi(6, 1))]))
    with pytest.raises(G.UnsupportedGraph) as e:
        G.load_mars(p)
    assert 'Relu' in str(e.value)
