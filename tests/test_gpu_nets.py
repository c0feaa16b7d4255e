"""GPU: the MFMA conv engine against the independent f32 torch-CPU restatement of each network,
on seeded synthetic weights (forward parity against the real TFLite models is unpinned: no
weights, no tflite_runtime -- see oracle/nets_torch.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def _assert_close(name, got, want, atol, rtol):
    """Per element |got - want| <= atol + rtol * |want|; prints how much of the bound the worst element uses."""
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    bound = atol + rtol * np.abs(want.astype(np.float64))
    worst = int(np.argmax(err / bound))
    used = float((err / bound).reshape(-1)[worst])
    print('%s: worst element uses %.0f %% of atol %.1e + rtol %.1e x |want| (|err| %.3e at |want| %.3e; max |err| %.3e)'
          % (name, 100 * used, atol, rtol, float(err.reshape(-1)[worst]), float(np.abs(want).reshape(-1)[worst]), float(err.max())))
    assert used <= 1.0, (name, used)


@pytest.fixture(scope='module')
def mars():
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    wd = nets.synthetic_mars_weights(1234)
    return wd, Net(nets.compile_mars(wd), max_batch=64)


def test_mars_forward_vs_oracle(mars):
    from oracle import nets_torch
    wd, net = mars
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (37, 64, 32, 3), dtype=np.uint8)
    x[0] = 0; x[1] = 255
    net.forward(x)
    got = net.read()[:, 0, 0, :]
    assert got.shape == (37, 128)
    want16 = nets_torch.mars_forward(wd, x, w16=True)
    want32 = nets_torch.mars_forward(wd, x, w16=False)
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
    # stated tolerance: f16 activations + f16 weights vs the f32 reference arithmetic
    cos16 = 1.0 - np.sum(got * want16, axis=1)
    cos32 = 1.0 - np.sum(got * want32, axis=1)
    assert np.abs(got - want16).max() < 5e-3, np.abs(got - want16).max()
    assert cos16.max() < 2e-4 and cos32.max() < 5e-4, (cos16.max(), cos32.max())
    # batch independence (split-K is chosen per layer shape, never per batch) + determinism
    net.forward(x[5:6])
    np.testing.assert_array_equal(net.read()[:, 0, 0, :], got[5:6])
    net.forward(x)
    np.testing.assert_array_equal(net.read()[:, 0, 0, :], got)


def test_mars_intermediate_layers(mars):
    """First conv / pool activations, layer by layer, to localise any engine bug."""
    import torch
    import torch.nn.functional as F
    from oracle import nets_torch as nt
    wd, net = mars
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, (3, 64, 32, 3), dtype=np.uint8)
    net.forward(x)
    xt = nt._t(x[..., ::-1].astype(np.float32)).permute(0, 3, 1, 2)
    a1 = F.elu(nt._conv_bn(xt, wd, 'conv1_1', w16=True))
    a2 = F.elu(nt._conv_bn(a1, wd, 'conv1_2', w16=True))
    a3 = F.max_pool2d(a2, 3, 2)
    ids = net.program.meta['tensors']
    for tid, want in ((ids['conv1_1'], a1), (ids['pool1'], a3)):          # conv1_2 is pooled inside its own launch
        got = net.read(tensor=tid).astype(np.float32)[..., :32]
        w = want.permute(0, 2, 3, 1).numpy()
        assert got.shape == w.shape, (got.shape, w.shape)
        assert _rel(got, w) < 4e-3, (tid, _rel(got, w))


def test_ssd_forward_vs_oracle():
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    from oracle import nets_torch
    wd = nets.synthetic_ssd_weights(1234)
    net = Net(nets.compile_ssd_mobilenet(wd), max_batch=2)
    rng = np.random.default_rng(2)
    x = rng.integers(0, 256, (2, 300, 300, 3), dtype=np.uint8)
    net.forward(x)
    got = net.read()[:, :, 0, :]
    want = nets_torch.ssd_forward(wd, x, w16=True)
    assert got.shape == want.shape == (2, 1917, 95)
    # stated tolerance, PER ELEMENT of the raw head (box encodings and class logits): f16 activations / f32 accumulation
    # against the f32 restatement with the same f16-rounded weights; measured worst case (scripts/probe_head_errors.py):
    # atol 4.4e-3 would do at this rtol
    _assert_close('ssd head vs f32 restatement, f16-rounded weights', got, want, atol=8e-3, rtol=1e-2)
    want32 = nets_torch.ssd_forward(wd, x, w16=False)                       # + the weight rounding itself (measured 6.2e-3)
    _assert_close('ssd head vs f32 restatement, f32 weights', got, want32, atol=1.2e-2, rtol=1e-2)


def test_yolov5s_forward_vs_oracle():
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    from oracle import nets_torch
    wd = nets.synthetic_yolov5s_weights(1234)
    net = Net(nets.compile_yolov5s(wd), max_batch=1)
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (1, 640, 640, 3), dtype=np.uint8)
    net.forward(x)
    got = net.read()[:, :, 0, :]
    want = nets_torch.yolov5s_forward(wd, x, w16=True)
    assert got.shape == want.shape == (1, 25200, 85)
    # per element of the decoded rows (xywh normalised, objectness, class scores in (0, 1)); measured: rtol 1e-2 alone
    # covers every element but a handful near zero (atol 2.1e-6)
    _assert_close('yolov5s rows vs f32 restatement, f16-rounded weights', got, want, atol=2e-4, rtol=1e-2)


def test_shared_activation_buffers_give_the_same_features():
    """dd_net_create_shared (buffers overlaid by lifetime, what the pipeline's encoder uses): same bits as one buffer per tensor at every
    launch shape the fusion rules distinguish, a third of the memory, and intermediate tensors are refused."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    from deepdish_amd._lib import DeepDishHipError
    wd = nets.synthetic_mars_weights(1234)
    prog = nets.compile_mars(wd)
    plain, shared = Net(prog, max_batch=2200), Net(prog, max_batch=2200, shared=True)
    assert shared.activation_bytes() < 0.45 * plain.activation_bytes(), (shared.activation_bytes(), plain.activation_bytes())
    rng = np.random.default_rng(4)
    for n in (1, 37, 200, 600, 1100, 2200):
        x = rng.integers(0, 256, (n, 64, 32, 3), dtype=np.uint8)
        plain.forward(x); shared.forward(x)
        np.testing.assert_array_equal(shared.read(), plain.read(), err_msg='%d crops' % n)
    with pytest.raises(DeepDishHipError):
        shared.read(tensor=prog.meta['tensors']['pool1'])


def test_box_encoder_loads_a_tflite_file(tmp_path):
    """create_box_encoder('.../mars-64x32x3.tflite') as deepdish.py:505-510 constructs it: the encoder written to disk in the interchange
    format gives, bit for bit, the features of the same weights handed over as named arrays (tools/tflite_reader.load_mars)."""
    import numpy as np
    from deepdish_amd import nets
    from deepdish_amd.tools import tflite_writer
    from deepdish_amd.tools.generate_detections import create_box_encoder, MarsImageEncoder
    wd = nets.synthetic_mars_weights(4321)
    path = str(tmp_path / 'mars-64x32x3.tflite')
    tflite_writer.write_mars(wd, path)
    enc = create_box_encoder(path, batch_size=32)
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    boxes = [[10, 20, 40, 90], [100, 50, 60, 120], [200, 30, 50, 100]]
    got = enc(frame, boxes)
    assert got.shape == (3, 128) and np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-3)
    np.savez(str(tmp_path / 'mars.npz'), **wd)
    ref = create_box_encoder(str(tmp_path / 'mars.npz'), batch_size=32)
    want = ref(frame, boxes)
    np.testing.assert_array_equal(got, want)


def test_create_box_encoder_reads_a_frozen_graph(tmp_path):
    """create_box_encoder('<file>.pb') (the `else` branch of generate_detections.py:182-189 upstream: ImageEncoder on a frozen graph): the
    64 x 32 encoder written as a GraphDef gives, bit for bit, the features of the same weights handed over as named arrays; a 128 x 64
    graph (mars-small128, freeze_model.py:200-201) takes 128 x 64 crops and agrees with the f32 restatement within the encoder's tolerance."""
    from deepdish_amd import nets
    from deepdish_amd.tools import graphdef
    from deepdish_amd.tools.generate_detections import create_box_encoder
    from oracle import nets_torch
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = [[30, 40, 60, 140], [300, 100, 80, 200], [500, 20, 50, 90]]
    wd = nets.synthetic_mars_weights(4321)
    path = str(tmp_path / 'mars-64x32x3.pb')
    graphdef.write_mars(wd, path, in_hw=(64, 32))
    np.savez(str(tmp_path / 'mars.npz'), **wd)
    enc, ref = create_box_encoder(path, batch_size=32), create_box_encoder(str(tmp_path / 'mars.npz'), batch_size=32)
    assert (enc.image_encoder.height, enc.image_encoder.width) == (64, 32)
    np.testing.assert_array_equal(enc(frame, boxes), ref(frame, boxes))
    wd['fc1/weights'] = (rng.standard_normal((16 * 8 * 128, 128)) * np.sqrt(2.0 / 16384)).astype(np.float32)
    path = str(tmp_path / 'mars-small128.pb')
    graphdef.write_mars(wd, path, in_hw=(128, 64))
    enc = create_box_encoder(path, batch_size=32)
    assert enc.image_encoder.image_shape == (128, 64, 3)
    from oracle import image_np
    patches = np.stack([image_np.extract_image_patch(frame, np.array(b), (128, 64)) for b in boxes])
    want = nets_torch.mars_forward(wd, patches)
    got = enc(frame, boxes)
    assert got.shape == (3, 128)
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-3)
    assert (1.0 - (got * want).sum(axis=1)).max() < 5e-4
