"""GPU: size-independent properties at the bench's full sizes, and equivalence of the specialised
kernels with the general ones they replace (where the oracle would take minutes or does not apply)."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ssd_batch_of_64_is_frame_independent():
    """A frame gives the same bits alone and as frame 0 / 63 of a 64-frame launch (bench batch per worker group)."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    net = Net(nets.compile_ssd_mobilenet(nets.synthetic_ssd_weights(1234)), max_batch=64)
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (64, 300, 300, 3), dtype=np.uint8)
    net.forward(x)
    full = net.read()
    assert np.isfinite(full).all()
    for i in (0, 31, 63):
        net.forward(x[i:i + 1])
        np.testing.assert_array_equal(net.read()[0], full[i])


def test_mars_batch_of_1280_is_crop_independent():
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    net = Net(nets.compile_mars(nets.synthetic_mars_weights(1234)), max_batch=1280)
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, (1280, 64, 32, 3), dtype=np.uint8)
    net.forward(x)
    full = net.read()[:, 0, 0, :]
    np.testing.assert_allclose(np.linalg.norm(full, axis=1), 1.0, atol=1e-5)
    for i in (0, 640, 1279):
        net.forward(x[i:i + 1])
        np.testing.assert_array_equal(net.read()[0, 0, 0, :], full[i])


def net_op_launches_(net):
    from deepdish_amd.profile import net_op_launches
    return net_op_launches(net)


@pytest.mark.parametrize('n', [159, 160, 170, 255, 256, 257, 511, 512, 800, 1023, 1024, 1030, 1600, 2100])
def test_mars_first_layers_give_the_same_bits_in_every_launch_shape(n):
    """conv1_1 + conv1_2 + pool: below 160 crops two launches (stem_conv3_k, tiled conv3x3_rw_k<POOL>), from 160 one
    launch of one wave per row range (conv3x3_pool_rows_k<STEM>, 4 / 2 / 1 units per crop by batch size); a program
    compiled without the fusion flag keeps the first layer separate and streams its rows by DMA.  Likewise the two
    residual units of conv2_x: two launches of conv3x3_rw_k each, or from 512 crops one res_unit_rows_k launch each
    (the intermediate tensor stays in LDS), or from 1024 crops BOTH units as one launch of wave pairs (res_pair_rows_k: the
    tensors between the units stay in LDS too; 1030 = some pairs own two images, most one).  All of them must give the bits
    of the single-crop forward."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    wd = nets.synthetic_mars_weights(1234)
    rng = np.random.default_rng(n)
    x = rng.integers(0, 256, (n, 64, 32, 3), dtype=np.uint8)
    x[1] = 0; x[2] = 255                                   # flat crops: every pooled window is a tie
    net = Net(nets.compile_mars(wd), max_batch=n)
    assert net.program.ops[0][30] == 1
    net.forward(x)
    full = net.read()[:, 0, 0, :].copy()
    pool = net.read(tensor=net.program.meta['tensors']['pool1']).copy()
    for i in (0, 1, 2, n // 2, n - 1):
        net.forward(x[i:i + 1])
        np.testing.assert_array_equal(net.read()[0, 0, 0, :], full[i])
        np.testing.assert_array_equal(net.read(tensor=net.program.meta['tensors']['pool1'])[0], pool[i])
    old = nets.Program.STEM_POOL_FUSE, nets.Program.RES_UNIT_FUSE
    try:
        nets.Program.STEM_POOL_FUSE = nets.Program.RES_UNIT_FUSE = False
        net2 = Net(nets.compile_mars(wd), max_batch=n)
    finally:
        nets.Program.STEM_POOL_FUSE, nets.Program.RES_UNIT_FUSE = old
    assert net2.program.ops[0][30] == 0 and not any(op[30] in (1, 2) for op in net2.program.ops)
    # conv1_1, conv2_1/1, conv2_1/2 (pair), conv2_3/1; 3 = the stride-2 layers of conv3_1 / conv4_1, whose projection may share their launch;
    # 4 = conv3_1's projection and conv3_3/1: with enough crops a conv3_x block is one launch (mars_pair64_k)
    assert [int(op[30]) for op in net.program.ops if op[30]] == [1, 1, 2, 1, 3, 4, 4, 3]
    net.forward(x)
    assert (16 in [int(c) for c in net_op_launches_(net)]) == (n >= 256)              # 16 = mars_pair64_k
    from deepdish_amd.profile import net_op_launches
    net.forward(x)
    assert (12 in [int(c) for c in net_op_launches(net)]) == (n >= 1024)         # 12 = res_pair_rows_k
    net2.forward(x)
    np.testing.assert_array_equal(net2.read()[:, 0, 0, :], full)
    np.testing.assert_array_equal(net2.read(tensor=net2.program.meta['tensors']['pool1']), pool)


@pytest.mark.parametrize('n', [15, 16, 100, 176])
def test_ssd_first_layers_give_the_same_bits_fused_and_separate(n):
    """conv0 + MobileNet block 1: from 16 frames one launch (ssd_front_k: strips of 30 columns streamed by single waves,
    neither the conv0 tensor nor the depthwise output leaves the CU), below that stem_conv3_k + dwpw_k; a program compiled
    without the flag always runs the two launches.  Same bits, and a frame's result does not depend on its batch (176: the
    pointwise layers of blocks 5-12 run on conv_ws_k from 160 frames)."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    wd = nets.synthetic_ssd_weights(1234)
    rng = np.random.default_rng(n)
    x = rng.integers(0, 256, (n, 300, 300, 3), dtype=np.uint8)
    x[1] = 0; x[2] = 255
    net = Net(nets.compile_ssd_mobilenet(wd), max_batch=n)
    assert net.program.ops[0][30] == 1
    net.forward(x)
    full = net.read().copy()
    for i in (0, 1, 2, n - 1):
        net.forward(x[i:i + 1])
        np.testing.assert_array_equal(net.read()[0], full[i])
    old = nets.Program.SSD_FRONT_FUSE
    try:
        nets.Program.SSD_FRONT_FUSE = False
        net2 = Net(nets.compile_ssd_mobilenet(wd), max_batch=n)
    finally:
        nets.Program.SSD_FRONT_FUSE = old
    assert net2.program.ops[0][30] == 0
    net2.forward(x)
    np.testing.assert_array_equal(net2.read(), full)


def test_fused_mobilenet_blocks_match_the_two_kernel_path():
    """dwpw_k (depthwise + pointwise in one launch) against dwconv3_k followed by the GEMM kernel: same f16
    rounding point between the two halves, so the head outputs agree to summation-order noise."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    wd = nets.synthetic_ssd_weights(1234)
    big = nets.Program.DWPW_BIG
    nets.Program.DWPW_BIG = True                 # the K-looped fused kernel for blocks 5-13 is opt-in (DD_DWPW_BIG=1): measured slower
    try:
        fused = Net(nets.compile_ssd_mobilenet(wd), max_batch=2)
    finally:
        nets.Program.DWPW_BIG = big
    saved = nets.Program.DWPW_SHAPES
    nets.Program.DWPW_SHAPES = set()
    try:
        prog = nets.compile_ssd_mobilenet(wd)
    finally:
        nets.Program.DWPW_SHAPES = saved
    kinds = [i['kernel'] for i in fused.program.info]
    assert sum(i['kernel'].startswith('dwpw') for i in prog.info) == 0 and kinds.count('dwpw_k') == 4
    assert kinds.count('dwpw_big_k') == 9 and 'dwconv3_k' not in kinds      # all thirteen MobileNet blocks are single launches
    plain = Net(prog, max_batch=2)
    x = np.random.default_rng(2).integers(0, 256, (2, 300, 300, 3), dtype=np.uint8)
    fused.forward(x); plain.forward(x)
    a, b = fused.read(), plain.read()
    assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()


def test_graph_replay_matches_eager():
    """Latency mode (dd_net_use_graph): a forward replayed as one hipGraph launch gives the bits of the eager launch
    train -- first call of an (input buffer, batch) key eager, second captured, later ones replayed -- and a one-stream
    pipeline ends with the same tracks and counts with graphs on and off."""
    import torch
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.synth import Scene
    wd = nets.synthetic_mars_weights(1234)
    net = Net(nets.compile_mars(wd), max_batch=32)
    rng = np.random.default_rng(5)
    xs = [torch.from_numpy(rng.integers(0, 256, (n, 64, 32, 3), dtype=np.uint8)).cuda() for n in (21, 7, 21)]
    want = []
    for x in xs:
        net.forward(x); want.append(net.read()[:, 0, 0, :].copy())
    net.use_graph(True)
    for rep in range(4):                                   # eager, capture, replay, replay -- per key
        for x, w in zip(xs, want):
            net.forward(x)
            np.testing.assert_array_equal(net.read()[:, 0, 0, :], w, err_msg=f'rep {rep}')
    xs[0].copy_(torch.from_numpy(rng.integers(0, 256, (21, 64, 32, 3), dtype=np.uint8)).cuda())    # same buffer, new contents
    torch.cuda.synchronize()                               # the copy runs on torch's stream, the forward on the engine's own
    net.forward(xs[0]); got = net.read()[:, 0, 0, :].copy()
    net.use_graph(False)
    net.forward(xs[0])
    np.testing.assert_array_equal(net.read()[:, 0, 0, :], got)
    sc = Scene(seed=3, n_obj=6, n_frames=24)
    res = []
    for graph in (False, True):
        mp = MultiStreamPipeline(1, graph=graph)
        assert mp.graph == graph
        for f in range(24):
            boxes, scores, _, _ = sc.detections(f)
            inj = mp.pack_injected([([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(x) for x in scores])])
            mp.step(torch.from_numpy(sc.frame(f)[None]).cuda(), inj)
        res.append((mp.tracker(0).table(), mp.counts().copy()))
    np.testing.assert_array_equal(res[0][0][0], res[1][0][0])
    np.testing.assert_array_equal(res[0][0][1], res[1][0][1])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][1].sum() > 0


def test_graph_replay_survives_a_larger_batch_in_between():
    """The split-K slab pointer is baked into a captured forward: it is sized for the engine's max_batch at creation, so a call
    with more crops between two replays of a small batch cannot free and reallocate it under the graph (round 2 sized it by
    the batch of the call).  3 crops twice (captured), 48 crops, 3 crops again (replayed) == the eager results."""
    import torch
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    wd = nets.synthetic_mars_weights(1234)
    rng = np.random.default_rng(11)
    small = torch.from_numpy(rng.integers(0, 256, (3, 64, 32, 3), dtype=np.uint8)).cuda()
    big = torch.from_numpy(rng.integers(0, 256, (48, 64, 32, 3), dtype=np.uint8)).cuda()
    ref = Net(nets.compile_mars(wd), max_batch=64)
    ref.forward(small); want_small = ref.read()[:, 0, 0, :].copy()
    ref.forward(big); want_big = ref.read()[:, 0, 0, :].copy()
    net = Net(nets.compile_mars(wd), max_batch=64)              # MARS at max_batch 64 splits K for fc1 and the 64-channel layers
    net.use_graph(True)
    for _ in range(3):                                          # eager, capture, replay
        net.forward(small)
        np.testing.assert_array_equal(net.read()[:, 0, 0, :], want_small)
    net.forward(big)
    np.testing.assert_array_equal(net.read()[:, 0, 0, :], want_big)
    for _ in range(2):
        net.forward(small)                                      # replays the graph captured before the larger batch ran
        np.testing.assert_array_equal(net.read()[:, 0, 0, :], want_small)


def test_reading_a_tensor_that_stayed_on_chip_is_an_error():
    """From 160 crops MARS conv1_1 runs inside conv1_2's launch and its tensor is never written: dd_net_read on it must fail
    loudly instead of handing back stale bytes; at a batch where the layer runs on its own the read works."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    net = Net(nets.compile_mars(nets.synthetic_mars_weights(1234)), max_batch=256)
    t = net.program.meta['tensors']['conv1_1']
    x = np.random.default_rng(12).integers(0, 256, (256, 64, 32, 3), dtype=np.uint8)
    net.forward(x[:8])
    assert np.isfinite(net.read(tensor=t)).all()
    net.forward(x)
    with pytest.raises(RuntimeError, match='not written'):
        net.read(tensor=t)
    net.read(tensor=net.program.meta['tensors']['pool1'])       # the launch's own output is there


def test_stream_order_does_not_matter():
    """Streams are independent units: permuting which slot a stream occupies permutes the results."""
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.synth import Scene
    S, F = 6, 25
    scenes = [Scene(seed=70 + z, n_obj=6 + z, n_frames=F) for z in range(S)]
    perm = [3, 0, 5, 1, 4, 2]
    res = []
    for order in (list(range(S)), perm):
        mp = MultiStreamPipeline(S, run_detector=False)
        for f in range(F):
            frames = np.stack([scenes[z].frame(f) for z in order])
            dets = []
            for z in order:
                boxes, scores, _, _ = scenes[z].detections(f)
                dets.append(([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(s) for s in scores]))
            mp.step(torch.from_numpy(frames).cuda(), mp.pack_injected(dets))
        res.append(([mp.tracker(k).table() for k in range(S)], mp.counts()))
    (tab0, cnt0), (tab1, cnt1) = res
    for k, z in enumerate(perm):
        np.testing.assert_array_equal(tab1[k][0], tab0[z][0])
        np.testing.assert_array_equal(tab1[k][1], tab0[z][1])
        np.testing.assert_array_equal(cnt1[k], cnt0[z])
    assert sum(len(t[0]) for t in tab0) > 0


def test_kalman_256_tracks_stay_symmetric_and_gate_is_nonnegative():
    from deepdish_amd.deep_sort.kalman_filter import KalmanFilter
    kf = KalmanFilter()
    rng = np.random.default_rng(3)
    meas = np.c_[rng.uniform(0, 4000, 256), rng.uniform(0, 3000, 256), rng.uniform(0.3, 0.6, 256), rng.uniform(60, 120, 256)]
    states = [kf.initiate(m) for m in meas]
    for step in range(20):
        nxt = []
        for (mean, cov), m in zip(states, meas):
            mean, cov = kf.predict(mean, cov)
            z = m + np.array([step * 1.5, step * 0.5, 0.0, 0.1 * step])
            d = kf.gating_distance(mean, cov, z[None])
            assert d.shape == (1,) and d[0] >= 0.0
            mean, cov = kf.update(mean, cov, z)
            assert np.abs(cov - cov.T).max() <= 1e-9 * np.abs(cov).max()
            assert np.all(np.linalg.eigvalsh((cov + cov.T) / 2) > -1e-9)
            nxt.append((mean, cov))
        states = nxt


@pytest.mark.parametrize('kind', ['mars', 'ssd'])
def test_first_layer_does_not_depend_on_the_frame_pointer_alignment(kind):
    """stem_conv3_k fills its input patch with 4-byte loads when the frame rows are 4-byte aligned and with byte loads
    otherwise (a frame tensor that starts at an odd address): same bits either way."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    if kind == 'mars':
        prog, shape, n = nets.compile_mars(nets.synthetic_mars_weights()), (64, 32), 37
    else:
        prog, shape, n = nets.compile_ssd_mobilenet(nets.synthetic_ssd_weights()), (300, 300), 2
    net = Net(prog, max_batch=n)
    x = np.random.default_rng(8).integers(0, 256, (n,) + shape + (3,), dtype=np.uint8)
    flat = torch.zeros(x.size + 16, dtype=torch.uint8, device='cuda')
    outs = []
    for off in (0, 1, 2):
        view = flat[off:off + x.size].view(x.shape)
        view.copy_(torch.from_numpy(x).cuda())
        torch.cuda.synchronize()                           # torch's stream, not the engine's
        assert view.data_ptr() % 4 == off
        net.forward(view)
        outs.append(net.read().copy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])


def test_nms_is_idempotent_and_order_free_at_full_size():
    """4096 boxes, dd_nms's single-pass capacity (the reference hands it a few hundred: what passes the detector's
    confidence threshold): suppressing the survivors again removes nothing, survivors come best-first, shuffling the
    input only relabels them; one box more is a loud error."""
    from deepdish_amd.deep_sort import preprocessing
    rng = np.random.default_rng(17)
    K = 4096
    boxes = np.c_[rng.integers(0, 600, K), rng.integers(0, 440, K), rng.integers(8, 120, K), rng.integers(8, 160, K)].astype(np.int64)
    scores = rng.permutation(K).astype(np.float64) / K                          # tie-free
    keep = preprocessing.non_max_suppression(boxes, 0.6, scores)
    assert 0 < len(keep) < K and len(set(keep)) == len(keep)
    assert np.all(np.diff(scores[keep]) < 0)
    again = preprocessing.non_max_suppression(boxes[keep], 0.6, scores[keep])
    assert again == list(range(len(keep)))
    perm = rng.permutation(K)
    keep_p = preprocessing.non_max_suppression(boxes[perm], 0.6, scores[perm])
    assert [int(perm[i]) for i in keep_p] == keep
    with pytest.raises(RuntimeError):
        preprocessing.non_max_suppression(np.concatenate([boxes, boxes[:1]]), 0.6, np.append(scores, 2.0))


def test_embeddings_are_unit_vectors_at_full_batch():
    """freeze_model.py:153-156 ends in an L2 normalisation: every row of a 3840-crop forward has norm 1."""
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    net = Net(nets.compile_mars(nets.synthetic_mars_weights()), max_batch=3840)
    x = np.random.default_rng(19).integers(0, 256, (3840, 64, 32, 3), dtype=np.uint8)
    net.forward(x)
    f = net.read().reshape(3840, -1)[:, :128]
    assert np.isfinite(f).all()
    np.testing.assert_allclose(np.linalg.norm(f.astype(np.float64), axis=1), 1.0, atol=1e-4)


@pytest.mark.parametrize('batch', [64, 176])
def test_weight_stationary_kernel_gives_the_same_bits(batch):
    """conv_ws_k (weight rows in registers, activation tiles through a counted-vmcnt LDS ring; on from 160 images per launch,
    DD_WS=1 / 0 forces it on / off) sums every output in the order conv_glds_k does, so the whole SSD forward is bit-identical
    either way (the switch is read once per process: two child processes)."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = []
    for ws in ('0', '1'):
        env = dict(os.environ, DD_WS=ws)
        r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'time_forward.py'), 'ssd', str(batch)], capture_output=True,
                           text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        sums.append(re.search(r'sha (\S+)', r.stdout).group(1))
    assert sums[0] == sums[1], sums


def test_pointwise_plus_next_depthwise_in_one_launch_gives_the_same_bits():
    """conv_ws_dw_k: from a batch that splits into whole frames per worker (256 frames on 128 workers) the pointwise layers of
    MobileNet blocks 6-10 and 12 run with the next block's depthwise 3x3 folded into their epilogue (the pointwise output lives
    in a per-wave LDS ring); DD_WS_DW_OFF=1 keeps the two launches.  Same bits."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = []
    for off in ('0', '1'):
        env = dict(os.environ, DD_WS_DW_OFF=off)
        r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'time_forward.py'), 'ssd', '256'], capture_output=True,
                           text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        sums.append(re.search(r'sha (\S+)', r.stdout).group(1))
    assert sums[0] == sums[1], sums


@pytest.mark.parametrize('n', [1, 5])
def test_yolo_focus_folded_into_the_first_conv_gives_the_same_bits(n):
    """YOLOv5s: the space-to-depth input op runs inside conv3x3_rw_k's patch fill (the u8 frame is sliced, normalised and
    zero-padded on the way into LDS).  Same decoded rows as a program compiled without the fold flag -- incl. frames whose
    last pixels differ (the fill reads the last focus pixel of a frame byte by byte) -- and the sliced tensor, which was never
    written, cannot be read."""
    import torch
    from deepdish_amd import nets
    from deepdish_amd.engine import Net
    wd = nets.synthetic_yolov5s_weights()
    rng = np.random.default_rng(11)
    x = rng.integers(0, 256, (n, 640, 640, 3), dtype=np.uint8)
    x[0, -2:, -2:] = (255, 0, 128)
    xd = torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    outs = {}
    for fuse in (True, False):
        old = nets.FOCUS_FUSE
        nets.FOCUS_FUSE = fuse
        try:
            prog = nets.compile_yolov5s(wd)
        finally:
            nets.FOCUS_FUSE = old
        net = Net(prog, max_batch=n)
        net.forward(xd)
        outs[fuse] = net.read().copy()
        focus_tensor = int(prog.ops[0][2])
        if fuse:
            with pytest.raises(Exception):
                net.read(tensor=focus_tensor)
        else:
            net.read(tensor=focus_tensor)
    np.testing.assert_array_equal(outs[True], outs[False])
