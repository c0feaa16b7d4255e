"""csrc/mog2.hip through the C ABI against oracle/mog2_np.py: masks AND the whole mixture model bit for bit
(f32, same operation order, no FMA contraction), frame after frame; the motion count per box; error behaviour.
Parity with OpenCV itself is unpinned (oracle header)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def frames_for(rng, z, h, w, t):
    yy, xx = np.mgrid[0:h, 0:w]
    bg = np.stack([(yy * 3 + xx * 2 + 40 * z) % 230, (yy + xx * 5) % 180 + 20, (yy * 7 + z * 9) % 250], -1)
    f = np.clip(bg + rng.integers(-6, 7, bg.shape), 0, 255).astype(np.uint8)
    x = (3 + 2 * t + 5 * z) % (w - 9)
    f[h // 4:h // 4 + 11, x:x + 9] = rng.integers(0, 256, (11, 9, 3))
    if t % 5 == 2:
        f[h // 2:h // 2 + 6] = (f[h // 2:h // 2 + 6] * 0.6).astype(np.uint8)     # shadow band
    if t == 7:
        f[-3:] = 0
    if t in (11, 12):
        f[:, : w // 3] = rng.integers(0, 256, (h, w // 3, 3))                     # burst: forces new modes
    return f


@pytest.mark.parametrize('kw,lr,shape', [
    (dict(), -1, (53, 71)),                                        # odd size: the last block is partial
    (dict(history=10, varThreshold=25, detectShadows=False), -1, (32, 64)),
    (dict(), 0.05, (40, 48)),
])
def test_masks_and_model_match_the_oracle_bit_for_bit(kw, lr, shape):
    from deepdish_amd.background import createBackgroundSubtractorMOG2
    from oracle.mog2_np import MOG2, live_state
    S, (h, w) = 3, shape
    sub = createBackgroundSubtractorMOG2(n_streams=S, **kw)
    ora = [MOG2(**kw) for _ in range(S)]
    rng = np.random.default_rng(11)
    seen = set()
    for t in range(30):
        fr = np.stack([frames_for(rng, z, h, w, t) for z in range(S)])
        got = sub.apply(fr, learningRate=lr)
        for z in range(S):
            want = ora[z].apply(fr[z], lr)
            np.testing.assert_array_equal(got[z], want, err_msg='mask, stream %d frame %d' % (z, t))
            seen |= set(np.unique(want).tolist())
        if t % 6 == 5 or t == 29:
            for z in range(S):
                for name, a, b in zip(('weight', 'variance', 'mean', 'nmodes'), sub.state(z), live_state(ora[z])):
                    np.testing.assert_array_equal(a, b, err_msg='%s, stream %d frame %d' % (name, z, t))
    assert seen == ({0, 127, 255} if kw.get('detectShadows', True) else {0, 255})
    assert max(o.nmodes.max() for o in ora) == 5                   # replace-the-weakest branch reached


def test_single_stream_has_the_cv2_call_shape_and_masking():
    from deepdish_amd.background import createBackgroundSubtractorMOG2
    from oracle.mog2_np import MOG2
    backSub = createBackgroundSubtractorMOG2()
    ora = MOG2()
    rng = np.random.default_rng(12)
    for t in range(8):
        frame = frames_for(rng, 0, 48, 64, t)
        fgMask = backSub.apply(frame)                              # deepdish.py:922
        assert fgMask.shape == (48, 64) and fgMask.dtype == np.uint8
        np.testing.assert_array_equal(fgMask, ora.apply(frame))
    # cv2.bitwise_and(frame, frame, mask=fgMask), deepdish.py:924
    frame = frames_for(rng, 0, 48, 64, 8)
    dev = backSub.ctx.to_device(frame[None])
    out = torch.empty_like(dev)
    mask = backSub.ctx.to_host(backSub.apply_device(dev, masked_out=out))[0]
    np.testing.assert_array_equal(mask, ora.apply(frame))
    np.testing.assert_array_equal(out.cpu().numpy()[0], np.where(mask[..., None] != 0, frame, 0))
    # a new frame size restarts the model, as cv2 does
    small = rng.integers(1, 256, (20, 30, 3), dtype=np.uint8)
    assert (backSub.apply(small) != 0).all()


def test_motion_counts_match_count_nonzero_and_bad_boxes_are_refused():
    from deepdish_amd.background import createBackgroundSubtractorMOG2, motion_filter
    from oracle.mog2_np import motion_box_filter
    S, h, w = 2, 60, 80
    sub = createBackgroundSubtractorMOG2(n_streams=S)
    rng = np.random.default_rng(13)
    for t in range(6):
        fr = np.stack([frames_for(rng, z, h, w, t) for z in range(S)])
        masks = sub.apply(fr)
    boxes = np.array([[0, 0, w, h], [5, 7, 0, 9], [3, 15, 30, 11], [w - 1, h - 1, 1, 1], [10, 10, 37, 41], [0, 15, w, 1], [79, 0, 1, 60]],
                     dtype=np.int32)
    zs = np.array([0, 1, 1, 0, 1, 0, 1], dtype=np.int32)
    got = sub.box_counts(boxes, zs)
    want = [int(np.count_nonzero(masks[z][y:y + bh, x:x + bw])) for (x, y, bw, bh), z in zip(boxes, zs)]
    assert got.tolist() == want
    for z in range(S):
        sel = zs == z
        assert motion_filter(got[sel], boxes[sel], 0.25) == motion_box_filter(masks[z], boxes[sel], 0.25)
    assert sub.box_counts(np.zeros((0, 4), np.int32)).size == 0
    with pytest.raises(RuntimeError):
        sub.box_counts(np.array([[70, 0, 20, 5]], np.int32))       # leaves the frame: an error, never a faulting kernel
    with pytest.raises(RuntimeError):
        sub.box_counts(np.array([[0, 0, 5, 5]], np.int32), np.array([2], np.int32))


def _scenes(S, F, W, H):
    from deepdish_amd.synth import Scene
    return [Scene(seed=200 + z, n_obj=7, n_frames=F, width=W, height=H, wrange=(14, 30), hrange=(30, 60), vmax=2.5)
            for z in range(S)]


def _dets(sc, f, phantom):
    boxes, scores, _, _ = sc.detections(f)
    b = [tuple(int(v) for v in bb) for bb in boxes] + list(phantom)       # phantoms sit on still background
    s = [float(v) for v in scores] + [0.55 - 0.01 * i for i in range(len(phantom))]
    return b, ['person'] * len(b), s


@pytest.mark.parametrize('masking', [False, True])
def test_pipeline_with_background_subtraction_equals_oracle_filter_in_front_of_a_plain_pipeline(masking):
    """deepdish.py:920-924,957 inside dd_pipeline_step2: the model update and the motion test sit between the
    detector and NMS.  Reference flow restated with the oracle: MOG2 on every frame, keep boxes with
    count_nonzero >= 0.25 w h, hand the survivors (and, with masking, the masked frames) to a pipeline that has
    background subtraction off.  Tracks, Kalman means and counts must be identical."""
    from deepdish_amd.multipipe import MultiStreamPipeline
    from oracle.mog2_np import MOG2, motion_box_filter
    S, F, W, H = 3, 36, 320, 240
    scenes = _scenes(S, F, W, H)
    phantom = [(6, 8, 30, 44), (250, 150, 40, 60)]
    a = MultiStreamPipeline(S, input_size=(W, H), run_detector=False, background_subtraction_ratio=0.25, background_masking=masking)
    b = MultiStreamPipeline(S, input_size=(W, H), run_detector=False)
    ora = [MOG2() for _ in range(S)]
    rejected = moving_kept = 0
    for f in range(F):
        frames = np.stack([sc.frame(f) for sc in scenes])
        dets = [_dets(sc, f, phantom) for sc in scenes]
        a.step(torch.from_numpy(frames).cuda(), a.pack_injected(dets))
        masks = np.stack([ora[z].apply(frames[z]) for z in range(S)])
        kept = []
        for z, (bx, lb, sc_) in enumerate(dets):
            ok = motion_box_filter(masks[z], bx, 0.25)
            rejected += ok.count(False)
            moving_kept += sum(ok[:len(bx) - len(phantom)])
            kept.append(([v for v, k in zip(bx, ok) if k], [v for v, k in zip(lb, ok) if k], [v for v, k in zip(sc_, ok) if k]))
        fb = np.where(masks[..., None] != 0, frames, 0).astype(np.uint8) if masking else frames
        b.step(torch.from_numpy(fb).cuda(), b.pack_injected(kept))
        for z in range(S):
            (ia, ma), (ib, mb) = a.tracker(z).table(), b.tracker(z).table()
            np.testing.assert_array_equal(ia, ib, err_msg='stream %d frame %d' % (z, f))
            np.testing.assert_array_equal(ma, mb, err_msg='stream %d frame %d' % (z, f))
    got_mask, got_rejected = a.motion_mask()
    np.testing.assert_array_equal(got_mask, masks)
    assert got_rejected == rejected and rejected > 2 * S * (F - 3) * 0.9      # the phantoms are dropped once the model settles
    assert moving_kept > 0 and sum(len(a.tracker(z).table()[0]) for z in range(S)) > 0
    np.testing.assert_array_equal(a.counts(), b.counts())
    assert b.motion_mask(read=False) == (None, 0)
    a.background_subtraction(None)                                           # --disable-background-subtraction
    with pytest.raises(RuntimeError):
        a.motion_mask()


def test_single_stream_host_path_applies_the_motion_test():
    from deepdish_amd.pipeline import HotPath
    from oracle.mog2_np import MOG2, motion_box_filter
    F, W, H = 14, 320, 240
    sc = _scenes(1, F, W, H)[0]
    phantom = [(6, 8, 30, 44)]
    on = HotPath(input_size=(W, H), run_detector=False, disable_background_subtraction=False)
    off = HotPath(input_size=(W, H), run_detector=False)
    ora = MOG2()
    for f in range(F):
        frame = sc.frame(f)
        bx, lb, sc_ = _dets(sc, f, phantom)
        on.step(torch.from_numpy(frame).cuda(), injected=(bx, lb, sc_))
        ok = motion_box_filter(ora.apply(frame), bx, 0.25)
        off.step(torch.from_numpy(frame).cuda(), injected=tuple([v for v, k in zip(seq, ok) if k] for seq in (bx, lb, sc_)))
        key = lambda hp: [(t.track_id, t.state, t.time_since_update, t.hits, tuple(t.mean)) for t in hp.tracker.tracks]
        assert key(on) == key(off), f
    assert len(on.tracker.tracks) > 0 and not ok[-1]


def test_masking_feeds_the_detector_masked_frames_and_ignores_lookahead():
    """--enable-background-masking with the detector running: the detector (and the crops) must see
    cv2.bitwise_and(frame, frame, mask=fgMask); a look-ahead request cannot be honoured (the next mask does not exist
    yet) and must not change anything.  Compared with a plain pipeline that is handed the masked frames."""
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.pipeline import DEFAULT_LABELS
    from deepdish_amd.synth import Scene
    from oracle.mog2_np import MOG2
    S, F = 2, 6
    scenes = [Scene(seed=300 + z, n_obj=8, n_frames=F) for z in range(S)]
    labels = [l.strip() for l in open(DEFAULT_LABELS)][1:]
    a = MultiStreamPipeline(S, wanted_labels=labels, background_subtraction_ratio=0.0, background_masking=True)
    b = MultiStreamPipeline(S, wanted_labels=labels)
    ora = [MOG2() for _ in range(S)]
    frames = [np.stack([sc.frame(f) for sc in scenes]) for f in range(F)]
    dev = [torch.from_numpy(x).cuda() for x in frames]
    seen = 0
    for f in range(F):
        a.step(dev[f], None, dev[f + 1] if f + 1 < F else None)
        masks = np.stack([ora[z].apply(frames[f][z]) for z in range(S)])
        b.step(torch.from_numpy(np.where(masks[..., None] != 0, frames[f], 0).astype(np.uint8)).cuda())
        for z in range(S):
            (ia, ma), (ib, mb) = a.tracker(z).table(), b.tracker(z).table()
            np.testing.assert_array_equal(ia, ib, err_msg='stream %d frame %d' % (z, f))
            np.testing.assert_array_equal(ma, mb)
            seen += len(ia)
    assert seen > 0                                                # ratio 0: every detector box passes the motion test


def test_full_size_frames_match_the_oracle():
    """BASELINE frame size (640x480), bench-style scenes: masks and mode counts bit-exact over the first frames, where
    modes are created, matched, re-sorted and pruned all over the image."""
    from deepdish_amd.background import createBackgroundSubtractorMOG2
    from deepdish_amd.synth import Scene
    from oracle.mog2_np import MOG2, live_state
    S, F = 3, 5
    scenes = [Scene(seed=500 + z, n_obj=20, n_frames=F) for z in range(S)]
    sub = createBackgroundSubtractorMOG2(n_streams=S)
    ora = [MOG2() for _ in range(S)]
    rng = np.random.default_rng(21)
    for f in range(F):
        fr = np.stack([np.clip(sc.frame(f).astype(np.int16) + rng.integers(-3, 4, (480, 640, 3)), 0, 255).astype(np.uint8) for sc in scenes])
        got = sub.apply(fr)
        for z in range(S):
            np.testing.assert_array_equal(got[z], ora[z].apply(fr[z]), err_msg='stream %d frame %d' % (z, f))
    for z in range(S):
        for name, a, b in zip(('weight', 'variance', 'mean', 'nmodes'), sub.state(z), live_state(ora[z])):
            np.testing.assert_array_equal(a, b, err_msg='%s, stream %d' % (name, z))
    assert 0 < (got != 0).mean() < 0.5
