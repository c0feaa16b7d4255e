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
