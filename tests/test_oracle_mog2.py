"""oracle/mog2_np.py (the CPU restatement of OpenCV's MOG2 -- parity with OpenCV unpinned, see its header):
the pixel-vectorised form against the statement-by-statement per-pixel loop, and the behaviour the reference
relies on (deepdish.py:922,957: settled background -> 0, moving object -> non-zero)."""
import numpy as np

from oracle.mog2_np import MOG2, MOG2Scalar, live_state, motion_box_filter


def scene(rng, h, w, t, noise=5):
    bg = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[..., None] % 200 + np.array([10, 30, 50])).astype(int)
    f = np.clip(bg + rng.integers(-noise, noise + 1, bg.shape), 0, 255).astype(np.uint8)
    x = (2 + t) % (w - 6)
    f[4:12, x:x + 6] = rng.integers(0, 256, (8, 6, 3))
    if t % 7 == 3:
        f[14:18, :] = f[14:18, :] // 2                            # a passing shadow
    if t == 9:
        f[:2] = 0                                                 # black rows: the zero-denominator branch
    return f


def test_vectorised_restatement_matches_the_per_pixel_loop():
    for kw, lr in ((dict(), -1), (dict(history=12, varThreshold=9, detectShadows=False), -1), (dict(), 0.2), (dict(), 0.0)):
        rng = np.random.default_rng(3)
        a, b = MOG2(**kw), MOG2Scalar(**kw)
        for t in range(22):
            f = scene(rng, 20, 24, t)
            ma, mb = a.apply(f, lr), b.apply(f, lr)
            np.testing.assert_array_equal(ma, mb, err_msg='frame %d %r' % (t, kw))
            for u, v in zip(live_state(a), live_state(b.p)):
                np.testing.assert_array_equal(u, v, err_msg='state, frame %d %r' % (t, kw))
        assert lr == 0.0 or a.nmodes.max() >= 3                                # the sort / replace / prune branches were exercised


def test_background_settles_and_motion_is_flagged():
    rng = np.random.default_rng(4)
    m = MOG2()
    first = m.apply(scene(rng, 20, 24, 0))
    assert set(np.unique(first)) <= {127, 255}                    # no model yet: nothing is background
    for t in range(1, 14):
        mask = m.apply(scene(rng, 20, 24, t))
    still = np.ones((20, 24), bool)
    still[4:12, :24] = False; still[14:18] = False; still[:2] = False
    assert (mask[still] == 0).mean() > 0.97
    x = 2 + 13
    assert np.count_nonzero(mask[4:12, x:x + 6]) >= 0.9 * 48
    assert motion_box_filter(mask, [(x, 4, 6, 8), (0, 18, 10, 2)], 0.25) == [True, False]


def test_model_restarts_when_the_frame_size_changes():
    rng = np.random.default_rng(5)
    m = MOG2()
    for t in range(3):
        m.apply(scene(rng, 20, 24, t))
    mask = m.apply(rng.integers(1, 256, (18, 30, 3), dtype=np.uint8))
    assert m.nframes == 1 and mask.shape == (18, 30) and (mask != 0).all()
