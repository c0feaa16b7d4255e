"""GPU: frame ingest ring -- pinned upload + cv2.flip + cv2.resize on the copy stream.
Resize parity is against the oracle's restatement of cv2 INTER_LINEAR (OpenCV itself is absent: unpinned)."""
import ctypes
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _read(ctx, addr, shape):
    """Device bytes at `addr` -> numpy (after the context's stream has drained)."""
    ctx.sync()
    out = torch.empty(shape, dtype=torch.uint8, device=f'cuda:{ctx.device}')
    hip = ctypes.CDLL('libamdhip64.so')
    n = int(np.prod(shape))
    rc = hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(addr), ctypes.c_size_t(n), 3)      # device -> device
    assert rc == 0
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize('src,dst,flip', [((1280, 720), (640, 480), False), ((1280, 960), (640, 480), True),
                                          ((480, 360), (640, 480), True), ((640, 480), (640, 480), False),
                                          ((640, 480), (640, 480), True)])
def test_ingest_flip_resize_vs_oracle(src, dst, flip):
    from deepdish_amd.ingest import FrameIngest
    from oracle import image_np
    S = 3
    ing = FrameIngest(S, src, dst, slots=2, flip=flip)
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, (S, src[1], src[0], 3), dtype=np.uint8)
    ing.host(1)[...] = raw
    ing.submit(1)
    got = _read(ing.ctx, ing.acquire(1), (S, dst[1], dst[0], 3))
    ing.release(1)
    for z in range(S):
        img = raw[z][::-1] if flip else raw[z]                 # cv2.flip(frame, 0), deepdish.py:864
        want = image_np.resize_linear_u8(np.ascontiguousarray(img), dst[0], dst[1]) if src != dst else img
        np.testing.assert_array_equal(got[z], want)


def test_ingest_ring_feeds_the_pipeline():
    """Frames that go host -> pinned slot -> copy stream -> hot path give the same tracks and counts as
    frames handed over as device tensors; slots are reused while earlier steps may still be running."""
    from deepdish_amd.ingest import FrameIngest
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.synth import Scene
    S, F = 3, 14
    scenes = [Scene(seed=90 + z, n_obj=8, n_frames=F) for z in range(S)]
    res = []
    for mode in ('direct', 'ring'):
        mp = MultiStreamPipeline(S, run_detector=False)
        ing = FrameIngest(S, (640, 480), slots=2, context=mp.ctx) if mode == 'ring' else None
        for f in range(F):
            frames = np.stack([sc.frame(f) for sc in scenes])
            dets = []
            for sc in scenes:
                boxes, scores, _, _ = sc.detections(f)
                dets.append(([tuple(int(v) for v in b) for b in boxes], ['person'] * len(boxes), [float(s) for s in scores]))
            inj = mp.pack_injected(dets)
            if ing is None:
                mp.step(torch.from_numpy(frames).cuda(), inj)
            else:
                slot = f % 2
                ing.host(slot)[...] = frames
                ing.submit(slot)
                mp.step(ing.frames(slot), inj)
                ing.release(slot)
        res.append(([mp.tracker(z).table() for z in range(S)], mp.counts()))
    for z in range(S):
        np.testing.assert_array_equal(res[0][0][z][0], res[1][0][z][0])
        np.testing.assert_array_equal(res[0][0][z][1], res[1][0][z][1])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert sum(len(t[0]) for t in res[0][0]) > 0


def test_ingest_ring_with_detector_look_ahead():
    """The bench's --ingest-host loop: the upload of frame t + 1 runs under step t, and the look-ahead detector run of frame t + 1 is the
    consumer that waits for it (`frames(slot, stream=pipe.detector_stream())`), not step t's own kernels.  The detector's own rows
    (no injected detections, every label wanted, uint8 model) and the tracks must be those of frames handed over as device tensors."""
    from deepdish_amd.ingest import FrameIngest
    from deepdish_amd.multipipe import MultiStreamPipeline
    from deepdish_amd.pipeline import DEFAULT_LABELS
    from deepdish_amd.synth import Scene
    S, F = 3, 7
    labels = [l.strip() for l in open(DEFAULT_LABELS)][1:]
    wanted = [l for l in labels if l and l != '???']
    scenes = [Scene(seed=40 + z, n_obj=6, n_frames=F) for z in range(S)]
    all_frames = [np.stack([sc.frame(f) for sc in scenes]) for f in range(F)]
    res = []
    for mode in ('direct', 'ring'):
        mp = MultiStreamPipeline(S, model='synthetic-ssd_mobilenet_v1-uint8', wanted_labels=wanted)
        rows = []
        if mode == 'direct':
            dev = [torch.from_numpy(fr).cuda() for fr in all_frames]
            for f in range(F):
                mp.step(dev[f], None, dev[f + 1] if f + 1 < F else None)
                rows.append([mp.detections(z) for z in range(S)])
        else:
            ing = FrameIngest(S, (640, 480), slots=F, context=mp.ctx)
            assert mp.detector_stream() is not None
            for f in range(F):
                ing.host(f)[...] = all_frames[f]
            ing.submit(0)
            for f in range(F):
                if f + 1 < F:
                    ing.submit(f + 1)
                nxt = ing.frames(f + 1, stream=mp.detector_stream()) if f + 1 < F else None
                mp.step(ing.frames(f), None, nxt)
                ing.release(f)
                rows.append([mp.detections(z) for z in range(S)])
        res.append((rows, [mp.tracker(z).table() for z in range(S)], mp.counts()))
    n_rows = 0
    for f in range(F):
        for z in range(S):
            a, b = res[0][0][f][z], res[1][0][f][z]
            assert list(a[1]) == list(b[1])
            np.testing.assert_array_equal(np.asarray(a[0], np.float64), np.asarray(b[0], np.float64))
            np.testing.assert_array_equal(np.asarray(a[2], np.float64), np.asarray(b[2], np.float64))
            n_rows += len(a[1])
    assert n_rows > 0                                            # random weights do emit rows with every label wanted
    for z in range(S):
        np.testing.assert_array_equal(res[0][1][z][0], res[1][1][z][0])
    np.testing.assert_array_equal(res[0][2], res[1][2])
