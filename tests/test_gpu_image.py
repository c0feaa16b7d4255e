"""GPU: image front-end kernels (crop + bilinear, Lanczos stretch, bilinear stretch) -- integer
arithmetic, so the bar is bit-exact against the oracle (and against Pillow itself for Lanczos)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from deepdish_amd.runtime import default_context
    return default_context()


def test_crop_resize_vs_oracle(ctx):
    from deepdish_amd.tools.generate_detections import crop_patches_device
    from oracle import image_np
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = np.c_[rng.integers(-20, 630, 60), rng.integers(-20, 470, 60), rng.integers(2, 120, 60), rng.integers(2, 200, 60)]
    extra = np.array([[100, 50, 41, 90], [100, 50, 40, 91], [10, 10, 64, 128], [0, 0, 640, 480], [639, 479, 5, 5],
                      [700, 10, 20, 40], [-50, -50, 20, 40], [5, 5, 1, 2], [300, 200, 16, 32]])
    boxes = np.concatenate([boxes, extra]).astype(np.int64)
    dev = ctx.to_device(img)
    out, valid = crop_patches_device(ctx, dev, 480, 640, boxes, 64, 32)
    out = ctx.to_host(out)
    n_valid = 0
    for i, b in enumerate(boxes):
        want = image_np.extract_image_patch(img, b, (64, 32))
        assert bool(valid[i]) == (want is not None), (i, b)
        if want is not None:
            n_valid += 1
            np.testing.assert_array_equal(out[i], want, err_msg=str(b))
        else:
            assert not out[i].any()
    assert n_valid > 40 and n_valid < len(boxes)
    # the exact-2x path (64x128 crop) must have been exercised
    assert image_np.crop_box(np.array([10, 10, 64, 128]), (64, 32), (480, 640)) == (10, 10, 74, 138)


def test_crop_resize_of_float_boxes_vs_oracle(ctx):
    """Boxes that come from CVAT annotations are floats (framerecords feeds them to the encoder): the reference then
    does generate_detections.py:64-74 in floating point and truncates once; the int path truncates earlier."""
    from deepdish_amd.tools.generate_detections import crop_patches_device, extract_image_patch
    from oracle import image_np
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    boxes = np.c_[rng.uniform(-20, 630, 80), rng.uniform(-20, 470, 80), rng.uniform(2, 120, 80), rng.uniform(2, 200, 80)]
    boxes = np.concatenate([boxes, [[40.5, 59.75, 30.0, 80.0], [100.9, 50.9, 40.9, 90.9], [-0.9, -0.9, 30.2, 60.2], [639.5, 10.0, 9.0, 9.0]]])
    out, valid = crop_patches_device(ctx, ctx.to_device(img), 480, 640, boxes, 64, 32)
    out = ctx.to_host(out)
    differs_from_int_path = 0
    for i, b in enumerate(boxes):
        want = image_np.extract_image_patch(img, b, (64, 32))
        assert bool(valid[i]) == (want is not None), (i, b)
        if want is not None:
            np.testing.assert_array_equal(out[i], want, err_msg=str(b))
            differs_from_int_path += image_np.crop_box(b, (64, 32), (480, 640)) != image_np.crop_box(b.astype(np.int64), (64, 32), (480, 640))
    assert differs_from_int_path > 10                           # the float arithmetic matters
    np.testing.assert_array_equal(extract_image_patch(img, boxes[80], (64, 32)), image_np.extract_image_patch(img, boxes[80], (64, 32)))


@pytest.mark.parametrize('shape', [(480, 640, 300, 300), (480, 640, 640, 640), (720, 1280, 300, 300), (97, 131, 300, 300),
                                   (200, 640, 150, 300)])     # last: rows not a multiple of 16 -> the LDS-staged scalar kernels
def test_lanczos_vs_pillow(ctx, shape):
    from PIL import Image
    from deepdish_amd._lib import lib, check
    from deepdish_amd.runtime import ptr
    H, W, h, w = shape
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    # what the reference does: BGR frame -> RGBA PIL image (deepdish.py:882) -> convert('RGB').resize(ANTIALIAS)
    rgba = np.dstack([bgr[..., ::-1], np.full((H, W, 1), 255, np.uint8)])
    want = np.asarray(Image.fromarray(rgba, 'RGBA').convert('RGB').resize((w, h), Image.LANCZOS))
    dst = ctx.empty((h, w, 3), torch.uint8)
    check(lib().dd_resize_lanczos(ctx.handle, ptr(ctx.to_device(bgr)), H, W, 3, 1, ptr(dst), h, w, None))
    np.testing.assert_array_equal(ctx.to_host(dst), want)
    check(lib().dd_resize_lanczos(ctx.handle, ptr(ctx.to_device(rgba)), H, W, 4, 0, ptr(dst), h, w, None))
    np.testing.assert_array_equal(ctx.to_host(dst), want)


def test_bilinear_stretch_vs_oracle(ctx):
    from deepdish_amd._lib import lib, check
    from deepdish_amd.runtime import ptr
    from oracle import image_np
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    for (h, w) in ((320, 320), (300, 300), (240, 320), (960, 1280)):
        dst = ctx.empty((h, w, 3), torch.uint8)
        check(lib().dd_resize_bilinear(ctx.handle, ptr(ctx.to_device(img)), 480, 640, 3, ptr(dst), h, w, None))
        np.testing.assert_array_equal(ctx.to_host(dst), image_np.resize_linear_u8(img, w, h))


def test_lanczos_batch_one_launch_vs_pillow(ctx):
    """A batch of frames through dd_resize_lanczos_batch (640x480 -> 300x300 runs both passes in ONE launch through LDS:
    lanczos_fused_k): every frame equals Pillow's bytes -- noise (overshoot both ways: the clamp instruction), constants at both
    ends of the range, a checkerboard, a ramp -- and a geometry the one-launch form does not take (720p) goes the two-pass way."""
    from PIL import Image
    from deepdish_amd._lib import lib, check
    from deepdish_amd.runtime import ptr
    rng = np.random.default_rng(5)
    for (H, W, h, w, n) in ((480, 640, 300, 300, 7), (480, 640, 320, 320, 2), (480, 640, 320, 512, 2),      # the third: one 64-byte window step
                            (512, 512, 128, 200, 2), (720, 1280, 300, 300, 2),
                            (640, 640, 640, 640, 3), (33, 35, 33, 35, 2)):      # same size: only the red / blue swap (four pixels per thread; 33 x 35 is not a multiple of four: byte kernel)
        frames = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
        frames[1] = 255
        if n > 3:
            frames[2] = 0
            yy, xx = np.mgrid[0:H, 0:W]
            frames[3] = (((yy // 3 + xx // 5) & 1) * 255).astype(np.uint8)[..., None]
            frames[4] = ((xx * 255) // (W - 1)).astype(np.uint8)[..., None]
        dst = ctx.empty((n, h, w, 3), torch.uint8)
        check(lib().dd_resize_lanczos_batch(ctx.handle, ptr(ctx.to_device(frames)), n, H, W, 3, 1, ptr(dst), h, w, None))
        got = ctx.to_host(dst)
        for i in range(n):
            rgba = np.dstack([frames[i][..., ::-1], np.full((H, W, 1), 255, np.uint8)])
            want = np.asarray(Image.fromarray(rgba, 'RGBA').convert('RGB').resize((w, h), Image.LANCZOS))
            np.testing.assert_array_equal(got[i], want, err_msg=f'{H}x{W}->{h}x{w} frame {i}')
