"""Re-ID box encoder with the reference's surface (tools/generate_detections.py:26-216 upstream):
`create_box_encoder(model_filename, batch_size=..., num_threads=...)` -> `encoder(image, boxes, timing)`.

Crop arithmetic + bilinear resample run in csrc/image.hip (one launch for all boxes), the MARS
forward pass in csrc/nets.hip (f16 MFMA).  `encoder.encode_device` keeps features in HBM for the
tracker; the reference-shaped call returns a host ndarray.
"""
from time import time
import numpy as np
import torch

from .._lib import lib, check
from ..runtime import default_context, ptr
from .. import nets
from ..engine import Net
from .weights_io import load_named_weights, load_mars_weights


def crop_patches_device(ctx, frame_dev, H, W, boxes_int64, ph, pw):
    """-> (patches u8 [n, ph, pw, 3] on device, valid int32 [n] on host)."""
    n = len(boxes_int64)
    out = ctx.empty((n, ph, pw, 3), torch.uint8)
    valid = np.ones(n, dtype=np.int32)
    if np.asarray(boxes_int64).dtype.kind == 'f':       # float boxes (CVAT annotations): the reference's float arithmetic
        b = np.ascontiguousarray(boxes_int64, dtype=np.float64).reshape(n, 4)
        check(lib().dd_crop_resize_f64(ctx.handle, ptr(frame_dev), H, W, ptr(b), n, ph, pw, ptr(out), ptr(valid), None),
              'dd_crop_resize_f64')
        return out, valid
    b = np.ascontiguousarray(boxes_int64, dtype=np.int64).reshape(n, 4)
    check(lib().dd_crop_resize(ctx.handle, ptr(frame_dev), H, W, ptr(b), n, ph, pw, ptr(out), ptr(valid), None),
          'dd_crop_resize')
    return out, valid


def extract_image_patch(image, bbox, patch_shape, context=None):
    """tools/generate_detections.py:40-84; returns None where the reference does."""
    ctx = context or default_context()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    H, W = img.shape[:2]
    bb = np.array(bbox)
    dev = ctx.to_device(img)
    out, valid = crop_patches_device(ctx, dev, H, W, bb.reshape(1, 4), patch_shape[0], patch_shape[1])
    if not valid[0]:
        return None
    return ctx.to_host(out)[0]


class MarsImageEncoder(object):
    """Counterpart of TFLiteImageEncoder / ImageEncoder (generate_detections.py:118-177)."""

    def __init__(self, model_filename, num_threads=1, max_batch=256, context=None):
        self.ctx = context or default_context()
        wd = load_mars_weights(model_filename)                   # .tflite (generate_detections.py:151-162), .npz of named arrays, or synthetic[:seed]
        self.weights = wd
        # a frozen graph states its crop size in the `images` placeholder (generate_detections.py:141-142; mars-small128.pb: 128 x 64),
        # the .tflite encoder the reference ships takes 64 x 32 (mars-64x32x3)
        self.height, self.width = (int(v) for v in wd.get('__in_hw__', (64, 32)))
        self.net = Net(nets.compile_mars(wd, self.height, self.width), max_batch=max_batch, context=self.ctx)
        self.image_shape = (self.height, self.width, 3)
        self.feature_dim = 128
        self.max_batch_size = max_batch

    def encode_device(self, patches_dev):
        """u8 [n, height, width, 3] on device -> f32 [n,128] on device."""
        n = int(patches_dev.shape[0])
        out = self.ctx.empty((n, 128), torch.float32)
        for s in range(0, n, self.max_batch_size):
            e = min(n, s + self.max_batch_size)
            self.net.forward(patches_dev[s:e])
            check(lib().dd_net_read(self.net._h, -1, e - s, ptr(out[s:e]), 1, None), 'dd_net_read')
        return out

    def __call__(self, data_in, batch_size=32):
        patches = self.ctx.to_device(np.asarray(data_in, dtype=np.uint8))
        return self.ctx.to_host(self.encode_device(patches))


class _FakeImageEncoder(object):
    """DummyImageEncoder / ConstantImageEncoder (generate_detections.py:86-116) on csrc/image.hip."""
    mode = 0

    def __init__(self, context=None):
        self.ctx = context or default_context()
        self.height, self.width = 16, 8
        self.image_shape = 16, 8, 3
        self.feature_dim = 128

    def encode_device(self, patches_dev):
        n = int(patches_dev.shape[0])
        out = self.ctx.empty((n, 128), torch.float32)
        check(lib().dd_fake_encode(self.ctx.handle, ptr(patches_dev), n, self.mode, ptr(out), None), 'dd_fake_encode')
        return out

    def __call__(self, data_in, batch_size=32):
        patches = self.ctx.to_device(np.asarray(data_in, dtype=np.uint8))
        return self.ctx.to_host(self.encode_device(patches))


class DummyImageEncoder(_FakeImageEncoder):
    mode = 0


class ConstantImageEncoder(_FakeImageEncoder):
    mode = 1


def create_box_encoder(model_filename, input_name="images", output_name="features", batch_size=32,
                       num_threads=1, context=None):
    if 'dummy' in model_filename:                                 # generate_detections.py:182-189
        image_encoder = DummyImageEncoder(context)
    elif 'constant' in model_filename:
        image_encoder = ConstantImageEncoder(context)
    else:
        image_encoder = MarsImageEncoder(model_filename, num_threads=num_threads, context=context)
    ctx = image_encoder.ctx
    ph, pw = image_encoder.image_shape[:2]

    def encode_device(frame_dev, H, W, boxes):
        """Hot path: frame already in HBM; returns (features f32 [n,128] on device, valid)."""
        patches, valid = crop_patches_device(ctx, frame_dev, H, W, boxes, ph, pw)
        return image_encoder.encode_device(patches), valid

    def encoder(image, boxes, timing=False):
        if len(boxes) == 0:                                       # generate_detections.py:193-197
            return (np.array([]), 0) if timing else np.array([])
        img = np.ascontiguousarray(image, dtype=np.uint8)
        frame_dev = ctx.to_device(img)
        t1 = time()
        arr = np.asarray([np.asarray(b) for b in boxes])
        feats, valid = encode_device(frame_dev, img.shape[0], img.shape[1],
                                     arr.astype(np.float64 if arr.dtype.kind == 'f' else np.int64))
        result = ctx.to_host(feats)
        t2 = time()
        for box, ok in zip(boxes, valid):
            if not ok:                                            # :201-204 (the reference substitutes noise)
                print("WARNING: Failed to extract image patch: %s." % str(box))
        return (result, t2 - t1) if timing else result

    encoder.image_encoder = image_encoder
    encoder.encode_device = encode_device
    encoder.width, encoder.height = image_encoder.width, image_encoder.height
    return encoder
