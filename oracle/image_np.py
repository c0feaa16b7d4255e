"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the image front-end of the hot path.

  * crop_box / extract_image_patch : tools/generate_detections.py:40-84 (upstream).  The integer box
    arithmetic is replayed with the very same numpy statements (in-place float->int64 truncation),
    so it is exact by construction.
  * resize_linear_u8 : cv2.resize(..., INTER_LINEAR) on u8 as generate_detections.py:83 and
    tools/tflite_object_detector.py:211 call it.  OpenCV is a third-party dependency that is absent
    from this image: the published fixed-point algorithm (11-bit coefficients, two passes, the
    exact 2x2 area shortcut) is restated here -- PARITY UNPINNED against a real cv2.
  * lanczos_resize_u8 : PIL Image.resize(size, ANTIALIAS) as tools/ssd_mobilenet.py:55 and
    tools/yolov5.py:99 call it (ANTIALIAS == LANCZOS).  Pillow's 8-bit two-pass resampler (22-bit
    coefficients, u8 rounding between passes) is restated; PINNED against Pillow itself
    (tests/test_oracle_image.py), which is installed here and on the GPU box.
"""
import math
import numpy as np


# ----------------------------------------------------------------------------- crops
def crop_box(bbox, patch_shape, image_hw):
    """generate_detections.py:63-80 -> (sx, sy, ex, ey) or None.  bbox keeps the caller's dtype."""
    bbox = np.array(bbox)
    if patch_shape is not None:
        target_aspect = float(patch_shape[1]) / patch_shape[0]
        new_width = target_aspect * bbox[3]
        bbox[0] -= (new_width - bbox[2]) / 2
        bbox[2] = new_width
    bbox[2:] += bbox[:2]
    bbox = bbox.astype(int)
    bbox[:2] = np.maximum(0, bbox[:2])
    bbox[2:] = np.minimum(np.asarray(image_hw[::-1]) - 1, bbox[2:])
    if np.any(bbox[:2] >= bbox[2:]):
        return None
    return tuple(int(v) for v in bbox)


def _linear_coeffs(dst, src):
    """OpenCV resize.cpp, linear branch: per destination index -> (source index, short alpha0, alpha1)."""
    scale = 1.0 / (float(dst) / float(src))
    idx = np.zeros(dst, dtype=np.int64)
    a0 = np.zeros(dst, dtype=np.int64)
    a1 = np.zeros(dst, dtype=np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = np.float32(f - np.float32(s))
        if s < 0:
            f, s = np.float32(0), 0
        if s >= src - 1:
            f, s = np.float32(0), src - 1
        c0, c1 = np.float32(1.0) - f, f
        idx[d] = s
        a0[d] = int(np.rint(np.float32(c0 * np.float32(2048))))      # saturate_cast<short>: round half even
        a1[d] = int(np.rint(np.float32(c1 * np.float32(2048))))
    return idx, a0, a1


def resize_linear_u8(img, out_w, out_h):
    """img u8 [h, w, c] -> u8 [out_h, out_w, c] like cv2.resize(img, (out_w, out_h))."""
    h, w = img.shape[:2]
    src = img.astype(np.int64)
    if w == 2 * out_w and h == 2 * out_h:                              # INTER_LINEAR -> INTER_AREA fast path
        s = src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2]
        return ((s + 2) >> 2).astype(np.uint8)
    xi, xa0, xa1 = _linear_coeffs(out_w, w)
    yi, ya0, ya1 = _linear_coeffs(out_h, h)
    x1 = np.minimum(xi + 1, w - 1)
    rows = src[:, xi] * xa0[None, :, None] + src[:, x1] * xa1[None, :, None]       # scale 2^11
    y1 = np.minimum(yi + 1, h - 1)
    r0, r1 = rows[yi], rows[y1]
    out = (((ya0[:, None, None] * (r0 >> 4)) >> 16) + ((ya1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def extract_image_patch(image, bbox, patch_shape):
    """generate_detections.py:40-84: None where the reference returns None."""
    box = crop_box(bbox, patch_shape, image.shape[:2])
    if box is None:
        return None
    sx, sy, ex, ey = box
    return resize_linear_u8(image[sy:ey, sx:ex], patch_shape[1], patch_shape[0])


# ----------------------------------------------------------------------------- Pillow Lanczos
PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def lanczos_coeffs(in_size, out_size):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (ksize, bounds[out,2], kk[out,ksize] int)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _resample_axis0(img, out_size):
    """Resample along axis 0 of an int array [n, ...] with Pillow's 8-bit accumulation."""
    ksize, bounds, kk = lanczos_coeffs(img.shape[0], out_size)
    out = np.zeros((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def lanczos_resize_u8(img, out_w, out_h):
    """img u8 [h, w, c] -> u8 [out_h, out_w, c]; horizontal pass first, u8 in between (Resample.c)."""
    h, w = img.shape[:2]
    tmp = img
    if out_w != w:
        tmp = _resample_axis0(np.ascontiguousarray(img.transpose(1, 0, 2)), out_w).transpose(1, 0, 2)
    if out_h != h:
        tmp = _resample_axis0(np.ascontiguousarray(tmp), out_h)
    return np.ascontiguousarray(tmp)
