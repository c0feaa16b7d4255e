"""TEST INFRASTRUCTURE ONLY -- integer restatement of the uint8 SSD-MobileNet-v1 forward as TFLite's reference
kernels evaluate it (what `interpreter.invoke()` does at tools/ssd_mobilenet.py:102-103 for the reference's
`ssdmobilenetv1.tflite`, a uint8-quantised model).

PARITY UNPINNED against the real reference: tflite_runtime (pins 2.5.0.post1 / 2.9.0 / 2.1.0.post1) and the model
blob are absent from this image and from the tree (.MISSING_LARGE_BLOBS); the reference holds no vector at this
boundary.  Restated from the published TFLite sources:
  * conv / depthwise conv, uint8: kernels/internal/reference/conv.h, depthwiseconv_uint8.h --
        acc(int32) = sum (in + in_offset) * (w + w_offset) over the taps inside the image;  acc += bias;
        acc = MultiplyByQuantizedMultiplier(acc, M, shift);  acc += out_offset;  clamp to [act_min, act_max]
  * MultiplyByQuantizedMultiplier: kernels/internal/common.h (gemmlowp SaturatingRoundingDoublingHighMul then
        RoundingDivideByPOT -- two roundings, written out literally below)
  * QuantizeMultiplier, GetQuantizedConvolutionMultipler, CalculateActivationRangeQuantized:
        kernels/internal/quantization_util.cc, kernels/kernel_util.cc
  * LOGISTIC uint8: kernels/activations.cc (256-entry table, TFLite >= 2.2)
  * TFLite_Detection_PostProcess with quantised inputs: kernels/detection_postprocess.cc
        (DequantizeBoxEncodings / DequantizeClassPredictions, then the float path of oracle/nets_torch.py)

The integer sums run as float64 matrix products: every product is < 2^16 and every sum < 2^27, so f64 is exact.
Nothing here is imported by the product (deepdish_amd/); the model description it reads (`QModel`, see
deepdish_amd/quantize.py) is plain data.
"""
import math
import numpy as np

MOBILENET_BLOCKS = 13
FEATURE_LAYERS = ['pw11', 'pw13', 'extra1_2', 'extra2_2', 'extra3_2', 'extra4_2']
ANCHORS_PER_MAP = [3, 6, 6, 6, 6, 6]


# ------------------------------------------------------------------------------------------- fixed point
def srdhm(a, b):
    """gemmlowp SaturatingRoundingDoublingHighMul on int32 arrays (b scalar)."""
    a = np.asarray(a, dtype=np.int64)
    ab = a * np.int64(b)
    nudge = np.where(ab >= 0, np.int64(1 << 30), np.int64(1 - (1 << 30)))
    q = ab + nudge
    res = np.where(q >= 0, q >> 31, -((-q) >> 31))               # C++ integer division truncates toward zero
    return np.where((a == b) & (a == -(1 << 31)), np.int64((1 << 31) - 1), res)


def rounding_divide_by_pot(x, exponent):
    x = np.asarray(x, dtype=np.int64)
    mask = np.int64((1 << exponent) - 1)
    rem = x & mask
    thr = (mask >> 1) + (x < 0)
    return (x >> exponent) + (rem > thr)


def multiply_by_quantized_multiplier(x, m, shift):
    left, right = (shift, 0) if shift > 0 else (0, -shift)
    return rounding_divide_by_pot(srdhm(np.asarray(x, dtype=np.int64) * (1 << left), m), right)


def quantize_multiplier(real):
    if real == 0.0:
        return 0, 0
    q, shift = math.frexp(real)
    qf = int(math.floor(q * (1 << 31) + 0.5))    # TfLiteRound = std::round: halves away from zero (q > 0 here), not Python's half-to-even
    if qf == (1 << 31):
        qf //= 2
        shift += 1
    if shift < -31:
        return 0, 0
    return qf, shift


def layer_fixed_point(L):
    prod = float(np.float32(L['in_scale']) * np.float32(L['w_scale']))       # float product, then double
    m, shift = quantize_multiplier(prod / float(np.float32(L['out_scale'])))
    if L['act'] == 'relu6':
        s = np.float32(L['out_scale'])
        rnd = lambda f: int(math.floor(float(np.float32(f) / s) + 0.5))
        lo, hi = max(0, L['out_zp'] + rnd(0.0)), min(255, L['out_zp'] + rnd(6.0))
    else:
        lo, hi = 0, 255
    return m, shift, lo, hi


# ------------------------------------------------------------------------------------------- layers
def _same_geometry(size, k, stride):
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2


def _finish(acc, L):
    m, shift, lo, hi = layer_fixed_point(L)
    acc = acc + L['bias'].astype(np.int64)
    acc = multiply_by_quantized_multiplier(acc, m, shift) + L['out_zp']
    return np.clip(acc, lo, hi).astype(np.uint8)


def conv_u8(x, L):
    """x u8 [N,H,W,Cin]; taps outside the image are skipped (reference conv.h), i.e. contribute nothing."""
    w = L['w'].astype(np.int64) - L['w_zp']                    # [KH,KW,Cin,Cout]
    k, stride = w.shape[0], L['stride']
    n, h, wd, cin = x.shape
    oh, pt = _same_geometry(h, k, stride)
    ow, pl = _same_geometry(wd, k, stride)
    xi = x.astype(np.int64) - L['in_zp']
    xp = np.zeros((n, (oh - 1) * stride + k, (ow - 1) * stride + k, cin), np.int64)       # 0 = (zp - zp): nothing added
    hh, ww = min(h, xp.shape[1] - pt), min(wd, xp.shape[2] - pl)
    xp[:, pt:pt + hh, pl:pl + ww] = xi[:, :hh, :ww]
    acc = np.zeros((n, oh, ow, w.shape[3]), np.float64)
    for dy in range(k):
        for dx in range(k):
            patch = xp[:, dy:dy + (oh - 1) * stride + 1:stride, dx:dx + (ow - 1) * stride + 1:stride, :]
            acc += patch.astype(np.float64) @ w[dy, dx].astype(np.float64)
    assert np.abs(acc).max() < 2 ** 52
    return _finish(acc.astype(np.int64), L)


def dwconv_u8(x, L):
    w = L['w'].astype(np.int64) - L['w_zp']                    # [3,3,C]
    stride = L['stride']
    n, h, wd, c = x.shape
    oh, pt = _same_geometry(h, 3, stride)
    ow, pl = _same_geometry(wd, 3, stride)
    xi = x.astype(np.int64) - L['in_zp']
    xp = np.zeros((n, (oh - 1) * stride + 3, (ow - 1) * stride + 3, c), np.int64)
    hh, ww = min(h, xp.shape[1] - pt), min(wd, xp.shape[2] - pl)
    xp[:, pt:pt + hh, pl:pl + ww] = xi[:, :hh, :ww]
    acc = np.zeros((n, oh, ow, c), np.int64)
    for dy in range(3):
        for dx in range(3):
            acc += xp[:, dy:dy + (oh - 1) * stride + 1:stride, dx:dx + (ow - 1) * stride + 1:stride, :] * w[dy, dx]
    return _finish(acc, L)


def logistic_table(in_scale, in_zp, out_scale=1.0 / 256.0, out_zp=0):
    tab = np.zeros(256, np.uint8)
    for q in range(256):
        x = np.float32(in_scale) * (np.float32(q) - np.float32(in_zp))
        with np.errstate(over='ignore'):
            y = np.float32(1.0) / (np.float32(1.0) + np.exp(-x))            # std::exp on a float: evaluated in f32
        r = float(y / np.float32(out_scale) + np.float32(out_zp))
        tab[q] = min(255, max(0, int(math.floor(r + 0.5)) if r >= 0 else int(math.ceil(r - 0.5))))
    return tab


# ------------------------------------------------------------------------------------------- network
def ssd_quant_forward(qm, img_rgb_u8, keep=()):
    """u8 [N,300,300,3] RGB -> (box u8 [N,1917,4], class logits u8 [N,1917,91], {name: tensor} for names in keep)."""
    Ls = qm['layers']
    x = np.asarray(img_rgb_u8, dtype=np.uint8)
    kept = {}

    def run(name, v):
        y = dwconv_u8(v, Ls[name]) if Ls[name]['kind'] == 'dw' else conv_u8(v, Ls[name])
        if name in keep:
            kept[name] = y
        return y

    x = run('conv0', x)
    feats = {}
    for i in range(1, MOBILENET_BLOCKS + 1):
        x = run(f'pw{i}', run(f'dw{i}', x))
        feats[f'pw{i}'] = x
    for j in range(1, 5):
        x = run(f'extra{j}_2', run(f'extra{j}_1', x))
        feats[f'extra{j}_2'] = x
    n = x.shape[0]
    box = np.concatenate([run(f'box{k}', feats[f]).reshape(n, -1, 4) for k, f in enumerate(FEATURE_LAYERS)], axis=1)
    cls = np.concatenate([run(f'cls{k}', feats[f]).reshape(n, -1, Ls[f'cls{k}']['w'].shape[3] // ANCHORS_PER_MAP[k])
                          for k, f in enumerate(FEATURE_LAYERS)], axis=1)
    return box, cls, kept


def ssd_quant_decode(qm, box_u8, cls_u8, anchors, score_thr=1e-8):
    """First stage of the post-process op on one image's quantised tensors: dequantise, LOGISTIC table, best class
    (background skipped, lowest class on ties), anchor decode -> (boxes f32 [A,4], score f32 [A], class int [A], key f32 [A])."""
    Lb, Lc = qm['layers']['box0'], qm['layers']['cls0']
    f = np.float32
    raw = f(Lb['out_scale']) * (box_u8.astype(np.float32) - f(Lb['out_zp']))
    tab = logistic_table(Lc['out_scale'], Lc['out_zp'], qm['logistic']['out_scale'], qm['logistic']['out_zp'])
    sq = tab[cls_u8[:, 1:]]                                   # class 0 = background
    best_c = sq.argmax(axis=1)                                # first maximum = lowest class
    best_q = sq[np.arange(len(sq)), best_c]
    score = f(qm['logistic']['out_scale']) * (best_q.astype(np.float32) - f(qm['logistic']['out_zp']))
    a = np.asarray(anchors, dtype=np.float32)
    yc = raw[:, 0] / f(10) * a[:, 2] + a[:, 0]
    xc = raw[:, 1] / f(10) * a[:, 3] + a[:, 1]
    hh = f(0.5) * np.exp(raw[:, 2] / f(5)) * a[:, 2]
    hw = f(0.5) * np.exp(raw[:, 3] / f(5)) * a[:, 3]
    boxes = np.stack([yc - hh, xc - hw, yc + hh, xc + hw], axis=1).astype(np.float32)
    keys = np.where(score >= f(score_thr), score, f(-1)).astype(np.float32)
    return boxes, score.astype(np.float32), best_c.astype(np.int32), keys


# ------------------------------------------------------------------------------------------- anchors
def ssd_anchors(in_size=300):
    """TF Object Detection API `ssd_anchor_generator` for ssd_mobilenet_v1 (anchor_generators/
    multiple_grid_anchor_generator.py create_ssd_anchors): num_layers 6, min_scale 0.2, max_scale 0.95, aspect ratios
    (1, 2, 1/2, 3, 1/3), interpolated_scale_aspect_ratio 1, reduce_boxes_in_lowest_layer; feature maps 19, 10, 5, 3, 2, 1
    for a 300 x 300 input; anchor stride 1/map, offset 0.5/map.  -> f32 [1917, 4] rows (ycenter, xcenter, h, w), ordered
    map, row, column, anchor -- written independently of deepdish_amd.nets.ssd_anchors (which tests compare it with)."""
    size = in_size
    maps = []
    strides = [2] + [1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1] + [2, 2, 2, 2]       # conv0, the 13 depthwise layers, 4 extra 3x3/2 layers
    for i, s in enumerate(strides):
        size = -(-size // s)
        if i in (11, 13, 14, 15, 16, 17):                     # after pw11 (19), pw13 (10), and each extra layer
            maps.append(size)
    num_layers = 6
    scales = [0.2 + (0.95 - 0.2) * i / (num_layers - 1) for i in range(num_layers)] + [1.0]
    rows = []
    for layer, fm in enumerate(maps):
        if layer == 0:
            layer_specs = [(0.1, 1.0), (scales[0], 2.0), (scales[0], 0.5)]
        else:
            layer_specs = [(scales[layer], ar) for ar in (1.0, 2.0, 0.5, 3.0, 1.0 / 3.0)]
            layer_specs.append((math.sqrt(scales[layer] * scales[layer + 1]), 1.0))
        for y in range(fm):
            for x in range(fm):
                for scale, ar in layer_specs:
                    r = math.sqrt(ar)
                    rows.append(((y + 0.5) / fm, (x + 0.5) / fm, scale / r, scale * r))
    return np.asarray(rows, dtype=np.float32)
