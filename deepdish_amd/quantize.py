"""Post-training quantiser: named f32 SSD-MobileNet-v1 weights -> the uint8 model the reference's detector file is.

The reference's `detectors/mobilenet/ssdmobilenetv1.tflite` is the COCO "quant" SSD-MobileNet-v1: every tensor
uint8 with one (scale, zero point) pair, int32 biases at scale in*w, ReLU6 as the clamp of the output range
(tools/ssd_mobilenet.py:102 feeds uint8; the blob is absent, .MISSING_LARGE_BLOBS).  This module builds a model
of exactly that form -- a `QModel` dict -- from f32 weights, so the integer kernels of csrc/netsq.hip have
something to run when no .tflite is at hand; tools/tflite_reader.py produces the same dict from a real file.

QModel:
  kind    'ssd_mobilenet_v1_uint8'
  input   {'scale', 'zp', 'size'}
  layers  name -> {'kind': 'conv' | 'dw', 'w': u8 HWIO (conv) / HWC (dw), 'w_scale', 'w_zp', 'bias': i32 [Cout],
                   'stride', 'act': 'relu6' | 'none', 'in_scale', 'in_zp', 'out_scale', 'out_zp'}
          names: conv0, dw1..13, pw1..13, extra{1..4}_{1,2}, box{0..5}, cls{0..5}
  logistic {'out_scale': 1/256, 'out_zp': 0}   (class scores; uint8 LOGISTIC is a 256-entry table in TFLite >= 2.2)

Scales follow the converter's conventions: ReLU6 outputs span [0, 6] (scale 6/255, zero point 0); weights
span [min(w, 0), max(w, 0)]; the six box tensors share one (scale, zp) and so do the six class tensors
(CONCATENATION needs equal parameters); those two ranges come from a float forward over seeded frames.
"""
import numpy as np

from . import nets

RELU6_SCALE = np.float32(6.0 / 255.0)


def _same_pad(x, k, stride):
    h, w = x.shape[1:3]
    oh, ow = -(-h // stride), -(-w // stride)
    ph, pw = max((oh - 1) * stride + k - h, 0), max((ow - 1) * stride + k - w, 0)
    return np.pad(x, ((0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0))), oh, ow


def _conv_f32(x, w_hwio, stride):
    """NHWC f32 conv, TF SAME padding, as k*k strided matmuls."""
    k = w_hwio.shape[0]
    xp, oh, ow = _same_pad(x, k, stride)
    out = np.zeros(x.shape[:1] + (oh, ow, w_hwio.shape[3]), np.float32)
    for dy in range(k):
        for dx in range(k):
            out += xp[:, dy:dy + (oh - 1) * stride + 1:stride, dx:dx + (ow - 1) * stride + 1:stride, :] @ w_hwio[dy, dx]
    return out


def _dw_f32(x, w_hwc, stride):
    xp, oh, ow = _same_pad(x, 3, stride)
    out = np.zeros(x.shape[:1] + (oh, ow, x.shape[3]), np.float32)
    for dy in range(3):
        for dx in range(3):
            out += xp[:, dy:dy + (oh - 1) * stride + 1:stride, dx:dx + (ow - 1) * stride + 1:stride, :] * w_hwc[dy, dx]
    return out


def folded_ssd_layers(wd):
    """(name, kind, f32 weights BN-folded, f32 bias, stride, act) in execution order; heads last."""
    out = []
    w, b = nets.fold_conv_bn(wd, 'conv0')
    out.append(('conv0', 'conv', w, b, 2, 'relu6'))
    for i, (c, st) in enumerate(nets.MOBILENET_V1, 1):
        s, t = nets.bn_affine(wd, f'dw{i}/bn')
        out.append((f'dw{i}', 'dw', (wd[f'dw{i}/weights'][:, :, :, 0] * s).astype(np.float32), t, st, 'relu6'))
        w, b = nets.fold_conv_bn(wd, f'pw{i}')
        out.append((f'pw{i}', 'conv', w, b, 1, 'relu6'))
    for j in range(1, 5):
        w, b = nets.fold_conv_bn(wd, f'extra{j}_1'); out.append((f'extra{j}_1', 'conv', w, b, 1, 'relu6'))
        w, b = nets.fold_conv_bn(wd, f'extra{j}_2'); out.append((f'extra{j}_2', 'conv', w, b, 2, 'relu6'))
    for k in range(6):
        out.append((f'box{k}', 'conv', wd[f'box{k}/weights'], wd[f'box{k}/biases'], 1, 'none'))
        out.append((f'cls{k}', 'conv', wd[f'cls{k}/weights'], wd[f'cls{k}/biases'], 1, 'none'))
    return out


def _float_heads(layers, frames_u8):
    """f32 forward of the folded network on u8 RGB frames -> (all box encodings, all class logits) flattened."""
    L = {n: (k, w, b, s, a) for n, k, w, b, s, a in layers}

    def run(name, x):
        k, w, b, s, a = L[name]
        y = (_dw_f32(x, w, s) if k == 'dw' else _conv_f32(x, w, s)) + b
        return np.clip(y, 0.0, 6.0) if a == 'relu6' else y

    x = (frames_u8.astype(np.float32) - 128.0) * np.float32(1.0 / 128.0)
    x = run('conv0', x)
    feats = []
    for i in range(1, 14):
        x = run(f'pw{i}', run(f'dw{i}', x))
        if i in (11, 13):
            feats.append(x)
    for j in range(1, 5):
        x = run(f'extra{j}_2', run(f'extra{j}_1', x))
        feats.append(x)
    box = np.concatenate([run(f'box{k}', f).reshape(-1) for k, f in enumerate(feats)])
    cls = np.concatenate([run(f'cls{k}', f).reshape(-1) for k, f in enumerate(feats)])
    return box, cls


def _range_params(lo, hi):
    """uint8 (scale, zero point) of a real range that must contain 0 (the converter's nudged form)."""
    lo, hi = min(float(lo), 0.0), max(float(hi), 0.0)
    if hi == lo:
        return np.float32(1.0), 0
    scale = (hi - lo) / 255.0
    zp = int(np.clip(np.round(-lo / scale), 0, 255))
    return np.float32(scale), zp


def calibration_frames(n=2, size=300, seed=77):
    """Seeded smooth-background frames with textured rectangles (SURVEY 8d's synthetic frames, at the net's input size)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, size, size, 3), np.uint8)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    for i in range(n):
        img = np.zeros((size, size, 3), np.float32)
        for c in range(3):
            a, b, p = rng.uniform(0.01, 0.05, 3)
            img[..., c] = 110 + 60 * np.sin(a * xx + p * 10) * np.cos(b * yy) + rng.normal(0, 6, (size, size))
        for _ in range(12):
            h, w = int(rng.integers(30, 90)), int(rng.integers(15, 50))
            y, x = int(rng.integers(0, size - h)), int(rng.integers(0, size - w))
            img[y:y + h, x:x + w] = rng.uniform(0, 255, 3) + rng.normal(0, 25, (h, w, 3))
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def quantize_ssd_mobilenet(wd, calib_frames=None, symmetric_weights=False):
    """f32 named weights (nets.synthetic_ssd_weights / an .npz) -> QModel.  symmetric_weights=True pins every weight
    zero point at 128 (ranges widened to +-max|w|): the form whose pointwise layers need no activation row sums."""
    layers = folded_ssd_layers(wd)
    frames = calibration_frames() if calib_frames is None else calib_frames
    box, cls = _float_heads(layers, frames)
    box_q = _range_params(box.min(), box.max())
    cls_q = _range_params(cls.min(), cls.max())
    qm = dict(kind='ssd_mobilenet_v1_uint8', input=dict(scale=np.float32(1.0 / 128.0), zp=128, size=300), layers={},
              logistic=dict(out_scale=np.float32(1.0 / 256.0), out_zp=0), order=[n for n, *_ in layers])
    outq = {}                                            # tensor produced by layer name -> (scale, zp)
    prev = ('input', qm['input']['scale'], qm['input']['zp'])
    feat_of = {}
    for name, kind, w, b, stride, act in layers:
        if name.startswith(('box', 'cls')):
            k = int(name[3:])
            src = ['pw11', 'pw13', 'extra1_2', 'extra2_2', 'extra3_2', 'extra4_2'][k]
            in_scale, in_zp = outq[src]
        else:
            in_scale, in_zp = prev[1], prev[2]
        if symmetric_weights:
            m = float(np.abs(w).max())
            w_scale, w_zp = np.float32(m / 127.0 if m > 0 else 1.0), 128
        else:
            w_scale, w_zp = _range_params(w.min(), w.max())
        wq = np.clip(np.round(w / w_scale) + w_zp, 0, 255).astype(np.uint8)
        bias_scale = np.float64(np.float32(in_scale) * np.float32(w_scale))
        bq = np.round(b.astype(np.float64) / bias_scale).astype(np.int64)
        assert np.abs(bq).max() < 2 ** 31
        if act == 'relu6':
            out_scale, out_zp = RELU6_SCALE, 0
        elif name.startswith('box'):
            out_scale, out_zp = box_q
        else:
            out_scale, out_zp = cls_q
        qm['layers'][name] = dict(kind=kind, w=wq, w_scale=np.float32(w_scale), w_zp=int(w_zp), bias=bq.astype(np.int32),
                                  stride=stride, act=act, in_scale=np.float32(in_scale), in_zp=int(in_zp),
                                  out_scale=np.float32(out_scale), out_zp=int(out_zp))
        outq[name] = (np.float32(out_scale), int(out_zp))
        if not name.startswith(('box', 'cls')):
            prev = (name, np.float32(out_scale), int(out_zp))
    return qm


def synthetic_ssd_quant_model(seed=1234, symmetric_weights=False):
    return quantize_ssd_mobilenet(nets.synthetic_ssd_weights(seed), symmetric_weights=symmetric_weights)


# ------------------------------------------------------------------------------------------- fixed-point parameters
def quantize_multiplier(real):
    """TFLite QuantizeMultiplier (kernels/internal/quantization_util.cc): real = q * 2^shift / 2^31, q in [2^30, 2^31)."""
    real = float(real)
    if real == 0.0:
        return 0, 0
    q, shift = np.frexp(real)
    q_fixed = int(np.floor(q * (1 << 31) + 0.5))          # TfLiteRound rounds halves away from zero (np.round: to even); q > 0
    if q_fixed == (1 << 31):
        q_fixed //= 2
        shift += 1
    if shift < -31:
        return 0, 0
    return int(q_fixed), int(shift)


def conv_multiplier(layer):
    """GetQuantizedConvolutionMultipler (kernels/kernel_util.cc): the in*w product is taken in float, the quotient in double."""
    prod = np.float64(np.float32(layer['in_scale']) * np.float32(layer['w_scale']))
    return quantize_multiplier(prod / np.float64(np.float32(layer['out_scale'])))


def activation_range(layer):
    """CalculateActivationRangeQuantized for uint8: the clamp of the output bytes."""
    if layer['act'] == 'relu6':
        s = np.float32(layer['out_scale'])
        q = lambda f: int(layer['out_zp']) + int(np.floor(np.float32(f) / s + np.float32(0.5)))      # TfLiteRound of a value >= 0
        return max(0, q(0.0)), min(255, q(6.0))
    return 0, 255


def logistic_table(in_scale, in_zp, out_scale=np.float32(1.0 / 256.0), out_zp=0):
    """uint8 LOGISTIC as TFLite >= 2.2 evaluates it (kernels/activations.cc PopulateLookupTable): one f32 evaluation per
    input byte, rescaled, rounded half away from zero, clamped."""
    q = np.arange(256, dtype=np.float32)
    x = np.float32(in_scale) * (q - np.float32(in_zp))
    with np.errstate(over='ignore'):
        y = np.float32(1.0) / (np.float32(1.0) + np.exp(-x, dtype=np.float32))
    r = y / np.float32(out_scale) + np.float32(out_zp)
    r = np.where(r >= 0, np.floor(r + np.float32(0.5)), np.ceil(r - np.float32(0.5)))
    return np.clip(r, 0, 255).astype(np.uint8)
