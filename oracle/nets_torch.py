"""TEST INFRASTRUCTURE ONLY -- f32 torch-CPU restatement of the three network forward passes.

PARITY UNPINNED against the real reference: the arithmetic lives in tflite_runtime (pins
2.5.0.post1 / 2.9.0 / 2.1.0.post1, Dockerfile:32, Dockerfile.rpi-tflite-armv7:46-47, pip3.lst:105) and
in weight blobs that are absent from the tree (.MISSING_LARGE_BLOBS), and the reference has no test
that pins results at that boundary.  What these functions pin is the build's own HIP path against
an independent f32 implementation of the same architecture on the same named weights:
  * mars_forward       tools/freeze_model.py:13-157 (+ :175-177 BGR->RGB); call contract
                       tools/generate_detections.py:164-177
  * ssd_forward        public TF-OD-API SSD-MobileNet-v1 (no in-tree description)
  * yolov5s_forward    detectors/yolov5/yolov5s.yaml:12-48 (+ public YOLOv5 module definitions)
Also used by bench.py as the timed CPU baseline of the detector / encoder stages ("port").

`w16=True` rounds the (BN-folded) conv weights to f16 first, i.e. uses exactly the weight values
the HIP path holds, so what remains is f16 activation storage and summation order.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _same_pad(x, k, stride):
    """TensorFlow 'SAME' zero padding (asymmetric: the extra pixel goes after)."""
    h, w = x.shape[2:]
    oh, ow = -(-h // stride), -(-w // stride)
    ph = max((oh - 1) * stride + k - h, 0)
    pw = max((ow - 1) * stride + k - w, 0)
    return F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))


def _affine(wd, scope):
    var, mean, beta = wd[scope + '/moving_variance'], wd[scope + '/moving_mean'], wd[scope + '/beta']
    gamma = wd.get(scope + '/gamma', np.ones_like(var))
    s = (gamma.astype(np.float64) / np.sqrt(var.astype(np.float64) + BN_EPS))
    return s.astype(np.float32), (beta - mean * s).astype(np.float32)


def _bn(x, wd, scope):
    s, t = _affine(wd, scope)
    return x * _t(s).view(1, -1, 1, 1) + _t(t).view(1, -1, 1, 1)


def _w(w_hwio, w16):
    w = _t(w_hwio).permute(3, 2, 0, 1).contiguous()          # OIHW
    return w.half().float() if w16 else w


def _conv(x, w_hwio, stride=1, pad='same', w16=False, groups=1):
    k = w_hwio.shape[0]
    if pad == 'same':
        x = _same_pad(x, k, stride)
        p = 0
    else:
        p = pad
    return F.conv2d(x, _w(w_hwio, w16), stride=stride, padding=p, groups=groups)


def _conv_bn(x, wd, scope, stride=1, pad='same', w16=False, bn=None):
    """conv without bias followed by inference BN (folded when w16 so the f16 rounding matches)."""
    s, t = _affine(wd, bn or scope + '/bn')
    if w16:
        return _conv(x, (wd[scope + '/weights'] * s).astype(np.float32), stride, pad, True) + _t(t).view(1, -1, 1, 1)
    return _bn(_conv(x, wd[scope + '/weights'], stride, pad), wd, bn or scope + '/bn')


# ------------------------------------------------------------------------------------------- MARS
def mars_forward(wd, patches_bgr_u8, w16=False):
    """patches u8 [N,64,32,3] BGR -> f32 [N,128] unit-norm (tools/freeze_model.py:88-157)."""
    x = _t(np.asarray(patches_bgr_u8)[..., ::-1].astype(np.float32)).permute(0, 3, 1, 2)     # :175-177 BGR -> RGB
    x = F.elu(_conv_bn(x, wd, 'conv1_1', w16=w16))                                           # :101-105
    x = F.elu(_conv_bn(x, wd, 'conv1_2', w16=w16))                                           # :106-110
    x = F.max_pool2d(x, 3, 2)                                                                # :116 (VALID)
    blocks = [('conv2_1', False, True), ('conv2_3', False, False), ('conv3_1', True, False),
              ('conv3_3', False, False), ('conv4_1', True, False), ('conv4_3', False, False)]
    for name, inc, first in blocks:                                                          # :118-137
        incoming = x
        pre = x if first else F.elu(_bn(x, wd, name + '/bn'))                                # :16-21
        h = F.elu(_conv_bn(pre, wd, name + '/1', stride=2 if inc else 1, w16=w16))           # :58-62
        h = _conv(h, wd[name + '/2/weights'], w16=w16) + _t(wd[name + '/2/biases']).view(1, -1, 1, 1)   # :68-72
        if inc:
            x = _conv(incoming, wd[name + '/projection/weights'], stride=2, w16=w16) + h     # :30-37
        else:
            x = incoming + h                                                                 # :39
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                                        # slim.flatten on NHWC
    s, t = _affine(wd, 'fc1/bn')
    wfc = wd['fc1/weights']
    if w16:
        x = x @ _t((wfc * s).astype(np.float32)).half().float() + _t(t)
    else:
        x = (x @ _t(wfc)) * _t(s) + _t(t)
    x = F.elu(x)                                                                             # :143-147
    s, t = _affine(wd, 'ball')
    x = x * _t(s) + _t(t)                                                                    # :152
    return (x / torch.sqrt(1e-8 + (x * x).sum(1, keepdim=True))).numpy()                     # :153-156


# ------------------------------------------------------------------------------------------- SSD-MobileNet-v1
MOBILENET_V1 = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1), (512, 1), (512, 1), (512, 1),
                (512, 1), (1024, 2), (1024, 1)]
ANCHORS_PER_MAP = [3, 6, 6, 6, 6, 6]


def ssd_forward(wd, img_rgb_u8, w16=False, n_classes=91):
    """u8 [N,300,300,3] RGB -> f32 [N,1917,4+91]: box encodings (ty,tx,th,tw) and class logits."""
    x = _t((np.asarray(img_rgb_u8).astype(np.float32) - 127.5) / 127.5).permute(0, 3, 1, 2)
    relu6 = lambda v: torch.clamp(v, 0.0, 6.0)
    x = relu6(_conv_bn(x, wd, 'conv0', stride=2, w16=w16))
    feats = []
    for i, (c, st) in enumerate(MOBILENET_V1, 1):
        s, t = _affine(wd, f'dw{i}/bn')
        wdw = wd[f'dw{i}/weights']                                        # [3,3,C,1]
        cin = wdw.shape[2]
        if w16:
            w = _t((wdw[:, :, :, 0] * s).astype(np.float32)).half().float().permute(2, 0, 1).unsqueeze(1)
            x = F.conv2d(_same_pad(x, 3, st), w, stride=st, groups=cin) + _t(t).view(1, -1, 1, 1)
        else:
            w = _t(wdw[:, :, :, 0]).permute(2, 0, 1).unsqueeze(1)
            x = _bn(F.conv2d(_same_pad(x, 3, st), w, stride=st, groups=cin), wd, f'dw{i}/bn')
        x = relu6(x)
        x = relu6(_conv_bn(x, wd, f'pw{i}', w16=w16))
        if i in (11, 13):
            feats.append(x)
    for j in range(1, 5):
        x = relu6(_conv_bn(x, wd, f'extra{j}_1', w16=w16))
        x = relu6(_conv_bn(x, wd, f'extra{j}_2', stride=2, w16=w16))
        feats.append(x)
    rows = []
    for k, (f, a) in enumerate(zip(feats, ANCHORS_PER_MAP)):
        n = f.shape[0]
        box = _conv(f, wd[f'box{k}/weights'], w16=w16) + _t(wd[f'box{k}/biases']).view(1, -1, 1, 1)
        cls = _conv(f, wd[f'cls{k}/weights'], w16=w16) + _t(wd[f'cls{k}/biases']).view(1, -1, 1, 1)
        box = box.permute(0, 2, 3, 1).reshape(n, -1, 4)                   # [N, H*W*A, 4]
        cls = cls.permute(0, 2, 3, 1).reshape(n, -1, n_classes)
        rows.append(torch.cat([box, cls], dim=2))
    return torch.cat(rows, dim=1).numpy()


def ssd_postprocess(raw, anchors, max_det=10, score_thr=1e-8, iou_thr=0.6):
    """TFLite_Detection_PostProcess (fast, class-agnostic NMS) for one image, in f32.
    raw [A, 4+C], anchors [A,4] (yc,xc,h,w) -> boxes [max_det,4] (ymin,xmin,ymax,xmax), classes, scores, count."""
    raw = np.asarray(raw, dtype=np.float32)
    a = np.asarray(anchors, dtype=np.float32)
    f = np.float32
    yc = raw[:, 0] / f(10) * a[:, 2] + a[:, 0]
    xc = raw[:, 1] / f(10) * a[:, 3] + a[:, 1]
    hh = f(0.5) * np.exp(raw[:, 2] / f(5)) * a[:, 2]
    hw = f(0.5) * np.exp(raw[:, 3] / f(5)) * a[:, 3]
    boxes = np.stack([yc - hh, xc - hw, yc + hh, xc + hw], axis=1).astype(np.float32)
    sc = (f(1) / (f(1) + np.exp(-raw[:, 5:]))).astype(np.float32)        # class 0 = background
    best_c = sc.argmax(axis=1)
    best = sc[np.arange(len(sc)), best_c]
    return ssd_postprocess_decoded(boxes, best, best_c, max_det, score_thr, iou_thr)


def ssd_postprocess_decoded(boxes, best, best_c, max_det=10, score_thr=1e-8, iou_thr=0.6):
    """Second stage of the op (`use_regular_nms` false: NonMaxSuppressionMultiClassFastHelper) on per-anchor decoded boxes, best scores
    and best classes -- what ssd_postprocess computes in f32 and what oracle/nets_quant.ssd_quant_decode computes from a uint8
    model's head tensors: candidates with score >= nms_score_threshold by descending score, greedy suppression at IoU >
    nms_iou_threshold, the first max_detections survivors."""
    f = np.float32
    boxes = np.asarray(boxes, dtype=np.float32)
    best = np.asarray(best, dtype=np.float32)
    best_c = np.asarray(best_c)
    cand = np.nonzero(best >= f(score_thr))[0]
    order = cand[np.lexsort((-cand, -best[cand]))]                       # score desc, ties: higher index first
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep, active = [], np.ones(len(order), dtype=bool)
    for i, oi in enumerate(order):
        if len(keep) >= max_det:
            break
        if not active[i]:
            continue
        keep.append(oi)
        rest = order[i + 1:]
        y0 = np.maximum(boxes[oi, 0], boxes[rest, 0]); x0 = np.maximum(boxes[oi, 1], boxes[rest, 1])
        y1 = np.minimum(boxes[oi, 2], boxes[rest, 2]); x1 = np.minimum(boxes[oi, 3], boxes[rest, 3])
        inter = np.maximum(y1 - y0, f(0)) * np.maximum(x1 - x0, f(0))
        with np.errstate(divide='ignore', invalid='ignore'):
            iou = inter / (area[oi] + area[rest] - inter)
        iou = np.where((area[oi] <= 0) | (area[rest] <= 0), f(0), iou)
        active[i + 1:] &= ~(iou > f(iou_thr))
    ob, oc, os_ = np.zeros((max_det, 4), np.float32), np.zeros(max_det, np.float32), np.zeros(max_det, np.float32)
    for j, k in enumerate(keep):
        ob[j], oc[j], os_[j] = boxes[k], best_c[k], best[k]
    return ob, oc, os_, len(keep)


# ------------------------------------------------------------------------------------------- YOLOv5s
YOLO_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def yolov5s_forward(wd, img_rgb_u8, w16=False, nc=80):
    """u8 [N,640,640,3] RGB -> f32 [N,25200,85] (xywh normalised to 0..1, obj, class scores)."""
    img = np.asarray(img_rgb_u8).astype(np.float32)
    size = img.shape[1]
    x = _t(img).permute(0, 3, 1, 2)
    silu = F.silu

    def cv(name, v, k=1, s=1):
        return silu(_conv_bn(v, wd, name, stride=s, pad=k // 2, w16=w16))

    def c3(name, v, n, shortcut):
        y = cv(name + '.cv1', v)
        for i in range(n):
            h = cv(f'{name}.m{i}.cv2', cv(f'{name}.m{i}.cv1', y), 3)
            y = y + h if shortcut else h
        return cv(name + '.cv3', torch.cat([y, cv(name + '.cv2', v)], 1))

    x = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)   # Focus
    x = cv('m0.focus', x, 3)
    x = cv('m1', x, 3, 2); x = c3('m2', x, 1, True)
    x = cv('m3', x, 3, 2); x4 = c3('m4', x, 3, True)
    x = cv('m5', x4, 3, 2); x6 = c3('m6', x, 3, True)
    x = cv('m7', x6, 3, 2)
    x = cv('m8.cv1', x)
    x = cv('m8.cv2', torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1))       # SPP
    x = c3('m9', x, 1, False)
    x10 = cv('m10', x)
    x = c3('m13', torch.cat([F.interpolate(x10, scale_factor=2, mode='nearest'), x6], 1), 1, False)
    x14 = cv('m14', x)
    p3 = c3('m17', torch.cat([F.interpolate(x14, scale_factor=2, mode='nearest'), x4], 1), 1, False)
    p4 = c3('m20', torch.cat([cv('m18', p3, 3, 2), x14], 1), 1, False)
    p5 = c3('m23', torch.cat([cv('m21', p4, 3, 2), x10], 1), 1, False)
    outs = []
    no = 5 + nc
    for i, p in enumerate((p3, p4, p5)):
        n, _, ny, nx = p.shape
        stride = size // ny
        y = _conv(p, wd[f'detect{i}/weights'], pad=0, w16=w16) + _t(wd[f'detect{i}/biases']).view(1, -1, 1, 1)
        y = torch.sigmoid(y.view(n, 3, no, ny, nx).permute(0, 1, 3, 4, 2))                # [n, a, y, x, no]
        gy, gx = torch.meshgrid(torch.arange(ny, dtype=torch.float32), torch.arange(nx, dtype=torch.float32), indexing='ij')
        anc = torch.tensor(YOLO_ANCHORS[i], dtype=torch.float32).view(3, 1, 1, 2)
        xy = (y[..., 0:2] * 2 - 0.5 + torch.stack([gx, gy], -1)) * stride / size
        wh = (y[..., 2:4] * 2) ** 2 * anc / size
        outs.append(torch.cat([xy, wh, y[..., 4:]], -1).reshape(n, -1, no))
    return torch.cat(outs, 1).numpy()


def yolov5_decode(raw, thr, img_w, img_h):
    """tools/yolov5.py:120-131 on one image's [N,85] rows -> (xyxy f32 [K,4], conf f32 [K], cls int [K])."""
    x = np.copy(np.asarray(raw, dtype=np.float32))
    boxes = np.copy(x[..., :4])
    boxes[..., 0] = x[..., 0] - x[..., 2] / 2
    boxes[..., 1] = x[..., 1] - x[..., 3] / 2
    boxes[..., 2] = x[..., 0] + x[..., 2] / 2
    boxes[..., 3] = x[..., 1] + x[..., 3] / 2
    x[..., 5:] *= x[..., 4:5]
    best = np.expand_dims(np.argmax(x[..., 5:], axis=-1), axis=-1)
    conf = np.take_along_axis(x, best + 5, axis=-1)
    y = np.concatenate((boxes, conf, best.astype(np.float32)), axis=-1)
    y = y[np.where(y[..., 4] >= thr)]
    y[..., :4] *= np.array([img_w, img_h, img_w, img_h])
    return y[:, :4], y[:, 4], y[:, 5].astype(np.int64)
