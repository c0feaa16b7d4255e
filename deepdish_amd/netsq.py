"""Compiler for uint8 models: QModel (deepdish_amd/quantize.py, or tools/tflite_reader.py from a .tflite) -> op program
for csrc/netsq.hip.  The analogue of handing `ssdmobilenetv1.tflite` to `Interpreter(model_path=...)`
(tools/ssd_mobilenet.py:35-38 upstream): the arithmetic is the file's own -- uint8 tensors, int32 accumulators, TFLite's
fixed-point requantisation.

Weight packing (see csrc/netsq.hip): w' = w - 128 as i8 in MFMA A-fragment order [16-row fragment][k step][lane][16 B];
k step = (tap, 64-channel slice); `cbias[c] = bias[c] - za' * sum_k w'[c][k] + K za' zw'` with za' = za - 128,
zw' = zw - 128; `zwc = 128 - zw` multiplies the device-side row sum of the operand bytes.
"""
import os
import numpy as np

from . import nets, quantize
from .nets import Program, same_pad, DT_U8

OP_QCONV0, OP_QCONV, OP_QDW, OP_QDWPW, OP_QSSD_DECODE = 16, 17, 18, 19, 20
QEPI_Q16, QEPI_ROWS = 0, 1
FUSE_BLOCKS = os.environ.get('DD_Q_FUSE', '1') != '0'      # MobileNet blocks as one launch each (q_dwpw_k); 0: depthwise and pointwise ops
SPLIT_PW = os.environ.get('DD_Q_SPLIT_PW', '1') != '0'      # pointwise filters of the blocks with <= SPLIT_PW_MAX_CIN input channels as hi + lo parts (no row sums)
SPLIT_PW_MAX_CIN = int(os.environ.get('DD_Q_SPLIT_PW_MAX_CIN', '128'))
SPLIT_PW_MIN_CIN = int(os.environ.get('DD_Q_SPLIT_PW_MIN_CIN', '64'))       # block 1 (32 channels) measured slower split: 308 vs 294 us
DUP32 = os.environ.get('DD_Q_DUP32', '1') != '0'              # block 1: split pointwise filter through the free half of the k slice
MERGE_HEADS = os.environ.get('DD_Q_MERGE_HEADS', '1') != '0'  # class + box predictor of a feature map with 512 / 1024 channels as one op
FUSED_SHAPES = {(32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 512, 2), (512, 512, 1)}
FEATURE_LAYERS = ['pw11', 'pw13', 'extra1_2', 'extra2_2', 'extra3_2', 'extra4_2']


def _req_words(L):
    m, shift = quantize.conv_multiplier(L)
    if shift > 0:
        raise ValueError('uint8 layer with a multiplier >= 1 (shift %d): not built' % shift)
    lo, hi = quantize.activation_range(L)
    linear = int(L['act'] == 'none')
    if not linear and lo < L['out_zp']:
        raise ValueError('activation clamp below the zero point')
    return {32: m, 33: -shift, 36: lo, 37: hi, 40: int(L['out_zp']), 41: linear}


def folded_addends(cb, rw):
    """Per-channel 64-bit addend of the ReLU-type requantisation z = ((acc + cb) M + C) >> (31 + e)  ->  (acc M + [cb M + C]) >> (31 + e),
    C = 2^30 + 2^(30+e) + (zo << (31+e)) (csrc/netsq.hip make_req): accumulators start at zero and v_mad_i64_i32 adds the channel's constant."""
    m, e, zo = int(rw[32]), int(rw[33]), int(rw[40])
    c = (1 << 30) + ((1 << (30 + e)) if e > 0 else 0) + (zo << (31 + e))
    out = [int(v) * m + c for v in np.asarray(cb).reshape(-1)]
    assert all(-(1 << 62) < v < (1 << 62) for v in out)
    return np.array(out, dtype=np.int64)


def _cbias(L, w_i8_ck):
    """w_i8_ck: [cout][K] int (w - 128) over the real taps."""
    za, zw = int(L['in_zp']) - 128, int(L['w_zp']) - 128
    k = w_i8_ck.shape[1]
    return (L['bias'].astype(np.int64) - za * w_i8_ck.sum(axis=1) + k * za * zw).astype(np.int64)


def pack_conv(L, epi, chan_map=None, cout_pad=None):
    """-> (packed i8 [n_mfrag][ksteps][64][16], cbias int32 [cout_pad], kc_per_tap).  chan_map[i] = source channel of output row i
    (or -1: a zero row)."""
    w = L['w'].astype(np.int64) - 128                       # [kh][kw][cin][cout]
    kh, kw, cin, cout = w.shape
    if chan_map is None:
        chan_map = np.arange(cout)
    cout_pad = cout_pad or (len(chan_map) + 15) // 16 * 16
    cmap = np.full(cout_pad, -1, np.int64)
    cmap[:len(chan_map)] = chan_map
    kcpt = (cin + 63) // 64
    wk = np.zeros((cout_pad, kh * kw, kcpt * 64), np.int64)
    src = np.transpose(w, (3, 0, 1, 2)).reshape(cout, kh * kw, cin)
    valid = cmap >= 0
    wk[valid, :, :cin] = src[cmap[valid]]
    cb = np.zeros(cout_pad, np.int64)
    cb[valid] = _cbias(L, src.reshape(cout, -1))[cmap[valid]]
    assert np.abs(cb).max() < 2 ** 31
    n_mfrag = cout_pad // 16
    ksteps = kh * kw * kcpt
    # rows: natural (ROWS) or fragment m of group mg holds row 4g + r = channel 64 mg + 16 g + 4 m + r (Q16)
    if epi == QEPI_Q16:
        assert cout_pad % 64 == 0
        f = np.arange(n_mfrag)[:, None]
        row = np.arange(16)[None, :]
        chan = 64 * (f // 4) + 16 * (row // 4) + 4 * (f % 4) + (row % 4)
    else:
        chan = 16 * np.arange(n_mfrag)[:, None] + np.arange(16)[None, :]
    wf = wk[chan]                                           # [n_mfrag][16 rows][taps][kcpt*64]
    wf = wf.reshape(n_mfrag, 16, kh * kw, kcpt, 4, 16)       # ... [kc][fq][16 bytes]
    packed = np.transpose(wf, (0, 2, 3, 4, 1, 5)).reshape(n_mfrag, ksteps, 64, 16)     # lane = fq * 16 + row
    return packed.astype(np.int8), cb.astype(np.int32), kcpt


def pack_conv0(L):
    """First layer: one k step, k group dy = the nine (dx, channel) bytes of filter row dy; two fragments, row 4g + r of fragment m =
    channel 8g + 4m + r.  -> (filter bytes, lo part or None, cbias): the filter is split w - zw = hi + lo (two MFMAs on one accumulator, no
    row sum of the window) unless zw = 128 already or a weight needs lo = 128."""
    def frags(w333o):                                       # [3][3][3][32] int -> [2][64][16] int8
        src = np.transpose(w333o, (3, 0, 1, 2)).reshape(32, 3, 9)
        wk = np.zeros((32, 4, 16), np.int64)
        wk[:, :3, :9] = src
        f = np.arange(2)[:, None]
        row = np.arange(16)[None, :]
        chan = 8 * (row // 4) + 4 * f + (row % 4)
        return np.transpose(wk[chan], (0, 2, 1, 3)).reshape(2, 64, 16).astype(np.int8)    # wk[chan]: [2][16][4][16]
    w = L['w'].astype(np.int64)
    assert w.shape == (3, 3, 3, 32)
    x = w - int(L['w_zp'])
    hi = np.clip(x, -128, 127)
    if int(L['w_zp']) != 128 and (x - hi).max() <= 127:
        cb = L['bias'].astype(np.int64) + (128 - int(L['in_zp'])) * x.reshape(27, 32).sum(axis=0)
        return frags(hi), frags(x - hi), cb.astype(np.int32)
    return frags(w - 128), None, _cbias(L, np.transpose(w - 128, (3, 0, 1, 2)).reshape(32, 27)).astype(np.int32)


def pack_dw(L):
    """Stand-alone depthwise op: int16 taps w - zw [9][C]; cbias = bias - (za - 128) * sum_t (w_t - zw) (the stored input bytes are a - 128)."""
    w = L['w'].astype(np.int64) - int(L['w_zp'])            # [3][3][C]
    c = w.shape[2]
    cb = L['bias'].astype(np.int64) - (int(L['in_zp']) - 128) * w.reshape(9, c).sum(axis=0)
    return w.reshape(9, c).astype(np.int16), cb.astype(np.int32)


def pack_dw_mfma(L):
    """Fused block's depthwise stage on the matrix pipe (csrc/netsq.hip q_dwpw_k): per plane of 16 channels and lane (g, r) the six
    bytes of the lane's diagonal element -- tap 4 ks + g of channel r for k steps ks = 0..2, split hi = clamp(w - zw, -128, 127),
    lo = (w - zw) - hi -- as uint32 [C / 16][64][2] (byte 3 of the lo word: the plane's mask of k steps whose lo part is not all zero).
    None when a weight needs lo = 128 (w - zw = 255)."""
    w9 = (L['w'].astype(np.int64) - int(L['w_zp'])).reshape(9, -1)       # [tap][C]
    c = w9.shape[1]
    hi = np.clip(w9, -128, 127)
    lo = w9 - hi
    if lo.max() > 127:
        return None
    tab = np.zeros((c // 16, 64, 2, 4), np.uint8)
    for ks in range(3):
        for g in range(4):
            t = 4 * ks + g
            if t < 9:
                tab[:, g * 16:(g + 1) * 16, 0, ks] = (hi[t].reshape(c // 16, 16) & 0xff).astype(np.uint8)
                tab[:, g * 16:(g + 1) * 16, 1, ks] = (lo[t].reshape(c // 16, 16) & 0xff).astype(np.uint8)
    # byte 3 of the lo word (every lane of the plane alike): which k steps of this plane have a lo part at all -- only the extreme weights of a
    # tensor overflow int8, so most planes have none and the kernels skip those MFMAs (bit ks = taps 4 ks .. 4 ks + 3)
    lo9 = np.zeros((12, c), np.int64)
    lo9[:9] = lo
    mask = sum(((lo9[4 * ks:4 * ks + 4].reshape(4, c // 16, 16) != 0).any(axis=(0, 2)).astype(np.uint8) << ks) for ks in range(3))
    tab[:, :, 1, 3] = mask[:, None]
    cb = L['bias'].astype(np.int64) - (int(L['in_zp']) - 128) * w9.sum(axis=0)
    return tab.reshape(c // 16, 64, 8).view(np.uint32).reshape(c // 16, 64, 2), cb.astype(np.int32)


def _model_need(cond, what):
    """A property of the MODEL (as opposed to an internal invariant): raised as tools.tflite_reader.UnsupportedModel, never an assert
    (python -O strips those, and a file that misses one would run with wrong arithmetic)."""
    if not cond:
        from .tools.tflite_reader import UnsupportedModel
        raise UnsupportedModel('uint8 SSD-MobileNet-v1: %s' % what)


def compile_ssd_mobilenet_quant(qm):
    """u8 RGB [n,300,300,3] -> box encodings u8 [n,1917,4] and class logits u8 [n,1917,96] (91 used per row), both in the
    tensors' own quantisation; dd_net_ssd_decode adds the post-process op's first stage (per-anchor arrays)."""
    size = int(qm['input']['size'])
    P = Program(size, size)
    Ls = qm['layers']
    anchors, maps = nets.ssd_anchors(size)
    if qm.get('anchors') is not None:                       # a model file carries its own (the post-process op's third input)
        _model_need(tuple(qm['anchors'].shape) == tuple(anchors.shape), 'anchors %s (this input size gives %s)' % (tuple(qm['anchors'].shape), tuple(anchors.shape)))
        anchors = np.ascontiguousarray(qm['anchors'], dtype=np.float32)
    n_anchors = len(anchors)

    def geom(t, k, stride):
        d = P.T(t)
        ho, pt = same_pad(d['h'], k, stride)
        wo, pl = same_pad(d['w'], k, stride)
        return ho, wo, pt, pl

    def info(kernel, flops, nbytes, wbytes, src_bytes=0):
        P.info[-1] = dict(kernel=kernel, flops=flops, bytes=nbytes, wbytes=wbytes, src_bytes=src_bytes)     # src_bytes: what the op reads (a folded op's share of its launch: profile.launches_of)

    # first layer
    L = Ls['conv0']
    ho, pt = same_pad(size, 3, L['stride'])
    wo, pl = same_pad(size, 3, L['stride'])
    x = P.qtensor(ho, wo, 32, L['out_zp'])
    wp, wl, cb = pack_conv0(L)
    raw = _req_words(L)
    P.add_blob(np.zeros(16, np.uint8))                      # (offset 0 means "absent" in the op words)
    raw.update({38: 0 if wl is not None else 128 - int(L['w_zp']), 39: int(L['in_zp']), 46: P.add_blob(folded_addends(cb, raw))})
    P._op(OP_QCONV0, dst=x, kh=3, kw=3, stride=L['stride'], pad_t=pt, pad_l=pl, cin=3, cout=32, cout_pad=32,
          w_off=P.add_blob(wp), b_off=P.add_blob(cb), aff_off=P.add_blob(wl) if wl is not None else 0, ho=ho, wo=wo, raw=raw)
    info('q_conv0_k', 2 * ho * wo * 27 * 32, 3 * size * size + ho * wo * 32, 27 * 32 + 4 * 32, 3 * size * size)

    def conv(src, name, epi=QEPI_Q16, dst=None, chan_map=None, cout_pad=None, row_bytes=0, base_off=0, cout_store=0):
        L = Ls[name]
        kh = L['w'].shape[0]
        ho, wo, pt, pl = geom(src, kh, L['stride'])
        s = P.T(src)
        _model_need(s['zp'] == L['in_zp'] and L['w'].shape[2] == s['c'], '%s: input zero point %s / channels %s, its producer writes %s / %s' % (name, L['in_zp'], L['w'].shape[2], s['zp'], s['c']))
        wp, cb, kcpt = pack_conv(L, epi, chan_map, cout_pad)
        if dst is None:
            dst = P.qtensor(ho, wo, L['w'].shape[3], L['out_zp'])
        raw = _req_words(L)
        raw.update({38: 128 - int(L['w_zp']), 39: int(L['in_zp']), 42: row_bytes, 43: base_off, 44: cout_store})
        # the per-channel 64-bit requantisation addends (q_pws_k): cbias * M + C of the ReLU form, or + 2^30 (the first rounding) without activation
        raw[46] = P.add_blob(folded_addends(cb, raw) if not raw[41] else np.array([int(v) * int(raw[32]) + (1 << 30) for v in cb], dtype=np.int64))
        P._op(OP_QCONV, src=src, dst=dst, kh=kh, kw=kh, stride=L['stride'], pad_t=pt, pad_l=pl, cin=s['c'], cout=L['w'].shape[3],
              cout_pad=len(cb), kpad=kcpt, epi=epi, w_off=P.add_blob(wp), b_off=P.add_blob(cb), ho=ho, wo=wo, raw=raw)
        cin, cout = L['w'].shape[2:]
        # (label: the kernel that runs the layer at the bench's 384 frames per launch -- csrc/netsq.hip picks q_pws_k from 8 192 pixels up)
        pws = kh == 1 and L['stride'] == 1 and cin in (512, 1024) and len(cb) >= 128
        info('q_pws_k' if pws else 'q_conv_k', 2 * ho * wo * kh * kh * cin * cout, s['h'] * s['w'] * cin + ho * wo * cout, kh * kh * cin * cout + 4 * cout)
        return dst

    def conv_heads(src, cname, bname, cls_args, box_args):
        """The class and box predictors of one feature map as ONE op (they read the same pixels; csrc/netsq.hip q_pws_k stores both, small
        batches run two launches of q_conv_k): fragments of the class layer first, then the box layer's; its parameters in words 20..25, 28."""
        Lc, Lb = Ls[cname], Ls[bname]
        s = P.T(src)
        ho, wo, pt, pl = geom(src, 1, 1)
        _model_need(s['zp'] == Lc['in_zp'] == Lb['in_zp'] and Lc['w'].shape[:3] == Lb['w'].shape[:3] == (1, 1, s['c']), '%s / %s: 1x1 predictors on one feature map with one input zero point' % (cname, bname))
        wc, cbc, kcpt = pack_conv(Lc, QEPI_ROWS, cls_args['chan_map'], cls_args['cout_pad'])
        wb, cbb, _ = pack_conv(Lb, QEPI_ROWS, None, None)
        rc, rb = _req_words(Lc), _req_words(Lb)
        _model_need(rc[41] and rb[41], '%s / %s: predictors carry no activation' % (cname, bname))
        lin = lambda cb, r: np.array([int(v) * int(r[32]) + (1 << 30) for v in cb], dtype=np.int64)
        raw = dict(rc)
        raw.update({38: 128 - int(Lc['w_zp']), 39: int(Lc['in_zp']), 42: cls_args['row_bytes'], 43: cls_args['base_off'], 44: cls_args['cout_store'],
                    46: P.add_blob(np.concatenate([lin(cbc, rc), lin(cbb, rb)])),
                    20: rb[32], 21: rb[33], 22: rb[40], 23: box_args['row_bytes'], 24: box_args['base_off'], 25: box_args['cout_store'], 28: 128 - int(Lb['w_zp'])})
        P._op(OP_QCONV, src=src, dst=cls_args['dst'], dst2=box_args['dst'], kh=1, kw=1, stride=1, pad_t=pt, pad_l=pl, cin=s['c'], cout=Lc['w'].shape[3] + Lb['w'].shape[3],
              cout_pad=len(cbc) + len(cbb), kpad=kcpt, act=len(cbc) // 16, epi=QEPI_ROWS, w_off=P.add_blob(np.concatenate([wc, wb])), b_off=P.add_blob(np.concatenate([cbc, cbb])),
              ho=ho, wo=wo, raw=raw)
        cin = s['c']
        cout = Lc['w'].shape[3] + Lb['w'].shape[3]
        info('q_pws_k', 2 * ho * wo * cin * cout, s['h'] * s['w'] * cin + ho * wo * cout, cin * cout + 4 * cout)

    def dw(src, name):
        L = Ls[name]
        ho, wo, pt, pl = geom(src, 3, L['stride'])
        s = P.T(src)
        _model_need(s['zp'] == L['in_zp'] and L['w'].shape[2] == s['c'], '%s: input zero point %s / channels %s, its producer writes %s / %s' % (name, L['in_zp'], L['w'].shape[2], s['zp'], s['c']))
        w16, cb = pack_dw(L)
        dst = P.qtensor(ho, wo, s['c'], L['out_zp'])
        mf = pack_dw_mfma(L)                                    # the matrix-pipe form's operand table (None: a weight needs w - zw = 255)
        P._op(OP_QDW, src=src, dst=dst, kh=3, kw=3, stride=L['stride'], pad_t=pt, pad_l=pl, cin=s['c'], cout=s['c'], cout_pad=s['c'],
              w_off=P.add_blob(w16), b_off=P.add_blob(cb), ho=ho, wo=wo, raw=_req_words(L),
              p=[P.add_blob(mf[0]), P.add_blob(mf[1])] if mf is not None else [0, 0])
        info('q_dw_k', 2 * ho * wo * 9 * s['c'], s['h'] * s['w'] * s['c'] + ho * wo * s['c'], 9 * s['c'] + 4 * s['c'])
        return dst

    def dwpw(src, dname, pname):
        """MobileNet block as one launch (csrc/netsq.hip q_dwpw_k) where a fused kernel exists for the shape, else two ops."""
        Ld, Lp = Ls[dname], Ls[pname]
        s = P.T(src)
        cin, cout, stride = s['c'], Lp['w'].shape[3], Ld['stride']
        ho, wo, pt, pl = geom(src, 3, stride)
        if not FUSE_BLOCKS or (cin, cout, stride) not in FUSED_SHAPES or wo < 19:
            return conv(dw(src, dname), pname)
        _model_need(s['zp'] == Ld['in_zp'] and Ld['out_zp'] == Lp['in_zp'] and Ld['w'].shape[2] == cin and Lp['w'].shape[2] == cin, '%s / %s: zero points and channel counts along the block do not agree' % (dname, pname))
        packed_dw = pack_dw_mfma(Ld)
        rd, rp = _req_words(Ld), _req_words(Lp)
        if packed_dw is None or rd[33] < 1 or rp[33] < 1:
            return conv(dw(src, dname), pname)                             # w - zw = 255, or a multiplier >= 0.5: the two-op form handles it
        dwa, dcb = packed_dw
        wp, cb, kcpt = pack_conv(Lp, QEPI_Q16)
        zwc = 128 - int(Lp['w_zp'])
        w_lo = 0
        dup = 0
        if DUP32 and cin == 32 and zwc != 0:
            # 32 channels fill half of the 64-byte k slice: the kernel writes the depthwise bytes into both halves, the filter's halves are the
            # hi and lo parts of w - zw -- the split filter in one MFMA, no activation row sums (block 1: 304 -> ~225 vector instructions per tile)
            x = Lp['w'].astype(np.int64) - int(Lp['w_zp'])
            hi = np.clip(x, -128, 127)
            if (x - hi).max() <= 127:
                wp, _, kcpt = pack_conv(dict(Lp, w=np.concatenate([hi + 128, x - hi + 128], axis=2).astype(np.int64)), QEPI_Q16)
                cb = (Lp['bias'].astype(np.int64) + (128 - int(Lp['in_zp'])) * x.reshape(cin, cout).sum(axis=0)).astype(np.int32)
                zwc, dup = 0, 1
        if SPLIT_PW and SPLIT_PW_MIN_CIN <= cin <= SPLIT_PW_MAX_CIN and zwc != 0:
            # few input channels: the filter as (w - zw) = hi + lo, two MFMAs per k slice instead of one plus the row-sum correction
            x = Lp['w'].astype(np.int64) - int(Lp['w_zp'])
            hi = np.clip(x, -128, 127)
            if (x - hi).max() <= 127:
                wp, _, _ = pack_conv(dict(Lp, w=(hi + 128).astype(np.int64)), QEPI_Q16)
                wl, _, _ = pack_conv(dict(Lp, w=(x - hi + 128).astype(np.int64)), QEPI_Q16)
                w_lo = P.add_blob(wl)
                cb = (Lp['bias'].astype(np.int64) + (128 - int(Lp['in_zp'])) * x.reshape(cin, cout).sum(axis=0)).astype(np.int32)
                zwc = 0
        dst = P.qtensor(ho, wo, cout, Lp['out_zp'])
        raw = dict(rp)
        raw.update({38: zwc, 39: int(Lp['in_zp']), 45: P.add_blob(folded_addends(dcb, rd)), 46: P.add_blob(folded_addends(cb, rp)), 47: dup})
        P._op(OP_QDWPW, src=src, dst=dst, kh=1, kw=1, stride=stride, pad_t=pt, pad_l=pl, cin=cin, cout=cout, cout_pad=cout, kpad=kcpt,
              w_off=P.add_blob(wp), b_off=P.add_blob(cb), aff_off=w_lo, ho=ho, wo=wo,
              p=[P.add_blob(dwa), P.add_blob(dcb), rd[32], rd[33], rd[36], rd[37]], bk=rd[40], raw=raw)
        info('q_dwpw_k', 2 * ho * wo * cin * (9 + cout), s['h'] * s['w'] * cin + ho * wo * cout, cin * (9 + cout) + 4 * (cin + cout), s['h'] * s['w'] * cin)
        return dst

    feats = {}
    for i in range(1, 14):
        x = dwpw(x, f'dw{i}', f'pw{i}')
        feats[f'pw{i}'] = x
    for j in range(1, 5):
        x = conv(conv(x, f'extra{j}_1'), f'extra{j}_2')
        feats[f'extra{j}_2'] = x
    n_cls = Ls['cls0']['w'].shape[3] // nets.SSD_ANCHORS_PER_MAP[0]
    cls_row = (n_cls + 15) // 16 * 16                        # 91 -> 96 bytes per anchor
    box_t = P.tensor(n_anchors, 1, 4, cs=4, dtype=DT_U8)
    cls_t = P.tensor(n_anchors, 1, n_cls, cs=cls_row, dtype=DT_U8)
    base = 0
    for k, (fname, a) in enumerate(zip(FEATURE_LAYERS, nets.SSD_ANCHORS_PER_MAP)):
        ft = feats[fname]
        fm = P.T(ft)['h']
        _model_need(fm == maps[k], 'feature map %d is %dx%d (the anchors are laid out for %d)' % (k, fm, fm, maps[k]))
        for other in (f'box{k}', f'cls{k}'):
            _model_need((Ls[other]['out_scale'], Ls[other]['out_zp']) == (Ls[other[:3] + '0']['out_scale'], Ls[other[:3] + '0']['out_zp']),
                        'the six %s tensors are concatenated: they need one (scale, zero point)' % other[:3])
        cmap = np.full(a * cls_row, -1, np.int64)
        for an in range(a):
            cmap[an * cls_row:an * cls_row + n_cls] = an * n_cls + np.arange(n_cls)
        box_args = dict(dst=box_t, row_bytes=4 * a, base_off=4 * base, cout_store=4 * a)
        cls_args = dict(dst=cls_t, chan_map=cmap, cout_pad=(a * cls_row + 15) // 16 * 16, row_bytes=a * cls_row, base_off=cls_row * base, cout_store=a * cls_row)
        if MERGE_HEADS and P.T(ft)['c'] in (512, 1024):
            conv_heads(ft, f'cls{k}', f'box{k}', cls_args, box_args)
        else:
            conv(ft, f'box{k}', epi=QEPI_ROWS, **box_args)
            conv(ft, f'cls{k}', epi=QEPI_ROWS, **cls_args)
        base += fm * fm * a
    _model_need(base == n_anchors, '%d anchors for predictors that emit %d rows' % (n_anchors, base))
    Lb, Lc = Ls['box0'], Ls['cls0']
    lut = quantize.logistic_table(Lc['out_scale'], Lc['out_zp'], qm['logistic']['out_scale'], qm['logistic']['out_zp'])
    P._op(OP_QSSD_DECODE, src=box_t, res=cls_t, w_off=P.add_blob(lut), p=[n_cls, n_anchors],
          rawf={32: float(Lb['out_scale']), 33: float(Lb['out_zp']), 34: float(qm['logistic']['out_scale']), 35: float(qm['logistic']['out_zp'])})
    info('q_ssd_decode_k', 0, n_anchors * (4 + cls_row + 24), 256)
    P.out_tensor = cls_t
    P.meta = dict(kind='ssd_mobilenet_v1_uint8', anchors=anchors, n_classes=n_cls, box_tensor=box_t, cls_tensor=cls_t, cls_row=cls_row,
                  feats=feats, quant=True)
    return P


def unpack_q16(raw, h, w, c):
    """dd_net_read of a bordered tensor ([n][h + 2][c / 16][w + 2][16] bytes) -> u8 NHWC [n][h][w][c] (the interior)."""
    n = raw.size // ((h + 2) * (w + 2) * c)
    a = np.asarray(raw, dtype=np.uint8).reshape(n, h + 2, c // 16, w + 2, 16)[:, 1:-1, :, 1:-1, :] ^ 0x80       # stored as a - 128
    return np.ascontiguousarray(np.transpose(a, (0, 1, 3, 2, 4)).reshape(n, h, w, c))


def borders_q16(raw, h, w, c):
    """The border bytes of a bordered tensor, flattened (tests: they must still hold the zero point after a forward)."""
    n = raw.size // ((h + 2) * (w + 2) * c)
    a = np.asarray(raw, dtype=np.uint8).reshape(n, h + 2, c // 16, w + 2, 16) ^ 0x80
    return np.concatenate([a[:, 0].reshape(-1), a[:, -1].reshape(-1), a[:, :, :, 0].reshape(-1), a[:, :, :, -1].reshape(-1)])
