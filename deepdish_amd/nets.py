"""Model compiler: named f32 weights -> (op program, weight blob) for csrc/nets.hip.

This is the build's analogue of the reference's model files (.tflite / .pb): the host describes the
network as a flat list of ops over NHWC f16 tensors, folds batch-norms into the conv weights, packs
the weights as f16 [Cout][KH*KW*Cin] rows (bias f32) and hands both to dd_net_create().

Architectures (upstream paths):
  * MARS re-ID encoder  -- tools/freeze_model.py:88-157 (+ BGR->RGB :175-177, call contract
    tools/generate_detections.py:151-177); 64x32x3 input per the file name mars-64x32x3.
  * SSD-MobileNet-v1    -- no in-tree description (blob absent): public TF-OD-API definition
    (MobileNet-v1 backbone, 4 extra feature layers, 1x1 box/class predictors, 1917 anchors).
  * YOLOv5s             -- detectors/yolov5/yolov5s.yaml (Focus / Conv / C3 / SPP / Detect, SiLU).

No weight blobs ship with the reference (.MISSING_LARGE_BLOBS); `synthetic_*` below create seeded
random weights of the right shapes so the HIP path and the oracle can be compared on equal terms.
"""
import math
import os
import numpy as np

OP_INPUT, OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_UPSAMPLE, OP_FC, OP_L2NORM, OP_STEM, OP_DWPW = 1, 2, 3, 4, 5, 6, 7, 8, 9
ACT_NONE, ACT_RELU6, ACT_ELU, ACT_SILU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3, 4, 5
EPI_F16, EPI_F32, EPI_SSD_HEAD, EPI_YOLO = 0, 1, 2, 3
DT_F16, DT_F32, DT_U8 = 0, 1, 2
OP_WORDS, TENSOR_WORDS = 48, 8
MAGIC = 0x314E4444
BN_EPS = 1e-3


# MFMA k slot of stem tap k = dy*9 + dx*3 + ch (csrc/nets.hip stem_koff): rem = k % 9 < 8 -> dy*8 + rem, else 24 + dy
STEM_K_SLOT = np.array([(k // 9) * 8 + k % 9 if k % 9 < 8 else 24 + k // 9 for k in range(27)])


def rup(x, m):
    return (x + m - 1) // m * m


def same_pad(size, k, stride):
    """TensorFlow 'SAME': (output size, pad before)."""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2


class Program:
    STEM_POOL_FUSE = os.environ.get('DD_STEM_POOL_FUSE', '1') != '0'   # MARS conv1_1 folded into conv1_2's launch
    RES_UNIT_FUSE = os.environ.get('DD_RES_UNIT_FUSE', '1') != '0'     # MARS conv2_x: both 3x3 layers of a residual unit in one launch
    RES_PAIR_FUSE = os.environ.get('DD_RES_PAIR_FUSE', '1') != '0'     # ... and the two units of the stage in one launch (res_pair_rows_k)
    PAIR64_FUSE = os.environ.get('DD_PAIR64_FUSE', '1') != '0'         # MARS conv3_x: both 3x3 layers of a block (and its projection) in one launch (csrc/mars_pair.hip)
    PROJ_FUSE = os.environ.get('DD_PROJ_FUSE', '1') != '0'             # MARS widening blocks: 3x3 stride-2 layer + 1x1 stride-2 projection in one launch
    SSD_FRONT_FUSE = os.environ.get('DD_SSD_FRONT_FUSE', '1') != '0'   # SSD conv0 + MobileNet block 1 in one launch
    PW_DW_FUSE = os.environ.get('DD_PW_DW_FUSE', '1') != '0'           # MobileNet: pointwise layer + the next block's depthwise layer in one launch

    def __init__(self, in_h, in_w):
        self.in_h, self.in_w = in_h, in_w
        self.tensors, self.bufs, self.ops = [], [], []
        self.keep = set()       # tensors somebody else reads later (feature maps): never folded away
        self.blob = bytearray()
        self.out_tensor = -1
        self.meta = {}
        self.info = []          # per op: kernel symbol + algorithmic flops / bytes per image

    # ---- storage
    def buffer(self, elems, dtype=DT_F16, fill=0):
        """fill: the byte every element of the buffer holds at engine creation (uint8 tensors: their zero point)."""
        self.bufs.append((int(elems), dtype | (int(fill) & 0xff) << 8))
        return len(self.bufs) - 1

    def qtensor(self, h, w, c, zp):
        """uint8 activation tensor in the bordered 16-channel-plane layout of csrc/netsq.hip: [h + 2][c / 16][w + 2][16] bytes
        a - 128, borders = the tensor's zero point (set once, never written)."""
        assert c % 16 == 0 and 0 <= zp <= 255
        buf = self.buffer((h + 2) * (w + 2) * c, DT_U8, fill=zp ^ 0x80)      # stored bytes are a - 128 (MFMA operands as they lie)
        self.tensors.append(dict(buf=buf, h=h, w=w, c=c, cs=c, coff=0, dtype=DT_U8, q16=1, zp=int(zp)))
        return len(self.tensors) - 1

    def tensor(self, h, w, c, cs=None, coff=0, buf=None, dtype=DT_F16):
        cs = rup(c, 8) if cs is None else cs
        if buf is None:
            buf = self.buffer(h * w * cs, dtype)
        self.tensors.append(dict(buf=buf, h=h, w=w, c=c, cs=cs, coff=coff, dtype=dtype))
        return len(self.tensors) - 1

    def view(self, t, coff, c):
        """Channel slice of an existing tensor (concat without copies)."""
        d = self.tensors[t]
        assert coff % 8 == 0 and c % 8 == 0
        return self.tensor(d['h'], d['w'], c, cs=d['cs'], coff=d['coff'] + coff, buf=d['buf'], dtype=d['dtype'])

    def T(self, t):
        return self.tensors[t]

    def add_blob(self, arr):
        while len(self.blob) % 256:
            self.blob.append(0)
        off = len(self.blob)
        self.blob += np.ascontiguousarray(arr).tobytes()
        return off

    def _op(self, kind, src=-1, dst=-1, res=-1, dst2=-1, **kw):
        w = np.zeros(OP_WORDS, dtype=np.int32)
        f = w.view(np.float32)
        w[0:5] = (kind, src, dst, res, dst2)
        names = dict(kh=5, kw=6, stride=7, pad_t=8, pad_l=9, cin=10, cout=11, cout_pad=12, kpad=13, act=14, epi=15,
                     w_off=16, b_off=17, aff_off=18, has_aff=19, ho=26, wo=27, bk=28, pool=29, fuse_next=30, aux_off=31)
        for k, v in kw.items():
            if k == 'p':
                w[20:20 + len(v)] = v
            elif k == 'f':
                f[32:32 + len(v)] = v
            elif k == 'raw':                      # {word index: int32} (uint8 programs, deepdish_amd/netsq.py)
                for i, x in v.items():
                    w[i] = x
            elif k == 'rawf':
                for i, x in v.items():
                    f[i] = x
            else:
                w[names[k]] = v
        self.ops.append(w)
        self.info.append(dict(kernel='?', flops=0, bytes=0, wbytes=0))

    # ---- ops
    def input(self, swap_rb, mean=0.0, scale=1.0, s2d=False, c_pad=0):
        """c_pad: channels of the output tensor (the real ones first, zeros behind them; 0 = 12 of 16 / 3 of 8)."""
        h, w = (self.in_h // 2, self.in_w // 2) if s2d else (self.in_h, self.in_w)
        t = self.tensor(h, w, c_pad if c_pad else 12 if s2d else 3, cs=c_pad if c_pad else 16 if s2d else 8)
        self._op(OP_INPUT, dst=t, kh=int(s2d), kw=int(swap_rb), f=[mean, scale])
        return t

    def stem(self, w_hwio, bias, stride, act, swap_rb, mean=0.0, scale=1.0):
        """First layer straight from the u8 image: (x - mean) * scale -> 3x3 conv over the 3 colour channels
        (TF SAME) -> bias -> act -> NHWC f16 with 32 channels (csrc/nets.hip stem_conv3_k).  The channel swap
        is folded into the weights; they are packed [32][32] f16, tap k = dy*9 + dx*3 + ch at MFMA k slot
        STEM_K_SLOT[k]: k-group dy holds the first eight taps of filter row dy (eight consecutive halves of an
        image row) and group 3 the ninth tap of each row -- a lane gathers its eight operands from one image row."""
        kh, kw, cin, cout = w_hwio.shape
        assert (kh, kw, cin) == (3, 3, 3) and cout <= 32
        ho, pt = same_pad(self.in_h, 3, stride)
        wo, pl = same_pad(self.in_w, 3, stride)
        w = w_hwio[:, :, ::-1, :] if swap_rb else w_hwio
        wp = np.zeros((32, 32), dtype=np.float16)
        wp[:cout, STEM_K_SLOT] = np.transpose(w, (3, 0, 1, 2)).reshape(cout, 27).astype(np.float16)
        bp = np.zeros(32, dtype=np.float32)
        bp[:cout] = bias
        dst = self.tensor(ho, wo, cout)
        self._op(OP_STEM, dst=dst, stride=stride, pad_t=pt, pad_l=pl, cout=cout, cout_pad=32, act=act,
                 w_off=self.add_blob(wp), b_off=self.add_blob(bp), ho=ho, wo=wo, f=[mean, scale])
        self.info[-1] = dict(kernel='stem_conv3_k', flops=2 * ho * wo * 27 * cout,
                             bytes=3 * self.in_h * self.in_w + 2 * ho * wo * cout, wbytes=2 * 27 * cout + 4 * cout,
                             src_bytes=3 * self.in_h * self.in_w)
        return dst

    def conv(self, src, w_hwio, bias, stride=1, pad=None, act=ACT_NONE, dst=None, res=-1, dst2=-1, aff2=None,
             epi=EPI_F16, p=(), f=(), out_hw=None, pool=False):
        """w_hwio f32 [KH,KW,Cin,Cout] (already BN-folded), bias f32 [Cout].  pool=True appends a 3x3 stride-2
        VALID max pool inside the same launch (conv3x3_rw_k; 3x3 32->32 layers only): dst is the pooled tensor."""
        s = self.T(src)
        kh, kw, cin, cout = w_hwio.shape
        assert cin == s['c'], (cin, s['c'])
        cin_pad = rup(cin, 8)
        if pad is None:                       # TF SAME
            ho, pt = same_pad(s['h'], kh, stride)
            wo, pl = same_pad(s['w'], kw, stride)
        else:                                 # explicit symmetric padding (PyTorch style)
            pt = pl = pad
            ho = (s['h'] + 2 * pad - kh) // stride + 1
            wo = (s['w'] + 2 * pad - kw) // stride + 1
        if out_hw is not None:
            assert (ho, wo) == tuple(out_hw), ((ho, wo), out_hw)
        cout_pad = rup(cout, 8)
        rows = 32 if cout_pad <= 32 else rup(cout_pad, 128 if cout_pad >= 128 else 64)
        bk = 32 if kh * kw * cin_pad <= 96 else 64
        kpad = rup(kh * kw * cin_pad, bk)
        wp = np.zeros((rows, kh * kw, cin_pad), dtype=np.float16)
        wp[:cout, :, :cin] = np.transpose(w_hwio, (3, 0, 1, 2)).reshape(cout, kh * kw, cin).astype(np.float16)
        wflat = np.zeros((rows, kpad), dtype=np.float16)
        wflat[:, :kh * kw * cin_pad] = wp.reshape(rows, -1)
        bp = np.zeros(rows, dtype=np.float32)
        bp[:cout] = bias
        if pool:
            assert dst is None and res < 0 and dst2 < 0 and epi == EPI_F16
            dst = self.tensor((ho - 3) // 2 + 1, (wo - 3) // 2 + 1, cout)
        elif dst is None:
            dst = self.tensor(ho, wo, cout, dtype=DT_F32 if epi == EPI_F32 else DT_F16)
        if epi in (EPI_F16, EPI_F32) and not pool:
            d = self.T(dst)
            assert (d['h'], d['w']) == (ho, wo) and d['c'] == cout, (d, ho, wo, cout)
        kw_ = dict(kh=kh, kw=kw, stride=stride, pad_t=pt, pad_l=pl, cin=cin_pad, cout=cout, cout_pad=cout_pad,
                   kpad=kpad, act=act, epi=epi, w_off=self.add_blob(wflat), b_off=self.add_blob(bp), p=list(p), f=list(f),
                   ho=ho, wo=wo, bk=bk, pool=int(pool))
        if dst2 >= 0:
            a = np.zeros((2, cout_pad), dtype=np.float32)
            a[0, :cout], a[1, :cout] = aff2
            kw_['aff_off'] = self.add_blob(a)
            kw_['has_aff'] = 1
        if epi == EPI_SSD_HEAD and (kh, kw, stride) == (1, 1, 1) and cin_pad % 64 == 0 and 4 + p[0] <= 96:
            # second copy of the head weights, one anchor per 96 rows (4 box encodings, C class logits, zero rows): the layout of
            # the kernel that decodes in its epilogue (csrc/nets.hip ssd_head_finish; dd_net_ssd_decode switches it on)
            per, a_n = 4 + p[0], p[5]
            assert cout == a_n * per and kpad == cin_pad
            w96 = np.zeros((a_n * 96, kpad), dtype=np.float16)
            b96 = np.zeros(a_n * 96, dtype=np.float32)
            for an in range(a_n):
                w96[an * 96:an * 96 + per] = wflat[an * per:(an + 1) * per]
                b96[an * 96:an * 96 + per] = bp[an * per:(an + 1) * per]
            kw_['aff_off'] = self.add_blob(w96)
            kw_['aux_off'] = self.add_blob(b96)
        if epi == EPI_YOLO and (kh, kw, stride) == (1, 1, 1) and cin_pad % 64 == 0 and p[0] <= 96 and cout % p[0] == 0:
            # second copy of the Detect weights, one anchor per 96 rows (x, y, w, h, objectness, C classes, zero rows): the layout of the
            # kernel that reduces a row to (box, confidence, class) in its epilogue (csrc/nets.hip yolo_head_finish; dd_net_yolo_decode)
            per, a_n = p[0], cout // p[0]
            assert kpad == cin_pad
            w96 = np.zeros((a_n * 96, kpad), dtype=np.float16)
            b96 = np.zeros(a_n * 96, dtype=np.float32)
            for an in range(a_n):
                w96[an * 96:an * 96 + per] = wflat[an * per:(an + 1) * per]
                b96[an * 96:an * 96 + per] = bp[an * per:(an + 1) * per]
            kw_['aff_off'] = self.add_blob(w96)
            kw_['aux_off'] = self.add_blob(b96)
        self._op(OP_CONV, src=src, dst=dst, res=res, dst2=dst2, **kw_)
        tile = '4,1,1,2' if cout_pad <= 32 else '2,2,2,2'
        rw = (kh, kw, stride, cin_pad, cout_pad, epi, pt, pl) == (3, 3, 1, 32, 32, EPI_F16, 1, 1)
        self.info[-1] = dict(kernel='conv3x3_pool_rows_k' if rw and pool else 'conv3x3_rw_k' if rw else 'conv_glds_k' if bk == 64 else 'conv_mfma_k<%s,%d>' % (tile, bk),
                             flops=2 * ho * wo * kh * kw * cin * cout,
                             bytes=2 * s['h'] * s['w'] * cin + (4 if epi != EPI_F16 else 2) * (self.T(dst)['h'] * self.T(dst)['w'] if pool else ho * wo) * cout
                             + (2 * ho * wo * cout if res >= 0 else 0) + (2 * ho * wo * cout if dst2 >= 0 else 0),
                             wbytes=2 * kh * kw * cin * cout + 4 * cout, src_bytes=2 * s['h'] * s['w'] * cin)
        return dst

    def dwconv(self, src, w_hwc, bias, stride, act, pad=None):
        s = self.T(src)
        c = s['c']
        assert w_hwc.shape == (3, 3, c)
        if pad is None:
            ho, pt = same_pad(s['h'], 3, stride)
            wo, pl = same_pad(s['w'], 3, stride)
        else:
            pt = pl = pad
            ho = (s['h'] + 2 * pad - 3) // stride + 1
            wo = (s['w'] + 2 * pad - 3) // stride + 1
        cp = rup(c, 8)
        wp = np.zeros((9, cp), dtype=np.float16)
        wp[:, :c] = w_hwc.reshape(9, c).astype(np.float16)
        bp = np.zeros(cp, dtype=np.float32)
        bp[:c] = bias
        dst = self.tensor(ho, wo, c)
        self._op(OP_DWCONV, src=src, dst=dst, stride=stride, pad_t=pt, pad_l=pl, cout_pad=cp, act=act,
                 w_off=self.add_blob(wp), b_off=self.add_blob(bp))
        self.info[-1] = dict(kernel='dwconv3_k', flops=2 * ho * wo * 9 * c, bytes=2 * (s['h'] * s['w'] * c + ho * wo * c), wbytes=2 * 9 * c + 4 * c,
                             src_bytes=2 * s['h'] * s['w'] * c)
        return dst

    DWPW_SHAPES = {(32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2)}     # (channels, pointwise cout, depthwise stride)
    DWPW_BIG = os.environ.get('DD_DWPW_BIG') is not None   # blocks with 256..1024 channels: K-looped fused kernel (dwpw_big_k), opt-in

    @classmethod
    def _fusable(cls, c, cout, stride):
        if (c, cout, stride) in cls.DWPW_SHAPES:
            return 'dwpw_k'
        if cls.DWPW_BIG and cls.DWPW_SHAPES and 256 <= c <= 1024 and c % 64 == 0 and cout % 256 == 0 and stride in (1, 2):
            return 'dwpw_big_k'
        return None

    def dwpw(self, src, dw_hwc, dw_bias, stride, dw_act, pw_hwio, pw_bias, pw_act):
        """Depthwise 3x3 (TF SAME) + pointwise 1x1 as one launch (csrc/nets.hip dwpw_k) for the block shapes
        it is instantiated for; any other shape falls back to the two separate ops."""
        s = self.T(src)
        c = s['c']
        cout = pw_hwio.shape[3]
        kernel = self._fusable(c, cout, stride)
        if kernel is None or c % 8 or cout % 8:
            # the previous op is the previous block's pointwise layer and this depthwise layer is its only reader: from 160
            # frames both can run as one launch (csrc/nets.hip conv_ws_dw_k; the executor checks shapes, batch and balance)
            if (self.PW_DW_FUSE and stride == 1 and self.ops and self.ops[-1][0] == OP_CONV and self.ops[-1][2] == src
                    and tuple(self.ops[-1][5:8]) == (1, 1, 1) and src not in self.keep):
                self.ops[-1][30] = 1
            x = self.dwconv(src, dw_hwc, dw_bias, stride, dw_act)
            return self.conv(x, pw_hwio, pw_bias, act=pw_act)
        assert dw_hwc.shape == (3, 3, c) and pw_hwio.shape[:3] == (1, 1, c)
        ho, pt = same_pad(s['h'], 3, stride)
        wo, pl = same_pad(s['w'], 3, stride)
        dwp = dw_hwc.reshape(9, c).astype(np.float16)
        wflat = np.ascontiguousarray(pw_hwio.reshape(c, cout).T.astype(np.float16))          # [cout][cin]
        dst = self.tensor(ho, wo, cout)
        self._op(OP_DWPW, src=src, dst=dst, kh=1, kw=1, stride=stride, pad_t=pt, pad_l=pl, cin=c, cout=cout, cout_pad=cout,
                 kpad=c, act=pw_act, epi=EPI_F16, w_off=self.add_blob(wflat), b_off=self.add_blob(pw_bias.astype(np.float32)),
                 p=[self.add_blob(dwp), self.add_blob(dw_bias.astype(np.float32)), dw_act], ho=ho, wo=wo)
        self.info[-1] = dict(kernel=kernel, flops=2 * ho * wo * c * (9 + cout), bytes=2 * (s['h'] * s['w'] * c + ho * wo * cout),
                             wbytes=2 * c * (9 + cout) + 4 * (c + cout), src_bytes=2 * s['h'] * s['w'] * c)
        return dst

    def maxpool(self, src, k, stride, pad, dst=None):
        s = self.T(src)
        ho = (s['h'] + 2 * pad - k) // stride + 1
        wo = (s['w'] + 2 * pad - k) // stride + 1
        if dst is None:
            dst = self.tensor(ho, wo, s['c'])
        self._op(OP_MAXPOOL, src=src, dst=dst, kh=k, stride=stride, pad_t=pad, cout_pad=rup(s['c'], 8))
        self.info[-1] = dict(kernel='maxpool_k', flops=0, bytes=2 * s['c'] * (s['h'] * s['w'] + ho * wo), wbytes=0)
        return dst

    def pool_cascade(self, src, k, n, dst_first):
        """n stride-1 k x k max pools of each other in one launch (csrc/nets.hip pool_cascade_k): result i goes to the channel slice i * c behind
        dst_first (a view of the concat tensor: YOLOv5's SPP)."""
        s = self.T(src)
        self._op(OP_MAXPOOL, src=src, dst=dst_first, kh=k, kw=n, stride=1, pad_t=k // 2, cout_pad=rup(s['c'], 8))
        self.info[-1] = dict(kernel='pool_cascade_k', flops=0, bytes=2 * s['c'] * s['h'] * s['w'] * (1 + n), wbytes=0)
        return dst_first

    def upsample2(self, src, dst):
        s = self.T(src)
        self._op(OP_UPSAMPLE, src=src, dst=dst, cout_pad=rup(s['c'], 8))
        return dst

    def fc(self, src, w_io, bias, act, aff2=None):
        """Fully connected layer on the NHWC-flattened source, run as a 1x1 conv over a [1,1,K] view
        (M = images, split-K in the launcher): out f32 [Cout] = aff2(act(x.w + b))."""
        s = self.T(src)
        assert s['cs'] == s['c'] and s['coff'] == 0
        k, cout = w_io.shape
        assert k == s['h'] * s['w'] * s['c'] and k % 8 == 0
        flat = self.tensor(1, 1, k, cs=k, buf=s['buf'])
        dst = self.tensor(1, 1, cout, cs=rup(cout, 8), dtype=DT_F32)
        self.conv(flat, w_io.reshape(1, 1, k, cout), bias, act=act, dst=dst, epi=EPI_F32)
        if aff2 is not None:
            op = self.ops[-1]
            a = np.zeros((2, rup(cout, 8)), dtype=np.float32)
            a[0, :cout], a[1, :cout] = aff2
            op[18] = self.add_blob(a)
            op[19] = 1
        return dst

    def l2norm(self, src, eps):
        s = self.T(src)
        dst = self.tensor(1, 1, s['c'], cs=s['cs'], dtype=DT_F32)
        self._op(OP_L2NORM, src=src, dst=dst, f=[eps])
        return dst

    def serialize(self):
        head = np.array([MAGIC, len(self.tensors), len(self.bufs), len(self.ops), self.in_h, self.in_w,
                         self.out_tensor, 0], dtype=np.int32)
        tw = np.array([[t['buf'], t['h'], t['w'], t['c'], t['cs'], t['coff'], t['dtype'], t.get('q16', 0)] for t in self.tensors],
                      dtype=np.int32).reshape(-1)
        bw = np.array(self.bufs, dtype=np.int32).reshape(-1)
        words = np.concatenate([head, tw, bw] + self.ops).astype(np.int32)
        return np.ascontiguousarray(words), bytes(self.blob)


# ------------------------------------------------------------------------------------------- folding
def bn_affine(wd, scope):
    """(scale, shift) of an inference batch-norm; gamma optional (slim default scale=False).  A model that arrives with its batch
    norms already folded (a .tflite file: tools/tflite_reader.py) has `<layer>/biases` instead: scale 1, shift = the bias."""
    if scope + '/scale' in wd:                                # a stand-alone batch norm as a .tflite file carries it: MUL + ADD constants
        return wd[scope + '/scale'].astype(np.float32), wd[scope + '/shift'].astype(np.float32)
    if scope + '/moving_variance' not in wd and scope.endswith('/bn') and scope[:-3] + '/biases' in wd:
        b = wd[scope[:-3] + '/biases'].astype(np.float32)
        return np.ones_like(b), b
    var, mean, beta = wd[scope + '/moving_variance'], wd[scope + '/moving_mean'], wd[scope + '/beta']
    gamma = wd.get(scope + '/gamma', np.ones_like(var))
    s = gamma / np.sqrt(var + BN_EPS)
    return s.astype(np.float32), (beta - mean * s).astype(np.float32)


def fold_conv_bn(wd, scope, bn_scope=None):
    """conv (no bias) followed by BN -> (w * s, shift)."""
    w = wd[scope + '/weights']
    s, t = bn_affine(wd, bn_scope or scope + '/bn')
    return (w * s).astype(np.float32), t


# ------------------------------------------------------------------------------------------- MARS
MARS_BLOCKS = [('conv2_1', 32, False, True), ('conv2_3', 32, False, False), ('conv3_1', 64, True, False),
               ('conv3_3', 64, False, False), ('conv4_1', 128, True, False), ('conv4_3', 128, False, False)]


def synthetic_mars_weights(seed=1234):
    """Seeded weights with the variable names/shapes of tools/freeze_model.py:88-157."""
    rng = np.random.default_rng(seed)
    wd = {}

    def conv(scope, kh, cin, cout, gain=1.0, bias=False, bn=True):
        wd[scope + '/weights'] = (rng.standard_normal((kh, kh, cin, cout)) * gain * math.sqrt(2.0 / (kh * kh * cin))).astype(np.float32)
        if bias:
            wd[scope + '/biases'] = (0.05 * rng.standard_normal(cout)).astype(np.float32)
        if bn:
            bnp(scope + '/bn', cout)

    def bnp(scope, c):
        wd[scope + '/beta'] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        wd[scope + '/moving_mean'] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        wd[scope + '/moving_variance'] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    conv('conv1_1', 3, 3, 32, gain=1.0 / 128)           # raw 0..255 pixels in
    conv('conv1_2', 3, 32, 32)
    cin = 32
    for name, c, inc, first in MARS_BLOCKS:
        if not first:
            bnp(name + '/bn', cin)
        conv(name + '/1', 3, cin, c)
        conv(name + '/2', 3, c, c, gain=0.5, bias=True, bn=False)
        if inc:
            wd[name + '/projection/weights'] = (rng.standard_normal((1, 1, cin, c)) * math.sqrt(1.0 / cin)).astype(np.float32)
        cin = c
    wd['fc1/weights'] = (rng.standard_normal((4096, 128)) * math.sqrt(2.0 / 4096)).astype(np.float32)
    bnp('fc1/bn', 128)
    bnp('ball', 128)
    return wd


def compile_mars(wd, in_h=64, in_w=32):
    """tools/freeze_model.py:88-157 as an op program; input u8 BGR [n,64,32,3] -> f32 [n,128]."""
    P = Program(in_h, in_w)
    w, b = fold_conv_bn(wd, 'conv1_1')
    x = c11 = P.stem(w, b, 1, ACT_ELU, swap_rb=bool(wd.get('__swap_rb__', True)))   # :175-177 BGR -> RGB, :101-105 (a .tflite graph says whether it reverses the channels)
    if Program.STEM_POOL_FUSE:       # conv1_1 is read by conv1_2 only: with >= 160 crops both run as one launch
        P.ops[-1][30] = 1            # (conv3x3_pool_rows_k<STEM>) and the conv1_1 tensor is not written
    w, b = fold_conv_bn(wd, 'conv1_2')
    if in_w == 32:
        x = pool = P.conv(x, w, b, act=ACT_ELU, pool=True)                     # :106-110 + :116 VALID pool, one launch (the fused kernel is 32 wide)
    else:                                                                      # other crop sizes (mars-small128.pb: 128 x 64): pool as its own op
        P.ops[-1][30] = 0
        x = pool = P.maxpool(P.conv(x, w, b, act=ACT_ELU), 3, 2, 0)
    raw, pre = x, x                  # raw = block input (skip path), pre = what conv "1" reads
    for i, (name, c, inc, first) in enumerate(MARS_BLOCKS):
        stride = 2 if inc else 1
        w, b = fold_conv_bn(wd, name + '/1')
        h1 = P.conv(pre, w, b, stride=stride, act=ACT_ELU)                      # :58-62
        if Program.RES_UNIT_FUSE and not inc and c == 32:
            P.ops[-1][30] = 1        # h1 is read by conv "2" only: both layers of the unit run as one launch (res_unit_rows_k)
        if inc:
            if Program.PROJ_FUSE:
                P.ops[-1][30] = 3        # the next op is this block's 1x1 stride-2 projection of `raw`: both may run as one launch (csrc/mars_tail.hip)
            skip = P.conv(raw, wd[name + '/projection/weights'], np.zeros(c, np.float32), stride=2)   # :30-36
        else:
            skip = raw
        nxt = MARS_BLOCKS[i + 1][0] if i + 1 < len(MARS_BLOCKS) else None
        if Program.PAIR64_FUSE and c == 64 and nxt is not None:
            # conv3_x: h1 (and the projection) are read by conv "2" only -- with enough crops the block's layers run as ONE launch
            # (csrc/mars_pair.hip) and neither tensor is written.  Marks the op in front of conv "2"; the widening block's stride-2 layer
            # already carries the projection marker (3), which chains it to the projection for the buffer lifetimes.
            P.ops[-1][30] = 4
        s = P.T(h1)
        out = P.tensor(s['h'], s['w'], c)
        if nxt is not None:          # the next block's BN+ELU pre-activation is a second epilogue output (:17-21)
            out2 = P.tensor(s['h'], s['w'], c)
            P.conv(h1, wd[name + '/2/weights'], wd[name + '/2/biases'], dst=out, res=skip, dst2=out2,
                   aff2=bn_affine(wd, nxt + '/bn'))                             # :68-72 + :37/:39 skip add
            nb = MARS_BLOCKS[i + 1]
            if Program.RES_UNIT_FUSE and Program.RES_PAIR_FUSE and not inc and c == 32 and not nb[2] and nb[1] == 32:
                P.ops[-1][30] = 2    # out / out2 are read by the next residual unit only (its skip and its first layer): with enough
                                     # crops both units run as one launch (res_pair_rows_k) and neither tensor is written
            raw, pre = out, out2
        else:
            P.conv(h1, wd[name + '/2/weights'], wd[name + '/2/biases'], dst=out, res=skip)
            raw = pre = out
    w, b = fold_conv_bn(wd, 'fc1', 'fc1/bn')                                    # :143-147 (w is [4096,128])
    f = P.fc(raw, w, b, ACT_ELU, aff2=bn_affine(wd, 'ball'))                    # :152 "ball" BN
    P.out_tensor = P.l2norm(f, 1e-8)                                           # :153-156
    P.meta = dict(kind='mars', out_dim=128, tensors=dict(conv1_1=c11, pool1=pool))
    return P


# ------------------------------------------------------------------------------------------- SSD-MobileNet-v1
MOBILENET_V1 = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1), (512, 1), (512, 1), (512, 1),
                (512, 1), (1024, 2), (1024, 1)]
SSD_EXTRAS = [(256, 512), (128, 256), (128, 256), (64, 128)]
SSD_ANCHORS_PER_MAP = [3, 6, 6, 6, 6, 6]
SSD_CLASSES = 91


def ssd_anchors(in_size=300):
    """TF-OD-API multiple-grid anchor generator for ssd_mobilenet_v1 (min 0.2, max 0.95, 6 maps,
    aspect ratios 1,2,1/2,3,1/3, reduced boxes in the lowest map) -> f32 [1917,4] (yc, xc, h, w)."""
    maps, s = [], in_size
    s = same_pad(s, 3, 2)[0]
    for c, st in MOBILENET_V1:
        s = same_pad(s, 3, st)[0]
        if (c, st) == (512, 1):
            f0 = s
    maps = [f0, s]
    for _ in SSD_EXTRAS:
        s = same_pad(s, 3, 2)[0]
        maps.append(s)
    scales = [0.2 + (0.95 - 0.2) * i / 5 for i in range(6)] + [1.0]
    out = []
    for k, fm in enumerate(maps):
        if k == 0:
            specs = [(0.1, 1.0), (scales[0], 2.0), (scales[0], 0.5)]
        else:
            specs = [(scales[k], ar) for ar in (1.0, 2.0, 0.5, 3.0, 1.0 / 3)] + [(math.sqrt(scales[k] * scales[k + 1]), 1.0)]
        for y in range(fm):
            for x in range(fm):
                for sc, ar in specs:
                    out.append(((y + 0.5) / fm, (x + 0.5) / fm, sc / math.sqrt(ar), sc * math.sqrt(ar)))
    return np.array(out, dtype=np.float32), maps


def synthetic_ssd_weights(seed=1234):
    rng = np.random.default_rng(seed)
    wd = {}

    def bnp(scope, c):
        wd[scope + '/gamma'] = rng.uniform(0.8, 1.2, c).astype(np.float32)
        wd[scope + '/beta'] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        wd[scope + '/moving_mean'] = (0.1 * rng.standard_normal(c)).astype(np.float32)
        wd[scope + '/moving_variance'] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    def conv(scope, k, cin, cout):
        wd[scope + '/weights'] = (rng.standard_normal((k, k, cin, cout)) * math.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        bnp(scope + '/bn', cout)

    conv('conv0', 3, 3, 32)
    cin = 32
    for i, (c, st) in enumerate(MOBILENET_V1, 1):
        wd[f'dw{i}/weights'] = (rng.standard_normal((3, 3, cin, 1)) * math.sqrt(2.0 / 9)).astype(np.float32)
        bnp(f'dw{i}/bn', cin)
        conv(f'pw{i}', 1, cin, c)
        cin = c
    feats = [512, 1024]
    for j, (c1, c2) in enumerate(SSD_EXTRAS, 1):
        conv(f'extra{j}_1', 1, cin, c1)
        conv(f'extra{j}_2', 3, c1, c2)
        cin = c2
        feats.append(c2)
    for k, (c, a) in enumerate(zip(feats, SSD_ANCHORS_PER_MAP)):
        wd[f'box{k}/weights'] = (rng.standard_normal((1, 1, c, a * 4)) * math.sqrt(1.0 / c)).astype(np.float32)
        wd[f'box{k}/biases'] = (0.1 * rng.standard_normal(a * 4)).astype(np.float32)
        wd[f'cls{k}/weights'] = (rng.standard_normal((1, 1, c, a * SSD_CLASSES)) * math.sqrt(1.0 / c)).astype(np.float32)
        wd[f'cls{k}/biases'] = (rng.standard_normal(a * SSD_CLASSES) - 3.0).astype(np.float32)
    return wd


def compile_ssd_mobilenet(wd, in_size=300, mean=127.5, std=127.5):
    """u8 RGB [n,300,300,3] -> f32 [n,1917,4+91] raw box encodings + class logits.  mean / std: the input normalisation (x - mean) / std, fused into
    the first layer (a model file's NormalizationOptions, tools/tflite_object_detector.py:124-131,222-224 upstream; 127.5 / 127.5 without one)."""
    P = Program(in_size, in_size)
    anchors, maps = ssd_anchors(in_size)
    if 'anchors' in wd:                                     # a model file carries its own (the post-process op's third input)
        assert wd['anchors'].shape == anchors.shape
        anchors = np.ascontiguousarray(wd['anchors'], dtype=np.float32)
    n_anchors = len(anchors)
    ld = 4 + SSD_CLASSES
    w, b = fold_conv_bn(wd, 'conv0')
    x = P.stem(w, b, 2, ACT_RELU6, swap_rb=False, mean=float(mean), scale=1.0 / float(std))
    feats = []
    for i, (c, st) in enumerate(MOBILENET_V1, 1):
        s, t = bn_affine(wd, f'dw{i}/bn')
        w, b = fold_conv_bn(wd, f'pw{i}')
        x = P.dwpw(x, wd[f'dw{i}/weights'][:, :, :, 0] * s, t, st, ACT_RELU6, w, b, ACT_RELU6)
        if i == 1 and Program.SSD_FRONT_FUSE and P.ops[-1][0] == OP_DWPW and P.ops[-2][0] == OP_STEM:
            P.ops[-2][30] = 1        # conv0 is read by block 1 only: one launch (csrc/nets.hip ssd_front_k), conv0's tensor is not written
        if i in (11, 13):
            feats.append(x)
            P.keep.add(x)
    for j in range(1, 5):
        w, b = fold_conv_bn(wd, f'extra{j}_1'); x = P.conv(x, w, b, act=ACT_RELU6)
        w, b = fold_conv_bn(wd, f'extra{j}_2'); x = P.conv(x, w, b, stride=2, act=ACT_RELU6)
        feats.append(x)
    out = P.tensor(n_anchors, 1, ld, cs=ld, dtype=DT_F32)
    base = 0
    for k, (ft, a) in enumerate(zip(feats, SSD_ANCHORS_PER_MAP)):
        fm = P.T(ft)['h']
        assert fm == maps[k]
        w = np.concatenate([wd[f'box{k}/weights'], wd[f'cls{k}/weights']], axis=3)      # one launch per feature map
        b = np.concatenate([wd[f'box{k}/biases'], wd[f'cls{k}/biases']])
        # output channels in the memory order of the head matrix: [anchor][4 box encodings + C class logits]
        order = np.concatenate([np.concatenate([np.arange(an * 4, an * 4 + 4),
                                                4 * a + np.arange(an * SSD_CLASSES, (an + 1) * SSD_CLASSES)]) for an in range(a)])
        w, b = w[..., order], b[order]
        P.conv(ft, w, b, dst=out, epi=EPI_SSD_HEAD, p=[SSD_CLASSES, n_anchors, base, ld, 4 * a, a])
        base += fm * fm * a
    assert base == n_anchors
    P.out_tensor = out
    P.meta = dict(kind='ssd_mobilenet_v1', anchors=anchors, n_classes=SSD_CLASSES)
    return P


# ------------------------------------------------------------------------------------------- YOLOv5s
YOLO_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]   # yolov5s.yaml:6-10
YOLO_NC = 80
FOCUS32 = os.environ.get('DD_YOLO_FOCUS32', '1') != '0'
FOCUS_FUSE = os.environ.get('DD_YOLO_FOCUS_FUSE', '1') != '0'
YOLO_SPP_FUSE = os.environ.get('DD_YOLO_SPP_FUSE', '1') != '0'      # SPP's three pools as one launch (same bits)
YOLO_C3_MERGE = os.environ.get('DD_YOLO_C3_MERGE', '1') != '0'      # C3 blocks: cv1 || cv2 as one layer into the concat tensor, bottlenecks in place (0: the round-3 program)


class _YoloNames:
    """Walks detectors/yolov5/yolov5s.yaml:12-48 (depth 0.33, width 0.50) and yields conv specs."""

    def __init__(self):
        self.convs = []          # (name, k, cin, cout)

    def conv(self, name, k, cin, cout):
        self.convs.append((name, k, cin, cout))

    def c3(self, name, c1, c2, n):
        c_ = c2 // 2
        self.conv(name + '.cv1', 1, c1, c_)
        self.conv(name + '.cv2', 1, c1, c_)
        self.conv(name + '.cv3', 1, 2 * c_, c2)
        for i in range(n):
            self.conv(f'{name}.m{i}.cv1', 1, c_, c_)
            self.conv(f'{name}.m{i}.cv2', 3, c_, c_)


def yolov5s_convs():
    y = _YoloNames()
    y.conv('m0.focus', 3, 12, 32)
    y.conv('m1', 3, 32, 64); y.c3('m2', 64, 64, 1)
    y.conv('m3', 3, 64, 128); y.c3('m4', 128, 128, 3)
    y.conv('m5', 3, 128, 256); y.c3('m6', 256, 256, 3)
    y.conv('m7', 3, 256, 512)
    y.conv('m8.cv1', 1, 512, 256); y.conv('m8.cv2', 1, 1024, 512)          # SPP
    y.c3('m9', 512, 512, 1)
    y.conv('m10', 1, 512, 256); y.c3('m13', 512, 256, 1)
    y.conv('m14', 1, 256, 128); y.c3('m17', 256, 128, 1)
    y.conv('m18', 3, 128, 128); y.c3('m20', 256, 256, 1)
    y.conv('m21', 3, 256, 256); y.c3('m23', 512, 512, 1)
    return y.convs


def synthetic_yolov5s_weights(seed=1234):
    rng = np.random.default_rng(seed)
    wd = {}
    for name, k, cin, cout in yolov5s_convs():
        gain = 1.0 / 255 if name == 'm0.focus' else 1.0         # raw 0..255 pixels in (tools/yolov5.py:100)
        wd[name + '/weights'] = (rng.standard_normal((k, k, cin, cout)) * gain * math.sqrt(2.0 / (k * k * cin))).astype(np.float32)
        wd[name + '/bn/gamma'] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
        wd[name + '/bn/beta'] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        wd[name + '/bn/moving_mean'] = (0.1 * rng.standard_normal(cout)).astype(np.float32)
        wd[name + '/bn/moving_variance'] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    for i, c in enumerate((128, 256, 512)):
        wd[f'detect{i}/weights'] = (rng.standard_normal((1, 1, c, 3 * (5 + YOLO_NC))) * math.sqrt(1.0 / c)).astype(np.float32)
        b = (0.5 * rng.standard_normal(3 * (5 + YOLO_NC))).astype(np.float32).reshape(3, -1)
        b[:, 4] -= 3.0                                              # most cells are background
        b[:, 5:] -= 1.0
        wd[f'detect{i}/biases'] = b.reshape(-1)
    return wd


def compile_yolov5s(wd, in_size=640):
    """u8 RGB [n,640,640,3] -> f32 [n,25200,85] decoded rows (xywh normalised, obj, cls) as the
    reference reads them at tools/yolov5.py:109."""
    P = Program(in_size, in_size)

    def cv(name, src, k=1, s=1, dst=None):
        w, b = fold_conv_bn(wd, name)
        return P.conv(src, w, b, stride=s, pad=k // 2, act=ACT_SILU, dst=dst)

    def c3(name, src, c2, n, shortcut, dst=None):
        c_ = c2 // 2
        s = P.T(src)
        cat = P.tensor(s['h'], s['w'], 2 * c_)
        left, right = P.view(cat, 0, c_), P.view(cat, c_, c_)
        if YOLO_C3_MERGE and n > 0:
            # cv1 || cv2 read the same tensor: ONE 1x1 layer with both filters writes [cv1 | cv2] straight into the concat tensor (one read of the
            # block input instead of two: 629 + 629 -> 839 MB at 160x160 and 128 frames), and the bottlenecks run IN PLACE on its left half --
            # their 3x3 layer reads the separate 1x1 output, the epilogue reads the residual element it is about to overwrite
            w1, b1 = fold_conv_bn(wd, name + '.cv1')
            w2, b2 = fold_conv_bn(wd, name + '.cv2')
            P.conv(src, np.concatenate([w1, w2], axis=3), np.concatenate([b1, b2]), pad=0, act=ACT_SILU, dst=cat)
            for i in range(n):
                h = cv(f'{name}.m{i}.cv1', left)
                w, b = fold_conv_bn(wd, f'{name}.m{i}.cv2')
                P.conv(h, w, b, pad=1, act=ACT_SILU, res=left if shortcut else -1, dst=left)
            return cv(name + '.cv3', cat, dst=dst)
        y = cv(name + '.cv1', src, dst=left if n == 0 else None)
        for i in range(n):
            h = cv(f'{name}.m{i}.cv1', y)
            w, b = fold_conv_bn(wd, f'{name}.m{i}.cv2')
            y = P.conv(h, w, b, pad=1, act=ACT_SILU, res=y if shortcut else -1,
                       dst=left if i == n - 1 else None)
        cv(name + '.cv2', src, dst=right)
        return cv(name + '.cv3', cat, dst=dst)

    # Focus slicing.  The 12 sliced channels are written as a 32-channel tensor (20 zero channels, zero weight columns): the 3x3 conv
    # behind it is then a 32 -> 32 layer for conv3x3_rw_k (whole filter in registers, input patch staged once, one tap = one
    # MFMA k slice) instead of conv_glds_k gathering 32-byte pixels nine times (881 us per 128 frames at 1.3 TB/s, the slowest
    # layer of the network; DD_YOLO_FOCUS32=0 keeps the 16-channel form).
    if FOCUS32:
        x = P.input(swap_rb=False, s2d=True, c_pad=32)
        if FOCUS_FUSE:
            P.ops[-1][30] = 1                                                 # only the conv behind it reads the sliced tensor: the executor folds the slicing into its patch fill
        w, b = fold_conv_bn(wd, 'm0.focus')
        w32 = np.zeros((3, 3, 32, w.shape[3]), dtype=w.dtype)
        w32[:, :, :12] = w
        x = P.conv(x, w32, b, pad=1, act=ACT_SILU)
        P.info[-1]['flops'] = 2 * 320 * 320 * 9 * 12 * 32 * (in_size // 640) ** 2      # algorithmic: the zero channels do not count
    else:
        x = P.input(swap_rb=False, s2d=True)
        x = cv('m0.focus', x, 3)
    x = cv('m1', x, 3, 2); x = c3('m2', x, 64, 1, True)
    x = cv('m3', x, 3, 2)
    s = P.T(x)
    cat17 = P.tensor(s['h'] // 2 * 2, s['w'] // 2 * 2, 256)                 # [up(m14) | m4]
    x4 = c3('m4', x, 128, 3, True, dst=P.view(cat17, 128, 128))
    x = cv('m5', x4, 3, 2)
    s = P.T(x)
    cat13 = P.tensor(s['h'], s['w'], 512)                                   # [up(m10) | m6]
    x6 = c3('m6', x, 256, 3, True, dst=P.view(cat13, 256, 256))
    x = cv('m7', x6, 3, 2)
    s = P.T(x)
    spp = P.tensor(s['h'], s['w'], 1024)                                    # [x | mp5 | mp9 | mp13]
    x = cv('m8.cv1', x, dst=P.view(spp, 0, 256))
    # 5x5, 9x9 and 13x13 stride-1 max pools of x as a cascade of 5x5 pools (a max over a window of windows; taps outside the map
    # are skipped, so the borders agree too): the same values with 75 taps per output instead of 275 -- and all three in one launch
    if YOLO_SPP_FUSE:
        P.pool_cascade(x, 5, 3, P.view(spp, 256, 256))
    else:
        src = x
        for i, k in enumerate((5, 9, 13)):
            src = P.maxpool(src, 5, 1, 2, dst=P.view(spp, 256 * (i + 1), 256))
    x = cv('m8.cv2', spp)
    x = c3('m9', x, 512, 1, False)
    cat23 = P.tensor(s['h'], s['w'], 512)                                   # [m21 | m10]
    x10 = cv('m10', x, dst=P.view(cat23, 256, 256))
    P.upsample2(x10, P.view(cat13, 0, 256))
    x = c3('m13', cat13, 256, 1, False)
    s = P.T(x)
    cat20 = P.tensor(s['h'], s['w'], 256)                                   # [m18 | m14]
    x14 = cv('m14', x, dst=P.view(cat20, 128, 128))
    P.upsample2(x14, P.view(cat17, 0, 128))
    p3 = c3('m17', cat17, 128, 1, False)
    cv('m18', p3, 3, 2, dst=P.view(cat20, 0, 128))
    p4 = c3('m20', cat20, 256, 1, False)
    cv('m21', p4, 3, 2, dst=P.view(cat23, 0, 256))
    p5 = c3('m23', cat23, 512, 1, False)
    no = 5 + YOLO_NC
    rows = sum(3 * P.T(t)['h'] * P.T(t)['w'] for t in (p3, p4, p5))
    out = P.tensor(rows, 1, no, cs=no, dtype=DT_F32)
    base = 0
    for i, t in enumerate((p3, p4, p5)):
        d = P.T(t)
        stride = in_size // d['h']
        P.conv(t, wd[f'detect{i}/weights'], wd[f'detect{i}/biases'], dst=out, epi=EPI_YOLO,
               p=[no, rows, base, 0, in_size, 0], f=[float(v) for v in YOLO_ANCHORS[i]] + [float(stride), float(in_size)])
        base += 3 * d['h'] * d['w']
    P.out_tensor = out
    P.meta = dict(kind='yolov5s', rows=rows, n_classes=YOLO_NC)
    return P
