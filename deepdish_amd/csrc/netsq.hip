// uint8 programs: the reference detector's own arithmetic.
//
// The reference's `ssdmobilenetv1.tflite` is a uint8-quantised model (tools/ssd_mobilenet.py:102 feeds uint8 and
// `interpreter.invoke()` at :103 runs TFLite's integer kernels).  This file runs such a model -- the `QModel` of
// deepdish_amd/quantize.py, or one read from a .tflite by deepdish_amd/tools/tflite_reader.py -- bit for bit as TFLite's
// reference kernels define it (kernels/internal/reference/conv.h, depthwiseconv_uint8.h, common.h):
//     acc = sum (a - za)(w - zw) + bias;   q = clamp(MultiplyByQuantizedMultiplier(acc, M, shift) + zo, lo, hi)
// with the contractions on v_mfma_i32_16x16x64_i8 and everything else in integer vector instructions.
//
// Layout of an activation tensor (H, W, C), C % 16 == 0, in HBM ("Q16"):
//     u8 [n][H + 2][C / 16][W + 2][16]
// i.e. planes of 16 channels, every image row and plane carrying one border pixel on each side, one border row above and
// below.  Border bytes hold the tensor's zero point from engine creation on (nobody writes them), so a 3x3 window never
// needs a bounds test and a padded tap contributes (za - za) = 0 as TFLite's skipped tap does.  A 16-pixel MFMA operand
// fragment is 4 planes x 16 pixels x 16 bytes: every lane loads 16 contiguous bytes, 16 lanes 256 contiguous bytes.
//
// Signed operands: the matrix instruction multiplies i8.  a' = a - 128 and w' = w - 128 (one XOR per 4 bytes / done on the
// host), za' = za - 128, zw' = zw - 128, and
//     sum (a - za)(w - zw) = sum a'w'  -  zw' * sum_k a'  -  za' * sum_k w'[c]  +  K za' zw'
// The last two terms are per output channel (host: `cbias`); the second needs the row sum of the pixel's operand bytes
// (v_dot4_i32_i8 on the fragments the MFMAs read) unless zw == 128.
//
// Requantisation: for multipliers < 1 (shift <= 0, e = -shift) the two roundings of MultiplyByQuantizedMultiplier,
//     y = (x M + 2^30) >> 31 (SaturatingRoundingDoublingHighMul; the truncating division and its sign-dependent nudge are
//     this floor), z = RoundingDivideByPOT(y, e),
// nest into ONE 64-bit multiply-add and shift whenever y >= 0:  z = (x M + 2^30 + 2^(30+e)) >> (31 + e).  For y < 0 both
// forms give z <= 0, which a ReLU-type clamp (lo >= zo) maps to the same byte; layers without activation (the SSD heads)
// take the literal two-step form.  tests/test_quant_host.py checks both against the literal gemmlowp statements.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "ssd_dev.h"
#include "net_priv.h"

namespace {

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

enum { OP_QCONV0 = 16, OP_QCONV = 17, OP_QDW = 18, OP_QDWPW = 19, OP_QSSD_DECODE = 20 };
enum { QEPI_Q16 = 0, QEPI_ROWS = 1 };
constexpr int OP_WORDS = 48;

struct QReq {            // requantisation of one layer (per-tensor parameters)
    int M;               // quantized multiplier, [2^30, 2^31)
    int e;               // right shift (>= 0)
    long long C;         // 2^30 + (e ? 2^(30+e) : 0) + (zo << (31 + e))   (ReLU-type layers)
    int zo, lo, hi;
    int linear;          // no activation: literal two-step rounding
};

__device__ __forceinline__ int q_requant(int x, const QReq &R) {
    if (R.linear) {
        const long long t = (long long)x * R.M + (1ll << 30);
        int y = (int)(t >> 31);
        if (R.e > 0) y = (y + (1 << (R.e - 1)) + (y >> 31)) >> R.e;          // RoundingDivideByPOT: half away from zero
        y += R.zo;
        return min(max(y, R.lo), R.hi);
    }
    const long long t = (long long)x * R.M + R.C;
    const int sh = 31 + R.e;
    const int z = sh >= 32 ? ((int)(t >> 32)) >> (sh - 32) : (int)(t >> 31);
    return min(max(z, R.lo), R.hi);
}

__device__ __forceinline__ int sdot4(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sdot4(a, b, c, false);
#else
    return c;
#endif
}

// ------------------------------------------------------------------------------------------------ generic conv
struct QConvP {
    const uint8_t *in; int H, W, c16_in;          // Q16 source
    int kh, kw, stride, off_y, off_x;             // source row of tap dy for output row y (bordered coordinates): y * stride + dy + off_y
    int ho, wo, m;                                // m = images * ho * wo
    const i4v *w;                                 // packed A fragments [m-frag][k-step][lane]
    const int *cbias;                             // [cout_pad], natural channel order
    int kc_per_tap, n_mfrag, mq;                  // 64-byte k slices per tap; 16-row fragments in all; fragments per wave item
    uint8_t *out; int epi;
    int Ho, Wo, c16_out;                          // QEPI_Q16: destination geometry (Ho == ho, Wo == wo)
    long long img_bytes_out; int row_bytes, base_off, cout_store;   // QEPI_ROWS: plain [pixel][row_bytes] rows at base_off of each image
    int zwc;                                      // 128 - zw
    QReq R;
};

// One wave item = MQ 16-channel fragments x two 16-pixel fragments; operands straight from L2 / HBM in fragment shape (Q16 is
// that shape), no LDS.  Small layers only (SSD extras and heads, 10x10 MobileNet blocks): the big ones run q_dwpw_k.
template <int MQ, bool ROWSUM>
__global__ __launch_bounds__(256) void q_conv_k(const QConvP P, const int n_items, const int n_mgroups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int item = blockIdx.x * 4 + wave;
    if (item >= n_items) return;                                   // whole waves leave; no barrier below
    const int mg = item % n_mgroups, pf = item / n_mgroups;
    const size_t PP = (size_t)(P.W + 2) * 16, RP = PP * P.c16_in;  // plane / row pitch of the source
    const uint8_t *base[2];
    int qn[2], qy[2], qx[2]; bool live[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int q = pf * 32 + j * 16 + fr;
        live[j] = q < P.m;
        q = min(q, P.m - 1);
        const int hw = P.ho * P.wo;
        qn[j] = q / hw; const int r = q - qn[j] * hw;
        qy[j] = r / P.wo; qx[j] = r - qy[j] * P.wo;
        base[j] = P.in + ((size_t)qn[j] * (P.H + 2) + qy[j] * P.stride + P.off_y) * RP + (size_t)(qx[j] * P.stride + P.off_x) * 16;
    }
    i4v acc[MQ][2];
#pragma unroll
    for (int m = 0; m < MQ; ++m) { acc[m][0] = i4v{0, 0, 0, 0}; acc[m][1] = acc[m][0]; }
    int rs[2] = {0, 0};
    const int ksteps = P.kh * P.kw * P.kc_per_tap;
    const i4v *wp = P.w + ((size_t)mg * MQ * ksteps) * 64 + lane;
    int ks = 0;
    for (int dy = 0; dy < P.kh; ++dy)
        for (int dx = 0; dx < P.kw; ++dx)
            for (int kc = 0; kc < P.kc_per_tap; ++kc, ++ks) {
                const int plane = 4 * kc + fq;
                const bool kv = plane < P.c16_in;
                const size_t off = (size_t)dy * RP + (size_t)dx * 16 + (size_t)min(plane, P.c16_in - 1) * PP;
                i4v b[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    b[j] = *reinterpret_cast<const i4v *>(base[j] + off);
                    b[j] ^= (int)0x80808080;
                    if (!kv) b[j] = i4v{0, 0, 0, 0};
                    if (ROWSUM) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) rs[j] = sdot4(b[j][d], 0x01010101, rs[j]);
                    }
                }
#pragma unroll
                for (int m = 0; m < MQ; ++m) {
                    const i4v a = (mg * MQ + m) < P.n_mfrag ? wp[((size_t)m * ksteps + ks) * 64] : i4v{0, 0, 0, 0};
                    acc[m][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[0], acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b[1], acc[m][1], 0, 0, 0);
                }
            }
    if (ROWSUM) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rs[j] += __shfl_xor(rs[j], 16, 64);
            rs[j] += __shfl_xor(rs[j], 32, 64);
            rs[j] *= P.zwc;
        }
    }
    if (P.epi == QEPI_Q16) {
        // the host packed fragment m's row 4g + r with channel 64 mg + 16 g + 4 m + r: this lane holds the 16 consecutive channels of plane 4 mg + fq
        if constexpr (MQ == 4) {
            const i4v *cb = reinterpret_cast<const i4v *>(P.cbias + 64 * mg + 16 * fq);
            const i4v c0 = cb[0], c1 = cb[1], c2 = cb[2], c3 = cb[3];
            const size_t PPo = (size_t)(P.Wo + 2) * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u4v o;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const i4v c = m == 0 ? c0 : m == 1 ? c1 : m == 2 ? c2 : c3;
                    unsigned wv = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) wv |= (unsigned)q_requant(acc[m][j][r] + rs[j] + c[r], P.R) << (8 * r);
                    o[m] = wv;
                }
                if (live[j]) {
                    uint8_t *dst = P.out + (((size_t)qn[j] * (P.Ho + 2) + qy[j] + 1) * P.c16_out + 4 * mg + fq) * PPo + (size_t)(qx[j] + 1) * 16;
                    *reinterpret_cast<u4v *>(dst) = o;
                }
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            const int ch = 16 * (mg * MQ + m) + 4 * fq;
            if (ch >= P.cout_store) continue;
            const i4v c = *reinterpret_cast<const i4v *>(P.cbias + ch);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned wv = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) wv |= (unsigned)q_requant(acc[m][j][r] + rs[j] + c[r], P.R) << (8 * r);
                if (live[j]) {
                    uint8_t *dst = P.out + (size_t)qn[j] * P.img_bytes_out + P.base_off + (size_t)(qy[j] * P.wo + qx[j]) * P.row_bytes + ch;
                    *reinterpret_cast<unsigned *>(dst) = wv;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ first layer
struct QConv0P {
    const uint8_t *src; int H, W; long long total_bytes;      // u8 [n][H][W][3]
    int stride, pad_t, pad_l, ho, wo, m;
    const i4v *w;                                             // [2 fragments][lane]: k group dy holds the 9 bytes (dx, channel) of filter row dy
    const int *cbias;                                         // [32]
    uint8_t *out; int in_zp, zwc;
    QReq R;
};

// 3x3 stride-2 conv over the 3 colour channels, 32 output channels: one MFMA k slice (3 filter rows x 9 bytes of an image row,
// 37 zero slots) per 16 pixels, two fragments of 16 channels.  Taps outside the image read the input zero point.
__global__ __launch_bounds__(256) void q_conv0_k(const QConv0P P, const int n_frags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int f = blockIdx.x * 4 + wave;
    if (f >= n_frags) return;
    int q = f * 16 + fr;
    const bool live = q < P.m;
    q = min(q, P.m - 1);
    const int hw = P.ho * P.wo, n = q / hw, r0 = q - n * hw, y = r0 / P.wo, x = r0 - y * P.wo;
    const unsigned zp4 = (unsigned)P.in_zp * 0x01010101u;
    unsigned d[3] = {zp4, zp4, zp4};
    const int row = y * P.stride + fq - P.pad_t, col = x * P.stride - P.pad_l;       // first of the three source pixels of this filter row
    if (fq < 3 && row >= 0 && row < P.H) {
        const long long a = ((long long)n * P.H + row) * P.W * 3 + (long long)col * 3;     // may be < 0 by up to 3 (col = -1): those bytes are replaced below
        const long long a4 = a & ~3ll;
        const int o = (int)(a - a4);
        unsigned w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const long long ai = min(max(a4 + 4 * i, 0ll), P.total_bytes - 4);
            w[i] = *reinterpret_cast<const unsigned *>(P.src + ai);
        }
        d[0] = __builtin_amdgcn_alignbyte(w[1], w[0], o);
        d[1] = __builtin_amdgcn_alignbyte(w[2], w[1], o);
        d[2] = __builtin_amdgcn_alignbyte(0u, w[2], o);
        // columns outside the image: pixel i of the three covers bytes 3i .. 3i + 2
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool in = col + i >= 0 && col + i < P.W;
            if (!in) {
#pragma unroll
                for (int b = 3 * i; b < 3 * i + 3; ++b) {
                    const unsigned msk = 0xffu << (8 * (b & 3));
                    d[b >> 2] = (d[b >> 2] & ~msk) | (zp4 & msk);
                }
            }
        }
    }
    i4v b;
    b[0] = (int)(d[0] ^ 0x80808080u); b[1] = (int)(d[1] ^ 0x80808080u); b[2] = (int)((d[2] ^ 0x80u) & 0xffu); b[3] = 0;
    if (fq == 3) b = i4v{0, 0, 0, 0};
    int rs = sdot4(b[0], 0x01010101, sdot4(b[1], 0x01010101, sdot4(b[2], 0x01010101, 0)));
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    rs *= P.zwc;
    i4v acc0 = {0, 0, 0, 0}, acc1 = acc0;
    acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(P.w[lane], b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(P.w[64 + lane], b, acc1, 0, 0, 0);
    // fragment m's row 4g + r was packed with channel 8g + 4m + r: this lane holds channels 8 fq .. 8 fq + 7
    const i4v c0 = *reinterpret_cast<const i4v *>(P.cbias + 8 * fq), c1 = *reinterpret_cast<const i4v *>(P.cbias + 8 * fq + 4);
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lo |= (unsigned)q_requant(acc0[r] + rs + c0[r], P.R) << (8 * r);
        hi |= (unsigned)q_requant(acc1[r] + rs + c1[r], P.R) << (8 * r);
    }
    if (live) {
        const size_t PPo = (size_t)(P.wo + 2) * 16;
        uint8_t *dst = P.out + (((size_t)n * (P.ho + 2) + y + 1) * 2 + (fq >> 1)) * PPo + (size_t)(x + 1) * 16 + (fq & 1) * 8;
        *reinterpret_cast<uint2 *>(dst) = make_uint2(lo, hi);
    }
}

// ------------------------------------------------------------------------------------------------ depthwise 3x3 (stand-alone)
struct QDwP {
    const uint8_t *in; int H, W, c16;
    int stride, off_y, off_x, ho, wo;
    const short *w;                 // [9][C]: w - zw
    const int *cbias;               // [C]: bias - za * sum_t (w_t - zw)
    uint8_t *out;
    long long total;                // images * ho * c16 * wo items
    QReq R;
};

// One lane = one pixel x one plane of 16 channels: nine 16-byte taps (the border makes every tap a plain load), 144 integer
// multiply-adds.  The stand-alone form runs where q_dwpw_k does not (10x10 maps).
__global__ __launch_bounds__(256) void q_dw_k(const QDwP P) {
    const long long it = (long long)blockIdx.x * 256 + threadIdx.x;
    if (it >= P.total) return;
    const int x = (int)(it % P.wo);
    long long t = it / P.wo;
    const int pl = (int)(t % P.c16); t /= P.c16;
    const int y = (int)(t % P.ho), n = (int)(t / P.ho);
    const size_t PP = (size_t)(P.W + 2) * 16, RP = PP * P.c16;
    const uint8_t *src = P.in + ((size_t)n * (P.H + 2) + y * P.stride + P.off_y) * RP + (size_t)pl * PP + (size_t)(x * P.stride + P.off_x) * 16;
    int acc[16];
    const i4v *cb = reinterpret_cast<const i4v *>(P.cbias + pl * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const i4v c = cb[i]; acc[4 * i] = c[0]; acc[4 * i + 1] = c[1]; acc[4 * i + 2] = c[2]; acc[4 * i + 3] = c[3]; }
    const int C = P.c16 * 16;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const u4v a = *reinterpret_cast<const u4v *>(src + (size_t)dy * RP + (size_t)dx * 16);
            const short *wt = P.w + (size_t)(dy * 3 + dx) * C + pl * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += (int)((a[i >> 2] >> (8 * (i & 3))) & 0xffu) * (int)wt[i];
        }
    u4v o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned wv = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) wv |= (unsigned)q_requant(acc[4 * d + r], P.R) << (8 * r);
        o[d] = wv;
    }
    uint8_t *dst = P.out + (((size_t)n * (P.ho + 2) + y + 1) * P.c16 + pl) * ((size_t)(P.wo + 2) * 16) + (size_t)(x + 1) * 16;
    *reinterpret_cast<u4v *>(dst) = o;
}

// ------------------------------------------------------------------------------------------------ SSD decode
// First stage of TFLite_Detection_PostProcess on the quantised head tensors (kernels/detection_postprocess.cc:
// DequantizeBoxEncodings, DequantizeClassPredictions behind the graph's uint8 LOGISTIC): sixteen lanes per anchor sweep its
// class bytes through the logistic table, a butterfly picks the best class (background skipped, lowest class on ties),
// one lane decodes the box with csrc/ssd_dev.h's statements.
struct QDecP {
    const uint8_t *box, *cls;              // [n][A][4], [n][A][cls_stride] (class bytes first)
    int n_anchors, n_classes, cls_stride;
    const uint8_t *lut;                    // 256 bytes
    float box_scale, box_zp, sc_scale, sc_zp, thr;
    const float *anchors;
    float *boxes, *score, *keys; int *cls_out;
};

__global__ __launch_bounds__(256) void q_ssd_decode_k(const QDecP P) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = P.lut[threadIdx.x];
    __syncthreads();
    const int a = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    const size_t z = blockIdx.y;
    const bool live = a < P.n_anchors;
    const uint8_t *c = P.cls + (z * P.n_anchors + (live ? a : 0)) * P.cls_stride;
    int best = -1, bi = 0x7fffffff;
    for (int k = 1 + sub; k < P.n_classes; k += 16) {          // class 0 = background
        const int v = lut[c[k]];
        if (v > best) { best = v; bi = k - 1; }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const int ob = __shfl_xor(best, o, 64), oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (live && sub == 0) {
#pragma clang fp contract(off)
        const uint8_t *b = P.box + (z * P.n_anchors + a) * 4;
        float r[4], an[4], bx[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { r[q] = P.box_scale * ((float)b[q] - P.box_zp); an[q] = P.anchors[a * 4 + q]; }
        (void)ssddev::decode_anchor(r, an, 0.f, bx);
        const float sc = P.sc_scale * ((float)best - P.sc_zp);
        const size_t o = z * P.n_anchors + a;
#pragma unroll
        for (int q = 0; q < 4; ++q) P.boxes[o * 4 + q] = bx[q];
        P.score[o] = sc;
        P.cls_out[o] = bi;
        P.keys[o] = sc >= P.thr ? sc : -1.f;
    }
}

QReq make_req(const int32_t *o) {
    QReq R;
    R.M = o[32]; R.e = o[33];
    R.zo = o[40]; R.lo = o[36]; R.hi = o[37]; R.linear = o[41];
    R.C = (1ll << 30) + (R.e > 0 ? (1ll << (30 + R.e)) : 0) + ((long long)R.zo << (31 + R.e));
    return R;
}

}  // namespace

int netq_prepare(dd_net *) { return DD_OK; }

int netq_run_op(dd_net *net, int i, const int32_t *o, const uint8_t *input, int nimg, hipStream_t s, int *handled) {
    const int kind = o[0], src = o[1], dst = o[2];
    *handled = kind >= OP_QCONV0 && kind <= OP_QSSD_DECODE;
    if (!*handled) return DD_OK;
    auto base = [&](int t) -> uint8_t * { return static_cast<uint8_t *>(net->bufs[net->tensors[t].buf]); };
    const TensorDesc *ts = src >= 0 ? &net->tensors[src] : nullptr;
    const TensorDesc *td = dst >= 0 ? &net->tensors[dst] : nullptr;
    char *W = net->d_weights;
    switch (kind) {
        case OP_QCONV0: {
            QConv0P P;
            P.src = input; P.H = net->in_h; P.W = net->in_w; P.total_bytes = (long long)nimg * P.H * P.W * 3;
            P.stride = o[7]; P.pad_t = o[8]; P.pad_l = o[9]; P.ho = td->h; P.wo = td->w; P.m = nimg * P.ho * P.wo;
            P.w = reinterpret_cast<const i4v *>(W + (size_t)(uint32_t)o[16]);
            P.cbias = reinterpret_cast<const int *>(W + (size_t)(uint32_t)o[17]);
            P.out = base(dst); P.in_zp = o[39]; P.zwc = o[38]; P.R = make_req(o);
            DD_REQUIRE(td->pad == 1 && td->cs == 32 && o[11] == 32 && (reinterpret_cast<uintptr_t>(input) & 3) == 0 && (P.total_bytes & 3) == 0 && !P.R.linear,
                       DD_E_ARG, "dd_net_forward: uint8 first layer: 32 channels into a bordered tensor from a 4-byte aligned batch");
            const int n_frags = dd_ceil_div(P.m, 16);
            hipLaunchKernelGGL(q_conv0_k, dim3(dd_ceil_div(n_frags, 4)), dim3(256), 0, s, P, n_frags);
            DD_LAUNCH_CHECK();
            return DD_OK;
        }
        case OP_QCONV: {
            QConvP P;
            memset(&P, 0, sizeof(P));
            DD_REQUIRE(ts->pad == 1 && ts->cs % 16 == 0, DD_E_ARG, "dd_net_forward: uint8 conv %d reads a tensor that is not in the bordered layout", i);
            P.in = base(src); P.H = ts->h; P.W = ts->w; P.c16_in = ts->cs / 16;
            P.kh = o[5]; P.kw = o[6]; P.stride = o[7]; P.off_y = 1 - o[8]; P.off_x = 1 - o[9];
            P.ho = o[26]; P.wo = o[27]; P.m = nimg * P.ho * P.wo;
            P.w = reinterpret_cast<const i4v *>(W + (size_t)(uint32_t)o[16]);
            P.cbias = reinterpret_cast<const int *>(W + (size_t)(uint32_t)o[17]);
            P.kc_per_tap = o[13]; P.n_mfrag = o[12] / 16; P.epi = o[15];
            P.out = base(dst); P.zwc = o[38]; P.R = make_req(o);
            DD_REQUIRE(P.off_y >= 0 && P.off_x >= 0 && (P.ho - 1) * P.stride + P.kh - 1 + P.off_y <= P.H + 1 && (P.wo - 1) * P.stride + P.kw - 1 + P.off_x <= P.W + 1,
                       DD_E_ARG, "dd_net_forward: uint8 conv %d reaches outside the one-pixel border", i);
            if (P.epi == QEPI_Q16) {
                DD_REQUIRE(td->pad == 1 && td->h == P.ho && td->w == P.wo && o[12] % 64 == 0 && td->cs == o[12] && !P.R.linear, DD_E_ARG,
                           "dd_net_forward: uint8 conv %d: bordered output needs a multiple of 64 channels and an activation", i);
                P.Ho = td->h; P.Wo = td->w; P.c16_out = td->cs / 16; P.mq = 4;
            } else {
                P.img_bytes_out = (long long)td->h * td->w * td->cs; P.row_bytes = o[42]; P.base_off = o[43]; P.cout_store = o[44];
                DD_REQUIRE(!td->pad && P.row_bytes % 4 == 0 && P.base_off % 4 == 0 && P.cout_store % 4 == 0 && P.cout_store <= P.row_bytes &&
                           (long long)P.base_off + (long long)P.ho * P.wo * P.row_bytes <= P.img_bytes_out, DD_E_ARG, "dd_net_forward: uint8 conv %d: row output geometry", i);
                P.mq = std::min(4, P.n_mfrag);
            }
            const int n_mgroups = dd_ceil_div(P.n_mfrag, P.mq);
            const long long n_items = (long long)n_mgroups * dd_ceil_div(P.m, 32);
            DD_REQUIRE(n_items < (1ll << 31), DD_E_CAPACITY, "dd_net_forward: uint8 conv %d: %lld wave items", i, n_items);
            const dim3 grid((unsigned)((n_items + 3) / 4));
            const bool rsum = P.zwc != 0;
#define DD_QC(MQ_) do { if (rsum) hipLaunchKernelGGL((q_conv_k<MQ_, true>), grid, dim3(256), 0, s, P, (int)n_items, n_mgroups); \
                        else hipLaunchKernelGGL((q_conv_k<MQ_, false>), grid, dim3(256), 0, s, P, (int)n_items, n_mgroups); } while (0)
            if (P.mq == 4) DD_QC(4); else if (P.mq == 3) DD_QC(3); else if (P.mq == 2) DD_QC(2); else DD_QC(1);
#undef DD_QC
            DD_LAUNCH_CHECK();
            return DD_OK;
        }
        case OP_QDW: {
            QDwP P;
            DD_REQUIRE(ts->pad == 1 && td->pad == 1 && ts->cs == td->cs && ts->cs % 16 == 0, DD_E_ARG, "dd_net_forward: uint8 depthwise %d: tensor layouts", i);
            P.in = base(src); P.H = ts->h; P.W = ts->w; P.c16 = ts->cs / 16;
            P.stride = o[7]; P.off_y = 1 - o[8]; P.off_x = 1 - o[9]; P.ho = td->h; P.wo = td->w;
            P.w = reinterpret_cast<const short *>(W + (size_t)(uint32_t)o[16]);
            P.cbias = reinterpret_cast<const int *>(W + (size_t)(uint32_t)o[17]);
            P.out = base(dst); P.R = make_req(o);
            P.total = (long long)nimg * P.ho * P.c16 * P.wo;
            DD_REQUIRE(P.off_y >= 0 && P.off_x >= 0 && (P.ho - 1) * P.stride + 2 + P.off_y <= P.H + 1 && (P.wo - 1) * P.stride + 2 + P.off_x <= P.W + 1 && !P.R.linear,
                       DD_E_ARG, "dd_net_forward: uint8 depthwise %d reaches outside the one-pixel border", i);
            DD_REQUIRE(P.total < (1ll << 31) * 256, DD_E_CAPACITY, "dd_net_forward: uint8 depthwise %d: too many items", i);
            hipLaunchKernelGGL(q_dw_k, dim3((unsigned)((P.total + 255) / 256)), dim3(256), 0, s, P);
            DD_LAUNCH_CHECK();
            return DD_OK;
        }
        case OP_QSSD_DECODE: {
            if (!net->ssd_dec) return DD_OK;                     // nobody asked for the per-anchor arrays: the head tensors are the output
            QDecP P;
            const TensorDesc &tb = net->tensors[o[1]], &tc = net->tensors[o[3]];
            P.box = base(o[1]); P.cls = base(o[3]);
            P.n_anchors = o[21]; P.n_classes = o[20]; P.cls_stride = tc.cs;
            DD_REQUIRE(P.n_anchors == net->dec_anchors && tb.cs == 4 && tb.h == P.n_anchors && tc.h == P.n_anchors && P.n_classes <= tc.cs, DD_E_ARG,
                       "dd_net_forward: uint8 decode of %d anchors, set up for %d", P.n_anchors, net->dec_anchors);
            P.lut = reinterpret_cast<const uint8_t *>(W + (size_t)(uint32_t)o[16]);
            const float *of = reinterpret_cast<const float *>(o);
            P.box_scale = of[32]; P.box_zp = of[33]; P.sc_scale = of[34]; P.sc_zp = of[35]; P.thr = net->dec_thr;
            P.anchors = net->d_anchors; P.boxes = net->dec_boxes; P.score = net->dec_score; P.keys = net->dec_keys; P.cls_out = net->dec_cls;
            hipLaunchKernelGGL(q_ssd_decode_k, dim3(dd_ceil_div(P.n_anchors * 16, 256), nimg), dim3(256), 0, s, P);
            DD_LAUNCH_CHECK();
            return DD_OK;
        }
        default:
            DD_REQUIRE(false, DD_E_ARG, "dd_net_forward: uint8 op kind %d at %d is not built", kind, i);
    }
    return DD_OK;
}
