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
//     i8 [n][H + 2][C / 16][W + 2][16],   stored byte = a - 128 (= a ^ 0x80)
// i.e. planes of 16 channels, every image row and plane carrying one border pixel on each side, one border row above and
// below.  Border bytes hold the tensor's zero point (minus 128) from engine creation on (nobody writes them), so a 3x3 window
// never needs a bounds test and a padded tap contributes (za - za) = 0 as TFLite's skipped tap does.  A 16-pixel MFMA operand
// fragment is 4 planes x 16 pixels x 16 bytes: every lane loads 16 contiguous bytes, 16 lanes 256 contiguous bytes.
//
// Signed operands: the matrix instruction multiplies i8, hence the stored a' = a - 128 (the producer's epilogue XORs the packed
// word: a quarter of an instruction per byte; consumers read operands as they lie) and w' = w - 128 (host).  With za' = za - 128,
// zw' = zw - 128:
//     sum (a - za)(w - zw) = sum a'w'  -  zw' * sum_k a'  -  za' * sum_k w'[c]  +  K za' zw'
// The last two terms are per output channel (host: `cbias`); the second needs the row sum of the pixel's operand bytes
// (v_dot4_i32_i8) unless zw == 128.  The head tensors (plain rows, read by the decode) hold the bytes themselves.
//
// Requantisation: for multipliers < 1 (shift <= 0, e = -shift) the two roundings of MultiplyByQuantizedMultiplier,
//     y = (x M + 2^30) >> 31 (SaturatingRoundingDoublingHighMul; the truncating division and its sign-dependent nudge are
//     this floor), z = RoundingDivideByPOT(y, e),
// nest into ONE 64-bit multiply-add and shift whenever y >= 0:  z = (x M + 2^30 + 2^(30+e)) >> (31 + e).  For y < 0 both
// forms give z <= 0, which a ReLU-type clamp (lo >= zo) maps to the same byte; layers without activation (the SSD heads)
// take the literal two-step form.  tests/test_quant_host.py checks both against the literal gemmlowp statements.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "ssd_dev.h"
#include "net_priv.h"
#include "netsq_dev.h"

namespace {


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
    unsigned hw_magic, wo_magic;                  // floor(2^32 / d) + 1 for ho * wo and wo; hw_magic = 0: not exact for this launch (plain divisions)
    QReq R;
    // QEPI_ROWS, two predictors on one feature map in one launch (as q_pws_k; the small feature maps and small batches run here): fragments
    // [n_frag_a, n_mfrag) -- the wave items' groups [groups_a, n_mgroups) -- belong to the second one, with its own requantisation, weight zero
    // point and destination rows.  A group never holds fragments of both.
    int n_frag_a, groups_a;                       // 0: one layer
    uint8_t *out_b; long long img_bytes_b; int row_bytes_b, base_off_b, cout_store_b, zwc_b;
    QReq Rb;
};

// One wave item = MQ 16-channel fragments x NPF (2 or 4) 16-pixel fragments; operands straight from L2 / HBM in fragment shape (Q16 is
// that shape), no LDS.  Small layers only (SSD extras and heads, 10x10 MobileNet blocks): the big ones run q_dwpw_k.
// SPLIT = 3 (3x3 layers with few wave items: the SSD extras): the block is ONE item, wave w sums the k steps of filter row w, the partial
// accumulators meet in LDS and wave 0 runs the epilogue -- three times the waves to cover the L2 round trips of a 36-step chain
// (the 10x10 -> 5x5 extra layer has 1 200 items for 1 024 SIMDs).  Integer sums: the same bits in any order.
template <int MQ, bool ROWSUM, int NPF, bool PIPE, int SPLIT = 1>
__global__ __launch_bounds__(256) void q_conv_k(const QConvP P, const int n_items, const int n_mgroups) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int item = SPLIT > 1 ? blockIdx.x : blockIdx.x * 4 + wave;
    if (item >= n_items) return;                                   // whole waves (SPLIT: whole blocks) leave
    const int mg = item % n_mgroups, pf = item / n_mgroups;
    const size_t PP = (size_t)(P.W + 2) * 16, RP = PP * P.c16_in;  // plane / row pitch of the source
    const uint8_t *base[NPF];
    int qn[NPF], qy[NPF], qx[NPF]; bool live[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        int q = pf * (16 * NPF) + j * 16 + fr;
        live[j] = q < P.m;
        q = min(q, P.m - 1);
        const int hw = P.ho * P.wo;
        int r;                                                      // (a division without the hardware's help is ~25 vector instructions: 100 per item, next to as few as 32 MFMAs)
        if (P.hw_magic) { qn[j] = (int)__umulhi((unsigned)q, P.hw_magic); r = q - qn[j] * hw; qy[j] = (int)__umulhi((unsigned)r, P.wo_magic); }
        else { qn[j] = q / hw; r = q - qn[j] * hw; qy[j] = r / P.wo; }
        qx[j] = r - qy[j] * P.wo;
        base[j] = P.in + ((size_t)qn[j] * (P.H + 2) + qy[j] * P.stride + P.off_y) * RP + (size_t)(qx[j] * P.stride + P.off_x) * 16;
    }
    i4v acc[MQ][NPF];
#pragma unroll
    for (int m = 0; m < MQ; ++m)
#pragma unroll
        for (int j = 0; j < NPF; ++j) acc[m][j] = i4v{0, 0, 0, 0};
    int rs[NPF];
#pragma unroll
    for (int j = 0; j < NPF; ++j) rs[j] = 0;
    const int ksteps = P.kh * P.kw * P.kc_per_tap;
    const int ks_lo = SPLIT > 1 ? wave * (ksteps / SPLIT) : 0, ks_hi = SPLIT > 1 ? ks_lo + ksteps / SPLIT : ksteps;     // this wave's k steps
    // (wave-uniform) the group's first fragment, the end of its predictor's fragments
    const bool is_b = P.n_frag_a != 0 && mg >= P.groups_a;
    const int frag0 = is_b ? P.n_frag_a + (mg - P.groups_a) * MQ : mg * MQ;
    const int frag_end = P.n_frag_a != 0 && !is_b ? P.n_frag_a : P.n_mfrag;
    const i4v *wp = P.w + ((size_t)frag0 * ksteps) * 64 + lane;
    // k step ks = (tap, 64-channel slice): the operands of step ks + 1 are requested before the MFMAs of step ks are issued (a step's
    // loads followed by its own MFMAs left every step waiting out an L2 round trip: 36 of them in a 3x3 layer with 256 channels)
    auto step_off = [&](int ks, bool &kv) -> size_t {
        const int tap = ks / P.kc_per_tap, kc = ks - tap * P.kc_per_tap, dy = tap / P.kw, dx = tap - dy * P.kw;
        const int plane = 4 * kc + fq;
        kv = plane < P.c16_in;
        return (size_t)dy * RP + (size_t)dx * 16 + (size_t)min(plane, P.c16_in - 1) * PP;
    };
    auto load_step = [&](int ks, i4v (&bb)[NPF], i4v (&aa)[MQ]) {
        bool kv;
        const size_t off = step_off(ks, kv);
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            bb[j] = *reinterpret_cast<const i4v *>(base[j] + off);      // stored as a - 128: the MFMA operand as it lies
            if (!kv) bb[j] = i4v{0, 0, 0, 0};
        }
#pragma unroll
        for (int m = 0; m < MQ; ++m) aa[m] = frag0 + m < frag_end ? wp[((size_t)m * ksteps + ks) * 64] : i4v{0, 0, 0, 0};
    };
    i4v b0[NPF], a0[MQ], b1[NPF], a1[MQ];
    if constexpr (!PIPE) {                                          // launches with thousands of wave items per CU: the waves cover each other's round trips,
        for (int ks = ks_lo; ks < ks_hi; ++ks) {                    // and the second operand set only costs occupancy (b13: 138 -> 155 us with it)
            load_step(ks, b0, a0);
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                if (ROWSUM) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) rs[j] = sdot4(b0[j][d], 0x01010101, rs[j]);
                }
#pragma unroll
                for (int m = 0; m < MQ; ++m) acc[m][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[m], b0[j], acc[m][j], 0, 0, 0);
            }
        }
    } else {
    load_step(ks_lo, b0, a0);
    for (int ks = ks_lo; ks < ks_hi; ks += 2) {
        if (ks + 1 < ks_hi) load_step(ks + 1, b1, a1);
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            if (ROWSUM) {
#pragma unroll
                for (int d = 0; d < 4; ++d) rs[j] = sdot4(b0[j][d], 0x01010101, rs[j]);
            }
#pragma unroll
            for (int m = 0; m < MQ; ++m) acc[m][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[m], b0[j], acc[m][j], 0, 0, 0);
        }
        if (ks + 1 >= ks_hi) break;
        if (ks + 2 < ks_hi) load_step(ks + 2, b0, a0);
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            if (ROWSUM) {
#pragma unroll
                for (int d = 0; d < 4; ++d) rs[j] = sdot4(b1[j][d], 0x01010101, rs[j]);
            }
#pragma unroll
            for (int m = 0; m < MQ; ++m) acc[m][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[m], b1[j], acc[m][j], 0, 0, 0);
        }
    }
    }
    if constexpr (SPLIT > 1) {
        __shared__ int xch[SPLIT - 1][MQ * NPF * 4 + NPF][64];
        if (wave > 0) {
#pragma unroll
            for (int m = 0; m < MQ; ++m)
#pragma unroll
                for (int j = 0; j < NPF; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xch[wave - 1][(m * NPF + j) * 4 + r][lane] = acc[m][j][r];
#pragma unroll
            for (int j = 0; j < NPF; ++j) xch[wave - 1][MQ * NPF * 4 + j][lane] = rs[j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < SPLIT - 1; ++w) {
#pragma unroll
            for (int m = 0; m < MQ; ++m)
#pragma unroll
                for (int j = 0; j < NPF; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][j][r] += xch[w][(m * NPF + j) * 4 + r][lane];
#pragma unroll
            for (int j = 0; j < NPF; ++j) rs[j] += xch[w][MQ * NPF * 4 + j][lane];
        }
    }
    if (ROWSUM) {
#pragma unroll
        for (int j = 0; j < NPF; ++j) {
            rs[j] += __shfl_xor(rs[j], 16, 64);
            rs[j] += __shfl_xor(rs[j], 32, 64);
            rs[j] *= is_b ? P.zwc_b : P.zwc;
        }
    }
    if (P.epi == QEPI_Q16) {
        // the host packed fragment m's row 4g + r with channel 64 mg + 16 g + 4 m + r: this lane holds the 16 consecutive channels of plane 4 mg + fq
        if constexpr (MQ == 4) {
            const i4v *cb = reinterpret_cast<const i4v *>(P.cbias + 64 * mg + 16 * fq);
            const i4v c0 = cb[0], c1 = cb[1], c2 = cb[2], c3 = cb[3];
            const size_t PPo = (size_t)(P.Wo + 2) * 16;
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                u4v o;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const i4v c = m == 0 ? c0 : m == 1 ? c1 : m == 2 ? c2 : c3;
                    unsigned wv = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) wv |= (unsigned)q_requant(acc[m][j][r] + rs[j] + c[r], P.R) << (8 * r);
                    o[m] = wv ^ 0x80808080u;
                }
                if (live[j]) {
                    uint8_t *dst = P.out + (((size_t)qn[j] * (P.Ho + 2) + qy[j] + 1) * P.c16_out + 4 * mg + fq) * PPo + (size_t)(qx[j] + 1) * 16;
                    *reinterpret_cast<u4v *>(dst) = o;
                }
            }
        }
    } else {
        const QReq R = is_b ? P.Rb : P.R;
        uint8_t *const out = is_b ? P.out_b : P.out;
        const long long img_bytes = is_b ? P.img_bytes_b : P.img_bytes_out;
        const int row_bytes = is_b ? P.row_bytes_b : P.row_bytes, base_off = is_b ? P.base_off_b : P.base_off, cout_store = is_b ? P.cout_store_b : P.cout_store;
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            const int ch = 16 * (frag0 + m - (is_b ? P.n_frag_a : 0)) + 4 * fq;          // the channel inside its predictor
            if (frag0 + m >= frag_end || ch >= cout_store) continue;
            const i4v c = *reinterpret_cast<const i4v *>(P.cbias + 16 * (frag0 + m) + 4 * fq);
#pragma unroll
            for (int j = 0; j < NPF; ++j) {
                unsigned wv = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) wv |= (unsigned)q_requant(acc[m][j][r] + rs[j] + c[r], R) << (8 * r);
                if (live[j]) {
                    uint8_t *dst = out + (size_t)qn[j] * img_bytes + base_off + (size_t)(qy[j] * P.wo + qx[j]) * row_bytes + ch;
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
    const i4v *w2;                                            // HL: the lo part of the split filter (w - zw = hi + lo), same layout
    const long long *cq;                                      // [32]: cbias * M + C, the requantisation's addend per channel
    uint8_t *out; long long out_bytes; int in_zp, zwc;
    QReq R;
};

// 3x3 stride-2 conv over the 3 colour channels, 32 output channels: one MFMA k slice (3 filter rows x 9 bytes of an image row,
// 37 zero slots) per 16 pixels, two fragments of 16 channels.  Taps outside the image read the input zero point.
// Bound by vector-instruction issue (177 per fragment against 2 MFMAs): the filter split hi + lo (HL: four MFMAs, no row sum of the window),
// per-channel 64-bit addends, saturating packs (q_requant_pack4), 32-bit addresses.
constexpr int C0F = 4;      // 16-pixel fragments per wave: twelve loads in flight per lane (one fragment per wave was bound by the HBM latency: 222 us per 384 frames)
template <bool HL, int SAT>
__global__ __launch_bounds__(256) void q_conv0_k(const QConv0P P, const int n_frags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int f0 = (blockIdx.x * 4 + wave) * C0F;
    if (f0 >= n_frags) return;
    const unsigned zp4 = (unsigned)P.in_zp * 0x01010101u;
    const int hw = P.ho * P.wo;
    // Range-checked buffer accesses: a window that starts before the batch (column -1 of the first row) or ends past it reads zeros instead of
    // faulting -- those bytes are replaced by the zero point below anyway -- and a pixel past the last one stores nothing; no clamps, no
    // 64-bit address arithmetic (the host checked that both tensors stay below 2^31 / 2^32 bytes).
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src), 0, (int)P.total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(P.out, 0, (int)P.out_bytes, 0x00020000);
    unsigned w[C0F][3]; int o_[C0F]; bool rowok[C0F]; int px[C0F]; unsigned so[C0F];
    // the lane's pixel of the first fragment by division (once), of the following ones by stepping 16 pixels along the raster
    int qn, qy, qx;
    {
        const int q = f0 * 16 + fr;
        qn = q / hw; const int r0 = q - qn * hw; qy = r0 / P.wo; qx = r0 - qy * P.wo;
    }
    const int PPo = (P.wo + 2) * 16;
#pragma unroll
    for (int j = 0; j < C0F; ++j) {
        px[j] = qx;
        const int row = qy * P.stride + fq - P.pad_t, col = qx * P.stride - P.pad_l;      // first of the three source pixels of this filter row
        rowok[j] = fq < 3 && row >= 0 && row < P.H;
        const int a = ((qn * P.H + row) * P.W + col) * 3;      // (< 0 or past the end only for rows / pixels whose bytes are not used)
        const int a4 = a & ~3;
        o_[j] = a - a4;
#pragma unroll
        for (int i = 0; i < 3; ++i) w[j][i] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, a4 + 4 * i, 0, 0);
        // a pixel past the end of the batch: an offset outside the tensor (the store is dropped)
        so[j] = (f0 + j) * 16 + fr < P.m ? (unsigned)((((qn * (P.ho + 2) + qy + 1) * 2 + (fq >> 1)) * PPo) + (qx + 1) * 16 + (fq & 1) * 8) : 0xfffffff0u;
        qx += 16;                                                   // (wo >= 16: at most one row step per fragment)
        if (qx >= P.wo) { qx -= P.wo; if (++qy == P.ho) { qy = 0; ++qn; } }
    }
    const i4v wa = P.w[lane], wb = P.w[64 + lane];
    i4v wal = {0, 0, 0, 0}, wbl = wal;
    if constexpr (HL) { wal = P.w2[lane]; wbl = P.w2[64 + lane]; }
    const int M0 = P.R.M, sh0 = P.R.e - 1, lo0 = P.R.lo, hi0 = P.R.hi;
    // fragment m's row 4g + r was packed with channel 8g + 4m + r: this lane holds channels 8 fq .. 8 fq + 7
    long long CQ[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) CQ[r] = P.cq[8 * fq + r];
#pragma unroll
    for (int j = 0; j < C0F; ++j) {
        unsigned d[3];
        d[0] = __builtin_amdgcn_alignbyte(w[j][1], w[j][0], o_[j]);
        d[1] = __builtin_amdgcn_alignbyte(w[j][2], w[j][1], o_[j]);
        d[2] = __builtin_amdgcn_alignbyte(0u, w[j][2], o_[j]);
        if (!rowok[j]) { d[0] = zp4; d[1] = zp4; d[2] = zp4; }
        const int col = px[j] * P.stride - P.pad_l;
        // columns outside the image: pixel i of the three covers bytes 3i .. 3i + 2 (only the waves at the left / right edge run this)
        if (__builtin_amdgcn_ballot_w64(col < 0 || col + 2 >= P.W) != 0ull) {
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
        // (bytes 9 .. 15 of a k group and the whole of group 3 meet zero weights: what they hold does not matter -- except to the row sum)
        i4v b;
        b[0] = (int)(d[0] ^ 0x80808080u); b[1] = (int)(d[1] ^ 0x80808080u); b[2] = (int)(d[2] ^ 0x80808080u); b[3] = 0;
        int rs = 0;
        if constexpr (!HL) {
            if (fq == 3) { b[0] = 0; b[1] = 0; b[2] = 0; }
            rs = sdot4(b[0], 0x01010101, sdot4(b[1], 0x01010101, sdot4(b[2] & 0xff, 0x01010101, 0)));
            rs += __shfl_xor(rs, 16, 64);
            rs += __shfl_xor(rs, 32, 64);
            rs *= P.zwc;
        }
        i4v acc0 = {0, 0, 0, 0}, acc1 = acc0;
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wb, b, acc1, 0, 0, 0);
        if constexpr (HL) {
            acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wal, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wbl, b, acc1, 0, 0, 0);
        }
        const unsigned lo = q_requant_pack4<SAT>(acc0[0] + rs, acc0[1] + rs, acc0[2] + rs, acc0[3] + rs, M0, CQ[0], CQ[1], CQ[2], CQ[3], sh0, lo0, hi0);
        const unsigned hi = q_requant_pack4<SAT>(acc1[0] + rs, acc1[1] + rs, acc1[2] + rs, acc1[3] + rs, M0, CQ[4], CQ[5], CQ[6], CQ[7], sh0, lo0, hi0);
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(u2v{lo ^ 0x80808080u, hi ^ 0x80808080u}, rdst, (int)so[j], 0, 0);
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
            for (int i = 0; i < 16; ++i) acc[i] += ((int)(a[i >> 2] << (24 - 8 * (i & 3))) >> 24) * (int)wt[i];      // the stored byte is a - 128, signed
        }
    u4v o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        unsigned wv = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) wv |= (unsigned)q_requant(acc[4 * d + r], P.R) << (8 * r);
        o[d] = wv ^ 0x80808080u;
    }
    uint8_t *dst = P.out + (((size_t)n * (P.ho + 2) + y + 1) * P.c16 + pl) * ((size_t)(P.wo + 2) * 16) + (size_t)(x + 1) * 16;
    *reinterpret_cast<u4v *>(dst) = o;
}

// The same layer on the matrix pipe (block-diagonal product, see q_dwpw_k): one wave item = one plane of 16 channels x four 16-pixel
// fragments, operands straight from the bordered planes in HBM / L2 (16 bytes per lane and tap group), the filter's six operand
// registers built once per item.  32 + 55 us -> for the two 10x10 layers of the SSD backbone that q_dwpw_k's ring does not fit.
struct QDwmP {
    const uint8_t *in; int H, W, c16;
    int stride, off_y, off_x, ho, wo, m;
    const uint2 *dw_a; const int *dw_cb;
    uint8_t *out;
    unsigned hw_magic, wo_magic, c16_magic;   // floor(2^32 / d) + 1; hw_magic = 0: the pixel index times hw does not fit 32 bits (large maps: plain divisions)
    QReq R;
};

template <bool SAT>
__global__ __launch_bounds__(256) void q_dwm_k(const QDwmP P, const int n_items) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int item = blockIdx.x * 4 + wave;
    if (item >= n_items) return;
    const int fg = P.hw_magic ? (int)__umulhi((unsigned)item, P.c16_magic) : item / P.c16, cg = item - fg * P.c16;   // planes fastest: the waves of a block read the same pixels' other planes
    const unsigned PP = (unsigned)(P.W + 2) * 16u, RP = PP * (unsigned)P.c16, PPo = (unsigned)(P.wo + 2) * 16u;    // (32-bit offsets: the launcher checks the tensor sizes)
    const uint2 ab = P.dw_a[cg * 64 + lane];
    const i4v cbv = *reinterpret_cast<const i4v *>(P.dw_cb + cg * 16 + 4 * fq);
    unsigned dmask[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dmask[d] = (fr >> 2) == d ? 0xffu << (8 * (fr & 3)) : 0u;
    const uint8_t *src[4]; uint8_t *dst[4]; bool live[4];
    const int hw = P.ho * P.wo;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        int q = (fg * 4 + f) * 16 + fr;
        live[f] = q < P.m;
        q = min(q, P.m - 1);
        // (a division without the hardware's help is ~25 vector instructions, and a wave item had eight of them around its 24 MFMAs)
        int n, y;
        if (P.hw_magic) { n = (int)__umulhi((unsigned)q, P.hw_magic); y = (int)__umulhi((unsigned)(q - n * hw), P.wo_magic); }
        else { n = q / hw; y = (q - n * hw) / P.wo; }
        const int r = q - n * hw, x = r - y * P.wo;
        src[f] = P.in + (size_t)((unsigned)(n * (P.H + 2) + y * P.stride + P.off_y) * RP + (unsigned)cg * PP + (unsigned)(x * P.stride + P.off_x) * 16u);
        dst[f] = P.out + (size_t)((unsigned)((n * (P.ho + 2) + y + 1) * P.c16 + cg) * PPo + (unsigned)(x + 1) * 16u + 4u * (unsigned)fq);
    }
    i4v acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] = cbv;
    i4v b[3][4];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int tp = min(4 * ks + fq, 8);
        const unsigned to = (unsigned)(tp / 3) * RP + (unsigned)(tp % 3) * 16u;
#pragma unroll
        for (int f = 0; f < 4; ++f) b[ks][f] = *reinterpret_cast<const i4v *>(src[f] + to);
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const unsigned sel = 0x01010101u * (unsigned)ks;
        const unsigned rh = __builtin_amdgcn_perm(ab.x, ab.x, sel), rl = __builtin_amdgcn_perm(ab.y, ab.y, sel);
        i4v Ah, Al;
#pragma unroll
        for (int d = 0; d < 4; ++d) { Ah[d] = (int)(rh & dmask[d]); Al[d] = (int)(rl & dmask[d]); }
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah, b[ks][f], acc[f], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Al, b[ks][f], acc[f], 0, 0, 0);
    }
    const int M = P.R.M, sh = P.R.e - 1, lo = P.R.lo, hi = P.R.hi;
    const long long C = P.R.C;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const unsigned packed = 0x80808080u ^ q_requant_pack4<SAT ? 1 : 0>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], M, C, C, C, C, sh, lo, hi);
        if (live[f]) *reinterpret_cast<unsigned *>(dst[f]) = packed;
    }
}

// ------------------------------------------------------------------------------------------------ SSD decode
// First stage of TFLite_Detection_PostProcess on the quantised head tensors (kernels/detection_postprocess.cc:
// DequantizeBoxEncodings, DequantizeClassPredictions behind the graph's uint8 LOGISTIC): the lanes of an anchor sweep its
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
    // eight lanes per anchor: six of them take 16 class bytes each (one load), table look-ups from LDS, a three-step butterfly
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = P.lut[threadIdx.x];
    __syncthreads();
    const int a = (blockIdx.x * 256 + threadIdx.x) >> 3, sub = threadIdx.x & 7;
    const size_t z = blockIdx.y;
    const bool live = a < P.n_anchors;
    const uint8_t *c = P.cls + (z * P.n_anchors + (live ? a : 0)) * P.cls_stride;
    int best = -1, bi = 0x7fffffff;
    if (sub * 16 < P.n_classes) {
        const u4v v = *reinterpret_cast<const u4v *>(c + sub * 16);         // cls_stride is a multiple of 16: aligned, inside the row
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = sub * 16 + i;                                      // class 0 = background
            const int q = lut[(v[i >> 2] >> (8 * (i & 3))) & 0xffu];
            if (k >= 1 && k < P.n_classes && q > best) { best = q; bi = k - 1; }
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
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

// ------------------------------------------------------------------------------------------------ MobileNet block in one launch
// depthwise 3x3 (stride 1 or 2) -> uint8 -> pointwise 1x1 -> uint8: one read of the block input, one write of its output, BOTH
// contractions on the matrix pipe.
//
// Why the depthwise stage is on MFMA too.  Measured (scripts/experiments/valu_rates.hip, profiles/r04_valu_rates.txt): v_perm_b32,
// v_dot2c_i32_i16, v_mad_i64_i32 cost 4.2 cycles of a SIMD per wave instruction however many waves share it (v_add_u32: 2.6), and
// a depthwise output on the vector ALU is 5 + 5 of them plus the requantisation: 53 cycles per 64 outputs -- the first form of this
// kernel spent 9.9 k of its 22 k cycles per tile there at 69 % VALU issue, and overlapping that stage with the matrix stage of the
// neighbouring tile (two wave teams) gained nothing: one wave per SIMD cannot issue faster than one instruction per 5 cycles
// (profiles/r04_layers_ssd_i8_b384_valu_dw.txt, ..._teams_valu_dw.txt).  As a block-diagonal product the same stage is 6
// v_mfma_i32_16x16x64_i8 per 16 pixels x 16 channels -- 24 matrix cycles per 64 outputs at 1/16 of the array's utilisation, still
// less than half the vector cost, and no unpacking at all: a k slice = four taps x 16 channels = exactly the 16 bytes per lane the
// bordered plane layout serves.
//
//   A operand (filter): lane (g, r) of k step ks holds tap 4 ks + g of channel r on the diagonal (byte r of its 16, zeros
//     elsewhere); w - zw spans 9 bits, so it is split hi = clamp(w - zw, -128, 127), lo = rest and the two products share one
//     accumulator.  Built in registers from six bytes per lane and plane (v_perm_b32 replicates, four masks select).
//   B operand (pixels): lane (g, p) reads the 16 channel bytes of tap 4 ks + g at pixel p straight from the LDS row ring.
//   accumulator: rows = channels, so a lane ends up with 4 consecutive channels of one pixel: requantise, pack, one
//     ds_write_b32 into the pointwise stage's operand tile [plane][pixel][16]; v_dot4_i32_i8 of the packed word feeds the row sum.
//
// A block of NW = (COUT / 64) * WP waves walks a contiguous range of 64-pixel tiles (raster order inside a frame; the last tile of
// a frame is partly filled).  Rows: the bordered input rows a tile's windows touch live in an LDS ring (slot = global row index
// % NR, a row = [CIN / 16][W + 2][16] bytes as it lies in HBM); the rows the NEXT tile adds are requested into registers at the
// start of the pointwise stage and written to the ring at its end (their slots are dead since barrier A): the ring holds one
// tile's rows, the HBM latency hides behind the MFMAs.  Pointwise stage: wave (wm, wp) keeps the A fragments of its 64 output
// channels for all of K in registers for the whole launch (<= 128 VGPRs), accumulates from its cbias registers, adds
// zwc * rowsum, requantises, stores 16 bytes per lane (fragment m's row 4g + r was packed with channel 64 wm + 16 g + 4 m + r).
// Two barriers per tile.
struct QDwpwP {
    const uint8_t *in; int H, W;
    int off_y, off_x, ho, wo, hw, tiles_per_frame;
    uint8_t *out; int c16_out;
    const uint2 *dw_a;               // [CIN / 16][64 lanes]: .x bytes 0..2 = hi parts of the lane's tap in k steps 0..2, .y = lo parts
    const long long *dw_cq;          // [CIN]: (bias - (za - 128) * sum_t (w_t - zw)) * M + C: the depthwise requantisation's 64-bit addend per channel
    const i4v *w;                    // [COUT / 64][4][KC][64 lanes]
    const i4v *w2;                   // HL: the second filter (see below), same layout
    const int *cbias;                // [COUT]
    const long long *cq;             // [COUT]: cbias * M + C (the pointwise requantisation's addend per channel; used when no row sums are)
    int zwc, NR;
    int dup;                         // CIN = 32 only: the depthwise outputs are written to k slots 32..63 as well (see the kernel)
    unsigned wo_magic, nr_magic, tpf_magic;   // floor(2^32 / d) + 1 for the output width, the ring size, the tiles per frame (exact for every value divided here)
    unsigned long long *dbg;         // DD_Q_STAMPS=1: per wave, cycles spent in each part of the tile loop (diagnostic launches only)
    QReq Rd, Rp;
};

constexpr int QT = 64;               // pixels per tile

// The tile loop is bound by vector-instruction issue (423 vector instructions per wave and tile in the 32-channel block against 28 MFMAs:
// scripts/experiments/isa_mix.py), so everything that does not depend on the tile stays in registers where they are to be had:
//   * per-channel requantisation addends (bias * M + C, 64 bits, computed by the host): accumulators start at zero (free: the first MFMA
//     takes the literal) and v_mad_i64_i32 adds the channel's constant -- no accumulator initialisation, no bias add;
//   * KEEP (<= 2 planes per wave): the depthwise A operands, built once per launch instead of per plane and tile (30 instructions each);
//   * FOLDP (no row sums, K <= 256): the pointwise addends, 32 registers.
// MW = 16-channel fragments of the pointwise filter per wave: 4 (64 channels, <= 128 VGPRs of filter, 256-register waves, two per SIMD) or
// 2 (32 channels, 128-register waves, four per SIMD -- measured slower, 131 vs 89 us on the 512-channel block: issue-bound, not latency-bound).
// P.dup (CIN = 32: half of the 64-byte k slice is free): the depthwise stage writes its bytes into BOTH halves and the host packs hi into the
// filter's first 32 k slots, lo into the other 32 -- the split filter in ONE MFMA per fragment, no row sums, no second filter registers.
// HL: the pointwise filter as w - zw split into hi = clamp(w - zw, -128, 127) and lo = rest, two MFMAs per fragment and k slice on one
// accumulator instead of one MFMA plus the zwc * rowsum correction: for the layers with few input channels the matrix pipe has the time
// and the row-sum machinery (dot products, cross-lane sums, LDS atomics, an add per output) goes.
template <int CIN, int COUT, int WP, int STRIDE, int LPT, bool ROWSUM, int MW = 4, int SAT = 0, bool HL = false, int QTT = QT>
__global__ __launch_bounds__((COUT / (16 * MW)) * WP * 64, (COUT / (16 * MW)) * WP <= 2 ? 3 : QTT > 64 ? 4 : 2) void q_dwpw_k(const QDwpwP P, const int n_tiles, const int tiles_per_block) {
    constexpr int WM = COUT / (16 * MW), NW = WM * WP, NT = NW * 64;
    static_assert(MW == 4 || MW == 2, "fragments per wave");
    static_assert(!(HL && ROWSUM), "the split filter needs no row sums");
    constexpr int KC = (CIN + 63) / 64, CINP = KC * 64, C16 = CIN / 16;
    // depthwise work units: (plane, four 16-pixel fragments); a tile of QTT pixels has QTT / 64 fragment quads per plane.  QTT = 128 with four
    // waves (block 1: 2 planes x 2 quads) does per wave and tile what QTT = 64 does with two, in 33 instead of 26.5 KB of LDS per block: 16 waves
    // per CU instead of 12, and that block's time goes as 1 / waves (533 / 443 / 338 / 282 / 243 us at 2..6 blocks of two waves per CU).
    constexpr int FPT = QTT / 16, UNITS = C16 * (FPT / 4);
    static_assert(QTT % 64 == 0 && UNITS % NW == 0, "depthwise units over waves");
    constexpr int CPW = UNITS / NW;                                   // units per wave
    static_assert(FPT == 4 || CPW == 1, "several units per wave: one fragment quad only");
    constexpr bool KEEP = CPW <= 2, FOLDP = !ROWSUM && KC <= 4 && QTT == 64;      // (128-pixel tiles run on 128 registers: the 32 of the folded addends are not to be had)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (the wave number in a scalar register)
    const int fr = lane & 15, fq = lane >> 4;
    const int wm = wave / WP, wp = wave % WP;
    const int RB = (P.W + 2) * CIN, PP = (P.W + 2) * 16;           // ring row / plane pitch (bytes)
    uint8_t *opnd = smem;                                           // [CINP / 16][QTT][16]
    int *rowsum = reinterpret_cast<int *>(opnd + QTT * CINP);        // [2][QTT]
    i4v *pinfo = reinterpret_cast<i4v *>(rowsum + 2 * QTT);          // [2][QTT]: ring offsets of the pixel's window (rows 0..2, first column), offset of its output
    int *cbl = reinterpret_cast<int *>(pinfo + 2 * QTT);             // [COUT]: the pointwise layer's per-channel constants (!FOLDP)
    uint2 *dwa_l = reinterpret_cast<uint2 *>(cbl + COUT);           // !KEEP: the depthwise tables in LDS ([C16][64] operand bytes, [CIN] addends): a global
    long long *dwq_l = reinterpret_cast<long long *>(dwa_l + (KEEP ? 0 : C16 * 64));     // load at the head of every plane is ~500 cycles before its first use
    uint8_t *ring = reinterpret_cast<uint8_t *>(dwq_l + (KEEP ? 0 : CIN));   // [NR][RB], last (>= 4 KB of other data below it: see pf_issue)

    // the wave's pointwise filter, once
    // wave wm's fragments: MW consecutive ones of the host's 4-fragment groups (fragment 4 mg + m holds channels 64 mg + 16 g + 4 m + r)
    i4v Wr[MW][KC];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) Wr[m][kc] = P.w[((size_t)(wm * MW + m) * KC + kc) * 64 + lane];
    const int mg = (wm * MW) / 4, m0 = (wm * MW) % 4;               // the 64-channel group and the first fragment inside it
    // Split filter (HL): only a tensor's extreme weights overflow int8, so the lo part is zero in nearly every (fragment, k slice).  wlm = the k
    // slices in which one of this wave's fragments has a lo part (wave-uniform, fixed for the launch); the lo fragments of the FIRST such slice stay
    // in registers (16 of them, not 16 per slice), those of any further one are fetched from L2 where they are used -- exact either way.
    unsigned wlm = 0;
    int kcl = -1;
    i4v Wl1[HL ? MW : 1];
    if constexpr (HL) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            int any = 0;
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                const i4v w = P.w2[((size_t)(wm * MW + m) * KC + kc) * 64 + lane];
                any |= w[0] | w[1] | w[2] | w[3];
            }
            if (__builtin_amdgcn_ballot_w64(any != 0) != 0ull) { wlm |= 1u << kc; if (kcl < 0) kcl = kc; }
        }
#pragma unroll
        for (int m = 0; m < MW; ++m) Wl1[m] = P.w2[((size_t)(wm * MW + m) * KC + max(kcl, 0)) * 64 + lane];
    }
    long long CP[FOLDP ? MW : 1][4];
    if constexpr (FOLDP) {
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) CP[m][r] = P.cq[64 * mg + 16 * fq + 4 * (m0 + m) + r];
    } else {
        for (int i = tid; i < COUT; i += NT) cbl[i] = P.cbias[i];
    }
    if constexpr (!KEEP) {
        for (int i = tid; i < C16 * 64; i += NT) dwa_l[i] = P.dw_a[i];
        for (int i = tid; i < CIN; i += NT) dwq_l[i] = P.dw_cq[i];
    }
    if (CIN < CINP) for (int i = tid; i < QTT * CINP / 16; i += NT) reinterpret_cast<u4v *>(opnd)[i] = u4v{0, 0, 0, 0};   // k slots without channels
    for (int i = tid; i < 2 * QTT; i += NT) rowsum[i] = 0;
    // depthwise lane constants: which byte of the 16 is this lane's diagonal element; which window column its tap of k step ks is
    // (k step 0: taps 0..3 = row 0 columns 0..2, row 1 column 0; step 1: taps 4..7 = row 1 columns 1, 2, row 2 columns 0, 1; step 2: tap 8 =
    // row 2 column 2, and the k slots past it carry zero weights: any valid address will do)
    unsigned dmask[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dmask[d] = (fr >> 2) == d ? 0xffu << (8 * (fr & 3)) : 0u;
    int tap_dx[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) tap_dx[ks] = (min(4 * ks + fq, 8) % 3) * 16;
    const bool row_up0 = fq == 3, row_up1 = fq >= 2;                // the lane's tap of k step 0 / 1 lies in the later of the step's two rows
    // (the hi parts of the wave's planes stay in registers; a lo part -- needed in few k steps of few planes, see the depthwise stage -- is built
    // from the table word where it is used)
    auto build_a = [&](int cg, i4v (&Ah)[3], unsigned &lo_w, long long (&Cq)[4], unsigned &lom) {
        const uint2 ab = P.dw_a[cg * 64 + lane];
        lo_w = ab.y;
        lom = (unsigned)__builtin_amdgcn_readfirstlane((int)(ab.y >> 24));     // which k steps of this plane have a lo part at all (netsq.pack_dw_mfma)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const unsigned rh = __builtin_amdgcn_perm(ab.x, ab.x, 0x01010101u * (unsigned)ks);
#pragma unroll
            for (int d = 0; d < 4; ++d) Ah[ks][d] = (int)(rh & dmask[d]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Cq[r] = P.dw_cq[cg * 16 + 4 * fq + r];
    };
    i4v KAh[KEEP ? CPW : 1][3];
    long long KCq[KEEP ? CPW : 1][4];
    unsigned Klom[KEEP ? CPW : 1] = {}, Klow[KEEP ? CPW : 1] = {};
    if constexpr (KEEP) {
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) build_a((wave * CPW + ci) % C16, KAh[ci], Klow[ci], KCq[ci], Klom[ci]);
    }

    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(n_tiles, t_begin + tiles_per_block);
    if (t_begin >= t_end) return;
    const int Md = P.Rd.M, shd = P.Rd.e - 1, Mp = P.Rp.M, shp = P.Rp.e - 1;
    const long long Cp = P.Rp.C;
    const int lod = P.Rd.lo, hid = P.Rd.hi, lop = P.Rp.lo, hip_ = P.Rp.hi;
    const int out_row = P.c16_out * ((P.wo + 2) * 16);               // bytes of one bordered output row (all planes)
    const unsigned ring_bytes = (unsigned)P.NR * (unsigned)RB;

    // (divisions by the layer's constants as multiply-high: a hardware-less integer division is ~40 scalar or ~25 vector instructions, and
    // a tile step had ten of them -- 320 scalar instructions per tile and wave in the 32-channel block)
    auto tile_rows = [&](int t, int &n, int &q0, int &q1, int &ga, int &gb) {
        n = (int)__umulhi((unsigned)t, P.tpf_magic);
        q0 = (t - n * P.tiles_per_frame) * QTT;
        q1 = min(q0 + QTT, P.hw) - 1;
        const int y0 = (int)__umulhi((unsigned)q0, P.wo_magic), y1 = (int)__umulhi((unsigned)q1, P.wo_magic);
        ga = n * (P.H + 2) + y0 * STRIDE + P.off_y;
        gb = n * (P.H + 2) + y1 * STRIDE + P.off_y + 2;
    };
    // threads 0 .. QTT - 1: where pixel tid of the tile reads and writes.  One wave's work on every tile's critical path: the row's ring slot
    // from the tile's first (a scalar division) plus the rows in between, 24-bit multiplies (the host checked the ranges).
    auto geometry = [&](int n, int q0, int q1, int ga, int buf) {
        const int q = q0 + tid;
        const int qc = min(q, q1);
        const int y = (int)__umulhi((unsigned)qc, P.wo_magic), x = qc - __mul24(y, P.wo);
        const int y0 = (int)__umulhi((unsigned)q0, P.wo_magic);
        const int sa = ga - (int)__umulhi((unsigned)ga, P.nr_magic) * P.NR;
        int s0 = sa + (y - y0) * STRIDE;
        s0 = s0 >= P.NR ? s0 - P.NR : s0;
        const int s1 = s0 + 1 == P.NR ? 0 : s0 + 1, s2 = s1 + 1 == P.NR ? 0 : s1 + 1;
        const int col = (x * STRIDE + P.off_x) * 16;
        const unsigned po = q <= q1 ? (unsigned)__mul24(n * (P.ho + 2) + y + 1, out_row) + (unsigned)(x + 1) * 16u : 0xffffffffu;
        pinfo[buf * QTT + tid] = i4v{__mul24(s0, RB) + col, __mul24(s1, RB) + col, __mul24(s2, RB) + col, (int)po};
    };
    auto load_rows_sync = [&](int lo, int hi) {                     // whole rows [lo, hi] into the ring (start of the block's range only)
        if (hi < lo) return;
        const unsigned nb = (unsigned)(hi - lo + 1) * RB;
        const uint8_t *src = P.in + (size_t)lo * RB;
        for (unsigned idx = tid * 16u; idx < nb; idx += NT * 16u) {
            const unsigned row = idx / (unsigned)RB, off = idx - row * RB;
            *reinterpret_cast<u4v *>(ring + (size_t)((lo + row) % P.NR) * RB + off) = *reinterpret_cast<const u4v *>(src + idx);
        }
    };
    // ---- depthwise stage: this wave's planes x the four pixel fragments of the tile whose geometry is in buffer gbuf
    auto dw_planes = [&](int gbuf, int rsb) __attribute__((always_inline)) {
        int rs[4] = {0, 0, 0, 0};
        constexpr int FB = QTT > 64 ? 2 : 4;                           // fragments in flight (see below)
        constexpr bool TAP_ONCE = FB == 4;                             // the ring offsets of all four fragments up front (shared by the wave's planes), or per pass
        int tapoff[3][4];                                            // ring offset of this lane's tap of k step ks at its pixel of fragment f (plane 0)
        const int fb = 4 * ((wave * CPW) / C16);                       // the wave's fragment quad (one per wave: see the assertion)
        auto taps = [&](int f0, int nfr) __attribute__((always_inline)) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (f < f0 || f >= f0 + nfr) continue;
                const i4v pi = pinfo[gbuf * QTT + 16 * (fb + f) + fr];
                tapoff[0][f] = (row_up0 ? pi[1] : pi[0]) + tap_dx[0];
                tapoff[1][f] = (row_up1 ? pi[2] : pi[1]) + tap_dx[1];
                tapoff[2][f] = pi[2] + tap_dx[2];
            }
        };
        if constexpr (TAP_ONCE) taps(0, 4);
        auto plane = [&](int unit, int ci) __attribute__((always_inline)) {                         // (ci: a literal after unrolling when KEEP)
            const int cg = unit % C16;
            const int pofs = cg * PP;
            uint2 ab = make_uint2(0u, 0u);
            if constexpr (!KEEP) ab = dwa_l[cg * 64 + lane];
            // Only a tensor's extreme weights overflow int8: most planes have no lo part in most k steps, and those MFMAs (half of the
            // stage's) are skipped -- a product with zeros, the same bits.
            unsigned lom;
            if constexpr (KEEP) lom = Klom[ci];
            else lom = (unsigned)__builtin_amdgcn_readfirstlane((int)(ab.y >> 24));
            long long Cq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (KEEP) Cq[r] = KCq[ci][r];
                else Cq[r] = dwq_l[cg * 16 + 4 * fq + r];
            }
            // FB fragments in flight: all four, or two at a time where the block runs four waves per SIMD on 128 registers each (their
            // operand double buffer is 16 registers instead of 32; the other waves cover the shorter chains)
#pragma unroll
            for (int f0 = 0; f0 < 4; f0 += FB) {
                if constexpr (!TAP_ONCE) taps(f0, FB);
                i4v acc[FB];
#pragma unroll
                for (int f = 0; f < FB; ++f) acc[f] = i4v{0, 0, 0, 0};
                i4v b[2][FB];
#pragma unroll
                for (int f = 0; f < FB; ++f) b[0][f] = *reinterpret_cast<const i4v *>(ring + tapoff[0][f0 + f] + pofs);
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    i4v Ah;
                    if constexpr (KEEP) Ah = KAh[ci][ks];
                    else {
                        const unsigned rh = __builtin_amdgcn_perm(ab.x, ab.x, 0x01010101u * (unsigned)ks);
#pragma unroll
                        for (int d = 0; d < 4; ++d) Ah[d] = (int)(rh & dmask[d]);
                    }
                    if (ks < 2) {                                    // the next k step's operands are on their way while this one multiplies
#pragma unroll
                        for (int f = 0; f < FB; ++f) b[(ks + 1) & 1][f] = *reinterpret_cast<const i4v *>(ring + tapoff[ks + 1][f0 + f] + pofs);
                    }
#pragma unroll
                    for (int f = 0; f < FB; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah, b[ks & 1][f], acc[f], 0, 0, 0);
                    if (lom & (1u << ks)) {
                        i4v Al;
                        unsigned lo_w;
                        if constexpr (KEEP) lo_w = Klow[ci]; else lo_w = ab.y;
                        const unsigned rl = __builtin_amdgcn_perm(lo_w, lo_w, 0x01010101u * (unsigned)ks);
#pragma unroll
                        for (int d = 0; d < 4; ++d) Al[d] = (int)(rl & dmask[d]);
#pragma unroll
                        for (int f = 0; f < FB; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Al, b[ks & 1][f], acc[f], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int f = 0; f < FB; ++f) {
                    unsigned packed = q_requant_pack4<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Md, Cq[0], Cq[1], Cq[2], Cq[3], shd, lod, hid);
                    packed ^= 0x80808080u;
                    if (ROWSUM) rs[f0 + f] = sdot4((int)packed, 0x01010101, rs[f0 + f]);
                    *reinterpret_cast<unsigned *>(opnd + ((size_t)cg * QTT + 16 * (fb + f0 + f) + fr) * 16 + 4 * fq) = packed;
                    if constexpr (2 * CIN == CINP && !ROWSUM && !HL) {   // (32 channels in a 64-byte k slice) the same bytes again in the slice's other half
                        if (P.dup) *reinterpret_cast<unsigned *>(opnd + ((size_t)(cg + C16) * QTT + 16 * (fb + f0 + f) + fr) * 16 + 4 * fq) = packed;
                    }
                }
            }
        };
#ifndef DD_DW_STREAM
#define DD_DW_STREAM 1
#endif
        if constexpr (TAP_ONCE && DD_DW_STREAM != 0 && !KEEP) {
            // The wave's planes as ONE stream of (plane, k step, fragment) operands with a rolling window of DW_W of them in flight: the next plane's
            // first operands (and its table words) are requested before this plane's requantisation starts and carried into the next round of the
            // loop.  Plane by plane (the form below) every plane began with a cold LDS round trip, and two waves per SIMD did not cover it: 7.6 k
            // cycles per tile for ~2 k of matrix and ~2 k of vector work per wave (512-channel block).  The loop stays rolled (unrolled, hipcc hoists
            // every plane's addresses and tables to the top: 200+ bytes of scratch).  The rare lo parts (lom != 0) are added behind the plane's
            // stream from re-read operands.
            constexpr int DW_W = 6;
            static_assert(12 % DW_W == 0, "the window slots of a plane's first operands are those of the next plane's");
            auto opnd_at = [&](int pofs, int j) __attribute__((always_inline)) {
                return *reinterpret_cast<const i4v *>(ring + tapoff[j / 4][j % 4] + pofs);
            };
            i4v b[DW_W];
            int cg = (wave * CPW) % C16;
            uint2 ab = dwa_l[cg * 64 + lane];
#pragma unroll
            for (int j = 0; j < DW_W; ++j) b[j] = opnd_at(cg * PP, j);
#pragma unroll 1
            for (int ci = 0; ci < CPW; ++ci) {
                const int pofs = cg * PP;
                i4v acc[4];
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    i4v Ah;
                    const unsigned rh = __builtin_amdgcn_perm(ab.x, ab.x, 0x01010101u * (unsigned)ks);
#pragma unroll
                    for (int d = 0; d < 4; ++d) Ah[d] = (int)(rh & dmask[d]);
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int j = ks * 4 + f;
                        if (ks == 0) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah, b[j % DW_W], i4v{0, 0, 0, 0}, 0, 0, 0);
                        else acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah, b[j % DW_W], acc[f], 0, 0, 0);
                        if (j + DW_W < 12) b[j % DW_W] = opnd_at(pofs, j + DW_W);
                    }
                }
                // the next plane's head (the last round re-requests its own: nobody reads them)
                const int cgn = (wave * CPW + min(ci + 1, CPW - 1)) % C16;
                const uint2 abn = dwa_l[cgn * 64 + lane];
                long long Cq[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) Cq[r] = dwq_l[cg * 16 + 4 * fq + r];
                const unsigned lom = (unsigned)__builtin_amdgcn_readfirstlane((int)(ab.y >> 24));
                if (lom) {                                           // (few planes: only a tensor's extreme weights overflow int8)
#pragma unroll
                    for (int ks = 0; ks < 3; ++ks) {
                        if (lom & (1u << ks)) {
                            const unsigned rl = __builtin_amdgcn_perm(ab.y, ab.y, 0x01010101u * (unsigned)ks);
                            i4v Al;
#pragma unroll
                            for (int d = 0; d < 4; ++d) Al[d] = (int)(rl & dmask[d]);
#pragma unroll
                            for (int f = 0; f < 4; ++f)
                                acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Al, *reinterpret_cast<const i4v *>(ring + tapoff[ks][f] + pofs), acc[f], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < DW_W; ++j) b[j] = opnd_at(cgn * PP, j);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    unsigned packed = q_requant_pack4<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Md, Cq[0], Cq[1], Cq[2], Cq[3], shd, lod, hid);
                    packed ^= 0x80808080u;
                    if (ROWSUM) rs[f] = sdot4((int)packed, 0x01010101, rs[f]);
                    *reinterpret_cast<unsigned *>(opnd + ((size_t)cg * QTT + 16 * (fb + f) + fr) * 16 + 4 * fq) = packed;
                }
                cg = cgn;
                ab = abn;
            }
        } else if constexpr (KEEP) {
#pragma unroll
            for (int ci = 0; ci < CPW; ++ci) plane(wave * CPW + ci, ci);
        } else {
#pragma unroll 1
            for (int ci = 0; ci < CPW; ++ci) plane(wave * CPW + ci, ci);
        }
        if (ROWSUM) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                int v = rs[f];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (fq == 0) atomicAdd(&rowsum[rsb * QTT + 16 * (fb + f) + fr], v);
            }
        }
    };
    // The rows the next tile adds go from HBM straight into the ring (LDS-DMA, global_load_lds_dwordx4: no registers -- a register-staged
    // prefetch held 32 per lane through the matrix stage and pushed the 512-channel block into scratch, where one reload after the requests
    // waits for all of them -- no ds_write pass, no per-lane ring addresses).  Rows lo .. lie one after the other in HBM and, but for the wrap,
    // in the ring (slot = row % NR, pitch RB): byte idx of the request goes to ring byte (slot(lo) * RB + idx) mod ring size.  One
    // wave-instruction moves 1 KB lane-linearly to a wave-uniform LDS base; a piece that straddles the end of the ring is issued twice with
    // complementary lane masks (the second base lies below the ring: the operand tile is there, the addresses the active lanes form are not).
    // The statement is assembly, not the builtin: hipcc would make every later ds_read that may alias an LDS-DMA in flight wait for it
    // (s_waitcnt vmcnt(0) at the head of the depthwise stage), and the rows requested BEFORE that stage -- into slots the current tile no
    // longer reads, see the tile loop -- are to stay in flight through it.  The waits are this kernel's own, at barrier B.  (M0, the
    // destination base, is the compiler's register: saved and restored in the statement that uses it.)
    auto glds16 = [&](const uint8_t *g, const uint8_t *l) {
        unsigned keep;
        const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)l;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    };
    auto pf_issue = [&](int lo, int hi) {                            // rows [lo, hi]
        if (hi < lo) return;
        const unsigned nb = (unsigned)(hi - lo + 1) * (unsigned)RB;
        const uint8_t *src = P.in + (size_t)lo * RB;
        const unsigned base = (unsigned)(lo - (int)__umulhi((unsigned)lo, P.nr_magic) * P.NR) * (unsigned)RB;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const unsigned cb = (unsigned)(i * NW + wave) * 1024u;  // (wave-uniform) this wave's piece of the request
            if (cb < nb) {
                unsigned a = base + cb;
                a = a >= ring_bytes ? a - ring_bytes : a;
                const unsigned idx = cb + (unsigned)lane * 16u;
                const uint8_t *g = src + idx;
                if (a + 1024u <= ring_bytes) {
                    if (idx < nb) glds16(g, ring + a);
                } else {
                    const bool first = a + (unsigned)lane * 16u < ring_bytes;
                    if (idx < nb && first) glds16(g, ring + a);
                    if (idx < nb && !first) glds16(g, ring + (int)(a - ring_bytes));
                }
            }
        }
    };
    // ---- pointwise stage of this wave's channels / fragments
    auto matrix = [&](int q0, int q1, int gbuf, int rsb) __attribute__((always_inline)) {
        const int nf = (q1 - q0) / 16 + 1;
        // A lone ds_read -> s_waitcnt -> 4 MFMAs per K slice leaves the matrix pipe idle for the LDS latency eight times per fragment
        // (2.2 k cycles per fragment measured, 0.5 k of MFMA): all K slices of a fragment are requested before its first MFMA, and the
        // next fragment's as soon as this one's MFMAs are issued, so they land during the epilogue.
        constexpr int KB = KC < 4 ? KC : 4;                             // K slices requested at a time (all eight of a 512-channel layer cost 32 registers: the prefetch
                                                                        // registers then spill, and a spilled HBM load is waited for on the spot)
        i4v b[KB];
        if (wp < nf) {
            const uint8_t *bp = opnd + ((size_t)fq * QTT + 16 * wp + fr) * 16;
#pragma unroll
            for (int kc = 0; kc < KB; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bp + (size_t)kc * 4 * QTT * 16);
        }
        for (int f = wp; f < nf; f += WP) {
            i4v acc[MW];
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                if constexpr (FOLDP) acc[m] = i4v{0, 0, 0, 0};
                else acc[m] = *reinterpret_cast<const i4v *>(cbl + 64 * mg + 16 * fq + 4 * (m0 + m));
            }
            const uint8_t *bp = opnd + ((size_t)fq * QTT + 16 * f + fr) * 16;
#pragma unroll
            for (int k0 = 0; k0 < KC; k0 += KB) {
#pragma unroll
                for (int kc = 0; kc < KB; ++kc) {
#pragma unroll
                    for (int m = 0; m < MW; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wr[m][k0 + kc], b[kc], acc[m], 0, 0, 0);
                    if (HL && ((wlm >> (k0 + kc)) & 1u)) {          // (the lo part of this k slice: zero for most of them)
                        if (k0 + kc == kcl) {
#pragma unroll
                            for (int m = 0; m < MW; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wl1[m], b[kc], acc[m], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int m = 0; m < MW; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(P.w2[((size_t)(wm * MW + m) * KC + (k0 + kc)) * 64 + lane], b[kc], acc[m], 0, 0, 0);
                        }
                    }
                }
                if (k0 + KB < KC) {
#pragma unroll
                    for (int kc = 0; kc < KB; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bp + (size_t)(k0 + KB + kc) * 4 * QTT * 16);
                }
            }
            if (f + WP < nf) {
                const uint8_t *bn = opnd + ((size_t)fq * QTT + 16 * (f + WP) + fr) * 16;
#pragma unroll
                for (int kc = 0; kc < KB; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bn + (size_t)kc * 4 * QTT * 16);
            }
            const int rsv = ROWSUM ? rowsum[rsb * QTT + 16 * f + fr] * P.zwc : 0;
            const unsigned po = (unsigned)pinfo[gbuf * QTT + 16 * f + fr][3];
            unsigned o[MW];
#pragma unroll
            for (int m = 0; m < MW; ++m) {
                if constexpr (FOLDP)
                    o[m] = 0x80808080u ^ q_requant_pack4<SAT>(acc[m][0], acc[m][1], acc[m][2], acc[m][3], Mp, CP[m][0], CP[m][1], CP[m][2], CP[m][3], shp, lop, hip_);
                else
                    o[m] = 0x80808080u ^ q_requant_pack4<SAT>(acc[m][0] + rsv, acc[m][1] + rsv, acc[m][2] + rsv, acc[m][3] + rsv, Mp, Cp, Cp, Cp, Cp, shp, lop, hip_);
            }
            if (po != 0xffffffffu) {
                uint8_t *dst = P.out + po + (size_t)(4 * mg + fq) * ((P.wo + 2) * 16) + 4 * m0;
                if constexpr (MW == 4) *reinterpret_cast<u4v *>(dst) = u4v{o[0], o[1], o[2], o[3]};
                else *reinterpret_cast<uint2 *>(dst) = make_uint2(o[0], o[1]);
            }
        }
    };
    int n, q0, q1, ga, gb;
    tile_rows(t_begin, n, q0, q1, ga, gb);
    load_rows_sync(ga, gb);
    if (tid < QTT) geometry(n, q0, q1, ga, t_begin & 1);
    int loaded_hi = gb;
    __syncthreads();
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tprev = P.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define Q_STAMP(k) do { if (P.dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st[k] += now_ - tprev; tprev = now_; } } while (0)

    for (int t = t_begin; t < t_end; ++t) {
        // The rows tile t + 1 adds.  Those whose ring slot holds a row BELOW this tile's first one (the ring is P.NR >= the largest span of one
        // tile; the launcher adds rows where LDS allows) are requested now and fly through both stages; the others wait for barrier A, when
        // the depthwise stage has read the rows they replace.  One tile's worth of requests in flight during the matrix stage only left HBM
        // idle half of the time.  (Measured: neutral -- with the stores or the requests removed the stride-2 blocks run in 124 / 132 us against 208 with
        // both: they move their 829 MB at the 4.0-4.2 TB/s every streaming kernel of this network reaches, extra ring rows (DD_Q_EXTRA) change nothing.)
        int lo = 0, hi = -1, e_hi = -1;
        int n2 = 0, q02 = 0, q12 = 0, ga2 = 0, gb2 = 0;
        if (t + 1 < t_end) {
            tile_rows(t + 1, n2, q02, q12, ga2, gb2);
            lo = max(loaded_hi + 1, ga2);
            hi = gb2;
            e_hi = min(hi, ga + P.NR - 1);
            loaded_hi = max(loaded_hi, gb2);
        }
        pf_issue(lo, e_hi);
        const int cur = t & 1;
        dw_planes(cur, cur);
        Q_STAMP(0);
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's part of the operand tile and of the row sums is written
        __builtin_amdgcn_s_barrier();                            // A: operand tile and row sums are complete; the ring is free (bare: requests stay in flight)
        Q_STAMP(1);
        pf_issue(max(lo, e_hi + 1), hi);                         // the rest of the next tile's rows
        if (tid < QTT) {
            if (ROWSUM) rowsum[(cur ^ 1) * QTT + tid] = 0;
            if (t + 1 < t_end) geometry(n2, q02, q12, ga2, cur ^ 1);
        }
        matrix(q0, q1, cur, cur);
        Q_STAMP(2);
        // This wave's pieces of the ring have landed: they were requested BEFORE the matrix stage's output stores and the vector-memory counter
        // retires in issue order, so the wait leaves as many operations outstanding as the wave stored fragments -- vmcnt(0) here also waited
        // for the stores' write acknowledgements (0.8 k cycles per tile in the 32-channel block).  The barrier is the bare instruction:
        // __syncthreads() would drain the counter again.
        {
            const int nf_t = (q1 - q0) / 16 + 1, ns = nf_t > wp ? (nf_t - wp + WP - 1) / WP : 0;     // (wave-uniform) fragments this wave stored
            if (ns >= 4) __builtin_amdgcn_s_waitcnt(0x0f74);
            else if (ns == 3) __builtin_amdgcn_s_waitcnt(0x0f73);
            else if (ns == 2) __builtin_amdgcn_s_waitcnt(0x0f72);
            else if (ns == 1) __builtin_amdgcn_s_waitcnt(0x0f71);
            else __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        Q_STAMP(3);
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's LDS traffic is done
        __builtin_amdgcn_s_barrier();                            // B: the operand tile is free, the ring holds the next tile's rows (every wave waited for its pieces)
        Q_STAMP(4);
        n = n2; q0 = q02; q1 = q12; ga = ga2;
    }
#undef Q_STAMP
    if (P.dbg && lane == 0) for (int k = 0; k < 5; ++k) P.dbg[((size_t)blockIdx.x * NW + wave) * 8 + k] = st[k];
}



// Ring rows and prefetch depth of q_dwpw_k for this geometry (tiles of one frame and the step into the next frame).
void dwpw_plan(int H, int W, int ho, int wo, int stride, int off_y, int cin, int nt, int qt, int *NR, int *lpt) {
    const int hw = ho * wo, tpf = dd_ceil_div(hw, qt), RB = (W + 2) * cin;
    int span = 0, hi = 0, mx = 0;
    for (int n = 0; n < 2; ++n)
        for (int k = 0; k < tpf; ++k) {
            const int q0 = k * qt, q1 = std::min(q0 + qt, hw) - 1;
            const int ga = n * (H + 2) + (q0 / wo) * stride + off_y, gb = n * (H + 2) + (q1 / wo) * stride + off_y + 2;
            span = std::max(span, gb - ga + 1);
            if (n || k) { const int lo = std::max(hi + 1, ga); mx = std::max(mx, gb - lo + 1); }
            hi = std::max(hi, gb);
        }
    *NR = span;
    *lpt = (int)(((long long)mx * RB + (long long)nt * 16 - 1) / ((long long)nt * 16));
}

template <int CIN, int COUT, int WP, int STRIDE, int LPT, int EXTRA = 0, int MW = 4, int QTT = QT>
int launch_q_dwpw(hipStream_t s, QDwpwP &P, int nimg, int device, bool *ok) {
    constexpr int NW = (COUT / (16 * MW)) * WP, NT = NW * 64, CINP = (CIN + 63) / 64 * 64;
    int lpt = 0;
    P.tiles_per_frame = dd_ceil_div(P.hw, QTT);
    dwpw_plan(P.H, P.W, P.ho, P.wo, STRIDE, P.off_y, CIN, NT, QTT, &P.NR, &lpt);
    static const int extra_env = getenv("DD_Q_EXTRA") ? atoi(getenv("DD_Q_EXTRA")) : -1;
    P.NR += extra_env >= 0 ? extra_env : EXTRA;                   // ring rows beyond one tile's span: the next tile's rows can be requested a stage earlier
    const int RB = (P.W + 2) * CIN;
    constexpr bool KEEP = ((CIN / 16) * (QTT / 64)) / NW <= 2;        // (as in the kernel) else the depthwise tables take LDS
    const size_t lds = (size_t)P.NR * RB + (size_t)QTT * CINP + (size_t)2 * QTT * sizeof(int) + 2 * QTT * 16 + COUT * sizeof(int) + (KEEP ? 0 : (CIN / 16) * 64 * 8 + CIN * 8);
    P.wo_magic = (unsigned)((1ull << 32) / (unsigned)P.wo) + 1u;
    P.nr_magic = (unsigned)((1ull << 32) / (unsigned)P.NR) + 1u;
    P.tpf_magic = (unsigned)((1ull << 32) / (unsigned)P.tiles_per_frame) + 1u;
    const long long lim24 = 1ll << 23;                              // the kernel's 24-bit multiplies
    *ok = (long long)nimg * P.tiles_per_frame * P.tiles_per_frame < (1ll << 32) && (long long)P.hw * P.wo < (1ll << 32) &&
          (long long)(nimg + 1) * (P.H + 2) * P.NR < (1ll << 32) && lpt <= LPT && lds <= 160 * 1024 && (long long)LPT * NT * 16 + (long long)P.NR * RB < (1ll << 32) &&
          RB < lim24 && (long long)P.c16_out * (P.wo + 2) * 16 < lim24 && (long long)(nimg + 1) * (P.ho + 2) < lim24 && P.hw < lim24;
    if (!*ok) return DD_OK;
    constexpr bool CAN_HL = CIN <= 128;                                // split pointwise filters are packed for the narrow blocks only
    const bool hl = P.w2 != nullptr;
    if (hl && !CAN_HL) { *ok = false; return DD_OK; }
    const bool rsum = P.zwc != 0 && !hl;
    // both clamps are the byte range: saturating packs (1); both shifts <= 8 as well: the packed 16-bit shift (2)
    const int sat = P.Rd.lo == 0 && P.Rd.hi == 255 && P.Rp.lo == 0 && P.Rp.hi == 255 ? (P.Rd.e <= 8 && P.Rp.e <= 8 ? 2 : 1) : 0;
#define DD_QK(R_, S_, H_) q_dwpw_k<CIN, COUT, WP, STRIDE, LPT, R_, MW, S_, (H_) && CAN_HL, QTT>
#define DD_QS(R_, H_) (sat == 2 ? &DD_QK(R_, 2, H_) : sat == 1 ? &DD_QK(R_, 1, H_) : &DD_QK(R_, 0, H_))
    void (*kern)(const QDwpwP, const int, const int) = hl ? DD_QS(false, true) : rsum ? DD_QS(true, false) : DD_QS(false, false);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        for (const void *f : {reinterpret_cast<const void *>(&DD_QK(false, 0, true)), reinterpret_cast<const void *>(&DD_QK(false, 1, true)), reinterpret_cast<const void *>(&DD_QK(false, 2, true)),
                              reinterpret_cast<const void *>(&DD_QK(true, 0, false)), reinterpret_cast<const void *>(&DD_QK(true, 1, false)), reinterpret_cast<const void *>(&DD_QK(true, 2, false)),
                              reinterpret_cast<const void *>(&DD_QK(false, 0, false)), reinterpret_cast<const void *>(&DD_QK(false, 1, false)), reinterpret_cast<const void *>(&DD_QK(false, 2, false))})
            DD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return DD_OK;
    });
#undef DD_QS
#undef DD_QK
    if (rc != DD_OK) return rc;
    static std::atomic<int> per_cu_cache[64];
    int per_cu = per_cu_cache[device & 63].load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int nb = 0;
        DD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kern), NT, lds));
        per_cu = std::max(1, std::min(8, nb));
        per_cu_cache[device & 63].store(per_cu, std::memory_order_relaxed);
    }
    static const int per_cu_env = getenv("DD_Q_PER_CU") ? atoi(getenv("DD_Q_PER_CU")) : 0;     // experiment: fewer blocks per CU than fit
    if (per_cu_env > 0) per_cu = std::min(per_cu, per_cu_env);
    const int n_tiles = nimg * P.tiles_per_frame;
    const int blocks = std::min(n_tiles, per_cu * 256);
    const int tpb = dd_ceil_div(n_tiles, blocks);
    const dim3 grid((unsigned)dd_ceil_div(n_tiles, tpb));
    static const bool stamps = getenv("DD_Q_STAMPS") && atoi(getenv("DD_Q_STAMPS")) != 0;
    const size_t n_st = (size_t)grid.x * NW * 8;
    if (stamps) { DD_HIP(hipMalloc(&P.dbg, n_st * 8)); DD_HIP(hipMemsetAsync(P.dbg, 0, n_st * 8, s)); }
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, P, n_tiles, tpb);
    DD_LAUNCH_CHECK();
    if (stamps) {                                                   // diagnostic: where the waves of this launch spent their cycles
        std::vector<unsigned long long> h(n_st);
        DD_HIP(hipStreamSynchronize(s));
        DD_HIP(hipMemcpy(h.data(), P.dbg, n_st * 8, hipMemcpyDeviceToHost));
        DD_HIP(hipFree(P.dbg));
        double sum[5] = {0, 0, 0, 0, 0};
        for (size_t w = 0; w < n_st / 8; ++w) for (int k = 0; k < 5; ++k) sum[k] += (double)h[w * 8 + k];
        const double nw = (double)(n_st / 8) * tpb;
        fprintf(stderr, "q_dwpw_k<%d,%d,%d> %d blocks/CU %d tiles/block: cycles per wave and tile: depthwise %.0f  barrier A %.0f  matrix stage %.0f  ring write %.0f  barrier B %.0f\n",
                CIN, COUT, STRIDE, per_cu, tpb, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw);
    }
    return DD_OK;
}

// ------------------------------------------------------------------------------------------------ pointwise layer, filter in registers
// 1x1 stride-1 layers with 512 or 1024 input channels after block 11 (b12 / b13 pointwise, the first extra layer, the two big class
// predictors: 245 of the tail's 285 GFLOP).  q_conv_k reads both operands of every MFMA from L2 in fragment shape -- 8 KB per 16 MFMAs and
// wave, the L1 path's 64 B per cycle and CU bound it at a quarter of the matrix rate (0.6-0.9 POP/s measured).  Here, as in q_dwpw_k's
// matrix stage (3.2 POP/s while it runs), a wave keeps the A fragments of its MW * 16 output channels for all of K in registers (128 VGPRs:
// MW = 4 at K = 512, MW = 2 at K = 1024) and the pixels come through LDS: tiles of 64 pixels as the B-fragment image [plane][pixel][16],
// filled by LDS-DMA with per-lane source addresses (lane = pixel, one wave instruction = one plane of the tile = 1 KB; the bordered layout
// makes a pixel's 16 channels of a plane one aligned 16-byte piece), two tiles deep, ONE barrier per tile.  Row sums (zw != 128) come off
// the matrix pipe as well: an A fragment of ones accumulates sum_k b into every row of a fifth accumulator, whose column is the lane's pixel.
struct QPwsP {
    const uint8_t *in; int H, W, c16_in;          // Q16 source; the layer's output has the same H x W
    int hw, m;                                    // H * W, images * H * W
    const i4v *w;                                 // [n_mfrag][K / 64][64 lanes]
    const int *cbias;                             // [16 * n_mfrag]
    const long long *cq;                          // [16 * n_mfrag]: cbias * M + C (Q16 epilogue)
    int n_mfrag, frags_per_group;                 // fragments in all; per block (= waves * MW)
    uint8_t *out; int c16_out;                    // QEPI_Q16
    long long img_bytes_out; int row_bytes, base_off, cout_store;   // QEPI_ROWS
    int zwc, tiles_per_block;
    unsigned hw_magic, w_magic;
    QReq R;
    // QEPI_ROWS, two predictors on one feature map in one launch (the SSD's class and box layers read the same pixels): fragments
    // [n_frag_a, n_mfrag) belong to the second one, with its own requantisation, weight zero point and destination rows
    int n_frag_a;                                 // 0: one layer
    uint8_t *out_b; long long img_bytes_b; int row_bytes_b, base_off_b, cout_store_b, zwc_b;
    QReq Rb;
};

template <int K, int MW, int EPI, bool ROWSUM, int SAT>
__global__ __launch_bounds__(512, 2) void q_pws_k(const QPwsP P, const int n_tiles) {
    constexpr int KC = K / 64, C16 = K / 16;
    static_assert(MW * KC * 4 <= 128, "the wave's filter is 128 registers at most");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *opnd = smem;                                            // [2][C16][QT][16]
    unsigned *psrc = reinterpret_cast<unsigned *>(smem + 2 * QT * K);   // [4][QT]: the pixel's offset in the source (plane 0)
    unsigned *pdst = psrc + 4 * QT;                                  // [4][QT]: its offset in the destination, ~0 past the end
    int *rowsum = reinterpret_cast<int *>(pdst + 4 * QT);            // [3][QT]: sum over k of the pixel's operand bytes (ROWSUM)
    unsigned *pdstb = reinterpret_cast<unsigned *>(rowsum + 3 * QT); // [4][QT]: the second predictor's destination offsets
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), NW = blockDim.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int frag0 = blockIdx.y * P.frags_per_group + wave * MW;     // this wave's first fragment
    const unsigned PP = (unsigned)(P.W + 2) * 16u;

    i4v Wr[MW][KC];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) Wr[m][kc] = frag0 + m < P.n_mfrag ? P.w[((size_t)(frag0 + m) * KC + kc) * 64 + lane] : i4v{0, 0, 0, 0};
    const int M = P.R.M, sh = P.R.e - 1, lo = P.R.lo, hi = P.R.hi;
    // Q16: fragment 4 mg + m holds channels 64 mg + 16 g + 4 m + r (MW = 2: the wave's pair is half of such a group)
    const int mg = frag0 / 4, m0 = frag0 % 4;
    // per-channel 64-bit addends (Q16: cbias * M + C of the ReLU form, channel order of the fragment rows; ROWS: cbias * M + 2^30, natural order)
    long long CQ[MW][4];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = EPI == QEPI_Q16 ? 64 * mg + 16 * fq + 4 * (m0 + m) + r : 16 * (frag0 + m) + 4 * fq + r;
            CQ[m][r] = frag0 + m < P.n_mfrag ? P.cq[ch] : 0ll;
        }
    // ROWS: which predictor fragment m belongs to (wave-uniform), its first channel, its constants
    const int nfa = EPI == QEPI_ROWS && P.n_frag_a ? P.n_frag_a : P.n_mfrag;
    bool hb[MW]; int chm[MW], Mm[MW], em[MW], k1m[MW], zwm[MW];
    int n_store_m = 0;                                               // fragments of this wave that store anything
#pragma unroll
    for (int m = 0; m < MW; ++m) {
        hb[m] = frag0 + m >= nfa;
        chm[m] = 16 * (frag0 + m - (hb[m] ? nfa : 0));
        const QReq &Rm = hb[m] ? P.Rb : P.R;
        Mm[m] = Rm.M; em[m] = Rm.e; k1m[m] = (1 << (Rm.e - 1)) + (Rm.zo << Rm.e); zwm[m] = hb[m] ? P.zwc_b : P.zwc;
        n_store_m += frag0 + m < P.n_mfrag && chm[m] < (hb[m] ? P.cout_store_b : P.cout_store) ? 1 : 0;
    }

    const int t_begin = blockIdx.x * P.tiles_per_block, t_end = min(n_tiles, t_begin + P.tiles_per_block);
    if (t_begin >= t_end) return;
    auto geometry = [&](int t) {                                     // threads 0 .. QT - 1
        const int q = t * QT + tid, qc = min(q, P.m - 1);
        const int n = (int)__umulhi((unsigned)qc, P.hw_magic), r = qc - n * P.hw;
        const int y = (int)__umulhi((unsigned)r, P.w_magic), x = r - y * P.W;
        const unsigned row = (unsigned)(n * (P.H + 2) + y + 1);
        psrc[(t & 3) * QT + tid] = row * (unsigned)P.c16_in * PP + (unsigned)(x + 1) * 16u;
        unsigned d;
        if constexpr (EPI == QEPI_Q16) d = row * (unsigned)P.c16_out * PP + (unsigned)(x + 1) * 16u;
        else d = (unsigned)((long long)n * P.img_bytes_out) + (unsigned)P.base_off + (unsigned)r * (unsigned)P.row_bytes;
        pdst[(t & 3) * QT + tid] = q < P.m ? d : 0xffffffffu;
        if (EPI == QEPI_ROWS && P.n_frag_a) pdstb[(t & 3) * QT + tid] = q < P.m ? (unsigned)((long long)n * P.img_bytes_b) + (unsigned)P.base_off_b + (unsigned)r * (unsigned)P.row_bytes_b : 0xffffffffu;
    };
    auto fill = [&](int t) {                                         // tile t's bytes on their way: plane c of the tile = one wave instruction
        const uint8_t *src = P.in + psrc[(t & 3) * QT + lane];
        uint8_t *dst = opnd + (size_t)(t & 1) * QT * K;
        for (int c = wave; c < C16; c += NW) {
            unsigned keep;
            const unsigned l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)(dst + (size_t)c * QT * 16));   // (wave-uniform: say so)
            const uint8_t *g = src + (size_t)c * PP;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(l) : "memory");
        }
    };

    // Row sums: every wave adds up the planes IT requested (64 pixels x 16 bytes each: lane = pixel) once its own requests have landed --
    // no barrier needed for that -- and adds the partial sums into the tile's slot of a three-slot ring (zeroed two tiles ahead, read one
    // barrier later).  An A fragment of ones on the matrix pipe gave the same numbers for 25 % (K = 512) / 50 % (K = 1024) more MFMAs.
    auto own_rowsum = [&](int t, int slot) {
        const uint8_t *ob = opnd + (size_t)(t & 1) * QT * K;
        int a = 0;
        for (int c = wave; c < C16; c += NW) {
            const i4v v = *reinterpret_cast<const i4v *>(ob + ((size_t)c * QT + lane) * 16);
#pragma unroll
            for (int d = 0; d < 4; ++d) a = sdot4(v[d], 0x01010101, a);
        }
        atomicAdd(&rowsum[slot * QT + lane], a);
    };
    auto wait_landed = [&](int ns) {                                 // this wave's requests have landed: they precede its last ns stores, the counter retires in order
        switch (ns) {
#define DD_VMCNT(N_) case N_: __builtin_amdgcn_s_waitcnt(0x0f70 | ((N_) & 15) | (((N_) >> 4) << 14)); break;
            DD_VMCNT(1) DD_VMCNT(2) DD_VMCNT(3) DD_VMCNT(4) DD_VMCNT(5) DD_VMCNT(6) DD_VMCNT(7) DD_VMCNT(8)
            DD_VMCNT(9) DD_VMCNT(10) DD_VMCNT(11) DD_VMCNT(12) DD_VMCNT(13) DD_VMCNT(14) DD_VMCNT(15) DD_VMCNT(16)
#undef DD_VMCNT
            default: __builtin_amdgcn_s_waitcnt(0x0f70);
        }
    };
    if (tid < QT) { geometry(t_begin); if (t_begin + 1 < t_end) geometry(t_begin + 1); }
    for (int i = tid; i < 3 * QT; i += blockDim.x) rowsum[i] = 0;
    __syncthreads();
    fill(t_begin);
    wait_landed(0);
    int slot = 0;                                                    // (t - t_begin) % 3
    if constexpr (ROWSUM) own_rowsum(t_begin, 0);
    int ns_prev = 0;                                                 // vector stores this wave issued since its last fill
    for (int t = t_begin; t < t_end; ++t) {
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // (and this wave's LDS reads of tile t - 1, its geometry writes)
        __builtin_amdgcn_s_barrier();                                // every wave's planes of tile t are in; tile t - 1's buffer is free
        if (t + 1 < t_end) fill(t + 1);
        if (t + 2 < t_end && tid < QT) geometry(t + 2);
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        if (ROWSUM && tid < QT) rowsum[slot2 * QT + tid] = 0;
        const uint8_t *ob = opnd + (size_t)(t & 1) * QT * K;
        const int nf = (min((t + 1) * QT, P.m) - t * QT + 15) / 16;
        constexpr int KB = KC < 4 ? KC : 4;
        i4v b[KB];
        {
            const uint8_t *bp = ob + ((size_t)fq * QT + fr) * 16;
#pragma unroll
            for (int kc = 0; kc < KB; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bp + (size_t)kc * 4 * QT * 16);
        }
        ns_prev = 0;
        for (int f = 0; f < nf; ++f) {
            i4v acc[MW];
#pragma unroll
            for (int m = 0; m < MW; ++m) acc[m] = i4v{0, 0, 0, 0};
            const uint8_t *bp = ob + ((size_t)fq * QT + 16 * f + fr) * 16;
#pragma unroll
            for (int k0 = 0; k0 < KC; k0 += KB) {
#pragma unroll
                for (int kc = 0; kc < KB; ++kc) {
#pragma unroll
                    for (int m = 0; m < MW; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wr[m][k0 + kc], b[kc], acc[m], 0, 0, 0);
                }
                if (k0 + KB < KC) {
#pragma unroll
                    for (int kc = 0; kc < KB; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bp + (size_t)(k0 + KB + kc) * 4 * QT * 16);
                }
            }
            if (f + 1 < nf) {
                const uint8_t *bn = ob + ((size_t)fq * QT + 16 * (f + 1) + fr) * 16;
#pragma unroll
                for (int kc = 0; kc < KB; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bn + (size_t)kc * 4 * QT * 16);
            }
            const int rsv = ROWSUM ? rowsum[slot * QT + 16 * f + fr] * P.zwc : 0;
            const unsigned po = pdst[(t & 3) * QT + 16 * f + fr];
            if constexpr (EPI == QEPI_Q16) {
                unsigned o[MW];
#pragma unroll
                for (int m = 0; m < MW; ++m)
                    o[m] = 0x80808080u ^ q_requant_pack4<SAT>(acc[m][0] + rsv, acc[m][1] + rsv, acc[m][2] + rsv, acc[m][3] + rsv, M, CQ[m][0], CQ[m][1], CQ[m][2], CQ[m][3], sh, lo, hi);
                if (po != 0xffffffffu && frag0 < P.n_mfrag) {
                    uint8_t *dst = P.out + po + (size_t)(4 * mg + fq) * PP + 4 * m0;
                    if constexpr (MW == 4) *reinterpret_cast<u4v *>(dst) = u4v{o[0], o[1], o[2], o[3]};
                    else *reinterpret_cast<uint2 *>(dst) = make_uint2(o[0], o[1]);
                }
                if (frag0 < P.n_mfrag) ++ns_prev;
            } else {
                const int rs0 = ROWSUM ? rowsum[slot * QT + 16 * f + fr] : 0;
                const unsigned pob = P.n_frag_a ? pdstb[(t & 3) * QT + 16 * f + fr] : 0xffffffffu;
#pragma unroll
                for (int m = 0; m < MW; ++m) {
                    const int ch = chm[m] + 4 * fq;
                    if (frag0 + m >= P.n_mfrag || ch >= (hb[m] ? P.cout_store_b : P.cout_store)) continue;
                    const int rv = rs0 * zwm[m];
                    const unsigned wv = q_requant_linear_pack4(acc[m][0] + rv, acc[m][1] + rv, acc[m][2] + rv, acc[m][3] + rv, Mm[m], CQ[m][0], CQ[m][1], CQ[m][2], CQ[m][3], em[m], k1m[m]);
                    const unsigned pm = hb[m] ? pob : po;
                    if (pm != 0xffffffffu) *reinterpret_cast<unsigned *>((hb[m] ? P.out_b : P.out) + pm + ch) = wv;
                }
                ns_prev += n_store_m;
            }
        }
        if (t + 1 < t_end) {
            wait_landed(ns_prev);                                    // tile t + 1: this wave's planes are in
            if constexpr (ROWSUM) own_rowsum(t + 1, slot1);
        }
        slot = slot1;
    }
}

// Shapes q_pws_k takes; *ok = false leaves the layer to q_conv_k.
template <int K, int MW>
int launch_q_pws(hipStream_t s, QPwsP &P, int nimg, int device, bool *ok) {
    const int n_tiles = dd_ceil_div(P.m, QT);
    const int waves = std::min(8, dd_ceil_div(P.n_mfrag, MW));         // per block; the fragments split evenly over the groups
    const int groups = dd_ceil_div(P.n_mfrag, waves * MW);
    const int wpb = dd_ceil_div(dd_ceil_div(P.n_mfrag, groups), MW);  // waves per block
    P.frags_per_group = wpb * MW;
    const size_t lds = (size_t)2 * QT * K + 15 * QT * sizeof(unsigned);
    const int blocks_x = std::max(1, std::min(n_tiles, 256 / groups));
    P.tiles_per_block = dd_ceil_div(n_tiles, blocks_x);
    P.hw_magic = (unsigned)((1ull << 32) / (unsigned)P.hw) + 1u;
    P.w_magic = (unsigned)((1ull << 32) / (unsigned)P.W) + 1u;
    const long long in_bytes = (long long)nimg * (P.H + 2) * P.c16_in * (P.W + 2) * 16;
    const long long out_bytes = P.img_bytes_out ? (long long)nimg * P.img_bytes_out : (long long)nimg * (P.H + 2) * P.c16_out * (P.W + 2) * 16;
    *ok = in_bytes < (1ll << 32) && out_bytes < (1ll << 32) && (long long)P.m * P.hw < (1ll << 32) && (long long)P.hw * P.W < (1ll << 32) && P.zwc >= -128 && P.zwc <= 128 &&
          lds <= 160 * 1024 && wpb >= 1 && wpb <= 8;
    if (!*ok) return DD_OK;
    const bool rsum = P.zwc != 0 || (P.n_frag_a && P.zwc_b != 0);
    const bool rows = P.img_bytes_out != 0;
    if (P.n_frag_a && !(P.Rb.linear && P.Rb.e >= 1 && P.Rb.e <= 30 && P.Rb.lo == 0 && P.Rb.hi == 255 && std::abs(P.Rb.zo) < 256 && P.zwc_b >= -128 && P.zwc_b <= 128 &&
                        (long long)nimg * P.img_bytes_b < (1ll << 32))) { *ok = false; return DD_OK; }
    if (rows && !(P.R.linear && P.R.e >= 1 && P.R.e <= 30 && P.R.lo == 0 && P.R.hi == 255 && std::abs(P.R.zo) < 256)) { *ok = false; return DD_OK; }
    const int sat = !rows && P.R.lo == 0 && P.R.hi == 255 ? (P.R.e <= 8 ? 2 : 1) : 0;
    void (*kern)(const QPwsP, const int) = nullptr;
#define DD_PW(E_, R_, S_) q_pws_k<K, MW, E_, R_, S_>
    if (rows) kern = rsum ? &DD_PW(QEPI_ROWS, true, 0) : &DD_PW(QEPI_ROWS, false, 0);
    else if constexpr (MW != 3) {
        if (sat == 2) kern = rsum ? &DD_PW(QEPI_Q16, true, 2) : &DD_PW(QEPI_Q16, false, 2);
        else if (sat == 1) kern = rsum ? &DD_PW(QEPI_Q16, true, 1) : &DD_PW(QEPI_Q16, false, 1);
        else kern = rsum ? &DD_PW(QEPI_Q16, true, 0) : &DD_PW(QEPI_Q16, false, 0);
    }
    if (!kern) { *ok = false; return DD_OK; }
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&DD_PW(QEPI_ROWS, true, 0)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&DD_PW(QEPI_ROWS, false, 0)), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        if constexpr (MW != 3) {
            for (const void *f : {reinterpret_cast<const void *>(&DD_PW(QEPI_Q16, true, 2)), reinterpret_cast<const void *>(&DD_PW(QEPI_Q16, false, 2)),
                                  reinterpret_cast<const void *>(&DD_PW(QEPI_Q16, true, 1)), reinterpret_cast<const void *>(&DD_PW(QEPI_Q16, false, 1)),
                                  reinterpret_cast<const void *>(&DD_PW(QEPI_Q16, true, 0)), reinterpret_cast<const void *>(&DD_PW(QEPI_Q16, false, 0))})
                DD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        return DD_OK;
    });
#undef DD_PW
    if (rc != DD_OK) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)dd_ceil_div(n_tiles, P.tiles_per_block), (unsigned)groups), dim3((unsigned)wpb * 64), lds, s, P, n_tiles);
    DD_LAUNCH_CHECK();
    return DD_OK;
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
            P.w2 = o[18] ? reinterpret_cast<const i4v *>(W + (size_t)(uint32_t)o[18]) : nullptr;
            P.cq = reinterpret_cast<const long long *>(W + (size_t)(uint32_t)o[46]);
            P.out = base(dst); P.in_zp = o[39]; P.zwc = o[38]; P.R = make_req(o);
            P.out_bytes = (long long)nimg * (P.ho + 2) * 2 * (P.wo + 2) * 16;
            DD_REQUIRE(td->w >= 16, DD_E_ARG, "dd_net_forward: uint8 first layer: output narrower than a fragment");
            DD_REQUIRE(td->pad == 1 && td->cs == 32 && o[11] == 32 && (reinterpret_cast<uintptr_t>(input) & 3) == 0 && (P.total_bytes & 3) == 0 && !P.R.linear && P.R.e >= 1,
                       DD_E_ARG, "dd_net_forward: uint8 first layer: 32 channels into a bordered tensor from a 4-byte aligned batch");
            const int n_frags = dd_ceil_div(P.m, 16);
            DD_REQUIRE(o[46] != 0 && P.total_bytes < (1ll << 31) - 16 && P.out_bytes < (1ll << 31) - 16, DD_E_ARG, "dd_net_forward: uint8 first layer: no folded requantisation constants in the program, or a batch beyond 31-bit offsets");
            const int sat0 = P.R.lo == 0 && P.R.hi == 255 ? (P.R.e <= 8 ? 2 : 1) : 0;
            const dim3 grid0(dd_ceil_div(n_frags, 4 * C0F));
#define DD_Q0(H_) do { if (sat0 == 2) hipLaunchKernelGGL((q_conv0_k<H_, 2>), grid0, dim3(256), 0, s, P, n_frags); \
                       else if (sat0 == 1) hipLaunchKernelGGL((q_conv0_k<H_, 1>), grid0, dim3(256), 0, s, P, n_frags); \
                       else hipLaunchKernelGGL((q_conv0_k<H_, 0>), grid0, dim3(256), 0, s, P, n_frags); } while (0)
            if (P.w2) DD_Q0(true); else DD_Q0(false);
#undef DD_Q0
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
            const int n_frag_a = o[14];                              // > 0: two predictors in one op (fragments [0, n_frag_a) the first: netsq.py conv_heads)
            QReq Rb = P.R;
            if (n_frag_a) {
                int32_t d[48] = {0}; d[32] = o[20]; d[33] = o[21]; d[40] = o[22]; d[36] = 0; d[37] = 255; d[41] = o[41]; Rb = make_req(d);
                DD_REQUIRE(o[4] >= 0 && P.epi == QEPI_ROWS && n_frag_a < P.n_mfrag && !net->tensors[o[4]].pad && o[23] % 4 == 0 && o[24] % 4 == 0 && o[25] % 4 == 0 && o[25] <= o[23] &&
                           (long long)o[24] + (long long)P.ho * P.wo * o[23] <= (long long)net->tensors[o[4]].h * net->tensors[o[4]].w * net->tensors[o[4]].cs, DD_E_ARG,
                           "dd_net_forward: uint8 conv %d: second predictor's row geometry", i);
            }
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
            // 1x1 stride-1 layers with 512 / 1024 input channels and pixels enough: the filter in registers, the pixels through LDS (q_pws_k)
            static const int pws_env = getenv("DD_Q_PWS") ? atoi(getenv("DD_Q_PWS")) : 1;
            if (pws_env && P.kh == 1 && P.kw == 1 && P.stride == 1 && P.off_y == 1 && P.off_x == 1 && (ts->cs == 512 || ts->cs == 1024) && P.m >= 8192 && P.n_mfrag >= 8 &&
                o[46] != 0 && (P.epi == QEPI_ROWS || (!P.R.linear && P.R.e >= 1))) {
                QPwsP Q;
                memset(&Q, 0, sizeof(Q));
                Q.in = P.in; Q.H = P.H; Q.W = P.W; Q.c16_in = P.c16_in; Q.hw = P.ho * P.wo; Q.m = P.m;
                Q.w = P.w; Q.cbias = P.cbias; Q.cq = reinterpret_cast<const long long *>(W + (size_t)(uint32_t)o[46]);
                Q.n_mfrag = P.n_mfrag; Q.out = P.out; Q.c16_out = P.c16_out; Q.zwc = P.zwc; Q.R = P.R;
                if (P.epi == QEPI_ROWS) { Q.img_bytes_out = P.img_bytes_out; Q.row_bytes = P.row_bytes; Q.base_off = P.base_off; Q.cout_store = P.cout_store; }
                if (n_frag_a) {
                    const TensorDesc *tb = &net->tensors[o[4]];
                    Q.n_frag_a = n_frag_a; Q.out_b = base(o[4]); Q.img_bytes_b = (long long)tb->h * tb->w * tb->cs; Q.row_bytes_b = o[23]; Q.base_off_b = o[24]; Q.cout_store_b = o[25];
                    Q.zwc_b = o[28]; Q.Rb = Rb;
                }
                bool ok = false;
                // (the 19x19 class predictor has 18 fragments: six waves of three load the SIMDs 6 / 6 / 3 / 3, five of four 8 / 4 / 4 / 2)
                const int rc = ts->cs == 512 ? (P.epi == QEPI_ROWS && (P.n_mfrag + 2) / 3 <= 8 && P.n_mfrag % 4 != 0 ? launch_q_pws<512, 3>(s, Q, nimg, net->ctx->device, &ok) : launch_q_pws<512, 4>(s, Q, nimg, net->ctx->device, &ok))
                                             : launch_q_pws<1024, 2>(s, Q, nimg, net->ctx->device, &ok);
                if (rc != DD_OK) return rc;
                if (ok) return DD_OK;
            }
            auto run_generic = [&](QConvP &P) -> int {
                P.hw_magic = P.wo > 1 && (long long)P.m * (P.ho * P.wo) < (1ll << 32) ? (unsigned)((1ull << 32) / (unsigned)(P.ho * P.wo)) + 1u : 0u;   // (a divisor of 1 has no 32-bit magic)
                P.wo_magic = (unsigned)((1ull << 32) / (unsigned)P.wo) + 1u;
                const int n_mgroups = P.n_frag_a ? P.groups_a + dd_ceil_div(P.n_mfrag - P.n_frag_a, P.mq) : dd_ceil_div(P.n_mfrag, P.mq);
                // Pixel fragments per wave item.  Two by default; four (half the weight fetches per MFMA, twice the registers) where the launch has
                // few wave items anyway and pixels enough -- measured per layer at 384 frames: extras 48 -> 38, 63 -> 39 us, 5x5 class predictor
                // 23 -> 16 us, but 53 -> 65, 86 -> 95, 68 -> 78 us on the layers with 19 k+ items (b12 / b13 pointwise, 19x19 class predictor).
                const long long items2 = (long long)n_mgroups * dd_ceil_div(P.m, 32);
                static const int npf_env = getenv("DD_Q_NPF") ? atoi(getenv("DD_Q_NPF")) : 0;
                // (end of round: with the register-filter kernel on the big 1x1 layers and the k split on the small 3x3 ones, two fragments win or tie on
                // every layer left here -- 87 -> 78 us over the eleven small layers; the rule used to be four below 8 192 items)
                const int npf = npf_env ? npf_env : 2;
                (void)items2;
                const long long n_items = (long long)n_mgroups * dd_ceil_div(P.m, 16 * npf);
                DD_REQUIRE(n_items < (1ll << 31), DD_E_CAPACITY, "dd_net_forward: uint8 conv %d: %lld wave items", i, n_items);
                const bool rsum = P.zwc != 0 || (P.n_frag_a && P.zwc_b != 0);
                const bool pipe = n_items < 8192;
                static const int split_env = getenv("DD_Q_SPLITK") ? atoi(getenv("DD_Q_SPLITK")) : 1;
                const bool split = split_env && P.kh * P.kw == 9 && n_items < 1024;       // (3x3: the k steps divide by the three filter rows; at 1 200 items the split costs: 40 -> 51 us)
                const dim3 grid(split ? (unsigned)n_items : (unsigned)((n_items + 3) / 4)), block(split ? 192 : 256);
#define DD_QC2(MQ_, R_) do { if (npf == 4 && split) hipLaunchKernelGGL((q_conv_k<MQ_, R_, 4, false, 3>), grid, block, 0, s, P, (int)n_items, n_mgroups); \
                             else if (npf == 4) hipLaunchKernelGGL((q_conv_k<MQ_, R_, 4, false>), grid, block, 0, s, P, (int)n_items, n_mgroups); \
                             else if (split) hipLaunchKernelGGL((q_conv_k<MQ_, R_, 2, true, 3>), grid, block, 0, s, P, (int)n_items, n_mgroups); \
                             else if (pipe) hipLaunchKernelGGL((q_conv_k<MQ_, R_, 2, true>), grid, block, 0, s, P, (int)n_items, n_mgroups); \
                             else hipLaunchKernelGGL((q_conv_k<MQ_, R_, 2, false>), grid, block, 0, s, P, (int)n_items, n_mgroups); } while (0)
#define DD_QC(MQ_) do { if (rsum) DD_QC2(MQ_, true); else DD_QC2(MQ_, false); } while (0)
                if (P.mq == 4) DD_QC(4); else if (P.mq == 3) DD_QC(3); else if (P.mq == 2) DD_QC(2); else DD_QC(1);
#undef DD_QC2
#undef DD_QC
                DD_LAUNCH_CHECK();
                return DD_OK;
            };
            if (n_frag_a == 0) return run_generic(P);
            // two predictors outside q_pws_k's shapes (small feature maps, small batches): one launch, the second one's fragments, constants and
            // destination rows chosen per wave item (DD_Q_HEADS_ONE=0: one launch each, the form until round 6 -- the same bits)
            const TensorDesc *tb = &net->tensors[o[4]];
            static const int heads_one = getenv("DD_Q_HEADS_ONE") ? atoi(getenv("DD_Q_HEADS_ONE")) : 1;
            if (heads_one) {
                P.mq = std::min(4, n_frag_a);
                P.n_frag_a = n_frag_a; P.groups_a = dd_ceil_div(n_frag_a, P.mq);
                P.out_b = base(o[4]); P.img_bytes_b = (long long)tb->h * tb->w * tb->cs; P.row_bytes_b = o[23]; P.base_off_b = o[24]; P.cout_store_b = o[25];
                P.zwc_b = o[28]; P.Rb = Rb;
                return run_generic(P);
            }
            QConvP P2 = P;
            P.n_mfrag = n_frag_a; P.mq = std::min(4, P.n_mfrag);
            int rc2 = run_generic(P);
            if (rc2 != DD_OK) return rc2;
            P2.w = P.w + (size_t)n_frag_a * (P.kh * P.kw * P.kc_per_tap) * 64; P2.cbias = P.cbias + 16 * n_frag_a;
            P2.n_mfrag -= n_frag_a; P2.mq = std::min(4, P2.n_mfrag);
            P2.out = base(o[4]); P2.img_bytes_out = (long long)tb->h * tb->w * tb->cs; P2.row_bytes = o[23]; P2.base_off = o[24]; P2.cout_store = o[25];
            P2.zwc = o[28]; P2.R = Rb;
            return run_generic(P2);
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
            static const bool no_mfma = getenv("DD_Q_DW_VALU") && atoi(getenv("DD_Q_DW_VALU")) != 0;       // A/B switch: the vector-ALU form
            if (o[20] && P.R.e >= 1 && !no_mfma) {                  // o[20], o[21]: the block-diagonal operand table and its constants (netsq.pack_dw_mfma)
                QDwmP Q;
                Q.in = P.in; Q.H = P.H; Q.W = P.W; Q.c16 = P.c16; Q.stride = P.stride; Q.off_y = P.off_y; Q.off_x = P.off_x; Q.ho = P.ho; Q.wo = P.wo;
                Q.m = nimg * P.ho * P.wo; Q.out = P.out; Q.R = P.R;
                Q.dw_a = reinterpret_cast<const uint2 *>(W + (size_t)(uint32_t)o[20]);
                Q.dw_cb = reinterpret_cast<const int *>(W + (size_t)(uint32_t)o[21]);
                const long long n_items = (long long)P.c16 * dd_ceil_div(Q.m, 64);
                DD_REQUIRE(n_items < (1ll << 31), DD_E_CAPACITY, "dd_net_forward: uint8 depthwise %d: %lld wave items", i, n_items);
                Q.hw_magic = (unsigned)((1ull << 32) / (unsigned)(P.ho * P.wo)) + 1u; Q.wo_magic = (unsigned)((1ull << 32) / (unsigned)P.wo) + 1u; Q.c16_magic = (unsigned)((1ull << 32) / (unsigned)P.c16) + 1u;
                if ((long long)Q.m * (P.ho * P.wo) >= (1ll << 32) || P.wo < 2 || P.c16 < 2) Q.hw_magic = 0;       // (a divisor of 1 has no 32-bit magic)
                DD_REQUIRE(n_items * P.c16 < (1ll << 32) && (long long)(nimg + 1) * (P.H + 2) * P.c16 * (P.W + 2) * 16 < (1ll << 32) &&
                           (long long)(nimg + 1) * (P.ho + 2) * P.c16 * (P.wo + 2) * 16 < (1ll << 32), DD_E_CAPACITY, "dd_net_forward: uint8 depthwise %d: beyond 32-bit offsets", i);
                if (P.R.lo == 0 && P.R.hi == 255) hipLaunchKernelGGL((q_dwm_k<true>), dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, s, Q, (int)n_items);
                else hipLaunchKernelGGL((q_dwm_k<false>), dim3((unsigned)((n_items + 3) / 4)), dim3(256), 0, s, Q, (int)n_items);
                DD_LAUNCH_CHECK();
                if (i < (int)net->op_launch.size()) net->op_launch[i] = 17;      // dd_net_op_launches: q_dwm_k ran (a layer table prints the kernel that ran, not the op's default)
                return DD_OK;
            }
            hipLaunchKernelGGL(q_dw_k, dim3((unsigned)((P.total + 255) / 256)), dim3(256), 0, s, P);
            DD_LAUNCH_CHECK();
            return DD_OK;
        }
        case OP_QDWPW: {
            QDwpwP P;
            memset(&P, 0, sizeof(P));
            DD_REQUIRE(ts->pad == 1 && td->pad == 1 && ts->cs % 16 == 0 && td->cs % 64 == 0, DD_E_ARG, "dd_net_forward: uint8 block %d: tensor layouts", i);
            const int stride = o[7], cin = o[10], cout = o[11];
            P.in = base(src); P.H = ts->h; P.W = ts->w;
            P.off_y = 1 - o[8]; P.off_x = 1 - o[9]; P.ho = td->h; P.wo = td->w; P.hw = P.ho * P.wo; P.tiles_per_frame = dd_ceil_div(P.hw, QT);
            P.out = base(dst); P.c16_out = td->cs / 16;
            P.dw_a = reinterpret_cast<const uint2 *>(W + (size_t)(uint32_t)o[20]);
            P.dw_cq = reinterpret_cast<const long long *>(W + (size_t)(uint32_t)o[45]);      // o[21] (the 32-bit constants) serves the two-op form only
            P.cq = reinterpret_cast<const long long *>(W + (size_t)(uint32_t)o[46]);
            P.w = reinterpret_cast<const i4v *>(W + (size_t)(uint32_t)o[16]);
            P.w2 = o[18] ? reinterpret_cast<const i4v *>(W + (size_t)(uint32_t)o[18]) : nullptr;       // o[18]: the lo part of a split pointwise filter
            P.cbias = reinterpret_cast<const int *>(W + (size_t)(uint32_t)o[17]);
            P.zwc = o[38]; P.Rp = make_req(o); P.dup = o[47];
            { int32_t d[48] = {0}; d[32] = o[22]; d[33] = o[23]; d[36] = o[24]; d[37] = o[25]; d[40] = o[28]; P.Rd = make_req(d); }
            DD_REQUIRE(!P.dup || (cin == 32 && P.zwc == 0 && !P.w2), DD_E_ARG, "dd_net_forward: uint8 block %d: duplicated operand bytes are for 32 input channels", i);
            DD_REQUIRE(o[45] != 0 && o[46] != 0, DD_E_ARG, "dd_net_forward: uint8 block %d: the program carries no folded requantisation constants (compiled by an older netsq.py?)", i);
            DD_REQUIRE(cin == ts->cs && cout == td->cs && P.Rd.e >= 1 && P.Rp.e >= 1 && !P.Rd.linear && !P.Rp.linear && P.off_y >= 0 && P.off_x >= 0 &&
                       (P.ho - 1) * stride + 2 + P.off_y <= P.H + 1 && (P.wo - 1) * stride + 2 + P.off_x <= P.W + 1, DD_E_ARG,
                       "dd_net_forward: uint8 block %d: shapes / multipliers the fused kernel does not take", i);
            DD_REQUIRE((double)nimg * (P.ho + 2) * (P.wo + 2) * td->cs < 4294967296.0, DD_E_CAPACITY, "dd_net_forward: uint8 block %d: output beyond 32-bit offsets", i);
            bool ok = false;
            int rc = DD_OK;
            const int dev = net->ctx->device;
#define DD_QB(CIN_, COUT_, WP_, S_, LPT_) launch_q_dwpw<CIN_, COUT_, WP_, S_, LPT_>(s, P, nimg, dev, &ok)
            static const int qt128 = getenv("DD_Q_QT128") ? atoi(getenv("DD_Q_QT128")) : 1;   // block 1: 128-pixel tiles, four waves (0: 64-pixel tiles, two waves)
            if (cin == 32 && cout == 64 && stride == 1) rc = qt128 ? launch_q_dwpw<32, 64, 4, 1, 6, 0, 4, 128>(s, P, nimg, dev, &ok) : DD_QB(32, 64, 2, 1, 8);
            else if (cin == 64 && cout == 128 && stride == 2) rc = DD_QB(64, 128, 2, 2, 8);
            else if (cin == 128 && cout == 128 && stride == 1) rc = DD_QB(128, 128, 2, 1, 8);
            else if (cin == 128 && cout == 256 && stride == 2) rc = DD_QB(128, 256, 2, 2, 8);
            else if (cin == 256 && cout == 256 && stride == 1) rc = DD_QB(256, 256, 2, 1, 6);
            else if (cin == 256 && cout == 512 && stride == 2) rc = DD_QB(256, 512, 1, 2, 12);
            else if (cin == 512 && cout == 512 && stride == 1) rc = DD_QB(512, 512, 1, 1, 8);
#undef DD_QB
            if (rc != DD_OK) return rc;
            DD_REQUIRE(ok, DD_E_ARG, "dd_net_forward: uint8 block %d (%d -> %d, stride %d, %d x %d): no fused kernel for this shape -- compile the program with the two-op form", i, cin, cout, stride, P.H, P.W);
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
            DD_REQUIRE(P.cls_stride % 16 == 0 && P.n_classes <= 128, DD_E_ARG, "dd_net_forward: uint8 decode: %d classes in rows of %d bytes", P.n_classes, P.cls_stride);
            hipLaunchKernelGGL(q_ssd_decode_k, dim3(dd_ceil_div(P.n_anchors * 8, 256), nimg), dim3(256), 0, s, P);
            DD_LAUNCH_CHECK();
            return DD_OK;
        }
        default:
            DD_REQUIRE(false, DD_E_ARG, "dd_net_forward: uint8 op kind %d at %d is not built", kind, i);
    }
    return DD_OK;
}
