// Convolutional network executor for the detector / re-ID encoder forward passes.
//
// Replaces tflite_runtime.Interpreter.invoke() at tools/ssd_mobilenet.py:102-109,
// tools/yolov5.py:107-109 and tools/generate_detections.py:169-171 (upstream paths).  The host
// (deepdish_amd/nets.py) compiles a model into a flat op program + one weight blob -- the analogue
// of the reference's .tflite file -- and this file runs it:
//   * activations NHWC f16 in HBM (channel stride padded to 8 so every tap is one 16-byte load),
//   * dense convs as implicit GEMM on v_mfma_f32_16x16x32_f16 (f32 accumulate): weights are the A
//     operand (rows = output channels) and pixels the B operand, so each lane ends up holding four
//     consecutive output channels of one pixel and the fused epilogue (bias, activation, residual,
//     second affine+ELU output, SSD / YOLO head scatter) stores 8 or 16 bytes per lane,
//   * operand tiles staged through double-buffered LDS (80-byte padded rows, ds_read_b128),
//   * depthwise 3x3, pooling, upsampling, input conversion as 16-byte-per-lane streaming kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include "common.h"
#include "ssd_dev.h"
#include "net_priv.h"
#include "mars_tail.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum { OP_INPUT = 1, OP_CONV = 2, OP_DWCONV = 3, OP_MAXPOOL = 4, OP_UPSAMPLE = 5, OP_FC = 6, OP_L2NORM = 7, OP_STEM = 8, OP_DWPW = 9 };
enum { ACT_NONE = 0, ACT_RELU6 = 1, ACT_ELU = 2, ACT_SILU = 3, ACT_RELU = 4, ACT_SIGMOID = 5 };
enum { EPI_F16 = 0, EPI_F32 = 1, EPI_SSD_HEAD = 2, EPI_YOLO = 3 };
// dd_net_op_launches: 0 = the op's own kernel, 1 = no launch (folded into the next op's), else the fused / special kernel
enum { OPK_DEFAULT = 0, OPK_FOLDED = 1, OPK_POOL_ROWS = 2, OPK_POOL_ROWS_STEM = 3, OPK_RES_UNIT = 4, OPK_SSD_FRONT = 5, OPK_C64_ROWS = 6, OPK_S2_ROWS = 7, OPK_CONV_WS = 8, OPK_WS_DW = 9, OPK_DWPW_ROWS = 10, OPK_SSD_HEAD_DEC = 11, OPK_RES_PAIR = 12, OPK_YOLO_HEAD_DEC = 13, OPK_MARS_WS = 14, OPK_FOLDED_PREV = 15, OPK_MARS_PAIR = 16, OPK_C64_STRIPS = 18, OPK_Q_FRONT = 19, OPK_Q_MID = 20 };   // (17: q_dwm_k, csrc/netsq.hip; 19: q_front_k, csrc/netsq_front.hip; 20: q_mid_k, csrc/netsq_mid.hip)
enum { DT_F16 = 0, DT_F32 = 1, DT_U8 = 2 };

constexpr int OP_WORDS = 48;       // int32 words per op record (see deepdish_amd/nets.py)
constexpr int TENSOR_WORDS = 8;

// 1 / (1 + exp(-v)) with v_rcp_f32 (1 ulp) instead of the correctly rounded quotient hipcc makes of `1.f / x` (v_div_scale x 2, v_rcp, four
// v_fma, v_div_fmas, v_div_fixup: eleven instructions per element -- 6 355 such sequences in this file before; every YOLOv5 layer ends in a SiLU
// and the conv kernels are bound by instruction issue).  The result is stored as f16 (SiLU) or compared at 2e-4 (Detect heads): the last f32
// bit does not reach either.  Used by SiLU and by both forms of the YOLOv5 head (matrix and fused decode: the same bits between them).
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

// YOLOv5 Detect box of one row (models/yolo.py Detect.forward as tools/yolov5.py consumes it): columns 0..3 of the decoded row from the four
// sigmoids; one function with contraction off, called by the matrix epilogue and by the fused decode, so both round alike.
__device__ __forceinline__ float yolo_box_col(int o, float s, float px, float py, float stride, float aw, float ah, float img_w, float img_h) {
#pragma clang fp contract(off)
    if (o == 0) return (s * 2.f - 0.5f + px) * stride / img_w;
    if (o == 1) return (s * 2.f - 0.5f + py) * stride / img_h;
    if (o == 2) return (s * 2.f) * (s * 2.f) * aw / img_w;
    return (s * 2.f) * (s * 2.f) * ah / img_h;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU6: return __builtin_amdgcn_fmed3f(v, 0.f, 6.f);      // one v_med3_f32 (fmin(fmax()) costs an extra canonicalising v_max)
        case ACT_ELU: {
            // max(v, exp(min(v, 0)) - 1): the min is the clamp modifier of v_exp_f32 (its result is clamped to [0, 1]: med3 with
            // 0 and 1 folds into the instruction), the max a v_med3 with a huge finite third operand (no canonicalising copy) -- four instructions per
            // element and no compare / VCC / select round trip with its wait states (v_mul, v_exp, v_add, v_cmp, s_nop, v_cndmask:
            // the row-streaming kernels are bound by instruction issue).  The product v * log2(e) is the one
            // __expf forms; exp(v) - 1: absolute error ~1e-7, far below the f16 the result is stored in.  For v > 0 the result is v.
            float e = __builtin_amdgcn_exp2f(v * 1.44269504088896340736f);
            e = __builtin_amdgcn_fmed3f(e, 0.f, 1.f);
            return __builtin_amdgcn_fmed3f(v, e - 1.f, 3.0e38f);   // (with +inf hipcc rewrites the med3 as a max and canonicalises v first: a fifth instruction)
        }
        case ACT_SILU: {
            // The product is pinned in a register: left to itself hipcc folds it into the f16 conversion that follows in SOME kernels
            // (v_fma_mixlo_f16: one rounding instead of two) or into a residual add -- and two kernels that can run the same layer
            // (the Focus fold, the row kernels, the tile variants) then differ in 2e-5 of their outputs.  A quotient could not fuse.
            float r = v * fast_sigmoid(v);
#if defined(__HIP_DEVICE_COMPILE__)
            asm("" : "+v"(r));
#endif
            return r;
        }
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

// Pooling maximum of values that are never NaN in a valid network: v_maximum3_f32 / v_maximum_f32 on gfx950 (IEEE-754-2019 maximum).  fmaxf is
// maxnum, which hipcc implements as canonicalise(a), canonicalise(b), v_max for operands it cannot prove quiet (loop-carried values,
// results of other maxima): three instructions instead of one in kernels bound by instruction issue.  Same result for non-NaN inputs.
__device__ __forceinline__ float pool_max(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_elementwise_maximum(a, b);
#else
    return a > b ? a : b;
#endif
}

struct ConvP {
    const _Float16 *in; int H, W, cs_in, coff_in, cin;
    const _Float16 *w; const float *bias; int kpad;
    int kh, kw, stride, pad_t, pad_l;
    int ho, wo, cout, m;
    int act;
    void *out; int cs_out, coff_out, epi;
    const _Float16 *res; int cs_res, coff_res;
    _Float16 *out2; int cs_out2, coff_out2; const float *aff2; int cout_pad;
    int post_aff;                   // EPI_F32 only: out = aff2.scale * act(v) + aff2.shift
    int splitk; float *slab;        // splitk > 1: raw partial sums go to slab[z][m][cout_pad]
    const _Float16 *zero;           // >= 16 bytes of zeros: source of out-of-image taps for direct-to-LDS fills
    int p[6]; float f[8];
    // spatially tiled kernels (conv3x3_rw_k, stem_conv3_k): th x tw output pixels of one image per block
    int th, tw, tiles_x, tiles_y;
    const uint8_t *src8; float in_mean, in_scale;      // stem: u8 [N][H][W][3] source, (x - mean) * scale
    // fused depthwise 3x3 -> pointwise (dwpw_k): the depthwise half (H, W, stride, pad_* describe it; kh = kw = 1)
    const _Float16 *dw_w; const float *dw_bias; int dw_act; int total_quads;
    // SSD head with the decode in its epilogue (ssd_head_finish): per-anchor outputs [max_batch][n_anchors] instead of the head matrix
    const float *anchors; float *dec_boxes, *dec_score, *dec_keys; int *dec_cls; float dec_thr;
};


// Everything after the K reduction for channels co..co+3 of output pixel m (v = raw sums).
__device__ __forceinline__ void conv_epilogue(const ConvP &P, int m, int co, float v[4], int hw) {
    const f4 bv = *reinterpret_cast<const f4 *>(P.bias + co);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bv[r];
    if (P.epi == EPI_YOLO) {
        const int n = m / hw, p = m - n * hw;
        const int py = p / P.wo, px = p - py * P.wo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = co + r;
            if (ch >= P.cout) continue;
            const int no = P.p[0];                     // 5 + classes
            const int an = ch / no, o = ch - an * no;
            const float s = fast_sigmoid(v[r]);
            float val = s;
            if (o < 4) val = yolo_box_col(o, s, (float)px, (float)py, P.f[6], P.f[2 * an], P.f[2 * an + 1], P.f[7], (float)P.p[4]);
            const size_t row = (size_t)n * P.p[1] + P.p[2] + (size_t)an * hw + p;
            static_cast<float *>(P.out)[row * no + o] = val;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], P.act);
    if (P.epi == EPI_SSD_HEAD) {                        // fused box + class predictor of one feature map; the host
        const int n = m / hw, p = m - n * hw;           // ordered the output channels [anchor][4 box + C class], which is
        const int A = P.p[5];                           // the memory order of the pixel's A rows of the head matrix
        float *dst = static_cast<float *>(P.out) + ((size_t)n * P.p[1] + P.p[2] + (size_t)p * A) * P.p[3];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co + r < P.cout) dst[co + r] = v[r];
        return;
    }
    if (P.res) {
        const h4 rv = *reinterpret_cast<const h4 *>(P.res + (size_t)m * P.cs_res + P.coff_res + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
    }
    if (P.epi == EPI_F32) {
        if (P.post_aff) {
            const f4 sc = *reinterpret_cast<const f4 *>(P.aff2 + co);
            const f4 sh = *reinterpret_cast<const f4 *>(P.aff2 + P.cout_pad + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = sc[r] * v[r] + sh[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co + r >= P.cout) v[r] = 0.f;
        *reinterpret_cast<f4 *>(static_cast<float *>(P.out) + (size_t)m * P.cs_out + P.coff_out + co) = f4{v[0], v[1], v[2], v[3]};
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (co + r >= P.cout) v[r] = 0.f;
    {
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (_Float16)v[r];
        *reinterpret_cast<h4 *>(static_cast<_Float16 *>(P.out) + (size_t)m * P.cs_out + P.coff_out + co) = o;
    }
    if (P.out2) {                                   // second view: ELU(scale * raw + shift)
        const f4 sc = *reinterpret_cast<const f4 *>(P.aff2 + co);
        const f4 sh = *reinterpret_cast<const f4 *>(P.aff2 + P.cout_pad + co);
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = co + r < P.cout ? apply_act(sc[r] * v[r] + sh[r], ACT_ELU) : 0.f;
            o[r] = (_Float16)t;
        }
        *reinterpret_cast<h4 *>(P.out2 + (size_t)m * P.cs_out2 + P.coff_out2 + co) = o;
    }
}

// acc + x.half * w.half in one instruction with f32 accumulation (the product of two halves is exact in f32, so
// this is bit-identical to converting and multiplying); left to itself the compiler converts most operands
// with separate v_cvt_f32_f16 (232 of them per depthwise item).
typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
template <int HI>
__device__ __forceinline__ float fma_mix_f16(unsigned x, unsigned w, float acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(w));
    else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(w));
#endif
    return acc;
}
__device__ __forceinline__ void dw_tap(float (&acc)[8], const h8 &x, const h8 &w) {
    const u4v_t xv = __builtin_bit_cast(u4v_t, x), wv = __builtin_bit_cast(u4v_t, w);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc[i] = (i & 1) ? fma_mix_f16<1>(xv[i >> 1], wv[i >> 1], acc[i]) : fma_mix_f16<0>(xv[i >> 1], wv[i >> 1], acc[i]);
}

// Plain NHWC f16 output, 8 consecutive channels of one pixel (v = raw sums): one 16-byte store, so
// eight neighbouring lanes write a full 128-byte line.  Epi8 = the per-channel constants of those
// eight channels (kernels whose lanes keep the same channels for every pixel load them once).
struct Epi8 { f4 b0, b1, s0, s1, t0, t1; };

__device__ __forceinline__ Epi8 epi8_load(const ConvP &P, int co) {
    Epi8 E;
    E.b0 = *reinterpret_cast<const f4 *>(P.bias + co);
    E.b1 = *reinterpret_cast<const f4 *>(P.bias + co + 4);
    E.s0 = E.s1 = E.t0 = E.t1 = f4{0.f, 0.f, 0.f, 0.f};
    if (P.out2) {
        E.s0 = *reinterpret_cast<const f4 *>(P.aff2 + co);
        E.s1 = *reinterpret_cast<const f4 *>(P.aff2 + co + 4);
        E.t0 = *reinterpret_cast<const f4 *>(P.aff2 + P.cout_pad + co);
        E.t1 = *reinterpret_cast<const f4 *>(P.aff2 + P.cout_pad + co + 4);
    }
    return E;
}

// ACT >= 0: the activation is known at compile time (the per-element switch on P.act would otherwise be
// compiled into a chain of branches around every value).
// EF (epilogue flavour) 0: no residual, no second output; 1: both; 2: residual only; -1: whatever P says.
#ifndef DD_EPI_NT
#define DD_EPI_NT true
#endif
template <int ACT = -1, bool BIAS = true, int EF = -1, bool NT = DD_EPI_NT, bool TAIL = true>   // BIAS = false: the accumulators were initialised with the bias;
__device__ __forceinline__ void conv_epilogue_f16x8(const ConvP &P, const Epi8 &E, int m, int co, float v[8], const h8 *res_pre = nullptr) {   // TAIL = false: cout % 8 == 0
    // res_pre: the residual vector of this (pixel, channel group), fetched by the caller before its first store -- a
    // load issued here cannot be moved above the stores of the caller's previous pixel (they may alias), so a loop of
    // epilogues would pay one memory round trip per pixel
    const int act = ACT < 0 ? P.act : ACT;
    const bool has_res = EF < 0 ? P.res != nullptr : EF >= 1, has_out2 = EF < 0 ? P.out2 != nullptr : EF == 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[r] = apply_act(BIAS ? v[r] + E.b0[r] : v[r], act);
        v[4 + r] = apply_act(BIAS ? v[4 + r] + E.b1[r] : v[4 + r], act);
    }
    if (has_res) {
        const h8 rv = res_pre ? *res_pre : *reinterpret_cast<const h8 *>(P.res + (size_t)m * P.cs_res + P.coff_res + co);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
    }
    const bool tail = TAIL && co + 8 > P.cout;     // only the last, partly padded channel group needs masking
    h8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (_Float16)v[r];
    if (tail) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (co + r >= P.cout) o[r] = (_Float16)0.f;
    }
    // streaming store: same-box A/B -0.7 % on both networks (the activations of a layer exceed the L2 anyway); the
    // depthwise kernel keeps plain stores -- its output is re-read at once by the following 1x1 layer (+4 % with nt)
    h8 *dst = reinterpret_cast<h8 *>(static_cast<_Float16 *>(P.out) + (size_t)m * P.cs_out + P.coff_out + co);
    if constexpr (NT) __builtin_nontemporal_store(o, dst);
    else *dst = o;
    if (has_out2) {                                 // second view: ELU(scale * raw + shift)
        h8 o2;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float sc = r < 4 ? E.s0[r] : E.s1[r - 4], sh = r < 4 ? E.t0[r] : E.t1[r - 4];
            o2[r] = (_Float16)apply_act(sc * v[r] + sh, ACT_ELU);
        }
        if (tail) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (co + r >= P.cout) o2[r] = (_Float16)0.f;
        }
        *reinterpret_cast<h8 *>(P.out2 + (size_t)m * P.cs_out2 + P.coff_out2 + co) = o2;
    }
}

template <int ACT = -1>
__device__ __forceinline__ void conv_epilogue_f16x8(const ConvP &P, int m, int co, float v[8]) {
    const Epi8 E = epi8_load(P, co);
    conv_epilogue_f16x8<ACT>(P, E, m, co, v);
}

// Rows of a block's f32 output tile staged in LDS -> activation etc. -> 16-byte stores.
template <int ACT, int BM, int BN, int T>
__device__ __forceinline__ void conv_finish_rows(const ConvP &P, const float *ot, int m0, int n0) {
    constexpr int OROW = BN + 4, G = BN / 8;                      // floats per staged pixel row, 8-channel groups per row
    static_assert(T % G == 0, "a thread keeps its channel group over the row loop");
    const int g = threadIdx.x % G, co = n0 + g * 8;
    if (co >= P.cout_pad) return;
    const Epi8 E = epi8_load(P, co);                              // once: inside the loop these loads could not move above
    for (int pl = threadIdx.x / G; pl < BM; pl += T / G) {         // the previous row's stores (they may alias)
        const int m = m0 + pl;
        if (m >= P.m) break;
        const f4 lo = *reinterpret_cast<const f4 *>(ot + pl * OROW + g * 8);
        const f4 hi = *reinterpret_cast<const f4 *>(ot + pl * OROW + g * 8 + 4);
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        conv_epilogue_f16x8<ACT>(P, E, m, co, v);
    }
}

// Everything after the K loop, shared by the register-staged and the direct-to-LDS kernels.
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void conv_finish(const ConvP &P, f4 (&acc)[NI][MI], _Float16 *lds, int m0, int n0, int hw) {
    constexpr int T = WM * WN * 64;
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    if (P.epi == EPI_F16 && P.splitk == 1) {
        // ---- transpose the accumulators through LDS (f32) so that stores are 16 bytes per lane and whole
        // 128-byte lines per pixel; the staging buffers are free after the loop's last barrier.
        constexpr int OROW = BN + 4;                              // floats per staged pixel row
        float *ot = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int a = 0; a < NI; ++a)
                *reinterpret_cast<f4 *>(ot + ((wm * MI + b) * 16 + fr) * OROW + (wn * NI + a) * 16 + fq * 4) = acc[a][b];
        __syncthreads();
        switch (P.act) {                                          // one straight-line copy of the row loop per activation
            case ACT_NONE: conv_finish_rows<ACT_NONE, BM, BN, T>(P, ot, m0, n0); break;
            case ACT_RELU6: conv_finish_rows<ACT_RELU6, BM, BN, T>(P, ot, m0, n0); break;
            case ACT_ELU: conv_finish_rows<ACT_ELU, BM, BN, T>(P, ot, m0, n0); break;
            case ACT_SILU: conv_finish_rows<ACT_SILU, BM, BN, T>(P, ot, m0, n0); break;
            default: conv_finish_rows<-1, BM, BN, T>(P, ot, m0, n0);
        }
        return;
    }
    if (P.epi == EPI_SSD_HEAD && P.splitk == 1) {
        // ---- head matrix rows: a pixel's A x (4 + C) floats are contiguous, so stage the tile in LDS and let
        // each wave write one pixel's run with consecutive lanes on consecutive floats (256-byte stores)
        constexpr int OROW = BN + 4;
        float *ot = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int a = 0; a < NI; ++a)
                *reinterpret_cast<f4 *>(ot + ((wm * MI + b) * 16 + fr) * OROW + (wn * NI + a) * 16 + fq * 4) = acc[a][b];
        __syncthreads();
        const int A = P.p[5];
        // a lane owns 4 consecutive columns: one ds_read_b128 + one 16-byte store (rows of A * (4 + C) floats are only
        // 4-byte aligned: the store type says so); bias loaded once -- inside the pixel loop the load could not move
        // above the previous pixel's stores (may alias)
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        constexpr int LPP = BN / 4, PPW = 64 / LPP;                // lanes per pixel row, pixels per wave pass
        const int lc = (lane % LPP) * 4, lp = lane / LPP;
        const int ch = n0 + lc;
        const int nlive = min(4, max(0, P.cout - ch));            // columns of this lane inside the layer
        f4 bias = f4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nlive; ++q) bias[q] = P.bias[ch + q];
        for (int pl = wave * PPW + lp; pl < BM; pl += WM * WN * PPW) {
            const int m = m0 + pl;
            if (m >= P.m || nlive == 0) continue;
            const int n = m / hw, p = m - n * hw;
            float *__restrict__ dst = static_cast<float *>(P.out) + ((size_t)n * P.p[1] + P.p[2] + (size_t)p * A) * P.p[3] + ch;
            const f4 v = *reinterpret_cast<const f4 *>(ot + pl * OROW + lc) + bias;
            if (nlive == 4) *reinterpret_cast<f4u *>(dst) = v;
            else for (int q = 0; q < nlive; ++q) dst[q] = v[q];
        }
        return;
    }
    if (P.epi == EPI_YOLO && P.splitk == 1) {
        // ---- decoded Detect rows: the (5 + C) floats of a (pixel, anchor) are contiguous, so stage the tile in LDS and let
        // consecutive lanes take consecutive 4-column groups of a pixel (one 16-byte store per lane where the group stays
        // inside one anchor's row; rows are only 4-byte aligned).  With a lane per (pixel, 4 channels) of the MFMA layout every
        // value was its own 4-byte store, 16 pixels x 340 bytes apart per instruction: the 80x80 head ran at 1.6 TB/s.
        // Same expressions per value as conv_epilogue.
        constexpr int OROW = BN + 4;
        float *ot = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int b = 0; b < MI; ++b)
#pragma unroll
            for (int a = 0; a < NI; ++a)
                *reinterpret_cast<f4 *>(ot + ((wm * MI + b) * 16 + fr) * OROW + (wn * NI + a) * 16 + fq * 4) = acc[a][b];
        __syncthreads();
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        constexpr int LPP = BN / 4, PPW = 64 / LPP;
        const int lc = (lane % LPP) * 4, lp = lane / LPP;
        const int ch = n0 + lc;
        const int nlive = min(4, max(0, P.cout - ch));
        const int no = P.p[0];
        f4 bias = f4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < nlive; ++q) bias[q] = P.bias[ch + q];
        int an[4], oo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { an[q] = (ch + q) / no; oo[q] = (ch + q) - an[q] * no; }
        const bool one_row = nlive == 4 && an[0] == an[3];
        // The four box columns of a row are 4 of 85: their arithmetic (a division each) sits behind one branch that two or three
        // lanes of a wave take, as ONE expression with per-lane constants -- as an if / else chain per value every wave ran all four
        // variants for all four values of every lane (the 80x80 head: 580 us, most of it this).
        bool box_q[4]; float mulc[4], divc[4];
        bool any_box = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            box_q[q] = oo[q] < 4;
            any_box |= box_q[q];
            mulc[q] = oo[q] < 2 ? P.f[6] : P.f[2 * an[q] + (oo[q] & 1)];
            divc[q] = (oo[q] & 1) ? (float)P.p[4] : P.f[7];
        }
        for (int pl = wave * PPW + lp; pl < BM; pl += WM * WN * PPW) {
            const int m = m0 + pl;
            if (m >= P.m || nlive == 0) continue;
            const int n = m / hw, p = m - n * hw;
            const f4 v = *reinterpret_cast<const f4 *>(ot + pl * OROW + lc) + bias;
            f4 val;
#pragma unroll
            for (int q = 0; q < 4; ++q) val[q] = fast_sigmoid(v[q]);
            if (any_box) {
#pragma clang fp contract(off)                                    // yolo_box_col's operations in its order, un-fused like them
                const int py = p / P.wo, px = p - py * P.wo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float sg = val[q];
                    // o = 0, 1: (s * 2 - 0.5 + grid) * stride / size;  o = 2, 3: (s * 2) * (s * 2) * anchor / size
                    const float num = oo[q] < 2 ? sg * 2.f - 0.5f + (float)((oo[q] & 1) ? py : px) : (sg * 2.f) * (sg * 2.f);
                    const float r = num * mulc[q] / divc[q];
                    if (box_q[q]) val[q] = r;
                }
            }
            float *__restrict__ out = static_cast<float *>(P.out);
            if (one_row) {
                *reinterpret_cast<f4u *>(out + ((size_t)n * P.p[1] + P.p[2] + (size_t)an[0] * hw + p) * no + oo[0]) = val;
            } else {
                for (int q = 0; q < nlive; ++q) out[((size_t)n * P.p[1] + P.p[2] + (size_t)an[q] * hw + p) * no + oo[q]] = val[q];
            }
        }
        return;
    }
    // ---- lane holds channels co..co+3 of pixel m
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = m0 + (wm * MI + b) * 16 + fr;
        if (m >= P.m) continue;
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int co = n0 + (wn * NI + a) * 16 + fq * 4;
            if (co >= P.cout_pad) continue;
            if (P.splitk > 1) {
                *reinterpret_cast<f4 *>(P.slab + ((size_t)blockIdx.z * P.m + m) * P.cout_pad + co) = acc[a][b];
            } else {
                float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
                conv_epilogue(P, m, co, v, hw);
            }
        }
    }
}

// SSD head tile whose 96 output channels are ONE anchor of the feature map (host layout: 4 box encodings, class 0 =
// background, classes 1 .. C-1, zero padding up to 96; nets.py packs the head weights that way for this kernel): the
// first stage of TFLite_Detection_PostProcess runs here, on the accumulators -- best class logit (lowest class on ties, as
// ssd_decode_k's scan + butterfly), anchor decode, sigmoid, score threshold (ssd_dev.h: the same arithmetic, the same bits)
// -- and the [n][1917][4 + C] f32 head matrix (140 MB per 384 frames, written here and read straight back by
// ssd_decode_k) never exists.  Two lanes per pixel scan the classes of the staged tile row, lane 0 of the pair decodes.
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void ssd_head_finish(const ConvP &P, f4 (&acc)[NI][MI], _Float16 *lds, int m0, int n0, int hw) {
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16, T = WM * WN * 64;
    static_assert(BN == 96 && T == 2 * BM, "one anchor per channel tile, two lanes per pixel");
    constexpr int OROW = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    float *ot = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int a = 0; a < NI; ++a)
            *reinterpret_cast<f4 *>(ot + ((wm * MI + b) * 16 + fr) * OROW + (wn * NI + a) * 16 + fq * 4) = acc[a][b];
    __syncthreads();
    const int pl = tid >> 1, h = tid & 1;
    const int m = m0 + pl;
    const int C = P.p[0];                                       // classes incl. background: columns 4 .. 4 + C - 1 of the anchor
    const float *row = ot + pl * OROW;
    const float *bias = P.bias + n0;
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    // lane h scans columns [4 + 48 h, 4 + 48 h + 48): ascending, strict `>` = the lowest class wins a tie
#pragma unroll
    for (int c4 = 0; c4 < 12; ++c4) {
        const int col = 4 + 48 * h + 4 * c4;
        const f4 v = (col < BN) ? *reinterpret_cast<const f4 *>(row + col) + *reinterpret_cast<const f4 *>(bias + col) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cls = col + q - 4;                        // class id (0 = background, skipped)
            if (cls >= 1 && cls < C && v[q] > best) { best = v[q]; bi = cls - 1; }
        }
    }
    {
        const float ob = __shfl_xor(best, 1, 64);
        const int oi = __shfl_xor(bi, 1, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (h == 0 && m < P.m) {
        const int n = m / hw, p = m - n * hw;
        const int A = P.p[5], K = P.p[1];
        const int a = P.p[2] + p * A + n0 / BN;                // anchor index inside the image
        const f4 rv = *reinterpret_cast<const f4 *>(row) + *reinterpret_cast<const f4 *>(bias);
        const float r[4] = {rv[0], rv[1], rv[2], rv[3]};
        const f4 av = *reinterpret_cast<const f4 *>(P.anchors + (size_t)a * 4);
        const float an[4] = {av[0], av[1], av[2], av[3]};
        float bx[4];
        const float sc = ssddev::decode_anchor(r, an, best, bx);
        const size_t o = (size_t)n * K + a;
        *reinterpret_cast<f4 *>(P.dec_boxes + o * 4) = f4{bx[0], bx[1], bx[2], bx[3]};
        P.dec_score[o] = sc;
        P.dec_cls[o] = bi;
        P.dec_keys[o] = sc >= P.dec_thr ? sc : -1.f;
    }
}

// YOLOv5 Detect head, one anchor per 96-channel tile (columns x, y, w, h, objectness, C classes, zeros): what tools/yolov5.py:120-128
// makes of a decoded row -- cls *= obj, argmax, confidence -- in the epilogue, so the [rows][5 + C] f32 matrix (8.6 MB per frame,
// written by the heads and read straight back by yolo_conf_k) never exists.  Per row: the box (conv_epilogue's expressions), the
// confidence and the class, np.argmax's rules as yolo_conf_k restates them (first maximum; a NaN product is the answer, the first
// one's index).  Two lanes per pixel scan 40 classes each.
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void yolo_head_finish(const ConvP &P, f4 (&acc)[NI][MI], _Float16 *lds, int m0, int n0, int hw) {
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16, T = WM * WN * 64;
    static_assert(BN == 96 && T == 2 * BM, "one anchor per channel tile, two lanes per pixel");
    constexpr int OROW = BN + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    float *ot = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
        for (int a = 0; a < NI; ++a)
            *reinterpret_cast<f4 *>(ot + ((wm * MI + b) * 16 + fr) * OROW + (wn * NI + a) * 16 + fq * 4) = acc[a][b];
    __syncthreads();
    const int pl = tid >> 1, h = tid & 1;
    const int m = m0 + pl;
    const int no = P.p[0], C = no - 5;
    const float *row = ot + pl * OROW;
    const float *bias = P.bias + n0;
    const float sobj = fast_sigmoid(row[4] + bias[4]);
    float best = -__builtin_inff();
    int bi = 0x7fffffff, nan_i = 0x7fffffff;
    const int half = (C + 1) >> 1;                              // lane h scans classes [h * half, min(C, h * half + half)), ascending
    const int c_lo = h * half, c_hi = min(C, c_lo + half);
    for (int c4 = (5 + c_lo) & ~3; c4 < 5 + c_hi; c4 += 4) {
        const f4 v = *reinterpret_cast<const f4 *>(row + c4) + *reinterpret_cast<const f4 *>(bias + c4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = c4 + q - 5;
            if (ci < c_lo || ci >= c_hi) continue;
            const float pq = fast_sigmoid(v[q]) * sobj;
            if (pq != pq) nan_i = min(nan_i, ci);
            if (pq > best) { best = pq; bi = ci; }
        }
    }
    {
        const float ob = __shfl_xor(best, 1, 64);
        const int oi = __shfl_xor(bi, 1, 64), on = __shfl_xor(nan_i, 1, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        nan_i = min(nan_i, on);
    }
    if (h == 0 && m < P.m) {
        const int n = m / hw, p = m - n * hw;
        const int py = p / P.wo, px = p - py * P.wo;
        const int an = n0 / BN;
        const size_t r = (size_t)n * P.p[1] + P.p[2] + (size_t)an * hw + p;
        const f4 rv = *reinterpret_cast<const f4 *>(row) + *reinterpret_cast<const f4 *>(bias);
        float sg[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sg[q] = fast_sigmoid(rv[q]);
        f4 bx;
#pragma unroll
        for (int q = 0; q < 4; ++q) bx[q] = yolo_box_col(q, sg[q], (float)px, (float)py, P.f[6], P.f[2 * an], P.f[2 * an + 1], P.f[7], (float)P.p[4]);
        *reinterpret_cast<f4 *>(P.dec_boxes + r * 4) = bx;
        const bool has_nan = nan_i != 0x7fffffff;
        P.dec_score[r] = has_nan ? __builtin_nanf("") : best;
        P.dec_cls[r] = has_nan ? nan_i : bi;
    }
}

// Epilogue without the LDS transposition: used when the weight rows of the tile were staged in the
// fragment order of rw_weight_row (conv_glds_k does that for plain f16 outputs), so the lane that owns
// rows fq*4.. of fragments 2g and 2g+1 holds 8 consecutive output channels of its pixel.
template <int WM, int WN, int MI, int NI, int ACT>
__device__ __forceinline__ void conv_finish_direct_act(const ConvP &P, f4 (&acc)[NI][MI], int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int g = 0; g < NI / 2; ++g) {
        const int co = n0 + (wn * NI + 2 * g) * 16 + fq * 8;
        if (co >= P.cout_pad) continue;
        const Epi8 E = epi8_load(P, co);
        h8 rp[MI];
        if (P.res) {
#pragma unroll
            for (int b = 0; b < MI; ++b) {
                const int m = min(m0 + (wm * MI + b) * 16 + fr, P.m - 1);
                rp[b] = *reinterpret_cast<const h8 *>(P.res + (size_t)m * P.cs_res + P.coff_res + co);
            }
        }
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = m0 + (wm * MI + b) * 16 + fr;
            if (m >= P.m) continue;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[2 * g][b][r]; v[4 + r] = acc[2 * g + 1][b][r]; }
            conv_epilogue_f16x8<ACT>(P, E, m, co, v, P.res ? &rp[b] : nullptr);
        }
    }
}

template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void conv_finish_direct(const ConvP &P, f4 (&acc)[NI][MI], int m0, int n0) {
    switch (P.act) {
        case ACT_NONE: conv_finish_direct_act<WM, WN, MI, NI, ACT_NONE>(P, acc, m0, n0); break;
        case ACT_RELU6: conv_finish_direct_act<WM, WN, MI, NI, ACT_RELU6>(P, acc, m0, n0); break;
        case ACT_ELU: conv_finish_direct_act<WM, WN, MI, NI, ACT_ELU>(P, acc, m0, n0); break;
        case ACT_SILU: conv_finish_direct_act<WM, WN, MI, NI, ACT_SILU>(P, acc, m0, n0); break;
        default: conv_finish_direct_act<WM, WN, MI, NI, -1>(P, acc, m0, n0);
    }
}

// Block tile: (WM*MI*16) pixels x (WN*NI*16) output channels, K step BK (32 for shallow K, else 64:
// two MFMA k-slices per barrier); blockIdx.z = K split.  Staged rows carry 8 halves of padding.
template <int WM, int WN, int MI, int NI, int BK>
__global__ __launch_bounds__(WM *WN * 64, 2) void conv_mfma_k(const ConvP P) {
    constexpr int LDS_ROW = BK + 8;
    constexpr int T = WM * WN * 64;
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    constexpr int CPR = BK / 8;                               // 16-byte chunks per staged row
    constexpr int XCH = (BM * CPR + T - 1) / T, WCH = (BN * CPR + T - 1) / T;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];         // 2 * (BM + BN) * LDS_ROW halves
    _Float16 *xs = lds, *ws = lds + 2 * BM * LDS_ROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // logical tile order: one contiguous slice per XCD, and inside it all channel tiles of a pixel tile
    // back to back -- the pixel tile is then fetched from HBM once and re-read by the other channel
    // tiles from that XCD's L2 (the weights, a few hundred KiB, stay L2-resident anyway)
    const unsigned lin = dd_xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int m0 = (int)(lin / gridDim.y) * BM, n0 = (int)(lin % gridDim.y) * BN;
    const int hw = P.ho * P.wo;
    const int ksteps = P.kpad / BK;
    const int per = (ksteps + P.splitk - 1) / P.splitk;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);

    // per-thread staging coordinates; (x_kh, x_kw, x_c) walk the K axis incrementally
    int x_n[XCH], x_iy0[XCH], x_ix0[XCH], x_kh[XCH], x_kw[XCH], x_c[XCH];
    bool x_ok[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * T;
        const int m = m0 + c / CPR;
        x_ok[i] = (c < BM * CPR) && (m < P.m);
        const int mm = x_ok[i] ? m : 0;
        x_n[i] = mm / hw;
        const int r = mm - x_n[i] * hw;
        const int oy = r / P.wo, ox = r - oy * P.wo;
        x_iy0[i] = oy * P.stride - P.pad_t;
        x_ix0[i] = ox * P.stride - P.pad_l;
        const int k = ks0 * BK + (c % CPR) * 8;
        const int tap = k / P.cin;
        x_c[i] = k - tap * P.cin;
        x_kh[i] = tap / P.kw;
        x_kw[i] = tap - x_kh[i] * P.kw;
    }

    h8 xr[XCH], wr[WCH];
    auto load_step = [&](int ks) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            const int iy = x_iy0[i] + x_kh[i], ix = x_ix0[i] + x_kw[i];
            if (x_ok[i] && x_kh[i] < P.kh && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W)
                v = *reinterpret_cast<const h8 *>(P.in + ((size_t)(x_n[i] * P.H + iy) * P.W + ix) * P.cs_in + P.coff_in + x_c[i]);
            xr[i] = v;
            x_c[i] += BK;                                     // advance to the next K step
            while (x_c[i] >= P.cin) {
                x_c[i] -= P.cin;
                if (++x_kw[i] == P.kw) { x_kw[i] = 0; ++x_kh[i]; }
            }
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + i * T;
            if (c < BN * CPR)
                wr[i] = *reinterpret_cast<const h8 *>(P.w + (size_t)(n0 + c / CPR) * P.kpad + ks * BK + (c % CPR) * 8);
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + i * T;
            if (c < BM * CPR) *reinterpret_cast<h8 *>(xs + (buf * BM + c / CPR) * LDS_ROW + (c % CPR) * 8) = xr[i];
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + i * T;
            if (c < BN * CPR) *reinterpret_cast<h8 *>(ws + (buf * BN + c / CPR) * LDS_ROW + (c % CPR) * 8) = wr[i];
        }
    };

    f4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    if (ks0 < ks1) {
        load_step(ks0);
        store_step(0);
        __syncthreads();
        for (int ks = ks0; ks < ks1; ++ks) {
            const int buf = (ks - ks0) & 1;
            if (ks + 1 < ks1) load_step(ks + 1);              // global loads in flight under the MFMAs
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                h8 xf[MI], wf[NI];
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    xf[b] = *reinterpret_cast<const h8 *>(xs + (buf * BM + (wm * MI + b) * 16 + fr) * LDS_ROW + kk * 32 + (fq << 3));
#pragma unroll
                for (int a = 0; a < NI; ++a)
                    wf[a] = *reinterpret_cast<const h8 *>(ws + (buf * BN + (wn * NI + a) * 16 + fr) * LDS_ROW + kk * 32 + (fq << 3));
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            }
            if (ks + 1 < ks1) store_step(buf ^ 1);
            __syncthreads();
        }
    }

    conv_finish<WM, WN, MI, NI>(P, acc, lds, m0, n0, hw);
}

// global_load_lds_dwordx4: each lane moves 16 bytes from its own global address to
// (wave-uniform LDS base) + lane * 16.  The builtin only exists in the device pass of hipcc.
__device__ __forceinline__ void lds_fill16(const _Float16 *g, _Float16 *lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(g, lds_wave_base, 16, 0, 0);
#endif
}

// Direct-to-LDS variant (K step 64, any padded Cin: each lane moves one 16-byte chunk = 8 channels of one
// filter tap, so a staged 128-byte row may be gathered from several taps).  Operand tiles go
// HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write); one wave instruction fills
// eight consecutive 128-byte rows.  Rows are unpadded, so the 16-byte chunks of a row are XOR-swizzled
// with (row & 7) -- on the SOURCE address for the fill and on the ds_read_b128 address for the
// fragments -- which makes every 16-lane read group hit 16 distinct 4-bank groups.
// FM (fill mode) 1: 1x1, stride 1, no padding, Cin % 64 == 0 -- the pixel operand is a plain row-major matrix: no
// tap walk, no bounds tests; 2: up to 3x3 with Cin % 64 == 0 -- a K step lies inside one filter tap, the same one
// for every lane, so the tap walk is scalar and a lane does one add and a two-bit test per row group; 0: anything
// else.  (A scalar walk over tap PAIRS for Cin = 32 -- YOLOv5's 3x3 stride-2 32 -> 64 layer -- changed nothing: 399 -> 407 us at 128 frames.)  (The general per-lane walk costs more issue slots per K step than the MFMAs: 19x19x512 -> 512 went
// 24.9 -> 21.8 us with FM 1, the MARS 16x8x64 -> 64 layers 28.8 -> 24.5 us with FM 2.)
template <int WM, int WN, int MI, int NI, int FM = 0, int DEC = 0>       // DEC: one anchor per 96-channel tile, decode in the epilogue (1 SSD head, 2 YOLOv5 Detect)
__global__ __launch_bounds__(WM *WN * 64, 2) void conv_glds_k(const ConvP P) {
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    constexpr int XG = BM / 8 / NW, WG = BN / 8 / NW;           // 8-row groups per wave
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split into 8-row groups per wave");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];         // [2][BM + BN][64] halves
    _Float16 *xs = lds, *ws = lds + 2 * BM * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const unsigned lin = dd_xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int m0 = (int)(lin / gridDim.y) * BM, n0 = (int)(lin % gridDim.y) * BN;      // channel tiles fastest (see conv_mfma_k)
    const int hw = P.ho * P.wo;
    const int ksteps = P.kpad >> 6;
    const int per = (ksteps + P.splitk - 1) / P.splitk;
    const int ks0 = blockIdx.z * per, ks1 = min(ksteps, ks0 + per);

    const int rr = lane >> 3, pp = lane & 7;                    // row inside the 8-row group, LDS chunk position
    const int gchunk = (pp ^ rr) * 8;                           // which 8 halves of the global row land there
    int x_n[XG], x_iy0[XG], x_ix0[XG];
    bool x_ok[XG];
#pragma unroll
    for (int i = 0; i < XG; ++i) {
        const int m = m0 + (wave * XG + i) * 8 + rr;
        x_ok[i] = m < P.m;
        const int mm = x_ok[i] ? m : 0;
        x_n[i] = mm / hw;
        const int r = mm - x_n[i] * hw;
        const int oy = r / P.wo, ox = r - oy * P.wo;
        x_iy0[i] = oy * P.stride - P.pad_t;
        x_ix0[i] = ox * P.stride - P.pad_l;
    }
    const bool direct = (NI % 2 == 0) && P.epi == EPI_F16 && P.splitk == 1;     // plain f16 output: no LDS transposition pass
    const _Float16 *wbase[WG];
#pragma unroll
    for (int i = 0; i < WG; ++i) {
        const int L = (wave * WG + i) * 8 + rr;                 // LDS row of the tile; direct: fragment order (see rw_weight_row)
        const int row = direct ? ((L & ~31) | (((L & 15) >> 2) << 3) | (((L >> 4) & 1) << 2) | (L & 3)) : L;
        wbase[i] = P.w + (size_t)(n0 + row) * P.kpad + gchunk;
    }

    // This lane always moves the same logical 16-byte chunk (8 channels) of a K step; (l_kh, l_kw, l_c)
    // walk the (tap, channel) position of that chunk along K, so any Cin that is a multiple of 8 works
    // (Cin = 32: a 64-wide step covers two taps, Cin = 8: eight taps).
    int l_kh, l_kw, l_c;
    {
        const int k = (ks0 << 6) + gchunk;
        const int tap = k / P.cin;
        l_c = k - tap * P.cin;
        l_kh = tap / P.kw;
        l_kw = tap - l_kh * P.kw;
    }
    const _Float16 *xbase[XG];                                  // FM 1/2: this lane's chunk of its pixel rows at tap (0,0), K = 0
    unsigned rmask[XG], cmask[XG];                              // FM 2: bit k set = filter row / column k of the pixel is inside the image
#pragma unroll
    for (int i = 0; i < XG; ++i) {
        const int m = m0 + (wave * XG + i) * 8 + rr;
        if constexpr (FM == 2) {
            xbase[i] = P.in + ((ptrdiff_t)(x_n[i] * P.H + x_iy0[i]) * P.W + x_ix0[i]) * P.cs_in + P.coff_in + gchunk;
            unsigned rm = 0, cm = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (x_ok[i] && x_iy0[i] + k >= 0 && x_iy0[i] + k < P.H) rm |= 1u << k;
                if (x_ix0[i] + k >= 0 && x_ix0[i] + k < P.W) cm |= 1u << k;
            }
            rmask[i] = rm; cmask[i] = cm;
        } else {
            xbase[i] = x_ok[i] ? P.in + (size_t)m * P.cs_in + P.coff_in + gchunk : nullptr;
        }
    }
    int s_c = 0, s_kh = 0, s_kw = 0;                            // FM 2: scalar (tap, channel) position of the K step
    if constexpr (FM == 2) {
        const int k = ks0 << 6, tap = k / P.cin;
        s_c = k - tap * P.cin; s_kh = tap / P.kw; s_kw = tap - s_kh * P.kw;
    }
    auto fill = [&](int ks, int buf) {                          // called with consecutive ks
        if constexpr (FM == 2) {                                 // Cin % 64 == 0: the K step lies inside one tap, the same for every lane
            const ptrdiff_t off = (ptrdiff_t)(s_kh * P.W + s_kw) * P.cs_in + s_c;      // wave-uniform
#pragma unroll
            for (int i = 0; i < XG; ++i)
                lds_fill16((((rmask[i] >> s_kh) & (cmask[i] >> s_kw)) & 1u) ? xbase[i] + off : P.zero,
                           xs + (size_t)(buf * BM + (wave * XG + i) * 8) * 64);
            s_c += 64;
            if (s_c >= P.cin) { s_c = 0; if (++s_kw == P.kw) { s_kw = 0; ++s_kh; } }
            const int k = ks << 6;
#pragma unroll
            for (int i = 0; i < WG; ++i) lds_fill16(wbase[i] + k, ws + (size_t)(buf * BN + (wave * WG + i) * 8) * 64);
            return;
        }
        if constexpr (FM == 1) {                                 // one add per row group and step
            const int k = ks << 6;
#pragma unroll
            for (int i = 0; i < XG; ++i)
                lds_fill16(xbase[i] ? xbase[i] + k : P.zero, xs + (size_t)(buf * BM + (wave * XG + i) * 8) * 64);
#pragma unroll
            for (int i = 0; i < WG; ++i) lds_fill16(wbase[i] + k, ws + (size_t)(buf * BN + (wave * WG + i) * 8) * 64);
            return;
        }
#pragma unroll
        for (int i = 0; i < XG; ++i) {
            const int iy = x_iy0[i] + l_kh, ix = x_ix0[i] + l_kw;
            const bool ok = x_ok[i] && l_kh < P.kh && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
            const _Float16 *g = ok ? P.in + ((size_t)(x_n[i] * P.H + iy) * P.W + ix) * P.cs_in + P.coff_in + l_c : P.zero;
            lds_fill16(g, xs + (size_t)(buf * BM + (wave * XG + i) * 8) * 64);
        }
        l_c += 64;
        while (l_c >= P.cin) {
            l_c -= P.cin;
            if (++l_kw == P.kw) { l_kw = 0; ++l_kh; }
        }
        const int k = ks << 6;
#pragma unroll
        for (int i = 0; i < WG; ++i) lds_fill16(wbase[i] + k, ws + (size_t)(buf * BN + (wave * WG + i) * 8) * 64);
    };

    f4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
    if (ks0 < ks1) {
        fill(ks0, 0);
        __syncthreads();                                          // waits for the fills (vmcnt(0) before s_barrier)
        for (int ks = ks0; ks < ks1; ++ks) {
            const int buf = (ks - ks0) & 1;
            if (ks + 1 < ks1) fill(ks + 1, buf ^ 1);              // lands while this step's MFMAs run
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                h8 xf[MI], wf[NI];
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    xf[b] = *reinterpret_cast<const h8 *>(xs + (size_t)(buf * BM + (wm * MI + b) * 16 + fr) * 64 + (((kk << 2) + fq) ^ sw) * 8);
#pragma unroll
                for (int a = 0; a < NI; ++a)
                    wf[a] = *reinterpret_cast<const h8 *>(ws + (size_t)(buf * BN + (wn * NI + a) * 16 + fr) * 64 + (((kk << 2) + fq) ^ sw) * 8);
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    if constexpr (DEC == 1) {
        ssd_head_finish<WM, WN, MI, NI>(P, acc, lds, m0, n0, hw);
    } else if constexpr (DEC == 2) {
        yolo_head_finish<WM, WN, MI, NI>(P, acc, lds, m0, n0, hw);
    } else if constexpr (MI * NI >= 12) {
        // 48 x 64 per wave and up: only the untransposed epilogue is compiled in (the launcher sends nothing else
        // here); with the general one the accumulator array stops being promoted to registers and every K step
        // stores all its fragments to scratch (seen with 64 x 64 per wave: 4x slower)
        conv_finish_direct<WM, WN, MI, NI>(P, acc, m0, n0);
    } else {
        if (direct) conv_finish_direct<WM, WN, MI, NI>(P, acc, m0, n0);
        else conv_finish<WM, WN, MI, NI>(P, acc, lds, m0, n0, hw);
    }
}

// ---------------------------------------------------------------------------------------------------
// Spatially tiled 3x3 kernels with register-resident weights.
//
// For 32 input channels one filter tap is exactly one MFMA k-slice, and 32 output channels are two
// 16-row weight fragments, so the whole 3x3x32x32 filter is 18 fragments = 72 VGPRs per lane: it is
// loaded once per wave and never staged.  The block copies its input patch (tile + halo) into LDS
// once -- every input byte crosses L2->LDS one time instead of nine -- and each pixel fragment is a
// single ds_read_b128 per tap.  LDS image: four 8-channel planes [chunk][patch pixel] x 16 bytes with
// a plane pitch that is a multiple of 256 bytes: the 16 lanes of every ds_read_b128 lane group then
// fall on 16 different 16-byte slots of the 256-byte bank row, whatever the tap offset.
//
// Weight rows are taken in the order (fr>>2)*8 + a*4 + (fr&3) for fragment a, so the lane that owns
// accumulator rows fq*4..fq*4+3 of fragments 2g and 2g+1 holds output channels g*32 + fq*8 .. +7:
// one 16-byte store per pixel and lane, 64 contiguous bytes per pixel, 1 KiB per fragment.
__device__ __forceinline__ int rw_weight_row(int a, int fr) { return (a >> 1) * 32 + (fr >> 2) * 8 + (a & 1) * 4 + (fr & 3); }

constexpr int RW_FILL = 11;                                    // 16 patch pixels x 4 chunks per wave pass: <= 704 patch pixels
constexpr int RW_MAX_PATCH = RW_FILL * 64;
constexpr int RW_MB = 4;                                       // pixel fragments in flight per wave

template <int NCO, int TW, int ACT, bool POOL, int EF = -1, bool S2D8 = false>     // TW: tile width when known at compile time (tap offsets
__global__ __launch_bounds__(256, 2) void conv3x3_rw_k(const ConvP P, const int total_tiles) {   // become ds_read immediates), else 0
    // S2D8: the input is the u8 frame [2 H][2 W][3] behind a space-to-depth-2 input op (YOLOv5's Focus) that was folded into this launch:
    // the patch fill slices, normalises and zero-pads it itself (input_s2d_chunks_k's arithmetic), so neither the input kernel nor its
    // 32-channel f16 tensor (6.5 MB per frame written and read back) exist.
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    // Pixel fragments in flight per wave.  The pooled conv1_2 tile is 17 x 32 pixels = 34 fragments: in groups of 4 that is
    // 9 groups for 4 waves -- three rounds of which the last is a quarter full; groups of 3 give 12 groups = three full
    // rounds of 3 (9 instead of 12 fragment slots per wave and tile).  The residual / two-output / run-time
    constexpr int MB = POOL ? 3 : EF == 0 ? RW_MB : 2;
                                                               // epilogues need the registers: 2 (with 4 they spilled 9-19 VGPRs to scratch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int tw = TW ? TW : P.tw;
    const int PW = tw + 2, npix = (P.th + 2) * PW;
    const int plane = (npix * 8 + 127) & ~127;                  // halves per 8-channel plane (256-byte multiple)
    _Float16 *stage = lds + 4 * plane;                          // POOL: the activated output tile [th*tw][NCO*16]
    // exact x / d for x < 1024, d <= 34 without an integer division per lane
    const unsigned rcp_pw = (65536u + PW - 1) / PW, rcp_tw = (65536u + tw - 1) / tw;
    const int tiles_per_image = P.tiles_x * P.tiles_y;

    h8 wf[9][NCO];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < NCO; ++a)
            wf[t][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)rw_weight_row(a, fr) * P.kpad + t * 32 + fq * 8);
    Epi8 E[NCO / 2];
#pragma unroll
    for (int g = 0; g < NCO / 2; ++g) E[g] = epi8_load(P, g * 32 + fq * 8);

    // Persistent blocks: tile t of this block is logical tile dd_xcd_remap(t) (each XCD walks one contiguous
    // range of images, so halo rows shared by neighbouring tiles meet in one L2).  The next tile's patch
    // is fetched into registers while the current one is multiplied.
    // Patch fill: a wave pass covers 16 consecutive patch pixels x 4 chunks; inside it each 8-lane group
    // takes 8 consecutive pixels of one chunk (the ds_write_b128 lane groups are 8 contiguous lanes).
    // S2D8: only chunks 0 and 1 carry data (12 of 32 channels); a wave pass covers 32 patch pixels x 2 chunks, the planes of chunks
    // 2 and 3 are zeroed once per block and never touched again
    const int fc = S2D8 ? (lane >> 3) & 1 : (lane >> 3) & 3, fpl = S2D8 ? (lane & 7) + ((lane >> 4) << 3) : (lane & 7) + ((lane >> 5) << 3);
    constexpr int PPP = S2D8 ? 32 : 16;                         // patch pixels per wave pass
    constexpr int NFILL = S2D8 ? (RW_FILL + 1) / 2 : RW_FILL;
    if constexpr (S2D8) {
        const h8 z8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        for (int i = tid; i < 2 * (plane >> 3); i += 256) *reinterpret_cast<h8 *>(lds + 2 * plane + i * 8) = z8;
    }
    h8 v[NFILL];
    auto fetch = [&](int t) {
        const int lt = (int)dd_xcd_remap((unsigned)t, (unsigned)total_tiles);
        const int n = lt / tiles_per_image, r = lt - n * tiles_per_image;
        const int ty = r / P.tiles_x, tx = r - ty * P.tiles_x;
        const int ys = ty * (POOL ? P.th - 1 : P.th) - P.pad_t, xs = tx * tw - P.pad_l;
        const _Float16 *img = P.in + (size_t)n * P.H * P.W * P.cs_in + P.coff_in + fc * 8;
        const uint8_t *img8 = S2D8 ? P.src8 + (size_t)n * P.H * P.W * 12 : nullptr;        // frame n: [2 H][2 W][3] bytes
#pragma unroll
        for (int i = 0; i < NFILL; ++i) {
            const int pix = (wave + 4 * i) * PPP + fpl;
            const int py = (int)((pix * rcp_pw) >> 16);
            const int y = ys + py, x = xs + pix - py * PW;
            const bool ok = pix < npix && y >= 0 && y < P.H && x >= 0 && x < P.W;      // else: a line of zeros, no branch
            if constexpr (S2D8) {
                // source pixels (2y, 2x), (2y, 2x + 1) are six adjacent bytes of row 2y, likewise in row 2y + 1: one unaligned 8-byte load
                // per row (the last focus pixel of a frame would read two bytes past it: it takes its bytes one by one).  Raw bytes
                // travel in the fill registers, the conversion happens at the LDS write.  Chunks 2 and 3 are the zero channels.
                typedef unsigned u2u __attribute__((ext_vector_type(2), aligned(1)));
                typedef unsigned u4t __attribute__((ext_vector_type(4)));
                u4t raw = {0u, 0u, 0u, 0u};
                if (ok) {
                    const uint8_t *p0 = img8 + ((size_t)(2 * y) * (2 * P.W) + 2 * x) * 3, *p1 = p0 + (size_t)P.W * 6;
                    if (y == P.H - 1 && x == P.W - 1) {
                        raw[0] = (unsigned)p0[0] | ((unsigned)p0[1] << 8) | ((unsigned)p0[2] << 16) | ((unsigned)p0[3] << 24);
                        raw[1] = (unsigned)p0[4] | ((unsigned)p0[5] << 8);
                        raw[2] = (unsigned)p1[0] | ((unsigned)p1[1] << 8) | ((unsigned)p1[2] << 16) | ((unsigned)p1[3] << 24);
                        raw[3] = (unsigned)p1[4] | ((unsigned)p1[5] << 8);
                    } else {
                        const u2u a = *reinterpret_cast<const u2u *>(p0), b = *reinterpret_cast<const u2u *>(p1);
                        raw[0] = a[0]; raw[1] = a[1]; raw[2] = b[0]; raw[3] = b[1];
                    }
                    raw[1] |= 0x80000000u;                        // marks a live chunk (bits the six bytes do not use)
                }
                v[i] = __builtin_bit_cast(h8, raw);
            } else {
                v[i] = *reinterpret_cast<const h8 *>(ok ? img + ((size_t)y * P.W + x) * P.cs_in : P.zero);
            }
        }
    };
    auto s2d_chunk = [&](const h8 &rawv) -> h8 {                  // chunk fc of a focus pixel from its six + six bytes
        typedef unsigned u4t __attribute__((ext_vector_type(4)));
        const u4t r = __builtin_bit_cast(u4t, rawv);
        h8 o = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (!(r[1] & 0x80000000u)) return o;
        auto byte = [&](int row, int k) { return (float)((k < 4 ? r[2 * row] >> (8 * k) : r[2 * row + 1] >> (8 * (k - 4))) & 255u); };
        auto cv = [&](float b) { return (_Float16)((b - P.in_mean) * P.in_scale); };
        if (fc == 0) {                                           // (y0x0) rgb, (y1x0) rgb, (y0x1) rg
            o[0] = cv(byte(0, 0)); o[1] = cv(byte(0, 1)); o[2] = cv(byte(0, 2));
            o[3] = cv(byte(1, 0)); o[4] = cv(byte(1, 1)); o[5] = cv(byte(1, 2));
            o[6] = cv(byte(0, 3)); o[7] = cv(byte(0, 4));
        } else {                                                 // (y0x1) b, (y1x1) rgb, zeros
            o[0] = cv(byte(0, 5));
            o[1] = cv(byte(1, 3)); o[2] = cv(byte(1, 4)); o[3] = cv(byte(1, 5));
        }
        return o;
    };

    const int tile_px = P.th * tw;
    const int nfrag = (tile_px + 15) >> 4;
    int t = blockIdx.x;
    fetch(t);
    for (;;) {
#pragma unroll
        for (int i = 0; i < NFILL; ++i) {
            const int pix = (wave + 4 * i) * PPP + fpl;
            if (pix < npix) *reinterpret_cast<h8 *>(lds + fc * plane + pix * 8) = S2D8 ? s2d_chunk(v[i]) : v[i];
        }
        __syncthreads();
        const int lt = (int)dd_xcd_remap((unsigned)t, (unsigned)total_tiles);
        const int n = lt / tiles_per_image, rt = lt - n * tiles_per_image;
        const int tile_y = rt / P.tiles_x;
        const int y0 = tile_y * (POOL ? P.th - 1 : P.th), x0 = (rt - tile_y * P.tiles_x) * tw;
        const int tn = t + gridDim.x;
        if (tn < total_tiles) fetch(tn);

        for (int f0 = wave * MB; f0 < nfrag; f0 += 4 * MB) {
            int base[MB], mrow[MB];
            bool ok[MB];
#pragma unroll
            for (int b = 0; b < MB; ++b) {
                const int p = (f0 + b) * 16 + fr;
                const int ty = (int)((p * rcp_tw) >> 16), tx = p - ty * tw;
                ok[b] = p < tile_px && y0 + ty < P.ho && x0 + tx < P.wo;
                base[b] = fq * plane + (ok[b] ? ty * PW + tx : 0) * 8;     // top-left tap of the pixel, in halves
                mrow[b] = (n * P.ho + y0 + ty) * P.wo + x0 + tx;
            }
            f4 acc[NCO][MB];                                  // start from the bias of the lane's output channels
#pragma unroll
            for (int a = 0; a < NCO; ++a)
#pragma unroll
                for (int b = 0; b < MB; ++b) acc[a][b] = (a & 1) ? E[a >> 1].b1 : E[a >> 1].b0;
            // residual rows of this group's pixels: requested before the MFMAs, so the round trip runs under them (a load
            // inside the epilogue cannot move above the previous pixel's stores -- they may alias -- and each pixel would
            // wait for memory on its own: PMC had the two-output variant parked on s_waitcnt 65 % of its cycles)
            h8 rp[MB][NCO / 2];
            const bool has_res = !POOL && (EF < 0 ? P.res != nullptr : EF == 1);
            if (has_res) {
#pragma unroll
                for (int b = 0; b < MB; ++b)
#pragma unroll
                    for (int g = 0; g < NCO / 2; ++g)
                        rp[b][g] = *reinterpret_cast<const h8 *>(P.res + (size_t)(ok[b] ? mrow[b] : 0) * P.cs_res + P.coff_res + g * 32 + fq * 8);
            }
            // The pixel fragments of tap t + 1 are requested BEFORE the MFMAs of tap t (two register sets, the order pinned
            // with sched_barrier): left to itself hipcc reads a fragment one or two instructions ahead of the MFMA pair that
            // uses it, so every 32 cycles of matrix work waited ~100 cycles for LDS (SQ_VALU_MFMA_BUSY 0.27-0.29 at two
            // waves per SIMD).
            h8 xf[2][MB];
#pragma unroll
            for (int b = 0; b < MB; ++b) xf[0][b] = *reinterpret_cast<const h8 *>(lds + base[b]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) {
                    const int off = (((t + 1) / 3) * PW + (t + 1) % 3) * 8;
#pragma unroll
                    for (int b = 0; b < MB; ++b) xf[(t + 1) & 1][b] = *reinterpret_cast<const h8 *>(lds + base[b] + off);
                }
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                for (int a = 0; a < NCO; ++a)
#pragma unroll
                    for (int b = 0; b < MB; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][a], xf[t & 1][b], acc[a][b], 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
#pragma unroll
            for (int b = 0; b < MB; ++b) {
                if (!ok[b]) continue;
#pragma unroll
                for (int g = 0; g < NCO / 2; ++g) {
                    float o[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { o[r] = acc[2 * g][b][r]; o[4 + r] = acc[2 * g + 1][b][r]; }
                    if constexpr (POOL) {                         // activated tile -> LDS, pooled below
                        const int act = ACT < 0 ? P.act : ACT;
                        h8 hv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            hv[r] = (_Float16)apply_act(o[r], act);
                            hv[4 + r] = (_Float16)apply_act(o[4 + r], act);
                        }
                        *reinterpret_cast<h8 *>(stage + ((f0 + b) * 16 + fr) * (NCO * 16) + g * 32 + fq * 8) = hv;
                    } else {
                        conv_epilogue_f16x8<ACT, false, EF>(P, E[g], mrow[b], g * 32 + fq * 8, o, has_res ? &rp[b][g] : nullptr);
                    }
                }
            }
        }
        if constexpr (POOL) {
            // 3x3 stride-2 VALID max pool of the tile: pooled row j of the tile covers tile rows 2j..2j+2
            // (the last tile row is the next tile's first -- one recomputed row per tile instead of a
            // round trip of the whole activation through HBM)
            __syncthreads();
            constexpr int CG = NCO * 2;                           // 8-channel groups
            const int prows = (P.th - 1) >> 1, ph = P.p[0], pw = P.p[1];
            for (int it = tid; it < prows * pw * CG; it += 256) {
                const int c = it % CG, q = it / CG;
                const int ox = q % pw, oyl = q / pw;
                const int oy = tile_y * prows + oyl;
                if (oy >= ph) continue;
                h8 best = *reinterpret_cast<const h8 *>(stage + ((2 * oyl) * tw + 2 * ox) * (NCO * 16) + c * 8);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if (dy == 0 && dx == 0) continue;
                        const h8 x = *reinterpret_cast<const h8 *>(stage + ((2 * oyl + dy) * tw + 2 * ox + dx) * (NCO * 16) + c * 8);
                        best = __builtin_elementwise_max(best, x);      // v_pk_max_f16
                    }
                *reinterpret_cast<h8 *>(static_cast<_Float16 *>(P.out) + ((size_t)(n * ph + oy) * pw + ox) * P.cs_out + P.coff_out + c * 8) = best;
            }
        }
        if (tn >= total_tiles) break;
        __syncthreads();                                          // every wave is done reading this patch
        t = tn;
    }
}

// ---------------------------------------------------------------------------------------------------
// 3x3 32->32 convolution (same padding) + activation + 3x3 stride-2 VALID max pool of a 32-pixel-wide image, ONE WAVE
// PER IMAGE streaming down its rows: no workgroup barrier anywhere, no staging registers, no activated tile in LDS.
//
//   * Round k computes conv rows 2k and 2k+1 from input rows 2k-1 .. 2k+2, which sit in the wave's own ring of eight
//     2 KiB row slots.  The two rows of round k+1 are requested at the top of round k by global_load_lds (4 DMAs of
//     1 KiB) and waited for with a counted vmcnt one round later.
//   * A row slot holds the even and the odd pixels separately, each as four 8-channel planes of 16 pixels
//     ([parity][plane][16] x 16 B = the lane order of the DMA).  An MFMA pixel fragment is the 16 even (x = 2 fr) or the
//     16 odd (x = 2 fr + 1) output pixels of a row, so the three taps of a filter row read
//         even outputs:  odd[fr-1]  even[fr]  odd[fr]            odd outputs:  even[fr]  odd[fr]  even[fr+1]
//     -- four distinct conflict-free ds_read_b128 per input row serve 12 (fragment, tap) pairs, and an input row is
//     read once for both conv rows that use it: 16 LDS reads per 72 MFMAs (conv3x3_rw_k: one read per MFMA pair --
//     there LDS bandwidth and the matrix pipe are the same bound).
//   * Horizontal pooling is lane-local: max(even[fr], odd[fr], even[fr+1]) with one DPP row shift; vertical pooling
//     carries max(h(2k), h(2k+1)) into the next round.  Activation and f16 rounding are monotonic, so they are applied
//     to the pooled maximum -- the same bits as pooling the activated f16 tile, with 4.3x fewer exponentials.
//   * Each accumulator sums its taps in the order t = 0..8 from the bias, as conv3x3_rw_k does: bit-identical output.
// The pooled row of round k is stored at the top of round k+1, after that round's wait: a store issued just before a
// counted wait would be waited for (stores and DMAs share vmcnt), this one has a whole round to complete.
//
// STEM = true: the input rows are not read at all -- the wave computes them from the u8 image (the network's first
// layer, stem_conv3_k's arithmetic in the same k slots: same bits) two rows ahead of the round that needs them.  Two image
// rows per round arrive as one dword per lane, are normalised to f16 into an 8-row ring (pitch 104 halves: x = -1 at
// halves 1..3, x = 0 at half 4, x = 32 at 100..102; the pad columns are zeroed once), and a stem fragment gathers its
// eight k operands with ds_read_u16 from one ring row per lane.  The 64x32x32 f16 tensor between the two layers (131 KB
// per image written and read back -- what bounds the DMA version) never exists.
// Round 5 (the kernel is bound by vector-instruction issue, and an MFMA and a vector instruction do not overlap on a SIMD:
// scripts/experiments/mfma_valu_overlap.hip, profiles/r05_mfma_valu_overlap.txt -- so every vector instruction that goes is time):
//   * the round loop is unrolled by four: a round consumes two of the eight ring slots, so inside a group of four rounds every slot
//     index -- of the row ring and of the crop ring -- is a literal and every LDS access is lane base + immediate offset (no address
//     arithmetic, no scalar bookkeeping per round);
//   * a row slot is 130 chunks: [64 even][64 odd][2 zero chunks]; the lane of pixel 0 / 31 points its left / right tap at a zero
//     chunk (the one in front of the slot / behind it: the bank phases of the chunks they replace, a read stays conflict free)
//     instead of masking what it read;
//   * the crop ring keeps rows 0..2 twice (rows 8..10), so the three image rows of a window are rows c, c+1, c+2 without a wrap;
//   * pooling maxima are v_maximum3_f32 (gfx950): the IEEE maxnum form canonicalises loop-carried and pooled operands first
//     (two extra v_max per maximum); the two differ only on NaN inputs.
//   * crop-ring pitch 160 halves = 80 banks = 16 (mod 32): lane group fq reads ring row c + fq, and the 16 lanes of a group sit 3 banks
//     apart (6 halves per pixel pair), so two groups of a 32-lane half are disjoint only at a row shift of 16 banks; at the tight pitch of
//     104 halves 39 % of the kernel's LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT), all of them these ds_read_u16 gathers.
constexpr int PR_SLOTS = 8, PR_SLOT_HALVES = 1040, PR_CROP_PITCH = 160, PR_CROP_ROWS = 11;
constexpr int pr_wave_halves(bool stem) { return 8 + PR_SLOTS * PR_SLOT_HALVES + (stem ? PR_CROP_ROWS * PR_CROP_PITCH : 0); }

template <int ACT, bool STEM>
__global__ __launch_bounds__(256, 2) void conv3x3_pool_rows_k(const ConvP P, const int n_units, const int split_dbg) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    _Float16 *ring = lds + (size_t)wave * pr_wave_halves(STEM) + 8;
    _Float16 *crop = ring + PR_SLOTS * PR_SLOT_HALVES;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};

    h8 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
            wf[t][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)rw_weight_row(a, fr) * P.kpad + t * 32 + fq * 8);
    const f4 bias0 = *reinterpret_cast<const f4 *>(P.bias + fq * 8), bias1 = *reinterpret_cast<const f4 *>(P.bias + fq * 8 + 4);
    h8 ws[2];
    f4 sb0, sb1;
    // the zero chunks: one in front of slot 0, two behind every slot (never written again: rows fill chunks 0..127 of a slot only)
    if (lane < 2 * PR_SLOTS) *reinterpret_cast<h8 *>(ring + (lane >> 1) * PR_SLOT_HALVES + 1024 + (lane & 1) * 8) = zero8;
    if (lane == 2 * PR_SLOTS) *reinterpret_cast<h8 *>(ring - 8) = zero8;
    if constexpr (STEM) {
#pragma unroll
        for (int a = 0; a < 2; ++a) ws[a] = *reinterpret_cast<const h8 *>(P.dw_w + rw_weight_row(a, fr) * 32 + fq * 8);
        sb0 = *reinterpret_cast<const f4 *>(P.dw_bias + fq * 8); sb1 = *reinterpret_cast<const f4 *>(P.dw_bias + fq * 8 + 4);
        if (lane < 2 * PR_CROP_ROWS)                              // pad columns of every crop-ring row, never written again
            *reinterpret_cast<h4 *>(crop + (lane >> 1) * PR_CROP_PITCH + ((lane & 1) ? 100 : 0)) = h4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    }
    const int ph = P.p[0], pw = P.p[1];
    const int H = P.H;
    // fragment addresses inside a row slot (halves): B = even[fr], C = odd[fr], A = odd[fr-1], D = even[fr+1]; x = -1 / x = 32 are the
    // zero chunk in front of the slot (phase 15, as odd[fr-1] of the other lanes' pattern) / behind it (phase 0)
    const _Float16 *pB = ring + (fq * 16 + fr) * 8, *pC = pB + 512;
    const _Float16 *pA = fr ? pC - 8 : ring - 8, *pD = fr < 15 ? pB + 8 : ring + 1024;
    const int act = ACT < 0 ? P.act : ACT;
    const int split = split_dbg & 255;
#ifdef DD_KERNEL_DBG
    const int dbg = split_dbg >> 8;                             // measurement aid (DD_PR_DBG): 1 no MFMAs, 2 no pooling epilogue, 4 no row production in the loop
#else
    constexpr int dbg = 0;                                      // compiled out of the product build: as a run-time value it put ~30 scalar branches
#endif                                                          // into every round of a kernel that is bound by instruction issue (build with -DDD_KERNEL_DBG)
    const int crow = lane >> 5, cdw = lane & 31;                // image rows: two per step, one dword per lane (24 of 32 live)
    // stem fragment gathers: lane group fq < 3 reads eight taps of crop-ring row c + fq, group 3 tap 8 of rows c .. c + 2 and the rest of row c + 2
    // (c = the ring row of image row y - 1, a literal inside the unrolled loop)
    const int cx = 1 + 6 * fr;                                  // window start (x - 1) of the even pixel x = 2 fr in a crop-ring row
    const _Float16 *gPb = crop + (fq < 3 ? fq : 2) * PR_CROP_PITCH + cx;
    const _Float16 *gA0 = fq == 3 ? crop + cx + 8 : gPb, *gA1 = fq == 3 ? crop + PR_CROP_PITCH + cx + 7 : gPb, *gA2 = fq == 3 ? crop + 2 * PR_CROP_PITCH + cx + 6 : gPb;
    _Float16 *cW = crop + 4 + cdw * 4;                           // where a lane's four normalised pixels-channels go inside a crop-ring row

    // unit = pooled rows [j0, j1) of one image (`split` units per image: few images still fill the chip; a unit's first
    // round only builds the carry, so a split costs one recomputed round per extra unit)
    for (int u = blockIdx.x * 4 + wave; u < n_units; u += gridDim.x * 4) {
        const int n = u / split, part = u - n * split;
        const int j0 = part * ph / split, j1 = (part + 1) * ph / split;
        const _Float16 *img = STEM ? nullptr : P.in + (size_t)n * H * 32 * P.cs_in + P.coff_in + fq * 8;
        const uint8_t *img8 = STEM ? P.src8 + (size_t)n * H * 96 : nullptr;
        auto fill_row = [&](int y, int slot) {                    // wave-uniform y (slot = (y + 1) & 7); rows outside the image: zero lines
            _Float16 *dst = ring + slot * PR_SLOT_HALVES;
            const bool ok = (unsigned)y < (unsigned)H;
            const _Float16 *src = img + ((size_t)(ok ? y : 0) * 32 + 2 * fr) * P.cs_in;
            lds_fill16(ok ? src : P.zero, dst);
            lds_fill16(ok ? src + P.cs_in : P.zero, dst + 512);
        };
        auto load_raw = [&](int r0) -> unsigned {                 // image rows r0, r0+1 (a select between addresses, not values)
            const int r = r0 + crow;
            const bool ok = (unsigned)r < (unsigned)H && cdw < 24;
            return *reinterpret_cast<const unsigned *>(ok ? img8 + (size_t)r * 96 + cdw * 4 : reinterpret_cast<const uint8_t *>(P.zero));
        };
        auto put_crop = [&](int r0, int row_lo, int row_hi, unsigned raw) {   // normalise image rows r0, r0+1 into crop-ring rows row_lo / row_hi = (r + 1) & 7
            const int r = r0 + crow;
            const bool ok = (unsigned)r < (unsigned)H;            // rows outside the image are the zero padding
            h4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (_Float16)(ok ? ((float)((raw >> (8 * q)) & 255u) - P.in_mean) * P.in_scale : 0.f);
            const int row = crow ? row_hi : row_lo;
            if (cdw < 24) {
                *reinterpret_cast<h4 *>(cW + row * PR_CROP_PITCH) = o;
                if (row < 3) *reinterpret_cast<h4 *>(cW + (row + 8) * PR_CROP_PITCH) = o;     // rows 0..2 a second time behind row 7
            }
        };
        auto stem_row = [&](int y, int slot, int c) {             // first-layer row y -> ring slot (y + 1) & 7 (wave-uniform y); c = y & 7
            _Float16 *dst = ring + slot * PR_SLOT_HALVES;
            if ((unsigned)y >= (unsigned)H) {                     // the second layer's zero padding
                *reinterpret_cast<h8 *>(dst + lane * 8) = zero8;
                *reinterpret_cast<h8 *>(dst + 512 + lane * 8) = zero8;
                return;
            }
            // image row y-1+d sits in crop-ring row c + d; k slots: group d < 3 = taps 0..7 of filter row d, group 3 = tap 8 of rows 0..2
            const int co = c * PR_CROP_PITCH;
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                h8 xf;
                xf[0] = gA0[co + 3 * par]; xf[1] = gA1[co + 1 + 3 * par]; xf[2] = gA2[co + 2 + 3 * par];
#pragma unroll
                for (int j = 3; j < 8; ++j) xf[j] = gPb[co + j + 3 * par];
                const f4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[0], xf, sb0, 0, 0, 0);       // onto the bias, as stem_conv3_k
                const f4 a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[1], xf, sb1, 0, 0, 0);
                h8 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[r] = (_Float16)apply_act(a0[r], act); o[4 + r] = (_Float16)apply_act(a1[r], act); }   // launcher: both layers ELU
                *reinterpret_cast<h8 *>(dst + par * 512 + (fq * 16 + fr) * 8) = o;
            }
        };
        const int kbeg = STEM ? j0 - 2 : j0;
        unsigned raw = 0;
        if constexpr (STEM) {
            put_crop(2 * j0 - 2, (2 * j0 - 1) & 7, (2 * j0) & 7, load_raw(2 * j0 - 2));
            raw = load_raw(2 * j0);
        } else {
            for (int y = 2 * j0 - 1; y <= 2 * j0 + 2; ++y) fill_row(y, (y + 1) & 7);
        }
        f4 carry[2];
        h8 pend;                                                  // pooled row of the previous round, not yet stored
        _Float16 *out_img = static_cast<_Float16 *>(P.out) + (size_t)n * ph * pw * P.cs_out + P.coff_out + fq * 8;
        for (int kk = kbeg & ~3; kk <= j1; kk += 4) {
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int k = kk + kq;                            // 2 k = 2 kq (mod 8): every slot index below is a literal
                if (k < kbeg || k > j1) continue;                 // wave-uniform
                if constexpr (STEM) {
                    // step k makes first-layer rows 2k+3 and 2k+4 (round k+1's new rows) from image rows 2k+2 .. 2k+5
                    if (k < j1 && !((dbg & 4) && k >= j0)) {
                        put_crop(2 * k + 4, (2 * kq + 5) & 7, (2 * kq + 6) & 7, raw);
                        raw = load_raw(2 * k + 6);
                        stem_row(2 * k + 3, (2 * kq + 4) & 7, (2 * kq + 3) & 7);
                        stem_row(2 * k + 4, (2 * kq + 5) & 7, (2 * kq + 4) & 7);
                    }
                    if (k < j0) continue;
                } else {
                    if (!(dbg & 4)) { fill_row(2 * k + 3, (2 * kq + 4) & 7); fill_row(2 * k + 4, (2 * kq + 5) & 7); }
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // rows up to 2k+2 have landed (this wave's own DMAs: no barrier)
#endif
                }
                if (k >= j0 + 2 && fr < pw) *reinterpret_cast<h8 *>(out_img + (size_t)((k - 2) * pw + fr) * P.cs_out) = pend;
                f4 acc[2][2][2];                                  // [conv row][parity][channel half], from the bias
#pragma unroll
                for (int cr = 0; cr < 2; ++cr)
#pragma unroll
                    for (int par = 0; par < 2; ++par) { acc[cr][par][0] = bias0; acc[cr][par][1] = bias1; }
                h8 X[2][4];
                auto read_row = [&](int i, h8 (&x)[4]) {
                    const int so = ((2 * kq + i) & (PR_SLOTS - 1)) * PR_SLOT_HALVES;      // slot of input row 2k-1+i
                    x[0] = *reinterpret_cast<const h8 *>(pA + so);
                    x[1] = *reinterpret_cast<const h8 *>(pB + so);
                    x[2] = *reinterpret_cast<const h8 *>(pC + so);
                    x[3] = *reinterpret_cast<const h8 *>(pD + so);
                };
                read_row(0, X[0]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // order: wait for row i (requested one row of MFMAs ago), THEN request row i+1, then multiply -- hipcc's wait
                    // before the first use of row i is lgkmcnt(0), so a request issued ahead of it would be waited for as well
                    h8 (&x)[4] = X[i & 1];
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    if (i + 1 < 4) read_row(i + 1, X[(i + 1) & 1]);
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int cr = 0; cr < 2; ++cr) {
                            const int dy = i - cr;                // filter row of conv row 2k+cr that meets input row 2k-1+i
                            if (dy < 0 || dy > 2 || (dbg & 1)) continue;
#pragma unroll
                            for (int par = 0; par < 2; ++par)
#pragma unroll
                                for (int a = 0; a < 2; ++a)
                                    acc[cr][par][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[dy * 3 + dx][a], x[dx + par], acc[cr][par][a], 0, 0, 0);
                        }
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                if (dbg & 2) { pend = __builtin_bit_cast(h8, acc[0][0][0] + acc[1][1][1] + acc[0][1][0] + acc[1][0][1]); continue; }
                // horizontal 3-max of each conv row: x = 2fr, 2fr+1 and 2fr+2 (the even fragment of lane fr+1)
                f4 hm[2][2];
#pragma unroll
                for (int cr = 0; cr < 2; ++cr)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = acc[cr][0][a][r];
                            float nx = e;
#if defined(__HIP_DEVICE_COMPILE__)
                            // lane 15 of a row gets 0 (bound_ctrl): it is pooled column 15, which does not exist and is never stored
                            nx = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, e), 0x101 /* row_shl:1 */, 0xf, 0xf, true));
#endif
                            hm[cr][a][r] = pool_max(pool_max(e, acc[cr][1][a][r]), nx);
                        }
                if (k > j0) {                                     // pooled row k-1 = rows 2k-2, 2k-1 (carried) and 2k
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            pend[a * 4 + r] = (_Float16)apply_act(pool_max(carry[a][r], hm[0][a][r]), act);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 4; ++r) carry[a][r] = pool_max(hm[0][a][r], hm[1][a][r]);
            }
        }
        if (fr < pw) *reinterpret_cast<h8 *>(out_img + (size_t)((j1 - 1) * pw + fr) * P.cs_out) = pend;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the look-ahead DMAs of the last rounds
#endif
}

// ---------------------------------------------------------------------------------------------------
// A whole residual unit of 3x3 32->32 layers on a 15-pixel-wide map as ONE launch, one wave per image streaming rows
// (tools/freeze_model.py:43-75 for MARS conv2_1 / conv2_3):
//     h1 = act(conv_A(pre) + bias_A)                         [conv3x3_rw_k<.., EF = 0>]
//     out = conv_B(h1) + bias_B + raw,  out2 = ELU(scale * out + shift)   [conv3x3_rw_k<.., EF = 1>]
// h1 lives only in a four-row LDS ring of the wave; both filters stay in registers (144 VGPRs).  Round r makes h1 rows
// 2r, 2r+1 from pre rows 2r-1 .. 2r+2 and then output rows 2r-1, 2r from h1 rows 2r-2 .. 2r+1.  pre (and raw, when it is
// a different tensor) rows arrive by global_load_lds one round ahead; the only wait is a vmcnt(0) just before the round's
// stores, by which time the DMAs issued at its top have long landed and the previous round's stores have retired.
// Row slot: four 8-channel planes of 16 pixel slots (1 KiB, the DMA's lane order); pixel slot 15 is always zero (the
// DMA reads the zero line for it, the h1 epilogue writes zeros), so the right tap of pixel 14 and -- one slot earlier
// in memory -- the left tap of pixel 0 of the next plane read the zero padding without a mask; a 16-byte zero pad in
// front of each ring does the same for plane 0 of slot 0.  Lane 15 of a fragment computes a pixel that does not exist
// and is never stored.  Every accumulator sums its taps in the order t = 0..8 from the bias and the epilogue is
// conv_epilogue_f16x8 itself: the bits of the two-launch path.
constexpr int RU_PRE_SLOTS = 8, RU_H1_SLOTS = 4, RU_RAW_SLOTS = 4, RU_SLOT = 512;     // halves
constexpr int ru_wave_halves(bool raw_sep) { return 8 + RU_PRE_SLOTS * RU_SLOT + 8 + RU_H1_SLOTS * RU_SLOT + 8 + (raw_sep ? RU_RAW_SLOTS * RU_SLOT : 0) + 8; }

template <int ACT_A, bool RAW_SEP>
__global__ __launch_bounds__(256, 2) void res_unit_rows_k(const ConvP PA, const ConvP PB, const int n_img) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    _Float16 *wbase = lds + (size_t)wave * ru_wave_halves(RAW_SEP);
    _Float16 *pre = wbase + 8, *h1 = pre + RU_PRE_SLOTS * RU_SLOT + 8, *rawr = h1 + RU_H1_SLOTS * RU_SLOT + 8;
    const h8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    // the whole region once: the pads, and the last pixel slot of rows no image has filled yet (an h1 row's left tap reads it)
    for (int i = lane * 8; i < ru_wave_halves(RAW_SEP); i += 64 * 8) *reinterpret_cast<h8 *>(wbase + i) = zero8;

    h8 wA[9][2], wB[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            wA[t][a] = *reinterpret_cast<const h8 *>(PA.w + (size_t)rw_weight_row(a, fr) * PA.kpad + t * 32 + fq * 8);
            wB[t][a] = *reinterpret_cast<const h8 *>(PB.w + (size_t)rw_weight_row(a, fr) * PB.kpad + t * 32 + fq * 8);
        }
    const f4 bA0 = *reinterpret_cast<const f4 *>(PA.bias + fq * 8), bA1 = *reinterpret_cast<const f4 *>(PA.bias + fq * 8 + 4);
    const Epi8 EB = epi8_load(PB, fq * 8);
    const int H = PA.H, W = PA.W;                                 // W = 15
    const int offC = (fq * 16 + fr) * 8;                          // centre tap; left = -8 halves, right = +8 halves
    const int actA = ACT_A < 0 ? PA.act : ACT_A;

    for (int n = blockIdx.x * 4 + wave; n < n_img; n += gridDim.x * 4) {
        const _Float16 *img = PA.in + (size_t)n * H * W * PA.cs_in + PA.coff_in + fq * 8;
        const _Float16 *rimg = PB.res + (size_t)n * H * W * PB.cs_res + PB.coff_res + fq * 8;
        auto fill_pre = [&](int y) {                              // wave-uniform y; outside the image and pixel slot 15: zero line
            const bool ok = (unsigned)y < (unsigned)H && fr < W;
            lds_fill16(ok ? img + ((size_t)y * W + fr) * PA.cs_in : PA.zero, pre + ((y + 1) & (RU_PRE_SLOTS - 1)) * RU_SLOT);
        };
        auto fill_raw = [&](int y) {
            const bool ok = (unsigned)y < (unsigned)H && fr < W;
            lds_fill16(ok ? rimg + ((size_t)y * W + fr) * PB.cs_res : PA.zero, rawr + (y & (RU_RAW_SLOTS - 1)) * RU_SLOT);
        };
        fill_pre(-1); fill_pre(0); fill_pre(1); fill_pre(2);
        if constexpr (RAW_SEP) { fill_raw(-1); fill_raw(0); }
        *reinterpret_cast<h8 *>(h1 + ((-1) & (RU_H1_SLOTS - 1)) * RU_SLOT + lane * 8) = zero8;       // h1 row -1: padding
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        for (int r = 0; r <= H / 2; ++r) {                          // output rows 2r-1, 2r
            fill_pre(2 * r + 3); fill_pre(2 * r + 4);
            if constexpr (RAW_SEP) { fill_raw(2 * r + 1); fill_raw(2 * r + 2); }
            // three fragments of input row `slot`: left / centre / right tap
            auto conv2 = [&](const _Float16 *ringp, int slot0, int mask, const h8 (&w)[9][2], f4 (&acc)[2][2]) {
                h8 X[2][3];
                auto rd = [&](int i, h8 (&x)[3]) {
                    const _Float16 *rs = ringp + ((slot0 + i) & mask) * RU_SLOT + offC;
                    x[0] = *reinterpret_cast<const h8 *>(rs - 8);
                    x[1] = *reinterpret_cast<const h8 *>(rs);
                    x[2] = *reinterpret_cast<const h8 *>(rs + 8);
                };
                rd(0, X[0]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    h8 (&x)[3] = X[i & 1];
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));      // row i has arrived before row i+1 is requested
                    __builtin_amdgcn_sched_barrier(0);
#endif
                    if (i + 1 < 4) rd(i + 1, X[(i + 1) & 1]);
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int cr = 0; cr < 2; ++cr) {
                            const int dy = i - cr;
                            if (dy < 0 || dy > 2) continue;
#pragma unroll
                            for (int a = 0; a < 2; ++a)
                                acc[cr][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy * 3 + dx][a], x[dx], acc[cr][a], 0, 0, 0);
                        }
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
            };
            // ---- layer A: h1 rows 2r, 2r+1 from pre rows 2r-1 .. 2r+2 (slots (y + 1) & 7)
            {
                f4 acc[2][2] = {{bA0, bA1}, {bA0, bA1}};
                conv2(pre, 2 * r, RU_PRE_SLOTS - 1, wA, acc);
#pragma unroll
                for (int cr = 0; cr < 2; ++cr) {
                    const int y = 2 * r + cr;
                    const bool live = y < H && fr < W;             // past the image / pixel slot 15: layer B's zero padding
                    h8 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[q] = (_Float16)apply_act(acc[cr][0][q], actA);
                        o[4 + q] = (_Float16)apply_act(acc[cr][1][q], actA);
                    }
                    u4v_t ou = __builtin_bit_cast(u4v_t, o);
                    ou &= live ? 0xFFFFFFFFu : 0u;
                    *reinterpret_cast<u4v_t *>(h1 + (y & (RU_H1_SLOTS - 1)) * RU_SLOT + offC) = ou;
                }
            }
            // ---- layer B: output rows 2r-1, 2r from h1 rows 2r-2 .. 2r+1 (slots y & 3)
            {
                f4 acc[2][2] = {{EB.b0, EB.b1}, {EB.b0, EB.b1}};
                conv2(h1, 2 * r - 2, RU_H1_SLOTS - 1, wB, acc);
                // the skip rows: read from their ring slots row by row (both at once cost the RAW_SEP variant its last registers:
                // 256 VGPRs + 20 bytes of scratch; the slots are not refilled before the next round's DMAs land)
                h8 rv0 = RAW_SEP ? *reinterpret_cast<const h8 *>(rawr + ((2 * r - 1) & (RU_RAW_SLOTS - 1)) * RU_SLOT + offC)
                                 : *reinterpret_cast<const h8 *>(pre + ((2 * r) & (RU_PRE_SLOTS - 1)) * RU_SLOT + offC);
#if defined(__HIP_DEVICE_COMPILE__)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next round's rows have landed, last round's stores have retired
#endif
#pragma unroll
                for (int cr = 0; cr < 2; ++cr) {
                    const int y = 2 * r - 1 + cr;
                    if (cr == 1) rv0 = RAW_SEP ? *reinterpret_cast<const h8 *>(rawr + (y & (RU_RAW_SLOTS - 1)) * RU_SLOT + offC)
                                               : *reinterpret_cast<const h8 *>(pre + ((y + 1) & (RU_PRE_SLOTS - 1)) * RU_SLOT + offC);
                    if ((unsigned)y >= (unsigned)H || fr >= W) continue;
                    float o[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { o[q] = acc[cr][0][q]; o[4 + q] = acc[cr][1][q]; }
                    conv_epilogue_f16x8<ACT_NONE, false, 1, true, false>(PB, EB, (n * H + y) * W + fr, fq * 8, o, &rv0);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// TWO consecutive residual units (MARS conv2_1 and conv2_3: the whole conv2_x stage, tools/freeze_model.py:118-121) as ONE
// launch.  res_unit_rows_k keeps a unit's inner tensor on the CU; between the units the block output `out` and its BN + ELU
// view `out2` (2 x 14.9 KB per crop) still went through HBM -- written by one launch, read back by the next.  Here a PAIR of
// waves owns an image: wave A runs the first unit exactly as res_unit_rows_k does, but its epilogue writes the rows of `out`
// and `out2` into two 8-row LDS rings instead of memory; wave B runs the second unit from those rings two rounds behind
// (it needs rows up to 2r + 2 of `out2` for its round r) and stores the stage's outputs.  The stage then reads its input
// once and writes its two outputs once: 685 MB instead of 1 828 MB per 7 680 crops.
//   * A block is four pairs: waves 0-3 are the A waves, 4-7 their B partners (the hardware deals a block's waves over the
//     SIMDs cyclically, so every SIMD hosts one A and one B wave); one raw s_barrier per round keeps the pairs in step --
//     all eight waves do the same amount of matrix work per round.
//   * The images of a pair form ONE stream of 32 rows each (31 map rows + the zero row they share as lower / upper
//     padding): stream row s sits in slot (s + 1) & 7 of the input-type rings and s & 3 of the h1 rings, round g of a wave
//     makes h1 rows 2g, 2g + 1 and output rows 2g - 1, 2g.  A writes rows 2g - 1, 2g while B (at its round g - 2) reads rows
//     2g - 5 .. 2g - 2: never the same slot.  Rows that are padding are written as zeros by A -- B reads them as its own padding.
//   * Same summation orders and the same epilogue statements as res_unit_rows_k / conv3x3_rw_k: the same bits.
constexpr int RP_X_SLOTS = 8;
constexpr int rp_pair_halves() { return 6 * 8 + (RU_PRE_SLOTS + 2 * RU_H1_SLOTS + 2 * RP_X_SLOTS) * RU_SLOT; }   // 16-byte zero pads around the rings

template <int ACT_A>
__global__ __launch_bounds__(512, 2) void res_pair_rows_k(const ConvP P1A, const ConvP P1B, const ConvP P2A, const ConvP P2B, const int n_img) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave & 3, role = wave >> 2;               // role 0 = A (first unit), 1 = B (second unit)
    const int fr = lane & 15, fq = lane >> 4;
    _Float16 *pbase = lds + (size_t)pair * rp_pair_halves();
    _Float16 *pre = pbase + 8;                                    // A: input rows (DMA)
    _Float16 *h1a = pre + RU_PRE_SLOTS * RU_SLOT + 8;             // A: rows between its two layers
    _Float16 *xo = h1a + RU_H1_SLOTS * RU_SLOT + 8;               // exchange: rows of the first unit's `out` (B's skip rows)
    _Float16 *xo2 = xo + RP_X_SLOTS * RU_SLOT + 8;                // exchange: rows of its `out2` (B's input rows)
    _Float16 *h1b = xo2 + RP_X_SLOTS * RU_SLOT + 8;               // B: rows between its two layers
    const h8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    for (int i = (role * 64 + lane) * 8; i < rp_pair_halves(); i += 128 * 8) *reinterpret_cast<h8 *>(pbase + i) = zero8;

    // this wave's unit: first-layer filter wA (ELU), second-layer filter wB (residual + second output)
    const _Float16 *wa_p = role ? P2A.w : P1A.w, *wb_p = role ? P2B.w : P1B.w;
    const int kpa = role ? P2A.kpad : P1A.kpad, kpb = role ? P2B.kpad : P1B.kpad;
    h8 wA[9][2], wB[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            wA[t][a] = *reinterpret_cast<const h8 *>(wa_p + (size_t)rw_weight_row(a, fr) * kpa + t * 32 + fq * 8);
            wB[t][a] = *reinterpret_cast<const h8 *>(wb_p + (size_t)rw_weight_row(a, fr) * kpb + t * 32 + fq * 8);
        }
    const float *ba_p = role ? P2A.bias : P1A.bias, *bb_p = role ? P2B.bias : P1B.bias, *af_p = role ? P2B.aff2 : P1B.aff2;
    const int cpad = role ? P2B.cout_pad : P1B.cout_pad;
    const f4 bA0 = *reinterpret_cast<const f4 *>(ba_p + fq * 8), bA1 = *reinterpret_cast<const f4 *>(ba_p + fq * 8 + 4);
    Epi8 EB;
    EB.b0 = *reinterpret_cast<const f4 *>(bb_p + fq * 8); EB.b1 = *reinterpret_cast<const f4 *>(bb_p + fq * 8 + 4);
    EB.s0 = *reinterpret_cast<const f4 *>(af_p + fq * 8); EB.s1 = *reinterpret_cast<const f4 *>(af_p + fq * 8 + 4);
    EB.t0 = *reinterpret_cast<const f4 *>(af_p + cpad + fq * 8); EB.t1 = *reinterpret_cast<const f4 *>(af_p + cpad + fq * 8 + 4);
    const int H = P1A.H, W = P1A.W;                               // 31, 15 (launcher)
    const int offC = (fq * 16 + fr) * 8;
    const int actA = ACT_A < 0 ? P1A.act : ACT_A;
    _Float16 *inr = role ? xo2 : pre, *h1r = role ? h1b : h1a, *resr = role ? xo : pre;

    const int n0 = blockIdx.x * 4 + pair, nstep = gridDim.x * 4;
    const int K = n0 < n_img ? (n_img - n0 + nstep - 1) / nstep : 0;                 // images of this pair: n0 + k * nstep
    const int base4 = blockIdx.x * 4;
    const int Kmax = base4 < n_img ? (n_img - base4 + nstep - 1) / nstep : 0;        // of the block's first pair: every wave loops alike
    const int GA = 16 * K, Gall = 16 * Kmax + 2;
    auto fill_pre = [&](int s) {                                  // stream row s of the pair (wave-uniform); padding rows / past the stream: zero lines
        const int kk = (s + 1) >> 5, y = ((s + 1) & 31) - 1;
        const bool ok = kk < K && y >= 0 && fr < W;
        const _Float16 *src = P1A.in + ((size_t)(n0 + (ok ? kk : 0) * nstep) * H * W + (size_t)(ok ? y : 0) * W + fr) * P1A.cs_in + P1A.coff_in + fq * 8;
        lds_fill16(ok ? src : P1A.zero, pre + ((s + 1) & (RU_PRE_SLOTS - 1)) * RU_SLOT);
    };
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                 // the rings are zero before the first row arrives
    asm volatile("" ::: "memory");
#endif
    if (role == 0 && K > 0) { fill_pre(-1); fill_pre(0); fill_pre(1); fill_pre(2); fill_pre(3); fill_pre(4); }   // rounds 0 and 1
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    // three fragments of an input row: left / centre / right tap (res_unit_rows_k's loop)
    auto conv2 = [&](const _Float16 *ringp, int slot0, int mask, const h8 (&w)[9][2], f4 (&acc)[2][2]) {
        h8 X[2][3];
        auto rd = [&](int i, h8 (&x)[3]) {
            const _Float16 *rs = ringp + ((slot0 + i) & mask) * RU_SLOT + offC;
            x[0] = *reinterpret_cast<const h8 *>(rs - 8);
            x[1] = *reinterpret_cast<const h8 *>(rs);
            x[2] = *reinterpret_cast<const h8 *>(rs + 8);
        };
        rd(0, X[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h8 (&x)[3] = X[i & 1];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (i + 1 < 4) rd(i + 1, X[(i + 1) & 1]);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int cr = 0; cr < 2; ++cr) {
                    const int dy = i - cr;
                    if (dy < 0 || dy > 2) continue;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        acc[cr][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[dy * 3 + dx][a], x[dx], acc[cr][a], 0, 0, 0);
                }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    // The two waves of a SIMD are an A and a B wave that leave every barrier together: run alike they would both multiply, then
    // both do their epilogue arithmetic, and the matrix pipe would idle half the time.  B therefore defers the epilogue of its
    // second layer (residual add, second BN + ELU, the stores) to the top of its next round: its vector work meets A's matrix work.
    f4 accC[2][2];
    int grC = -1;
    auto b_store = [&]() {
#pragma unroll
        for (int cr = 0; cr < 2; ++cr) {
            const int s = 2 * grC - 1 + cr;
            const int kk = (s + 1) >> 5, y = ((s + 1) & 31) - 1;
            if (y < 0 || fr >= W) continue;
            float o[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { o[q] = accC[cr][0][q]; o[4 + q] = accC[cr][1][q]; }
            // the skip row is still in its ring slot: A is four rows further on (slots (s + 1) & 7 of rows s and s + 6 .. s + 7 differ)
            const h8 rv = *reinterpret_cast<const h8 *>(resr + ((s + 1) & (RU_PRE_SLOTS - 1)) * RU_SLOT + offC);
            conv_epilogue_f16x8<ACT_NONE, false, 1, true, false>(P2B, EB, ((n0 + kk * nstep) * H + y) * W + fr, fq * 8, o, &rv);
        }
    };
    for (int g = 0; g < Gall; ++g) {
        const int gr = role ? g - 2 : g;                          // this wave's round
        const bool active = role ? (gr >= 0 && gr < GA) : (K > 0 && gr <= GA);   // A's round GA only writes the trailing padding row
        if (grC >= 0) { b_store(); grC = -1; }
        if (active) {
            // A's input rows arrive TWO rounds ahead (the eight slots hold rows 2gr - 1 .. 2gr + 6: four being read, two landed, two in flight): a
            // round is ~1.4 us, about one HBM round trip under load -- requested one round ahead, the wait at the end of the round still met them
            if (role == 0) { fill_pre(2 * gr + 5); fill_pre(2 * gr + 6); }
            // ---- first layer: h1 stream rows 2gr, 2gr + 1 from input rows 2gr - 1 .. 2gr + 2
            {
                f4 acc[2][2] = {{bA0, bA1}, {bA0, bA1}};
                conv2(inr, 2 * gr, RU_PRE_SLOTS - 1, wA, acc);
#pragma unroll
                for (int cr = 0; cr < 2; ++cr) {
                    const int s = 2 * gr + cr;
                    const bool live = ((s + 1) & 31) != 0 && fr < W;   // the padding row between images / pixel slot 15: zeros
                    h8 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[q] = (_Float16)apply_act(acc[cr][0][q], actA);
                        o[4 + q] = (_Float16)apply_act(acc[cr][1][q], actA);
                    }
                    u4v_t ou = __builtin_bit_cast(u4v_t, o);
                    ou &= live ? 0xFFFFFFFFu : 0u;
                    *reinterpret_cast<u4v_t *>(h1r + (s & (RU_H1_SLOTS - 1)) * RU_SLOT + offC) = ou;
                }
            }
            // ---- second layer: output stream rows 2gr - 1, 2gr from h1 rows 2gr - 2 .. 2gr + 1
            {
                f4 acc[2][2] = {{EB.b0, EB.b1}, {EB.b0, EB.b1}};
                conv2(h1r, 2 * gr - 2, RU_H1_SLOTS - 1, wB, acc);
                h8 rv[2];
#pragma unroll
                for (int cr = 0; cr < 2; ++cr) rv[cr] = *reinterpret_cast<const h8 *>(resr + ((2 * gr + cr) & (RU_PRE_SLOTS - 1)) * RU_SLOT + offC);   // row 2gr - 1 + cr
                if (role == 0) {
#pragma unroll
                    for (int cr = 0; cr < 2; ++cr) {
                        const int s = 2 * gr - 1 + cr;
                        const bool live = ((s + 1) & 31) != 0 && fr < W;
                        // conv_epilogue_f16x8<ACT_NONE, false, 1>'s statements: out = f16(v + res), out2 = f16(ELU(scale * (v + res) + shift))
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[q] = acc[cr][0][q]; v[4 + q] = acc[cr][1][q]; }
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += (float)rv[cr][q];
                        h8 o, o2;
#pragma unroll
                        for (int q = 0; q < 8; ++q) o[q] = (_Float16)v[q];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float sc = q < 4 ? EB.s0[q] : EB.s1[q - 4], sh = q < 4 ? EB.t0[q] : EB.t1[q - 4];
                            o2[q] = (_Float16)apply_act(sc * v[q] + sh, ACT_ELU);
                        }
                        u4v_t ou = __builtin_bit_cast(u4v_t, o), ou2 = __builtin_bit_cast(u4v_t, o2);
                        ou &= live ? 0xFFFFFFFFu : 0u;
                        ou2 &= live ? 0xFFFFFFFFu : 0u;
                        *reinterpret_cast<u4v_t *>(xo + ((s + 1) & (RP_X_SLOTS - 1)) * RU_SLOT + offC) = ou;
                        *reinterpret_cast<u4v_t *>(xo2 + ((s + 1) & (RP_X_SLOTS - 1)) * RU_SLOT + offC) = ou2;
                    }
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // the next round's input rows have landed (this round's two requests may still be in flight; an A wave issues nothing else that counts)
#endif
                } else {                                            // B keeps its sums: their epilogue opens its NEXT round (see b_store)
#pragma unroll
                    for (int cr = 0; cr < 2; ++cr) { accC[cr][0] = acc[cr][0]; accC[cr][1] = acc[cr][1]; }
                    grC = gr;
                }
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this round's ring rows are written ...
        __builtin_amdgcn_s_barrier();                                 // ... before the partner reads them (stores to memory are not waited for)
        asm volatile("" ::: "memory");
#endif
    }
    if (grC >= 0) b_store();
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // A: the look-ahead DMAs past the stream
#endif
}

// ---------------------------------------------------------------------------------------------------
// 3x3 64->64 convolution (same padding, stride 1) on 8-pixel-wide maps (MARS conv3_x, 16x8), weight-stationary: a wave
// owns 32 output channels of one image and keeps their whole filter in registers (9 taps x 2 k slices x 2 fragments =
// 144 VGPRs); the image streams through the wave's own ring of row slots, filled by global_load_lds one round ahead.
// A pixel fragment is two map rows (16 pixels).  Row slot: eight 8-channel planes of ten pixel slots (zero, 8 pixels,
// zero -- the DMA reads the zero line for the two pad slots, so no tap needs a mask), pitch 1408 B: consecutive rows
// sit 128 B apart modulo the 256-B bank row, so the two halves of a fragment read never meet on a bank.
// conv_glds_k's summation order (zero, taps in order, two k slices per tap, bias in the epilogue) and its epilogue
// call: the same bits as the generic kernel, which runs these layers at 370-540 TFLOP/s (one barrier per 8 MFMAs).
// The rows of a wave's images form ONE stream (a zero row between images serves as the lower padding of one and the
// upper padding of the next): stream row g sits in ring slot g % 11 and the DMAs run three rounds (6-7 rows) ahead of
// the MFMAs, across image boundaries -- a round is ~1k cycles, a DMA under load takes several times that.  Nothing in
// the loop but DMAs and stores touches vmcnt (the residual rows come by DMA too; an ordinary load would make hipcc drain
// the queue at its first use), so the wait is counted: everything younger than the issue group of three rounds ago,
// i.e. 3 x (4 row DMAs [+ 1 residual DMA]) + 3 x (1 [+ 1] stores); a round that issues a fifth and sixth row DMA (first
// round of an image) only makes the wait stricter.
constexpr int C64_PITCH = 704, C64_SLOTS = 11, C64_RES_SLOTS = 4, C64_LEAD = 3;   // 11: 4 rows being read + up to 7 requested (a group that crosses an image boundary)
constexpr int c64_wave_halves() { return C64_SLOTS * C64_PITCH + C64_RES_SLOTS * 512; }

// STRIP: the same kernel on maps wider than 8 pixels (YOLOv5's 3x3 64 -> 64 layers, 80 pixels wide): a work item is an 8-column strip of an
// image instead of an image -- its two pad slots take the neighbouring strips' pixels (the zero line only at the image's own border), its
// pixels sit W, not 8, apart in memory; everything else (the row stream with a shared zero row between items, the ring, the counted waits,
// the summation order) is the 8-wide kernel.  Each strip re-reads two of ten columns (L2 hits: the neighbours run on the same CU or XCD).
template <int ACT, int EF, bool STRIP = false>                  // EF 0: plain layer; 1: residual input and second output; 2: residual only
__global__ __launch_bounds__(256, 2) void conv3x3_c64_rows_k(const ConvP P, const int n_img) {          // n_img: work items (images, or strips of images)
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int j = fr >> 3, x = fr & 7;                            // fragment pixel: row j of the pair, column x
    _Float16 *ring = lds + (size_t)wave * c64_wave_halves(), *resr = ring + C64_SLOTS * C64_PITCH;
    const int H = P.H, S = H + 1, RPI = H / 2;                    // stream rows / rounds per image
    constexpr int NRES = EF ? 1 : 0;                              // residual DMAs per round
    (void)NRES;
    const int NSX = STRIP ? P.W >> 3 : 1;                         // strips per image

    const int half = wave & 1;                                    // a wave keeps its channel half for the whole launch: one filter load
    h8 wf[9][2][2];                                               // [tap][k slice][fragment]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                wf[t][ks][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)(half * 32 + rw_weight_row(a, fr)) * P.kpad + t * 64 + ks * 32 + fq * 8);
    const Epi8 E = epi8_load(P, half * 32 + fq * 8);
    const int lane_off = (fq * 10 + x) * 8;                       // plane fq of k slice 0, pixel slot x (= column x - 1)
    const int n0 = blockIdx.x * 2 + (wave >> 1), nstep = gridDim.x * 2;
    const int K = n0 < n_img ? (n_img - n0 + nstep - 1) / nstep : 0;      // images of this wave: n0 + k * nstep
    const int T = K * S + 1, Q = K * RPI;                         // stream rows, rounds

    // per-lane constants of the two row DMAs (chunk c = 16 * c2 + lane -> plane c / 10, pixel slot c % 10)
    int dma_off[2];
    bool dma_px[2], dma_l[2], dma_r[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
        const int c = 16 * c2 + lane, pl = c / 10, sl = c - pl * 10;
        dma_px[c2] = sl >= 1 && sl <= 8;
        dma_l[c2] = sl == 0; dma_r[c2] = sl == 9;
        dma_off[c2] = (sl - 1) * P.cs_in + pl * 8;
    }
    int dq = 0, dq_slot = 0, dq_k = 0, dq_y = -1;                 // next stream row to request: its ring slot, image, map row
    int dq_f = n0 / NSX, dq_sx = n0 - dq_f * NSX;                 // STRIP: image and strip of item n0 + dq_k * nstep
    auto issue_row = [&]() {                                      // past the stream / between images: zero lines
        const bool rok = dq < T && dq_y >= 0;
        const _Float16 *row = STRIP ? P.in + (((size_t)dq_f * H + (rok ? dq_y : 0)) * P.W + dq_sx * 8) * P.cs_in + P.coff_in
                                    : P.in + ((size_t)(n0 + dq_k * nstep) * H + (rok ? dq_y : 0)) * 8 * P.cs_in + P.coff_in;
        _Float16 *dst = ring + dq_slot * C64_PITCH;
        // two full-wave DMAs: chunks 0..63 and 16..79 (48 chunks twice, with the same bytes).  A DMA under a divergent
        // `if (lane < 16)` is not safe: hipcc threads consecutive such branches and the copies of the wave-wide DMA
        // between them then take M0 from readfirstlane of a per-path value -- one path's lanes land in another row.
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const bool ok = rok && (dma_px[c2] || (STRIP && ((dma_l[c2] && dq_sx > 0) || (dma_r[c2] && dq_sx + 1 < NSX))));
            lds_fill16(ok ? row + dma_off[c2] : P.zero, dst + 128 * c2);
        }
        ++dq;
        dq_slot = dq_slot + 1 == C64_SLOTS ? 0 : dq_slot + 1;
        if (++dq_y == H) {
            dq_y = -1; ++dq_k;
            if constexpr (STRIP) { const int it = n0 + dq_k * nstep; dq_f = it / NSX; dq_sx = it - dq_f * NSX; }
        }
    };
    int ga_k = 0, ga_r = 0;                                       // round q + LEAD: image, round inside the image
    int ga_f = n0 / NSX, ga_sx = n0 - ga_f * NSX;
    const int res_px = lane & 15, res_c = lane >> 4;
    auto issue_group = [&](int qa) {                              // what round qa = q + LEAD needs: rows up to its last, its residual rows
        const int last = ga_k * S + 2 * ga_r + 3;
        while (dq <= last) issue_row();                           // 2 rows, 3 when round qa opens an image (first group: 4)
        if constexpr (EF != 0) {                                  // residual rows 2 ga_r, 2 ga_r + 1 of image ga_k: [chunk fq][pixel] x 16 B
            const bool ok = qa < Q;
            const _Float16 *src = STRIP ? P.res + ((((size_t)ga_f * H + 2 * ga_r + (res_px >> 3)) * P.W + ga_sx * 8 + (res_px & 7))) * P.cs_res + P.coff_res + half * 32 + res_c * 8
                                        : P.res + ((size_t)((n0 + ga_k * nstep) * H + 2 * ga_r) * 8 + res_px) * P.cs_res + P.coff_res + half * 32 + res_c * 8;
            lds_fill16(ok ? src : P.zero, resr + (qa & (C64_RES_SLOTS - 1)) * 512);
        }
        if (++ga_r == RPI) {
            ga_r = 0; ++ga_k;
            if constexpr (STRIP) { const int it = n0 + ga_k * nstep; ga_f = it / NSX; ga_sx = it - ga_f * NSX; }
        }
    };
    for (int qa = 0; qa < C64_LEAD; ++qa) issue_group(qa);
    int k = 0, r = 0, bslot = 0;                                  // this round: image, round inside it, ring slot of its first row (map row 2r - 1)
    int k_f = n0 / NSX, k_sx = n0 - k_f * NSX;
    for (int q = 0; q < Q; ++q) {
        issue_group(q + C64_LEAD);
#if defined(__HIP_DEVICE_COMPILE__)
        if (q < C64_LEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C64_LEAD * (4 + NRES)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(C64_LEAD * (4 + NRES) + C64_LEAD * (1 + (EF == 1 ? 1 : 0))) : "memory");
#endif
        const int m = STRIP ? ((k_f * H + 2 * r + j) * P.W + k_sx * 8 + x)
                            : ((n0 + k * nstep) * H + 2 * r + j) * 8 + x; // this lane's output pixel
        f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        const _Float16 *rowp[3];                                   // this lane's row of filter row dy: stream row first + dy + j
        {
            int sl = bslot;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int s1 = sl + 1 == C64_SLOTS ? 0 : sl + 1;
                rowp[dy] = ring + (j ? s1 : sl) * C64_PITCH + lane_off;
                sl = s1;
            }
        }
        // (reads two taps ahead, as inline-asm ds_read with hand-counted lgkmcnt, changed nothing: 50.8 vs 51.6 us -- the round
        // is bound by what it issues besides its 36 MFMAs: 4 DMAs at 100+ cycles each, the epilogue, the stores)
        h8 X[2][2];
        auto rd = [&](int t, h8 (&xv)[2]) {                       // tap t: pixel slot x + dx of that row, both k slices
            const int dy = t / 3, dx = t - dy * 3;
            xv[0] = *reinterpret_cast<const h8 *>(rowp[dy] + dx * 8);
            xv[1] = *reinterpret_cast<const h8 *>(rowp[dy] + dx * 8 + 4 * 10 * 8);
        };
        rd(0, X[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            h8 (&xv)[2] = X[t & 1];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(xv[0]), "+v"(xv[1]));          // tap t has arrived before tap t+1 is requested
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (t + 1 < 9) rd(t + 1, X[(t + 1) & 1]);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][ks][a], xv[ks], acc[a], 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        h8 rp;
        if constexpr (EF != 0) rp = *reinterpret_cast<const h8 *>(resr + (q & (C64_RES_SLOTS - 1)) * 512 + (fq * 16 + fr) * 8);
        float o[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = acc[0][i]; o[4 + i] = acc[1][i]; }
        conv_epilogue_f16x8<ACT, true, EF, true, false>(P, E, m, half * 32 + fq * 8, o, EF != 0 ? &rp : nullptr);
        // next round: two rows on, three across an image boundary
        const int adv = r + 1 == RPI ? 3 : 2;
        bslot += adv; if (bslot >= C64_SLOTS) bslot -= C64_SLOTS;
        if (++r == RPI) {
            r = 0; ++k;
            if constexpr (STRIP) { const int it = n0 + k * nstep; k_f = it / NSX; k_sx = it - k_f * NSX; }
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the look-ahead DMAs past the last round
#endif
}

// ---------------------------------------------------------------------------------------------------
// 3x3 stride-2 convolution 32 -> 64 channels from a 31x15 map to 16x8 (MARS conv3_1/1), one wave per image with the
// whole filter in registers (9 taps x 4 fragments = 144 VGPRs), rows streamed as in conv3x3_c64_rows_k: an image is 32
// stream rows (31 + the zero row it shares with the next image) = 8 rounds of 4, so the DMA queue never breaks.  A row
// slot keeps the odd and the even columns apart ([odd | even][plane][8] x 16 B, one DMA; odd slot 7 = column 15 is the
// zero line, and with the 16 zero bytes in front of the row every left tap finds its padding without a mask); the pitch
// of 1088 B puts the two map rows of a fragment read on different halves of the 256-B bank row.
constexpr int S2_PITCH = 544, S2_SLOTS = 16, S2_LEAD = 2;         // halves per row slot (64 B pad + 1 KiB), ring depth, rounds of DMA lead

template <int ACT>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_rows_k(const ConvP P, const int n_img) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int j = fr >> 3, ox = fr & 7;                           // fragment pixel: output row j of the pair, column ox
    _Float16 *ring = lds + (size_t)wave * (S2_SLOTS * S2_PITCH);
    const h8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    for (int i = lane * 8; i < S2_SLOTS * S2_PITCH; i += 64 * 8) *reinterpret_cast<h8 *>(ring + i) = zero8;     // the pads in front of the rows
    const int H = P.H, W = P.W, S = H + 1, RPI = S / 4;           // 31, 15, 32 stream rows and 8 rounds per image

    h8 wf[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[t][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)rw_weight_row(a, fr) * P.kpad + t * 32 + fq * 8);
    Epi8 E[2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) E[g2] = epi8_load(P, g2 * 32 + fq * 8);
    const int n0 = blockIdx.x * 4 + wave, nstep = gridDim.x * 4;
    const int K = n0 < n_img ? (n_img - n0 + nstep - 1) / nstep : 0;      // images of this wave: n0 + k * nstep
    const int T = K * S + 1, Q = K * RPI;                         // stream rows, rounds

    // DMA lane -> chunk: [set: 0 odd columns, 1 even][plane][8]: column 2 * idx + (set ? 0 : 1)
    const int dma_col = 2 * (lane & 7) + ((lane >> 5) ? 0 : 1);
    const int dma_off = dma_col * P.cs_in + ((lane >> 3) & 3) * 8;
    const bool dma_px = dma_col < W;
    int dq = 0, dq_k = 0, dq_y = -1;                              // next stream row to request: image, map row
    auto issue_row = [&]() {                                      // past the stream / between images: zero lines
        const bool rok = dq < T && dq_y >= 0;
        const _Float16 *row = P.in + ((size_t)(n0 + dq_k * nstep) * H + (rok ? dq_y : 0)) * W * P.cs_in + P.coff_in;
        lds_fill16(rok && dma_px ? row + dma_off : P.zero, ring + (dq & (S2_SLOTS - 1)) * S2_PITCH + 32);
        ++dq;
        if (++dq_y == H) { dq_y = -1; ++dq_k; }
    };
    auto issue_group = [&](int qa) { while (dq <= 4 * qa + 4) issue_row(); };     // round qa reads stream rows 4 qa .. 4 qa + 4
    for (int qa = 0; qa < S2_LEAD; ++qa) issue_group(qa);
    // per-lane offsets inside a row slot (halves, after the 64-byte pad): left / centre / right tap of output column ox
    const int off_dx[3] = {32 + (fq * 8 + ox - 1) * 8, 32 + ((4 + fq) * 8 + ox) * 8, 32 + (fq * 8 + ox) * 8};
    int k = 0, r = 0;
    for (int q = 0; q < Q; ++q) {
        issue_group(q + S2_LEAD);
#if defined(__HIP_DEVICE_COMPILE__)
        if (q < S2_LEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(S2_LEAD * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(S2_LEAD * 4 + S2_LEAD * 2) : "memory");
#endif
        const int m = ((n0 + k * nstep) * P.ho + 2 * r + j) * P.wo + ox;       // this lane's output pixel
        f4 acc[4] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
        const _Float16 *rowp[3];                                   // this lane's input row of filter row dy: stream row 4q + 2j + dy
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rowp[dy] = ring + ((4 * q + 2 * j + dy) & (S2_SLOTS - 1)) * S2_PITCH;
        h8 X[2];
        auto rd = [&](int t) -> h8 { return *reinterpret_cast<const h8 *>(rowp[t / 3] + off_dx[t % 3]); };
        X[0] = rd(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            h8 &xv = X[t & 1];
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(xv));                          // tap t has arrived before tap t+1 is requested
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (t + 1 < 9) X[(t + 1) & 1] = rd(t + 1);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][a], xv, acc[a], 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[i] = acc[2 * g2][i]; o[4 + i] = acc[2 * g2 + 1][i]; }
            conv_epilogue_f16x8<ACT, true, 0, true, false>(P, E[g2], m, g2 * 32 + fq * 8, o);
        }
        if (++r == RPI) { r = 0; ++k; }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the look-ahead DMAs past the last round
#endif
}

// First layer of a network straight from the u8 image: (x - mean) * scale, 3x3 conv over 3 channels
// (K = 27, one MFMA k-slice padded to 32; tap dy*9 + dx*3 + ch, so one filter row of a pixel is nine
// consecutive halves of the LDS patch; k slots as in stem_conv_pool_rows_k, which must produce the same bits), bias, activation -> NHWC f16 with 32 channels.  Replaces the
// separate input-conversion pass and its 8-channel f16 tensor.  Weights [32][32] f16, channel swap
// (BGR -> RGB) already folded into them by the host.
template <int STRIDE, int ACT>
__global__ __launch_bounds__(256, 2) void stem_conv3_k(const ConvP P) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    int bi = blockIdx.x;
    const int tile_x = bi % P.tiles_x; bi /= P.tiles_x;
    const int tile_y = bi % P.tiles_y;
    const int n = bi / P.tiles_y;
    const int y0 = tile_y * P.th, x0 = tile_x * P.tw;
    const int PW3 = ((P.tw - 1) * STRIDE + 3) * 3, PH = (P.th - 1) * STRIDE + 3;
    const unsigned rcp_tw = (65536u + P.tw - 1) / P.tw;         // exact x / tw for x < 1024, tw <= 32

    h8 wf[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const h8 *>(P.w + rw_weight_row(a, fr) * 32 + fq * 8);

    // LDS patch: PH rows of `pitch` halves; a pixel's filter row is nine consecutive halves starting at column
    // (tx * STRIDE) * 3 + shift.  Dword fill (image rows 4-byte aligned, P.p[2]): a row starts at the aligned byte at
    // or before the patch's first byte, shift = 0..3; byte fill: pitch = PW3, shift = 0.
    const int xs3 = (x0 * STRIDE - P.pad_l) * 3, w3 = P.W * 3;
    const bool dwords = P.p[2] != 0;
    const int a0 = dwords ? (xs3 & ~3) : xs3;                       // & ~3 rounds towards -inf in two's complement
    const int shift = xs3 - a0;
    const int pitch = dwords ? ((PW3 + 6) & ~3) : PW3;
    const uint8_t *img = P.src8 + (size_t)n * P.H * w3;
    if (dwords) {
        // eight independent 4-byte loads in flight per lane (a dependent load per element would leave the block waiting
        // on one memory round trip after another); an aligned dword lies entirely inside or outside the image row
        const int DW = pitch >> 2, total = PH * DW;
        const unsigned rcp = 0xFFFFFFFFu / (unsigned)DW + 1u;      // __umulhi(i, rcp) == i / DW for i < 2^16
        for (int i0 = tid; i0 < total; i0 += 8 * 256) {
            unsigned raw[8];
            bool in[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i0 + j * 256;
                const int py = (int)__umulhi((unsigned)idx, rcp), dq = idx - py * DW;
                const int y = y0 * STRIDE - P.pad_t + py, x3 = a0 + dq * 4;
                in[j] = idx < total && y >= 0 && y < P.H && x3 >= 0 && x3 < w3;
                // a select between two ADDRESSES: "in ? load : 0" makes hipcc branch around every load and wait for each
                // one separately (18 s_waitcnt vmcnt(0) for the 8 loads of an iteration)
                raw[j] = *reinterpret_cast<const unsigned *>(in[j] ? img + (size_t)y * w3 + x3 : reinterpret_cast<const uint8_t *>(P.zero));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i0 + j * 256;
                if (idx >= total) continue;
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                h4 o;                                             // zero padding is applied after normalisation
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = (_Float16)(in[j] ? ((float)((raw[j] >> (8 * q)) & 255u) - P.in_mean) * P.in_scale : 0.f);
                *reinterpret_cast<h4 *>(lds + idx * 4) = o;
            }
        }
    } else {
        const int total = PH * PW3;
        const unsigned rcp = 0xFFFFFFFFu / (unsigned)PW3 + 1u;     // __umulhi(i, rcp) == i / PW3 for i < 2^16
        for (int i0 = tid; i0 < total; i0 += 8 * 256) {
            int raw[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i0 + j * 256;
                const int py = (int)__umulhi((unsigned)idx, rcp), r = idx - py * PW3;
                const int y = y0 * STRIDE - P.pad_t + py, x3 = xs3 + r;
                const bool ok = idx < total && y >= 0 && y < P.H && x3 >= 0 && x3 < w3;
                const int v = (int)*(ok ? img + (size_t)y * w3 + x3 : reinterpret_cast<const uint8_t *>(P.zero));
                raw[j] = ok ? v : -1;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i0 + j * 256;                     // zero padding is applied after normalisation
                if (idx < total) lds[idx] = (_Float16)(raw[j] < 0 ? 0.f : ((float)raw[j] - P.in_mean) * P.in_scale);
            }
        }
    }
    int koff[8];                                                  // this lane's eight k positions inside a pixel's window
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // k slot (fq, j) -> tap (deepdish_amd/nets.py STEM_K_SLOT): group dy < 3 = taps 0..7 of filter row dy, group 3 =
        // the ninth tap of rows 0..2; the other five slots have zero weights, any finite value will do
        koff[j] = fq < 3 ? fq * pitch + j : j < 3 ? j * pitch + 8 : 0;
    }
    __syncthreads();

    const Epi8 E = epi8_load(P, fq * 8);
    const int tile_px = P.th * P.tw;
    const int nfrag = (tile_px + 15) >> 4;
    for (int f0 = wave * RW_MB; f0 < nfrag; f0 += 4 * RW_MB) {
        h8 xf[RW_MB];
        int mrow[RW_MB];
        bool ok[RW_MB];
#pragma unroll
        for (int b = 0; b < RW_MB; ++b) {
            const int p = (f0 + b) * 16 + fr;
            const int ty = (int)((p * rcp_tw) >> 16), tx = p - ty * P.tw;
            ok[b] = p < tile_px && y0 + ty < P.ho && x0 + tx < P.wo;
            const int e0 = ok[b] ? ty * STRIDE * pitch + tx * STRIDE * 3 + shift : shift;
            mrow[b] = (n * P.ho + y0 + ty) * P.wo + x0 + tx;
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[b][j] = lds[e0 + koff[j]];
        }
#pragma unroll
        for (int b = 0; b < RW_MB; ++b) {
            const f4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], xf[b], E.b0, 0, 0, 0);     // accumulate onto the bias
            const f4 a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], xf[b], E.b1, 0, 0, 0);
            if (!ok[b]) continue;
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            conv_epilogue_f16x8<ACT, false, 0>(P, E, mrow[b], fq * 8, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// First layer + first MobileNet block of the SSD in ONE launch, rows streamed by single waves (no workgroup barrier):
//     u8 frame -> 3x3 stride-2 conv (3 -> 32) + act -> depthwise 3x3 + act -> pointwise 32 -> 64 + act -> f16 NHWC.
// The 150x150x32 tensor between the two layers (1.44 MB per frame written and read back) and the depthwise output
// stay in the wave's LDS.  A wave owns a strip of SF_SW = 30 output columns and a range of rows of one frame:
//   * frame rows arrive two per step as aligned dwords (one per lane), are normalised to f16 into an 8-row ring;
//   * the stem row y+1 (32 columns: the strip + one halo column each side; columns outside the map are the depthwise
//     layer's zero padding) is computed with stem_conv3_k's arithmetic (same k slots, MFMA onto the bias) into a
//     4-row ring of [channel group][pixel] x 16 B;
//   * the depthwise row y is computed as dwpw_k does it (bias, then taps kh-major with v_fma_mix), one (pixel pair,
//     channel group) per lane, into a 32-pixel MFMA operand tile; the pointwise layer is one k slice: 8 MFMAs per row.
// Same summation orders and the same epilogue function as stem_conv3_k + dwpw_k: the same bits.
constexpr int SF_SW = 30, SF_CROP_PITCH = 208, SF_CROP_SLOTS = 8, SF_STEM_SLOTS = 4, SF_ROW = 32 * 32;
constexpr int sf_wave_halves() { return SF_CROP_SLOTS * SF_CROP_PITCH + SF_STEM_SLOTS * SF_ROW + SF_ROW + 64; }

template <int ACT>                                              // all three activations (launcher: ReLU6)
__global__ __launch_bounds__(256, 2) void ssd_front_k(const ConvP PS, const ConvP P, const int n_tasks, const int parts) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    _Float16 *crop = lds + (size_t)wave * sf_wave_halves();
    _Float16 *stemr = crop + SF_CROP_SLOTS * SF_CROP_PITCH, *xt = stemr + SF_STEM_SLOTS * SF_ROW;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const h8 zero8 = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    for (int i = lane * 8; i < sf_wave_halves(); i += 64 * 8) *reinterpret_cast<h8 *>(crop + i) = zero8;

    h8 ws[2], wd[9], wp[4];
#pragma unroll
    for (int a = 0; a < 2; ++a) ws[a] = *reinterpret_cast<const h8 *>(PS.w + rw_weight_row(a, fr) * 32 + fq * 8);
    const f4 sb0 = *reinterpret_cast<const f4 *>(PS.bias + fq * 8), sb1 = *reinterpret_cast<const f4 *>(PS.bias + fq * 8 + 4);
    const int cg = fq, pp = fr;                                  // depthwise item of this lane: channel group, pixel pair
#pragma unroll
    for (int t = 0; t < 9; ++t) wd[t] = *reinterpret_cast<const h8 *>(P.dw_w + (size_t)t * 32 + cg * 8);
    const f4 db0 = *reinterpret_cast<const f4 *>(P.dw_bias + cg * 8), db1 = *reinterpret_cast<const f4 *>(P.dw_bias + cg * 8 + 4);
#pragma unroll
    for (int a = 0; a < 4; ++a) wp[a] = *reinterpret_cast<const h8 *>(P.w + (size_t)rw_weight_row(a, fr) * P.kpad + fq * 8);
    Epi8 E[2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) E[g2] = epi8_load(P, g2 * 32 + fq * 8);

    const int Hin = PS.H, Win3 = PS.W * 3, Ho = P.ho, Wo = P.wo;
    const int strips = (Wo + SF_SW - 1) / SF_SW;
    for (int task = blockIdx.x * 4 + wave; task < n_tasks; task += gridDim.x * 4) {
        const int part = task % parts, t2 = task / parts;
        const int strip = t2 % strips, n = t2 / strips;
        const int x0 = strip * SF_SW, y0 = part * Ho / parts, y1 = (part + 1) * Ho / parts;
        // frame bytes of the strip's rows: from column 2 (x0 - 1) - pad_l, 66 columns; aligned dwords from a0
        const int b0 = (2 * (x0 - 1) - PS.pad_l) * 3, a0 = b0 & ~3, shift = b0 - a0;
        const uint8_t *img8 = PS.src8 + (size_t)n * Hin * Win3;
        const int bx = a0 + 4 * lane;                             // this lane's dword in a frame row
        const bool bx_ok = lane < SF_CROP_PITCH / 4 && bx >= 0 && bx < Win3;
        auto load_row = [&](int r) -> unsigned {                  // a select between addresses, not values
            const bool ok = bx_ok && (unsigned)r < (unsigned)Hin;
            return *reinterpret_cast<const unsigned *>(ok ? img8 + (size_t)r * Win3 + bx : reinterpret_cast<const uint8_t *>(PS.zero));
        };
        auto put_row = [&](int r, unsigned raw) {                 // zero padding is applied after normalisation
            const bool ok = bx_ok && (unsigned)r < (unsigned)Hin;
            h4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (_Float16)(ok ? ((float)((raw >> (8 * q)) & 255u) - PS.in_mean) * PS.in_scale : 0.f);
            if (lane < SF_CROP_PITCH / 4) *reinterpret_cast<h4 *>(crop + (r & (SF_CROP_SLOTS - 1)) * SF_CROP_PITCH + 4 * lane) = o;
        };
        auto stem_row = [&](int ys) {                             // stem row ys of the strip (32 columns) -> ring slot ys & 3
            _Float16 *dst = stemr + (ys & (SF_STEM_SLOTS - 1)) * SF_ROW;
            if ((unsigned)ys >= (unsigned)Ho) {                   // the depthwise layer's zero padding
                *reinterpret_cast<h8 *>(dst + lane * 8) = zero8;
                *reinterpret_cast<h8 *>(dst + 512 + lane * 8) = zero8;
                return;
            }
            const int r0 = 2 * ys - PS.pad_t;                     // frame row of filter row 0
            const int cx = shift + 6 * fr;
            const int L0 = (r0 & (SF_CROP_SLOTS - 1)) * SF_CROP_PITCH + cx, L1 = ((r0 + 1) & (SF_CROP_SLOTS - 1)) * SF_CROP_PITCH + cx,
                      L2 = ((r0 + 2) & (SF_CROP_SLOTS - 1)) * SF_CROP_PITCH + cx;
            const int Pb = fq == 0 ? L0 : fq == 1 ? L1 : L2;
            const int A0 = fq == 3 ? L0 + 8 : Pb, A1 = fq == 3 ? L1 + 7 : Pb, A2 = fq == 3 ? L2 + 6 : Pb;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                h8 xf;
                xf[0] = crop[A0 + 96 * f]; xf[1] = crop[A1 + 1 + 96 * f]; xf[2] = crop[A2 + 2 + 96 * f];
#pragma unroll
                for (int j = 3; j < 8; ++j) xf[j] = crop[Pb + j + 96 * f];
                const f4 c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[0], xf, sb0, 0, 0, 0);
                const f4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ws[1], xf, sb1, 0, 0, 0);
                const int xs = x0 - 1 + 16 * f + fr;
                h8 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) { o[q] = (_Float16)apply_act(c0[q], ACT); o[4 + q] = (_Float16)apply_act(c1[q], ACT); }
                u4v_t ou = __builtin_bit_cast(u4v_t, o);
                ou &= (unsigned)xs < (unsigned)Wo ? 0xFFFFFFFFu : 0u;   // columns outside the map: zero padding
                *reinterpret_cast<u4v_t *>(dst + (fq * 32 + 16 * f + fr) * 8) = ou;
            }
        };
        // rows for stem rows y0-1 and y0, then the two rows step y0 will add
        const int rp = 2 * (y0 - 1) - PS.pad_t;
#pragma unroll
        for (int i = 0; i < 5; ++i) put_row(rp + i, load_row(rp + i));
        unsigned raw0 = load_row(rp + 5), raw1 = load_row(rp + 6);
        stem_row(y0 - 1); stem_row(y0);
        for (int y = y0; y < y1; ++y) {
            const int rn = 2 * (y + 1) - PS.pad_t;                // stem row y+1: frame rows rn (already there), rn+1, rn+2
            put_row(rn + 1, raw0); put_row(rn + 2, raw1);
            raw0 = load_row(rn + 3); raw1 = load_row(rn + 4);
            stem_row(y + 1);
            // ---- depthwise row y: output pixels 2pp, 2pp+1 of the strip = stem columns 2pp .. 2pp+3 of rows y-1 .. y+1
            {
                h8 x[3][4];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        x[kh][c] = *reinterpret_cast<const h8 *>(stemr + ((y - 1 + kh) & (SF_STEM_SLOTS - 1)) * SF_ROW + (cg * 32 + 2 * pp + c) * 8);
                float acc[2][8];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc[j][i] = db0[i]; acc[j][4 + i] = db1[i]; }
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                        for (int j = 0; j < 2; ++j) dw_tap(acc[j], x[kh][j + kw], wd[kh * 3 + kw]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    h8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = (_Float16)apply_act(acc[j][i], ACT);
                    *reinterpret_cast<h8 *>(xt + (cg * 32 + 2 * pp + j) * 8) = o;
                }
            }
            // ---- pointwise row y: [32 pixels] x [64 channels] x 32
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const h8 xf = *reinterpret_cast<const h8 *>(xt + (fq * 32 + 16 * f + fr) * 8);
                f4 acc[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[a], xf, (a & 1) ? E[a >> 1].b1 : E[a >> 1].b0, 0, 0, 0);
                const int px = 16 * f + fr;
                if (px < SF_SW && x0 + px < Wo) {
                    const int m = (n * Ho + y) * Wo + x0 + px;
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        float o[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { o[q] = acc[2 * g2][q]; o[4 + q] = acc[2 * g2 + 1][q]; }
                        conv_epilogue_f16x8<ACT, false, 0, false, false>(P, E[g2], m, g2 * 32 + fq * 8, o);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// MobileNet block with 128 -> 128 channels (depthwise 3x3 stride 1 + pointwise; block 3 of the SSD at 75x75) as strips
// streamed by single waves, like ssd_front_k: a wave owns 6 output columns (8 input columns) of a row range of one frame
// and produces two output rows per round -- one 16-pixel MFMA fragment.  Input rows arrive by global_load_lds (a row
// slot is [8 pixels][16 channel groups] x 16 B = 2 KiB, two DMAs, a pixel's 256 bytes contiguous on both sides); the
// depthwise stage is dwpw_k's arithmetic with three output pixels per lane (lane = channel group x (row, half)), its
// f16 result goes into a [channel group][17 pixel slots] operand tile (the odd pitch keeps the 16 writing lanes on 16
// different bank groups), the pointwise stage is 4 k slices x 8 fragments with the whole 128 x 128 filter in registers.
// dwpw_k runs this block at 3.2 TB/s of algorithmic traffic with 44 % of its wave cycles parked on memory.
constexpr int DR_SW = 6, DR_SLOTS = 6, DR_ROW = 8 * 16 * 8, DR_XT = 16 * 17 * 8;        // halves
constexpr int dr_wave_halves() { return DR_SLOTS * DR_ROW + DR_XT; }

template <int ACT>
__global__ __launch_bounds__(256, 2) void dwpw_rows_k(const ConvP P, const int n_tasks, const int parts) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    _Float16 *ring = lds + (size_t)wave * dr_wave_halves(), *xt = ring + DR_SLOTS * DR_ROW;
    _Float16 *dwt = lds + 4 * dr_wave_halves();                  // block-shared: depthwise taps [9][128] f16, then biases (f32)
    float *dwb = reinterpret_cast<float *>(dwt + 9 * 128), *pwb = dwb + 128;
    if (tid < 9 * 16) *reinterpret_cast<h8 *>(dwt + tid * 8) = *reinterpret_cast<const h8 *>(P.dw_w + (size_t)(tid >> 4) * 128 + (tid & 15) * 8);
    if (tid < 128) { dwb[tid] = P.dw_bias[tid]; pwb[tid] = P.bias[tid]; }
    __syncthreads();

    h8 wp[4][8];                                                  // [k slice][fragment]: the pointwise filter
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int a = 0; a < 8; ++a) wp[ks][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)rw_weight_row(a, fr) * P.kpad + ks * 32 + fq * 8);
    Epi8 E;                                                       // the epilogue takes no constants (bias in the accumulators, no second output)
    E.b0 = E.b1 = E.s0 = E.s1 = E.t0 = E.t1 = f4{0.f, 0.f, 0.f, 0.f};
    const int H = P.H, W = P.W;
    const int strips = (W + DR_SW - 1) / DR_SW;
    // depthwise item of this lane: channel group cg, output row dj of the pair, pixels 3 * dh .. 3 * dh + 2
    const int cg = lane & 15, dj = lane >> 5, dh = (lane >> 4) & 1;
    // pointwise fragment pixel of this lane: row pj of the pair, column px
    const int pj = fr >> 3, px = fr & 7;

    for (int task = blockIdx.x * 4 + wave; task < n_tasks; task += gridDim.x * 4) {
        const int part = task % parts, t2 = task / parts;
        const int strip = t2 % strips, n = t2 / strips;
        const int x0 = strip * DR_SW;
        int y0 = (int)((long long)part * H / parts), y1 = (int)((long long)(part + 1) * H / parts);
        y0 &= ~1; if (part + 1 < parts) y1 &= ~1;                 // parts start on even rows (rounds are row pairs)
        const _Float16 *img = P.in + (size_t)n * H * W * P.cs_in + P.coff_in;
        auto fill_row = [&](int y) {                              // wave-uniform y; outside the map: zero lines
            _Float16 *dst = ring + ((y + 1) % DR_SLOTS) * DR_ROW;
            const bool rok = (unsigned)y < (unsigned)H;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int pxl = 4 * k + (lane >> 4), xin = x0 - 1 + pxl;
                const bool ok = rok && (unsigned)xin < (unsigned)W;
                lds_fill16(ok ? img + ((size_t)y * W + xin) * P.cs_in + (lane & 15) * 8 : P.zero, dst + 512 * k);
            }
        };
        fill_row(y0 - 1); fill_row(y0); fill_row(y0 + 1); fill_row(y0 + 2);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        for (int y = y0; y < y1; y += 2) {
            fill_row(y + 3); fill_row(y + 4);
            // ---- depthwise rows y, y+1: this lane's row y + dj, output pixels 3 dh .. 3 dh + 2 = input pixel slots 3 dh .. 3 dh + 4
            {
                float acc[3][8];
                const f4 b0 = *reinterpret_cast<const f4 *>(dwb + cg * 8), b1 = *reinterpret_cast<const f4 *>(dwb + cg * 8 + 4);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) { acc[i][c] = b0[c]; acc[i][4 + c] = b1[c]; }
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const _Float16 *rs = ring + ((y + dj + kh) % DR_SLOTS) * DR_ROW + (3 * dh * 16 + cg) * 8;      // map row y + dj + kh - 1
                    h8 xv[5];
#pragma unroll
                    for (int c = 0; c < 5; ++c) xv[c] = *reinterpret_cast<const h8 *>(rs + c * 16 * 8);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const h8 wv = *reinterpret_cast<const h8 *>(dwt + (kh * 3 + kw) * 128 + cg * 8);
#pragma unroll
                        for (int i = 0; i < 3; ++i) dw_tap(acc[i], xv[i + kw], wv);
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    h8 o;
#pragma unroll
                    for (int c = 0; c < 8; ++c) o[c] = (_Float16)apply_act(acc[i][c], ACT);
                    *reinterpret_cast<h8 *>(xt + (cg * 17 + dj * 8 + 3 * dh + i) * 8) = o;
                }
            }
            // ---- pointwise rows y, y+1: [16 pixel slots] x [128 channels] x 128, from the bias
            f4 acc[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = *reinterpret_cast<const f4 *>(pwb + (a >> 1) * 32 + fq * 8 + (a & 1) * 4);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h8 xf = *reinterpret_cast<const h8 *>(xt + ((ks * 4 + fq) * 17 + fr) * 8);
#pragma unroll
                for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[ks][a], xf, acc[a], 0, 0, 0);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next round's rows have landed, last round's stores have retired
#endif
            if (px < DR_SW && x0 + px < W && y + pj < y1) {
                const int m = (n * H + y + pj) * W + x0 + px;
#pragma unroll
                for (int g2 = 0; g2 < 4; ++g2) {
                    float o[8];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { o[c] = acc[2 * g2][c]; o[4 + c] = acc[2 * g2 + 1][c]; }
                    conv_epilogue_f16x8<ACT, false, 0, false, false>(P, E, m, g2 * 32 + fq * 8, o);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// MobileNet block in one launch: depthwise 3x3 (+bias, activation) -> pointwise 1x1 (+bias, activation).
// The depthwise result of a pixel tile never leaves the CU: it is rounded to f16 (exactly what the
// two-kernel path stores) into the LDS image the MFMA fragments are read from, so the layer pair
// costs one read of the block input and one write of the block output instead of two of each.
//
// Pixels are taken in "quad order": a quad is 4 consecutive output pixels of one row (the last quad of
// a row may hang over the edge), a block owns BM/4 consecutive quads and each thread computes one
// (quad, 8-channel group) item from a 3 x (3 + 3*stride) window of 16-byte loads, as dwconv3_k does.
// The whole pointwise weight panel [BN][CIN] is resident in LDS (fetched while the depthwise part
// runs); its rows are stored in fragment order of rw_weight_row, so a lane ends up with 8 consecutive
// output channels of a pixel and stores 16 bytes without a transposition pass.
// Halves offset of 16-byte chunk c of logical row `row` in a [rows][CIN] LDS image.  64-byte rows (CIN = 32)
// are additionally stored at row ^ ((row >> 2) & 1): the depthwise phase writes rows r and r + 4 from one
// 8-lane ds_write_b128 group, and unpermuted they sit 256 bytes apart on the same banks (PMC: 27 % of the
// LDS cycles of this kernel were bank conflicts); every aligned quartet of rows stays a quartet, so the
// fragment reads remain conflict-free.
template <int CIN>
__device__ __forceinline__ int dwpw_swz(int c, int row);
template <int CIN>
__device__ __forceinline__ int dwpw_at(int c, int row) {
    const int prow = CIN == 32 ? row ^ ((row >> 2) & 1) : row;
    return prow * CIN + dwpw_swz<CIN>(c, row) * 8;
}
template <int CIN>
__device__ __forceinline__ int dwpw_swz(int c, int row) {         // position of 16-byte chunk c inside LDS row `row`
    if (CIN == 32) return c ^ ((4 - ((row >> 2) & 3)) & 3);
    if (CIN == 64) return c ^ (row & 7);
    return (c & ~15) | ((c & 15) ^ (row & 15));
}

template <int WM, int WN, int MI, int CIN, int STRIDE, int DACT, int ACT>
__global__ __launch_bounds__(256, 2) void dwpw_k(const ConvP P, const int n_tiles) {
    constexpr int NI = 4, BM = WM * MI * 16, BN = WN * 64, G = CIN / 8, KS = CIN / 32;
    constexpr int TX = 4, NCOL = (TX - 1) * STRIDE + 3;
    static_assert(WM * WN == 4 && (BM / 4) * G == 256, "one (quad, channel group) item per thread");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    _Float16 *xs = lds, *ws = lds + BM * CIN;
    int *mrow = reinterpret_cast<int *>(ws + BN * CIN);             // output pixel index of each tile row, -1 = none
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.y * BN;

    // ---- pointwise weight panel -> LDS (rows in fragment order), ONCE per block: blocks are persistent and walk the
    // pixel tiles (tile t of the launch is logical tile dd_xcd_remap(t): each XCD keeps to one contiguous range).  With
    // one tile per block the panel was re-fetched for every 64-256 pixels -- 30 % (128 -> 128 channels) to 37 % (128 -> 256)
    // of everything the block pulled through the texture path, which is what bounds these kernels (10-14 B/clk/CU).
    {
        constexpr int CH = BN * G / 256;
        h8 wv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int idx = tid + i * 256;
            const int j = idx / G, c = idx % G;
            const int co = n0 + (j & ~63) + rw_weight_row((j >> 4) & 3, j & 15);
            wv[i] = *reinterpret_cast<const h8 *>(P.w + (size_t)co * P.kpad + c * 8);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int idx = tid + i * 256;
            const int j = idx / G, c = idx % G;
            *reinterpret_cast<h8 *>(ws + dwpw_at<CIN>(c, j)) = wv[i];
        }
    }

    Epi8 E[2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) E[g2] = epi8_load(P, n0 + wn * 64 + g2 * 32 + fq * 8);
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const int q0 = (int)dd_xcd_remap((unsigned)t, (unsigned)n_tiles) * (BM / 4);
    // ---- depthwise part: this thread's quad and channel group
    {
        const int ql = tid / G, g = tid % G;
        const int Q = q0 + ql;
        const int wo4 = (P.wo + TX - 1) / TX;
        const bool qok = Q < P.total_quads;
        const int Qc = qok ? Q : 0;
        const int oxq = Qc % wo4, t2 = Qc / wo4;
        const int oy = t2 % P.ho, n = t2 / P.ho;
        const int ox0 = oxq * TX;
        float acc[TX][8];
        {
            const f4 b0 = *reinterpret_cast<const f4 *>(P.dw_bias + g * 8), b1 = *reinterpret_cast<const f4 *>(P.dw_bias + g * 8 + 4);
#pragma unroll
            for (int j = 0; j < TX; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[j][i] = b0[i]; acc[j][4 + i] = b1[i]; }
        }
        const int ix0 = ox0 * STRIDE - P.pad_l;
        const _Float16 *img = P.in + (size_t)n * P.H * P.W * P.cs_in + P.coff_in + g * 8;
        // Out-of-image taps read a zero line instead of branching around the load.  HOIST: all 3 x NCOL window loads and
        // the nine filter taps are issued before the first multiply (as in dwconv3_k) -- pays only for the 32-channel
        // block (232 -> 223 us at 192 frames); the others lose more to the registers it costs (block 2: 176 -> 231 us,
        // block 3: 197 -> 229 us), so they keep hipcc's row-by-row order (three batches of loads per item).
        constexpr bool HOIST = CIN == 32;
        if constexpr (HOIST) {
            h8 x[3][NCOL], w[9];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int iy = oy * STRIDE - P.pad_t + kh;
                const bool rok = qok && (unsigned)iy < (unsigned)P.H;
                const int roff = ((rok ? iy : 0) * P.W + ix0) * P.cs_in;   // 32-bit offset inside the image
#pragma unroll
                for (int cx = 0; cx < NCOL; ++cx)
                    x[kh][cx] = *reinterpret_cast<const h8 *>((rok && (unsigned)(ix0 + cx) < (unsigned)P.W) ? img + (roff + cx * P.cs_in) : P.zero);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const h8 *>(P.dw_w + (size_t)t * CIN + g * 8);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int j = 0; j < TX; ++j) dw_tap(acc[j], x[kh][j * STRIDE + kw], w[kh * 3 + kw]);
        } else {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int iy = oy * STRIDE - P.pad_t + kh;
                const bool rok = qok && (unsigned)iy < (unsigned)P.H;
                const int roff = ((rok ? iy : 0) * P.W + ix0) * P.cs_in;   // 32-bit offset inside the image
                h8 x[NCOL];
#pragma unroll
                for (int cx = 0; cx < NCOL; ++cx)
                    x[cx] = *reinterpret_cast<const h8 *>((rok && (unsigned)(ix0 + cx) < (unsigned)P.W) ? img + (roff + cx * P.cs_in) : P.zero);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const h8 w = *reinterpret_cast<const h8 *>(P.dw_w + (size_t)(kh * 3 + kw) * CIN + g * 8);
#pragma unroll
                    for (int j = 0; j < TX; ++j) dw_tap(acc[j], x[j * STRIDE + kw], w);
                }
            }
        }
        const int dact = DACT < 0 ? P.dw_act : DACT;
#pragma unroll
        for (int j = 0; j < TX; ++j) {
            const int prow = ql * TX + j;
            const bool pok = qok && ox0 + j < P.wo;
            h8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (_Float16)apply_act(acc[j][i], dact);
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            u4v ou = __builtin_bit_cast(u4v, o);                   // rows past the edge are zeroed on the packed words
            ou &= pok ? 0xFFFFFFFFu : 0u;                          // (a per-element select gets compiled into 32 branches)
            *reinterpret_cast<u4v *>(xs + dwpw_at<CIN>(g, prow)) = ou;
            if (g == 0) mrow[prow] = pok ? (n * P.ho + oy) * P.wo + ox0 + j : -1;
        }
    }
    __syncthreads();

    // ---- pointwise part: [BM pixels] x [BN channels] x CIN, everything already in LDS
    f4 acc[NI][MI];                                             // start from the bias of the lane's output channels
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (a & 1) ? E[a >> 1].b1 : E[a >> 1].b0;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int c = kk * 4 + fq;
        h8 xf[MI], wf[NI];
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int r = (wm * MI + b) * 16 + fr;
            xf[b] = *reinterpret_cast<const h8 *>(xs + dwpw_at<CIN>(c, r));
        }
#pragma unroll
        for (int a = 0; a < NI; ++a) {
            const int r = (wn * NI + a) * 16 + fr;
            wf[a] = *reinterpret_cast<const h8 *>(ws + dwpw_at<CIN>(c, r));
        }
#pragma unroll
        for (int a = 0; a < NI; ++a)
#pragma unroll
            for (int b = 0; b < MI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = mrow[(wm * MI + b) * 16 + fr];
        if (m < 0) continue;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = acc[2 * g2][b][r]; o[4 + r] = acc[2 * g2 + 1][b][r]; }
            conv_epilogue_f16x8<ACT, false, 0, false>(P, E[g2], m, n0 + wn * 64 + g2 * 32 + fq * 8, o);
        }
    }
    __syncthreads();                                              // the next tile overwrites the pixel image and mrow
    }
}

// ---------------------------------------------------------------------------------------------------
// MobileNet block with many channels (Cin = 256 / 512 / 1024) in one launch: depthwise 3x3 -> pointwise 1x1 as a
// K-looped GEMM whose pixel operand never exists in HBM.  For every 64-channel slab of K the block computes the
// depthwise result of its 128 pixels (+bias, activation, rounded to f16 exactly as dwconv3_k stores it) straight
// into the LDS image the MFMA fragments are read from, while the slab's [256][64] weight panel arrives by
// global_load_lds.  One read of the block input (plus halo, served by L1 / L2) and one write of the block output
// instead of two of each.
//   * 512 threads = 8 waves; tile 128 pixels x 256 output channels (wave: 64 x 64, 64 accumulator registers).
//   * pixels in "pair order": a pair is 2 consecutive output pixels of one row; thread (pair, 8-channel group)
//     owns one depthwise item per slab: 3 x (3 + stride) window loads of 16 bytes through a buffer resource whose
//     range check turns out-of-image taps into zeros (offsets are fixed per thread; the slab advances the scalar
//     offset), nine filter taps from an LDS copy of the depthwise weights.
//   * schedule per slab ks (one barrier): issue the window loads and the weight fill of slab ks+1, MFMAs of slab
//     ks, depthwise of slab ks+1 into the other X buffer.  The loads of a slab are in flight under the MFMAs of
//     the slab before it.
//   * Cout > 256: the depthwise part is recomputed by each of the Cout / 256 channel tiles (they run back to back on
//     one XCD, so the input is read from HBM once).
constexpr int DWB_BM = 128, DWB_BN = 256;

template <int STRIDE, int DACT, int ACT>
__global__ __launch_bounds__(512) void dwpw_big_k(const ConvP P) {
    constexpr int BM = DWB_BM, BN = DWB_BN, MI = 4, NI = 4, TX = 2, NCOL = (TX - 1) * STRIDE + 3;
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int C = P.cin;
    _Float16 *xs = lds;                                          // [2][BM][64]
    _Float16 *ws = lds + 2 * BM * 64;                            // [2][BN][64]
    _Float16 *dww = ws + 2 * BN * 64;                            // [9][C]
    float *dwb = reinterpret_cast<float *>(dww + 9 * C);         // [C]
    int *mrow = reinterpret_cast<int *>(dwb + C);                // [BM] output pixel of each tile row, -1 = none
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
    const unsigned lin = dd_xcd_remap(blockIdx.x, gridDim.x);
    const int gy = P.cout_pad / BN;
    const int p0 = (int)(lin / gy) * (BM / TX), n0 = (int)(lin % gy) * BN;       // first pair of the tile; channel tiles fastest
    const int KS = C >> 6;

    // ---- depthwise weights and bias of the whole layer -> LDS
    for (int i = tid; i < 9 * C / 8; i += 512) reinterpret_cast<h8 *>(dww)[i] = reinterpret_cast<const h8 *>(P.dw_w)[i];
    for (int i = tid; i < C / 4; i += 512) reinterpret_cast<f4 *>(dwb)[i] = reinterpret_cast<const f4 *>(P.dw_bias)[i];

    // ---- this thread's depthwise item: pair pl of the tile, channel group g of the slab
    const int pl = tid >> 3, g = tid & 7;
    unsigned woff[3][NCOL];                                      // byte offsets of the window taps at slab 0; 0x80000000 = outside
    {
        const int wo2 = (P.wo + TX - 1) / TX;
        const int Q = p0 + pl;
        const bool qok = Q < P.total_quads;
        const int Qc = qok ? Q : 0;
        const int oxq = Qc % wo2, t2 = Qc / wo2;
        const int oy = t2 % P.ho, n = t2 / P.ho;
        const int ox0 = oxq * TX;
        const int ix0 = ox0 * STRIDE - P.pad_l, iy0 = oy * STRIDE - P.pad_t;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int cx = 0; cx < NCOL; ++cx) {
                const int iy = iy0 + kh, ix = ix0 + cx;
                const bool ok = qok && (unsigned)iy < (unsigned)P.H && (unsigned)ix < (unsigned)P.W;
                woff[kh][cx] = ok ? (unsigned)((((n * P.H + iy) * P.W + ix) * P.cs_in + P.coff_in + g * 8) * 2) : 0x80000000u;
            }
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < TX; ++j) mrow[pl * TX + j] = (qok && ox0 + j < P.wo) ? (n * P.ho + oy) * P.wo + ox0 + j : -1;
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(P.in), 0, P.p[0], 0x00020000);
#endif
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    u4v win[3][NCOL];
    auto load_window = [&](int ks) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int cx = 0; cx < NCOL; ++cx) win[kh][cx] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, woff[kh][cx], ks * 128, 0);
#endif
    };
    // weight panel rows in fragment order (see rw_weight_row), 16-byte chunks XOR-swizzled on the source side
    const int rr = lane >> 3, pp = lane & 7;
    const _Float16 *wbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = (wave * 4 + i) * 8 + rr;
        const int row = (L & ~31) | (((L & 15) >> 2) << 3) | (((L >> 4) & 1) << 2) | (L & 3);
        wbase[i] = P.w + (size_t)(n0 + row) * P.kpad + (pp ^ rr) * 8;
    }
    auto fill_w = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_fill16(wbase[i] + (ks << 6), ws + (size_t)(buf * BN + (wave * 4 + i) * 8) * 64);
    };
    auto depthwise = [&](int ks, int buf) {                      // win[] holds slab ks -> X[buf]
        const int c0 = (ks << 6) + g * 8;
        float acc[TX][8];
        {
            const f4 b0 = *reinterpret_cast<const f4 *>(dwb + c0), b1 = *reinterpret_cast<const f4 *>(dwb + c0 + 4);
#pragma unroll
            for (int j = 0; j < TX; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[j][i] = b0[i]; acc[j][4 + i] = b1[i]; }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const h8 w = *reinterpret_cast<const h8 *>(dww + (kh * 3 + kw) * C + c0);
#pragma unroll
                for (int j = 0; j < TX; ++j) dw_tap(acc[j], __builtin_bit_cast(h8, win[kh][j * STRIDE + kw]), w);
            }
        const int dact = DACT < 0 ? P.dw_act : DACT;
#pragma unroll
        for (int j = 0; j < TX; ++j) {
            const int prow = pl * TX + j;
            h8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (_Float16)apply_act(acc[j][i], dact);
            u4v ou = __builtin_bit_cast(u4v, o);
            ou &= mrow[prow] >= 0 ? 0xFFFFFFFFu : 0u;               // rows past the edge of the image / batch are zero
            *reinterpret_cast<u4v *>(xs + (size_t)(buf * BM + prow) * 64 + ((g ^ (prow & 7)) << 3)) = ou;
        }
    };

    // ---- pointwise accumulators start from the bias of the lane's output channels
    Epi8 E[2];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) E[g2] = epi8_load(P, n0 + wn * 64 + g2 * 32 + fq * 8);
    f4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = (a & 1) ? E[a >> 1].b1 : E[a >> 1].b0;

    load_window(0);
    fill_w(0, 0);
    __syncthreads();                                              // dww / dwb / mrow visible (and slab 0 landed)
    depthwise(0, 0);
    __syncthreads();
    for (int ks = 0; ks < KS; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < KS) { load_window(ks + 1); fill_w(ks + 1, buf ^ 1); }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 xf[MI], wf[NI];
#pragma unroll
            for (int b = 0; b < MI; ++b)
                xf[b] = *reinterpret_cast<const h8 *>(xs + (size_t)(buf * BM + (wm * MI + b) * 16 + fr) * 64 + (((kk << 2) + fq) ^ sw) * 8);
#pragma unroll
            for (int a = 0; a < NI; ++a)
                wf[a] = *reinterpret_cast<const h8 *>(ws + (size_t)(buf * BN + (wn * NI + a) * 16 + fr) * 64 + (((kk << 2) + fq) ^ sw) * 8);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        if (ks + 1 < KS) depthwise(ks + 1, buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < MI; ++b) {
        const int m = mrow[(wm * MI + b) * 16 + fr];
        if (m < 0) continue;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = acc[2 * g2][b][r]; o[4 + r] = acc[2 * g2 + 1][b][r]; }
            conv_epilogue_f16x8<ACT, false, 0>(P, E[g2], m, n0 + wn * 64 + g2 * 32 + fq * 8, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight-stationary pointwise GEMM for the 1x1 layers with many pixels (MobileNet pointwise convs).
//
// conv_glds_k moves BOTH operands of every K step from L2 to LDS and is bound by that fill rate (17 B/clk per CU
// measured).  Here the weight operand never touches LDS: a block of NW waves owns a slice of 32 * NW
// output channels, every wave keeps the [32 channels][K] rows it needs as MFMA A-fragments in registers for the
// whole launch (K = 512: 128 VGPRs), and persistent blocks stream 64-pixel tiles of the activation matrix through
// an LDS ring of D stages, one 64-wide K slab (8 KiB) per stage:
//   * fills are global_load_lds_dwordx4 issued D - 1 stages ahead and waited for with a counted vmcnt (never 0 inside
//     the loop) followed by a raw s_barrier, so (D - 2) x 8 KiB stay in flight across barriers;
//   * a filled byte feeds 32 * NW output channels (NW = 8: 256 FLOP per byte, twice what a 256 x 256 tile of the
//     two-operand kernel gets) and the LDS is read for the pixel fragments only;
//   * pixel fragments are read one K slice ahead of the MFMAs that use them (the first slice of stage s + 1 under the
//     last MFMAs of stage s), so an LDS read latency is never exposed behind a barrier;
//   * blocks that share a pixel range but own different channel slices sit on one XCD and walk their tiles in step,
//     so the activations leave HBM once.
// Epilogue stores count in vmcnt like the fills; the loop's wait is the store-free count, which only shortens the
// prefetch distance for the few stages after a tile's stores (never too short a wait).
constexpr int WS_BM = 64;

template <int KS, int NW, int D, int ACT, int SPB = 1>
__global__ __launch_bounds__(NW * 64, 2) void conv_ws_k(const ConvP P, const int n_slices) {
    static_assert((SPB == 1 || SPB == 2) && KS % SPB == 0 && D >= 2 * SPB + 2, "stages per barrier");
    constexpr int BM = WS_BM, MI = 4, NI = 2, G = 8 / NW;       // G: fills per wave and stage
    static_assert((KS & (KS - 1)) == 0 && (D & (D - 1)) == 0 && KS % 2 == 0 && 8 % NW == 0, "stage arithmetic uses masks");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];         // [D][64 pixels][64 halves]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slice = j % n_slices, worker = (j / n_slices) * 8 + xcd;
    const int n_workers = ((int)(gridDim.x >> 3) / n_slices) * 8;
    const int T = (P.m + BM - 1) / BM;
    const int ntiles = worker < T ? (T - worker + n_workers - 1) / n_workers : 0;
    const int cbase = slice * (32 * NW) + wave * 32;

    h8 wf[KS * 2][NI];                                          // this wave's weight rows, every K slice
#pragma unroll
    for (int k = 0; k < KS * 2; ++k)
#pragma unroll
        for (int a = 0; a < NI; ++a)
            wf[k][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)(cbase + rw_weight_row(a, fr)) * P.kpad + k * 32 + fq * 8);
    const Epi8 E = epi8_load(P, cbase + fq * 8);
    f4 acc[NI][MI];                                             // from zero, bias in the epilogue: the summation order of
#pragma unroll                                                  // conv_glds_k, so a frame gives the same bits whichever kernel
    for (int a = 0; a < NI; ++a)                                // its batch size selects
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#ifdef DD_KERNEL_DBG
    const int dbg = P.p[5];                                      // measurement aid (DD_WS_MODE), bits: 1 = no fills in the loop, 2 = no MFMAs, 4 = no epilogue, 8 = no fragment reads
#else
    constexpr int dbg = 0;                                       // compiled out of the product build (-DDD_KERNEL_DBG brings it back)
#endif

    const int rr = lane >> 3, pp = lane & 7;
    const int gch = (pp ^ rr) * 8;
    auto fill = [&](int st) {                                    // stage st of this block: tile st / KS, K slab st % KS
        const int ti = st / KS, ks = st & (KS - 1);
        const int m0 = (worker + ti * n_workers) * BM;
        _Float16 *dst = lds + (size_t)(st & (D - 1)) * (BM * 64);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int grp = wave * G + i;
            const int m = m0 + grp * 8 + rr;
            const _Float16 *src = (ti < ntiles && m < P.m) ? P.in + (size_t)m * P.cs_in + P.coff_in + (ks << 6) + gch : P.zero;
            lds_fill16(src, dst + grp * 8 * 64);                 // past the last tile: a zero line, so the wait count stays uniform
        }
    };
    h8 xa[MI], xb[MI];                                           // pixel fragments: K slice 0 of the next stage / slice 1 of this one
    auto read_frags = [&](int st, int kk, h8 (&x)[MI]) {
        const _Float16 *xs = lds + (size_t)(st & (D - 1)) * (BM * 64);
#pragma unroll
        for (int b = 0; b < MI; ++b)
            x[b] = *reinterpret_cast<const h8 *>(xs + (b * 16 + fr) * 64 + (((kk << 2) + fq) ^ sw) * 8);
    };
    // SPB stages per barrier (DD_WS_SPB).  SPB = 2 = one wait + barrier per TWO K slabs, 32 MFMAs per wave between barriers instead of
    // 16 (fills D - 2 .. D - 1 stages ahead, D - 5 stages left in flight by the wait): built to test whether barrier skew between the
    // four waves of a block is what a stage pays for -- it is not: no change on either layer (round 3, same-box A/B).  Default 1.
    for (int st = 0; st < D - SPB; ++st) fill(st);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G * (D - SPB - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
    read_frags(0, 0, xa);
    for (int ti = 0; ti < ntiles; ++ti) {
        const int m0 = (worker + ti * n_workers) * BM;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int st = ti * KS + ks;
            if (ks % SPB == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
                // this wave's fills of stages st + 1 .. st + SPB have landed and its reads of the stages before st have returned ...
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(G * (D - 2 * SPB - 1)) : "memory");
                __builtin_amdgcn_s_barrier();                    // ... everybody's have
                asm volatile("" ::: "memory");
#endif
                if (!(dbg & 1)) {
#pragma unroll
                    for (int q = 0; q < SPB; ++q) fill(st + D - SPB + q);   // into the buffers stages st - SPB .. st - 1 have left
                }
            }
            if (!(dbg & 8)) read_frags(st, 1, xb);               // lands under the MFMAs of slice 0
            if (!(dbg & 2)) {
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks * 2][a], xa[b], acc[a][b], 0, 0, 0);
            }
            if (!(dbg & 8)) read_frags(st + 1, 0, xa);           // next stage's slice 0, under the MFMAs of slice 1
            if (!(dbg & 2)) {
#pragma unroll
                for (int a = 0; a < NI; ++a)
#pragma unroll
                    for (int b = 0; b < MI; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks * 2 + 1][a], xb[b], acc[a][b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = m0 + b * 16 + fr;
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[r] = acc[0][b][r]; o[4 + r] = acc[1][b][r]; }
            // Plain stores, not the streaming ones of the other GEMM epilogues: a wave writes 64-byte halves of a pixel's lines here, and
            // the counted waits of the K loop count stores too -- a store that has to reach HBM holds the next stages' fills back.
            // Ablation at 384 frames, 19x19x512 -> 512 (-DDD_KERNEL_DBG, DD_WS_MODE bits): all 98 us, no MFMAs 94, no fills 65, no
            // epilogue 70, fills alone 44, loop skeleton 19: the matrix work is free, the layer is its memory path (146 MB read +
            // 182 MB written at the fabric for 142 + 142); plain stores 100.5 -> 89.6 us (38x38x256: 122.9 -> 111.0), same bits.
            if (m < P.m && !(dbg & 4)) conv_epilogue_f16x8<ACT, true, 0, false>(P, E, m, cbase + fq * 8, o);
            acc[0][b] = acc[1][b] = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the look-ahead fills past the last tile
#endif
}

// conv_ws_k with the pixel tiles staged THROUGH REGISTERS instead of by LDS-DMA.  In-kernel stamps (round 3, -DDD_KERNEL_STAMP build
// of conv_ws_k: scripts/stamp_conv_ws.sh) put the two global_load_lds of a stage at 336 of its 876 cycles per wave -- with the
// address chain hoisted out of the loop still 233: an LDS-DMA costs the issuing wave ~115 cycles here, and a weight-stationary wave
// issues two of them per 16 MFMAs.  A global_load_dwordx4 to registers costs the wave a few cycles to issue and a ds_write_b128 13;
// the price is 32 VGPRs for four stages in flight.  Stage s is loaded LA + 2 = 6 stages before its MFMAs, written to LDS two
// stages before them (behind that stage's barrier: the buffer's last reader is two barriers back), read like conv_ws_k reads it.
// hipcc counts these loads itself (no LDS-DMA in the kernel), so the waits in front of the ds_writes are exact.  Same LDS image,
// same fragment reads, same MFMA order, same epilogue: the same bits.
template <int KS, int NW, int ACT, int NG = 1>
__global__ __launch_bounds__(NW * 64) void conv_wsr_k(const ConvP P, const int n_slices) {
    constexpr int BM = WS_BM, MI = 4, NI = 2 * NG, G = 8 / NW, D = 4, LA = 4;
    static_assert(KS % LA == 0 && 8 % NW == 0, "register sets rotate with the unrolled K loop");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];         // [D][64 pixels][64 halves]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int slice = j % n_slices, worker = (j / n_slices) * 8 + xcd;
    const int n_workers = ((int)(gridDim.x >> 3) / n_slices) * 8;
    const int T = (P.m + BM - 1) / BM;
    const int ntiles = worker < T ? (T - worker + n_workers - 1) / n_workers : 0;
    const int cbase = slice * (32 * NG * NW) + wave * (32 * NG);

    h8 wf[KS * 2][NI];
#pragma unroll
    for (int k = 0; k < KS * 2; ++k)
#pragma unroll
        for (int a = 0; a < NI; ++a)
            wf[k][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)(cbase + rw_weight_row(a, fr)) * P.kpad + k * 32 + fq * 8);
    Epi8 E[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) E[g] = epi8_load(P, cbase + g * 32 + fq * 8);
    f4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

    const int rr = lane >> 3, pp = lane & 7;
    const int gch = (pp ^ rr) * 8;
    // per-lane source pointers of a tile's G row groups (K chunk gch of pixel row grp * 8 + rr), once per tile; a load is then the
    // pointer plus an immediate K-slab offset.  Rows past the tensor / tiles past the last one read pixel 0 (never stored).
    auto tile_base = [&](int ti, const _Float16 *(&tb)[G]) {
        const int m0 = (worker + ti * n_workers) * BM;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int m = m0 + (wave * G + i) * 8 + rr;
            tb[i] = P.in + (size_t)((ti < ntiles && m < P.m) ? m : 0) * P.cs_in + P.coff_in + gch;
        }
    };
    h8 rs[LA][G];                                                // stages in flight: set s % LA holds stage s between its load and its LDS write
    auto load_stage = [&](const _Float16 *const (&tb)[G], int ksp, h8 (&r)[G]) {
#pragma unroll
        for (int i = 0; i < G; ++i) r[i] = *reinterpret_cast<const h8 *>(tb[i] + ksp * 64);
    };
    auto write_stage = [&](int buf, const h8 (&r)[G]) {
        _Float16 *dst = lds + (size_t)buf * (BM * 64);
#pragma unroll
        for (int i = 0; i < G; ++i) *reinterpret_cast<h8 *>(dst + ((wave * G + i) * 8 + rr) * 64 + pp * 8) = r[i];
    };
    h8 xa[MI], xb[MI];
    auto read_frags = [&](int st, int kk, h8 (&x)[MI]) {
        const _Float16 *xs = lds + (size_t)(st & (D - 1)) * (BM * 64);
#pragma unroll
        for (int b = 0; b < MI; ++b)
            x[b] = *reinterpret_cast<const h8 *>(xs + (b * 16 + fr) * 64 + (((kk << 2) + fq) ^ sw) * 8);
    };
    constexpr int AH = LA + 2;                                   // a stage is loaded AH stages ahead: tile ti + OA or ti + OA + 1
    constexpr int OA = AH / KS;
    const _Float16 *tbA[G], *tbB[G];
    {   // stages 0, 1 -> LDS; stages 2 .. AH - 1 -> their register sets
        const _Float16 *t[G];
#pragma unroll
        for (int tt = 0; tt * KS < AH; ++tt) {
            tile_base(tt, t);
#pragma unroll
            for (int st = tt * KS; st < (tt + 1) * KS && st < AH; ++st) {
                load_stage(t, st - tt * KS, rs[st % LA]);
                if (st < 2) write_stage(st, rs[st % LA]);
            }
        }
    }
    tile_base(OA, tbA);
    tile_base(OA + 1, tbB);
    __syncthreads();
    read_frags(0, 0, xa);
    for (int ti = 0; ti < ntiles; ++ti) {
        const int m0 = (worker + ti * n_workers) * BM;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int st = ti * KS + ks;
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my writes of stage st + 1 (last stage) and my reads of stage st - 1 are done ...
            __builtin_amdgcn_s_barrier();                        // ... everybody's are
            asm volatile("" ::: "memory");
#endif
            write_stage((st + 2) & (D - 1), rs[(ks + 2) % LA]);  // stage st + 2 (its loads were issued LA stages ago)
            if ((ks + AH) / KS == OA) load_stage(tbA, (ks + AH) % KS, rs[(ks + 2) % LA]);   // stage st + AH into the set just written out
            else load_stage(tbB, (ks + AH) % KS, rs[(ks + 2) % LA]);
            read_frags(st, 1, xb);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks * 2][a], xa[b], acc[a][b], 0, 0, 0);
            read_frags(st + 1, 0, xa);
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks * 2 + 1][a], xb[b], acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            const int m = m0 + b * 16 + fr;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float o[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[r] = acc[2 * g][b][r]; o[4 + r] = acc[2 * g + 1][b][r]; }
                if (m < P.m) conv_epilogue_f16x8<ACT, true, 0>(P, E[g], m, cbase + g * 32 + fq * 8, o);
                acc[2 * g][b] = acc[2 * g + 1][b] = f4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) tbA[i] = tbB[i];
        tile_base(ti + OA + 2, tbB);
    }
}

struct DwP {
    const _Float16 *in; int H, W, cs_in, coff_in;
    const _Float16 *w; const float *bias;
    int stride, pad_t, pad_l, ho, wo, c, m, act;
    _Float16 *out; int cs_out, coff_out;
    const _Float16 *zero;           // >= 16 bytes of zeros: what out-of-image taps read
};

// conv_ws_k with the NEXT block's depthwise 3x3 (stride 1) folded into its epilogue -- a depthwise convolution is per
// channel, so the wave that computes 32 output channels of the pointwise layer for a run of whole frames can apply the
// depthwise taps to exactly those channels without sharing anything: its activated f16 outputs go into a private ring of
// 128 pixels x 4 channel groups in LDS instead of memory, and once a 64-pixel tile is in, the 64 pixels that now have their
// whole 3x3 neighbourhood (one map row + one pixel behind) get their nine taps from the ring (taps outside the frame read a
// zero chunk), bias, activation and a 16-byte store.  The pointwise output tensor and the depthwise launch disappear.
// Workers own whole frames (the pixel stream of a worker is contiguous, so the ring just keeps turning across frames);
// four LDS stages instead of eight leave room for the rings at two blocks per CU (+0.5 % on the SSD forward by itself).
// Same summation orders as conv_glds_k / conv_ws_k and dwconv3_k: same bits as the two launches.
// The ring has 128 slots per channel group plus 48 that mirror slots 0..47: the nine taps of a pixel then sit at compile-time
// offsets (kh * W + kw) * 16 from the slot of its top-left tap with no wrap-around arithmetic -- the depthwise step is 72
// multiply-adds, 18 LDS reads, nine address selects (an invalid tap points at a zero chunk) and little else.
constexpr int WSD_RING = 128, WSD_MIRROR = 48, WSD_SLOTS = WSD_RING + WSD_MIRROR;

template <int KS, int WMAP, int ACT, int DACT>
__global__ __launch_bounds__(256, 2) void conv_ws_dw_k(const ConvP P, const DwP Q, const int n_slices, const int n_frames) {
    constexpr int NW = 4, D = 4, BM = WS_BM, MI = 4, NI = 2, G = 8 / NW;
    static_assert((KS & (KS - 1)) == 0 && KS % 2 == 0 && KS >= D, "stage arithmetic uses masks");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];         // [D][64 pixels][64 halves] | 4 x y ring | zero chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slice = jb % n_slices, worker = (jb / n_slices) * 8 + xcd;
    const int n_workers = ((int)(gridDim.x >> 3) / n_slices) * 8;
    constexpr int Wm = WMAP, L = Wm + 1;
    static_assert(2 * WMAP + 2 < WSD_MIRROR && 64 + 2 * (WMAP + 1) <= WSD_RING, "ring sizes");
    const int HW = Q.H * Wm, Hm = Q.H;
    const int f0 = (int)((long long)worker * n_frames / n_workers), f1 = (int)((long long)(worker + 1) * n_frames / n_workers);
    const int px0 = f0 * HW, px1 = f1 * HW;                        // this worker's pixels (whole frames)
    const int ntiles = (px1 - px0 + BM - 1) / BM;
    const int cbase = slice * (32 * NW) + wave * 32;
    _Float16 *yring = lds + D * BM * 64 + wave * (4 * WSD_SLOTS * 8);
    _Float16 *zchunk = lds + D * BM * 64 + NW * (4 * WSD_SLOTS * 8), *dwt = zchunk + 8;      // zero chunk | depthwise taps [9][128 channels of the slice]
    if (tid < 8) zchunk[tid] = (_Float16)0.f;
    if (tid < 9 * 16) {                                         // (an ordinary global load inside the loop would make hipcc drain the fill queue at its use)
        const int t = tid >> 4, c8 = tid & 15;
        *reinterpret_cast<h8 *>(dwt + t * 128 + c8 * 8) = *reinterpret_cast<const h8 *>(Q.w + (size_t)t * Q.c + slice * 128 + c8 * 8);
    }
    __syncthreads();

    h8 wf[KS * 2][NI];                                          // this wave's weight rows, every K slice
#pragma unroll
    for (int k = 0; k < KS * 2; ++k)
#pragma unroll
        for (int a = 0; a < NI; ++a)
            wf[k][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)(cbase + rw_weight_row(a, fr)) * P.kpad + k * 32 + fq * 8);
    const f4 pb0 = *reinterpret_cast<const f4 *>(P.bias + cbase + fq * 8), pb1 = *reinterpret_cast<const f4 *>(P.bias + cbase + fq * 8 + 4);
    f4 acc[NI][MI];
#pragma unroll
    for (int a = 0; a < NI; ++a)
#pragma unroll
        for (int b = 0; b < MI; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

#ifdef DD_KERNEL_DBG
    const int dbg = P.p[5];                                      // measurement aid (DD_WS_MODE), bits: 1 no fills in the loop, 2 no MFMAs, 4 no depthwise steps, 8 no ring writes
#else
    constexpr int dbg = 0;
#endif
    const int rr = lane >> 3, pp = lane & 7;
    const int gch = (pp ^ rr) * 8;
    auto fill = [&](int st) {                                    // stage st of this block: tile st / KS, K slab st % KS
        const int ti = st / KS, ks = st & (KS - 1);
        const int m0 = px0 + ti * BM;
        _Float16 *dst = lds + (size_t)(st & (D - 1)) * (BM * 64);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int grp = wave * G + i;
            const int m = m0 + grp * 8 + rr;
            const _Float16 *src = (ti < ntiles && m < px1) ? P.in + (size_t)m * P.cs_in + P.coff_in + (ks << 6) + gch : P.zero;
            lds_fill16(src, dst + grp * 8 * 64);                 // past the last tile: a zero line, so the wait count stays uniform
        }
    };
    h8 xa[MI], xb[MI];
    auto read_frags = [&](int st, int kk, h8 (&x)[MI]) {
        const _Float16 *xs = lds + (size_t)(st & (D - 1)) * (BM * 64);
#pragma unroll
        for (int b = 0; b < MI; ++b)
            x[b] = *reinterpret_cast<const h8 *>(xs + (b * 16 + fr) * 64 + (((kk << 2) + fq) ^ sw) * 8);
    };
    // depthwise pass: the 64 pixels [p_lo, p_lo + 64) of this worker's stream, 16 per step (lane = pixel x channel group)
    const unsigned inv_hw = 0xFFFFFFFFu / (unsigned)HW + 1u, inv_w = 0xFFFFFFFFu / (unsigned)Wm + 1u;     // exact q / d for q < 2^16 * ... (checked by the launcher)
    const f4 db0 = *reinterpret_cast<const f4 *>(Q.bias + cbase + fq * 8), db1 = *reinterpret_cast<const f4 *>(Q.bias + cbase + fq * 8 + 4);
    const int x_step = 16 % Wm, y_step = 16 / Wm;                 // what 16 pixels further means for (y, x)
    const char *lds_c = reinterpret_cast<const char *>(lds);
    const unsigned ring_b = (unsigned)((yring - lds) * 2) + (unsigned)fq * (WSD_SLOTS * 16), zero_b = (unsigned)((zchunk - lds) * 2);
    // The depthwise pass of a tile (64 pixels = four steps of 16) runs INSIDE the next tile's K loop, a step every KS / 4
    // stages: its ~300 vector instructions issue while the matrix pipe works through the stage's 32 MFMAs instead of
    // after them.  (The ring slots it reads are only overwritten by the next tile's epilogue, which comes after its K loop.)
    const _Float16 *wtap = dwt + wave * 32 + fq * 8;               // this lane's taps (its channel group): re-read per use, 36 registers short
    int dq = 0, dy = 0, dx = 0, dleft = 0;                        // pending pass: next pixel of this lane, its (y, x), steps left
    auto dw_begin = [&](int p_lo) {                               // (y, x) of the lane's first pixel by division, the rest by stepping
        dq = p_lo + fr;
        const unsigned qc = (unsigned)(dq + HW);                 // one frame up: the first pass starts a row + a pixel before the stream (never
        const unsigned f = __umulhi(qc, inv_hw), r = qc - f * (unsigned)HW;      // stored, but it must step into the first real pixels correctly)
        dy = (int)__umulhi(r, inv_w); dx = (int)r - dy * Wm;
        dleft = 4;
    };
    auto dw_step = [&]() {
        if (dleft == 0) return;                                  // wave-uniform
        --dleft;
        const int q = dq, y = dy, x = dx;
        const bool live = q >= px0 && q < px1;
        // nine reads (taps outside the frame read the zero chunk), then the 72 multiply-adds.  32-bit LDS byte offsets: the
        // top-left tap of pixel q sits in ring slot (q - px0 - L) & 127 of plane fq, tap (kh, kw) 16 * (kh * W + kw) bytes on
        const unsigned p_tl = ring_b + (((unsigned)(q - px0 - L)) & (WSD_RING - 1)) * 16u;
        const bool top = y > 0, bot = y < Hm - 1, lef = x > 0, rig = x < Wm - 1;
        float a[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = db0[i]; a[4 + i] = db1[i]; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            h8 xv[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                constexpr int dummy = 0; (void)dummy;
                const bool ok = (kh == 0 ? top : kh == 2 ? bot : true) && (kw == 0 ? lef : kw == 2 ? rig : true);
                const unsigned toff = (unsigned)(kh * Wm + kw) * 16u;                  // compile-time: becomes the ds_read offset
                xv[kw] = *reinterpret_cast<const h8 *>(lds_c + (ok ? p_tl : zero_b - toff) + toff);
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) dw_tap(a, xv[kw], *reinterpret_cast<const h8 *>(wtap + (kh * 3 + kw) * 128));
        }
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (_Float16)apply_act(a[i], DACT);
        if (live) *reinterpret_cast<h8 *>(Q.out + (size_t)q * Q.cs_out + Q.coff_out + cbase + fq * 8) = o;
        dq += 16;
        dx += x_step; dy += y_step;
        if (dx >= Wm) { dx -= Wm; ++dy; }
        if (dy >= Hm) dy -= Hm;                                  // into the next frame
    };

    for (int st = 0; st < D - 1; ++st) fill(st);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(G * (D - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#endif
    read_frags(0, 0, xa);
    for (int ti = 0; ti < ntiles; ++ti) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int st = ti * KS + ks;
#if defined(__HIP_DEVICE_COMPILE__)
            // (the depthwise steps' stores count in vmcnt too: the wait is the store-free count, i.e. at most one fill stricter than
            // needed; an exact count with an unconditional store per step -- lanes without a pixel writing to a dump -- was slower)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(G * (D - 3)) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#endif
            if (!(dbg & 1)) fill(st + D - 1);
            read_frags(st, 1, xb);
            if (!(dbg & 2)) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks * 2][a], xa[b], acc[a][b], 0, 0, 0);
            }
            read_frags(st + 1, 0, xa);
            if (!(dbg & 2)) {
#pragma unroll
            for (int a = 0; a < NI; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks * 2 + 1][a], xb[b], acc[a][b], 0, 0, 0);
            }
            if ((ks + 1) % (KS / 4) == 0 && !(dbg & 4)) dw_step();             // the previous tile's depthwise pass, a quarter at a time
        }
        // pointwise epilogue into the ring (bias, activation, f16: conv_epilogue_f16x8's arithmetic)
#pragma unroll
        for (int b = 0; b < MI; ++b) {
            h8 yv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                yv[r] = (_Float16)apply_act(acc[0][b][r] + pb0[r], ACT);
                yv[4 + r] = (_Float16)apply_act(acc[1][b][r] + pb1[r], ACT);
            }
            const int slot = (ti * BM + b * 16 + fr) & (WSD_RING - 1);
            if (!(dbg & 8)) *reinterpret_cast<h8 *>(yring + (fq * WSD_SLOTS + slot) * 8) = yv;
            if (slot < WSD_MIRROR && !(dbg & 8)) *reinterpret_cast<h8 *>(yring + (fq * WSD_SLOTS + slot + WSD_RING) * 8) = yv;
            acc[0][b] = acc[1][b] = f4{0.f, 0.f, 0.f, 0.f};
        }
        dw_begin(px0 + ti * BM - L);
    }
    for (int i = 0; i < 4; ++i) dw_step();                       // the last tile's pass,
    dw_begin(px0 + ntiles * BM - L);                             // then the last map row + pixel of the worker's last frame
    for (int i = 0; i < 4; ++i) dw_step();
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the look-ahead fills past the last tile
#endif
}

// Split-K tail: sum the partial slabs in a fixed order (bitwise reproducible) and run the epilogue.
__global__ __launch_bounds__(256) void conv_splitk_finish_k(const ConvP P) {
    const int groups = P.cout_pad >> 2;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;   // launcher: m * groups < 2^31 (64-bit divisions cost ~100 instructions each)
    if (idx >= (unsigned)P.m * (unsigned)groups) return;
    const int m = (int)(idx / (unsigned)groups), co = (int)(idx % (unsigned)groups) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < P.splitk; ++z) {
        const f4 s = *reinterpret_cast<const f4 *>(P.slab + ((size_t)z * P.m + m) * P.cout_pad + co);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += s[r];
    }
    conv_epilogue(P, m, co, v, P.ho * P.wo);
}

// Depthwise 3x3: one lane per (pixel, 8-channel group); weights [3][3][C] f16, bias f32 [C].

// Each lane produces TX = 4 consecutive output pixels of one row for one 8-channel group: the
// 3 x (3 + 3*stride) input window is loaded once (18 or 27 sixteen-byte loads instead of 36) and the
// nine filter taps stay in registers.
template <int STRIDE, int ACT, int TY = 1>                    // ACT >= 0: activation known at compile time; TY output rows per item
__global__ __launch_bounds__(256) void dwconv3_k(const DwP P) {
    constexpr int TX = 4, NCOL = (TX - 1) * STRIDE + 3, NROW = (TY - 1) * STRIDE + 3;
    const int groups = P.c >> 3;
    const int wo4 = (P.wo + TX - 1) / TX, hoy = (P.ho + TY - 1) / TY;
    const unsigned total = (unsigned)(P.m / (P.wo * P.ho)) * hoy * wo4 * groups;     // < 2^31 (checked by the launcher)
    const unsigned idx = dd_xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % (unsigned)groups);                          // 32-bit: a 64-bit division costs ~100 instructions
    unsigned t = idx / (unsigned)groups;
    const int ox0 = (int)(t % (unsigned)wo4) * TX;
    t /= (unsigned)wo4;
    const int oy0 = (int)(t % (unsigned)hoy) * TY, n = (int)(t / (unsigned)hoy);
    float acc[TY][TX][8];
    {
        const f4 b0 = *reinterpret_cast<const f4 *>(P.bias + g * 8), b1 = *reinterpret_cast<const f4 *>(P.bias + g * 8 + 4);
#pragma unroll
        for (int r = 0; r < TY; ++r)
#pragma unroll
            for (int j = 0; j < TX; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[r][j][i] = b0[i]; acc[r][j][4 + i] = b1[i]; }
    }
    const int ix0 = ox0 * STRIDE - P.pad_l;
    const _Float16 *img = P.in + (size_t)n * P.H * P.W * P.cs_in + P.coff_in + g * 8;
    // Every load of the item is issued before the first multiply: left to itself hipcc sinks each load next to its use
    // to save registers (66 VGPRs) and the item becomes a chain of ~25 dependent memory round trips (PMC: waves parked
    // on s_waitcnt 78-85 % of their cycles, 3.4-4 TB/s); with all window loads and the nine filter taps in flight at
    // once a wave waits for memory once.  TY = 2 (stride 1): two output rows share their four input rows -- 24 loads
    // per 8 output pixels instead of 18 per 4, i.e. a third fewer bytes through the texture path per output.
    h8 x[NROW][NCOL], w[9];
#pragma unroll
    for (int kh = 0; kh < NROW; ++kh) {
        const int iy = oy0 * STRIDE - P.pad_t + kh;
        const bool rok = (unsigned)iy < (unsigned)P.H;        // out-of-image taps read the zero line: no branches, all loads in flight
        const int roff = ((rok ? iy : 0) * P.W + ix0) * P.cs_in;     // 32-bit offset inside the image (64-bit index math per load doubled the address code)
#pragma unroll
        for (int cx = 0; cx < NCOL; ++cx)
            x[kh][cx] = *reinterpret_cast<const h8 *>((rok && (unsigned)(ix0 + cx) < (unsigned)P.W) ? img + (roff + cx * P.cs_in) : P.zero);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const h8 *>(P.w + (size_t)t * P.c + g * 8);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int r = 0; r < TY; ++r)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)                        // per output the order (kh, kw) of the one-row form: same bits
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int j = 0; j < TX; ++j) dw_tap(acc[r][j], x[r * STRIDE + kh][j * STRIDE + kw], w[kh * 3 + kw]);
#pragma unroll
    for (int r = 0; r < TY; ++r) {
        if (oy0 + r >= P.ho) break;
#pragma unroll
        for (int j = 0; j < TX; ++j) {
            if (ox0 + j >= P.wo) break;
            h8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (_Float16)apply_act(acc[r][j][i], ACT < 0 ? P.act : ACT);
            const size_t m = ((size_t)n * P.ho + oy0 + r) * P.wo + ox0 + j;
            *reinterpret_cast<h8 *>(P.out + m * P.cs_out + P.coff_out + g * 8) = o;
        }
    }
}

struct PoolP {
    const _Float16 *in; int H, W, cs_in, coff_in;
    int k, stride, pad, ho, wo, c, m;
    _Float16 *out; int cs_out, coff_out;
};

__global__ __launch_bounds__(256) void maxpool_k(const PoolP P) {
    const int groups = P.c >> 3;
    const unsigned idx = dd_xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)P.m * (unsigned)groups) return;
    const int g = (int)(idx % (unsigned)groups);
    const int m = (int)(idx / (unsigned)groups);
    const int hw = P.ho * P.wo;
    const int n = m / hw, r = m - n * hw;
    const int oy = r / P.wo, ox = r - oy * P.wo;
    float best[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) best[i] = -65504.f;
    for (int kh = 0; kh < P.k; ++kh) {
        const int iy = oy * P.stride - P.pad + kh;
        if (iy < 0 || iy >= P.H) continue;
        for (int kw = 0; kw < P.k; ++kw) {
            const int ix = ox * P.stride - P.pad + kw;
            if (ix < 0 || ix >= P.W) continue;
            const h8 x = *reinterpret_cast<const h8 *>(P.in + ((size_t)(n * P.H + iy) * P.W + ix) * P.cs_in + P.coff_in + g * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) best[i] = fmaxf(best[i], (float)x[i]);
        }
    }
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)best[i];
    *reinterpret_cast<h8 *>(P.out + (size_t)m * P.cs_out + P.coff_out + g * 8) = o;
}

// A cascade of N stride-1 k x k max pools (YOLOv5's SPP: 5, 9 and 13 as three 5 x 5 pools of each other) in ONE launch: a block holds the
// H x W plane of one image and 8-channel group in LDS, pools it separably (row maxima, then column maxima: 2 k instead of k^2 taps), writes
// the result into channel slice i of the destination and pools that again.  Three maxpool_k launches re-read their input from HBM at
// 1.1 TB/s (47 us each at 128 frames); max of f16 values is exact either way: the same bits.
constexpr int PC_GROUPS = 4;       // 8-channel groups per block: a pixel's 64 bytes are one sector of the NHWC row (one group alone would fetch four times its bytes)
__global__ __launch_bounds__(256) void pool_cascade_k(const PoolP P, const int n_casc) {
    extern __shared__ __attribute__((aligned(16))) _Float16 pl[];
    const int gblocks = (P.c >> 3) / PC_GROUPS, hw = P.H * P.W, items = hw * PC_GROUPS;
    const int n = blockIdx.x / gblocks, g0 = (blockIdx.x - n * gblocks) * PC_GROUPS;
    h8 *a = reinterpret_cast<h8 *>(pl), *b = a + items;            // [pixel][group]
    for (int it = threadIdx.x; it < items; it += 256) {
        const int p = it / PC_GROUPS, gs = it - p * PC_GROUPS;
        a[it] = *reinterpret_cast<const h8 *>(P.in + ((size_t)n * hw + p) * P.cs_in + P.coff_in + (g0 + gs) * 8);
    }
    __syncthreads();
    const int rad = P.k / 2;
    const float inv_w = 1.f / (float)P.W;
    auto row_of = [&](int p) { return (int)(((float)p + 0.5f) * inv_w); };      // p / W (exact: p < 4096)
    for (int i = 0; i < n_casc; ++i) {
        for (int it = threadIdx.x; it < items; it += 256) {         // row maxima (v_pk_max_f16: four instructions per tap)
            const int p = it / PC_GROUPS, y = row_of(p), x = p - y * P.W;
            h8 m = a[it];
            for (int d = -rad; d <= rad; ++d) {
                if (d == 0 || x + d < 0 || x + d >= P.W) continue;
                m = __builtin_elementwise_max(m, a[it + d * PC_GROUPS]);
            }
            b[it] = m;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < items; it += 256) {         // column maxima of those
            const int p = it / PC_GROUPS, gs = it - p * PC_GROUPS, y = row_of(p);
            h8 m = b[it];
            for (int d = -rad; d <= rad; ++d) {
                if (d == 0 || y + d < 0 || y + d >= P.H) continue;
                m = __builtin_elementwise_max(m, b[it + d * P.W * PC_GROUPS]);
            }
            a[it] = m;
            *reinterpret_cast<h8 *>(P.out + ((size_t)n * hw + p) * P.cs_out + P.coff_out + i * P.c + (g0 + gs) * 8) = m;
        }
        __syncthreads();
    }
}

// nearest x2 upsample into a (possibly wider) destination channel slice
__global__ __launch_bounds__(256) void upsample2_k(const _Float16 *__restrict__ in, int H, int W, int cs_in, int coff_in,
                                                   int c, int m_out, _Float16 *__restrict__ out, int cs_out, int coff_out) {
    const int groups = c >> 3;
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)m_out * (unsigned)groups) return;
    const int g = (int)(idx % (unsigned)groups);
    const int m = (int)(idx / (unsigned)groups);
    const int wo = W * 2, hw = H * 2 * wo;
    const int n = m / hw, r = m - n * hw;
    const int oy = r / wo, ox = r - oy * wo;
    const h8 x = *reinterpret_cast<const h8 *>(in + ((size_t)(n * H + (oy >> 1)) * W + (ox >> 1)) * cs_in + coff_in + g * 8);
    *reinterpret_cast<h8 *>(out + (size_t)m * cs_out + coff_out + g * 8) = x;
}

// u8 [N][H][W][3] -> f16 [N][Ho][Wo][cs]: optional channel swap, (x - mean) * scale, optional
// space-to-depth 2 (YOLOv5 Focus: channel order (y0x0, y1x0, y0x1, y1x1) x rgb).
__global__ __launch_bounds__(256) void input_k(const uint8_t *__restrict__ src, int H, int W, int swap_rb, float mean,
                                               float scale, int s2d, int m_out, _Float16 *__restrict__ out, int cs_out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= m_out) return;
    const int ho = s2d ? H / 2 : H, wo = s2d ? W / 2 : W;
    const int hw = ho * wo;
    const int n = m / hw, r = m - n * hw;
    const int oy = r / wo, ox = r - oy * wo;
    _Float16 *o = out + (size_t)m * cs_out;
    if (!s2d) {
        const uint8_t *p = src + ((size_t)(n * H + oy) * W + ox) * 3;
        h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        v[0] = (_Float16)(((float)p[swap_rb ? 2 : 0] - mean) * scale);
        v[1] = (_Float16)(((float)p[1] - mean) * scale);
        v[2] = (_Float16)(((float)p[swap_rb ? 0 : 2] - mean) * scale);
        *reinterpret_cast<h8 *>(o) = v;
    } else {
        h8 v0 = {0, 0, 0, 0, 0, 0, 0, 0}, v1 = v0;
        float t[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int dy = q & 1, dx = q >> 1;
            const uint8_t *p = src + ((size_t)(n * H + 2 * oy + dy) * W + 2 * ox + dx) * 3;
            t[q * 3 + 0] = ((float)p[swap_rb ? 2 : 0] - mean) * scale;
            t[q * 3 + 1] = ((float)p[1] - mean) * scale;
            t[q * 3 + 2] = ((float)p[swap_rb ? 0 : 2] - mean) * scale;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v0[i] = (_Float16)t[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) v1[i] = (_Float16)t[8 + i];
        *reinterpret_cast<h8 *>(o) = v0;
        *reinterpret_cast<h8 *>(o + 8) = v1;
        const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 16; c < cs_out; c += 8) *reinterpret_cast<h8 *>(o + c) = z;      // a wider tensor: zero channels behind the 12 slices
    }
}

// x / sqrt(eps + sum x^2) over rows of `c` f32 (c <= 256); one wave per row.
// input_k's space-to-depth form with one thread per 16-byte chunk of the output instead of one per pixel: consecutive lanes write
// consecutive chunks (a thread per 64-byte pixel put the lanes of every store 64 bytes apart: 2.9 TB/s for the 32-channel Focus
// tensor).  Same arithmetic per value.  Chunk 0 = slices (y0x0, y1x0) rgb + (y0x1) rg, chunk 1 = (y0x1) b + (y1x1) rgb + zeros,
// further chunks zeros.
__global__ __launch_bounds__(256) void input_s2d_chunks_k(const uint8_t *__restrict__ src, int H, int W, int swap_rb, float mean,
                                                          float scale, int n_chunks, int cpp, _Float16 *__restrict__ out, int cs_out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_chunks) return;
    const int m = t / cpp, ch = t - m * cpp;
    const int ho = H / 2, wo = W / 2, hw = ho * wo;
    const int n = m / hw, r = m - n * hw;
    const int oy = r / wo, ox = r - oy * wo;
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ch < 2) {
        const uint8_t *p0 = src + ((size_t)(n * H + 2 * oy) * W + 2 * ox) * 3, *p1 = p0 + (size_t)W * 3;      // rows 2 oy, 2 oy + 1; pixels 2 ox, 2 ox + 1
        const int c0 = swap_rb ? 2 : 0, c2 = swap_rb ? 0 : 2;
        auto cv = [&](uint8_t b) { return (_Float16)(((float)b - mean) * scale); };
        if (ch == 0) {
            v[0] = cv(p0[c0]); v[1] = cv(p0[1]); v[2] = cv(p0[c2]);
            v[3] = cv(p1[c0]); v[4] = cv(p1[1]); v[5] = cv(p1[c2]);
            v[6] = cv(p0[3 + c0]); v[7] = cv(p0[3 + 1]);
        } else {
            v[0] = cv(p0[3 + c2]);
            v[1] = cv(p1[3 + c0]); v[2] = cv(p1[3 + 1]); v[3] = cv(p1[3 + c2]);
        }
    }
    *reinterpret_cast<h8 *>(out + (size_t)m * cs_out + ch * 8) = v;
}

__global__ __launch_bounds__(256) void l2norm_k(const float *__restrict__ in, int n_rows, int c, float eps, float *__restrict__ out) {
    const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    float v[4], ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        v[i] = j < c ? in[(size_t)row * c + j] : 0.f;
        ss += v[i] * v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = sqrtf(eps + ss);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        if (j < c) out[(size_t)row * c + j] = v[i] / nrm;
    }
}


}  // namespace



static void net_drop_graphs(dd_net *net) {
    for (auto &kv : net->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    net->graphs.clear();
}

namespace {

inline size_t dtype_size(int dt) { return dt == DT_F16 ? 2 : (dt == DT_F32 ? 4 : 1); }

// K split of a dense conv layer (1 = none): a function of the layer shape and the engine's max_batch only.
int conv_splitk(int ho, int wo, int cout_pad, int ksteps, int max_batch) {
    int splitk = 1;
    const int gy64 = dd_ceil_div(cout_pad, 64);
    const int blocks_per_image = dd_ceil_div(ho * wo, 64) * gy64;
    const long long blocks_full = (long long)dd_ceil_div(max_batch * ho * wo, 64) * gy64;
    if (blocks_per_image <= 8 && ksteps >= 8 && blocks_full < 256) {        // fewer blocks than CUs at full batch
        splitk = std::min(16, ksteps / 4);
        const int per = dd_ceil_div(ksteps, splitk);
        splitk = dd_ceil_div(ksteps, per);                    // no empty split
    }
    return splitk;
}

// Few blocks and a long K axis: split K over blockIdx.z so the chip is busy and each block's serial
// chain of (load -> barrier -> MFMA) steps is short; partial sums go through an f32 slab.
// GLDS selects the direct-to-LDS kernel (needs padded Cin % 64 == 0 and BK == 64).
template <int WM, int WN, int MI, int NI, int BK, bool GLDS>
int launch_conv(hipStream_t s, ConvP &P, DevBuf &slab, int max_batch, int device, bool *slab_moved = nullptr) {
    constexpr int BM = WM * MI * 16, BN = WN * NI * 16;
    const int gx = dd_ceil_div(P.m, BM), gy = dd_ceil_div(P.cout_pad, BN);
    const int ksteps = P.kpad / BK;
    // The split depends on the layer shape and on the engine's max_batch only, never on the batch of the
    // call: every output element is then summed in the same order whatever else shares the launch
    // (streams stay independent).  An engine sized for many images has enough blocks without splitting,
    // and the f32 partial-sum slabs (splitk x the layer output, written and re-read) would dominate.
    // Counted in 64 x 64 tiles whatever tile this launch uses: the tile is picked from the batch of the call,
    // the split must not be.
    const int splitk = conv_splitk(P.ho, P.wo, P.cout_pad, ksteps, max_batch);
    P.splitk = splitk;
    P.slab = nullptr;
    if (splitk > 1) {
        // Sized for the engine's max_batch, never for the batch of the call: the slab pointer is a kernel argument, and a
        // forward captured into a hipGraph (dd_net_use_graph) must not find it freed and reallocated because a later call
        // with more images grew it.  After the first forward of an engine the slab never moves again.
        void *before = slab.p;
        int rc = slab.reserve((size_t)splitk * max_batch * P.ho * P.wo * P.cout_pad * sizeof(float));
        if (rc != DD_OK) return rc;
        if (before && slab.p != before && slab_moved) *slab_moved = true;     // a later layer needs more than an earlier one reserved
        P.slab = slab.as<float>();
    }
    constexpr size_t stage_bytes = (size_t)2 * (BM + BN) * (GLDS ? 64 : BK + 8) * sizeof(_Float16);
    constexpr size_t out_bytes = BM * BN > 128 * 128 ? 0 : (size_t)BM * (BN + 4) * sizeof(float);   // big tiles: direct epilogue only
    constexpr size_t lds_bytes = stage_bytes > out_bytes ? stage_bytes : out_bytes;
    if (lds_bytes > 65536) {                                  // > 64 KiB of LDS needs the opt-in attribute, per device
        static DevOnce once;
        const int rc = once.run(device, [&]() -> int {
            if constexpr (GLDS) {
                DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_glds_k<WM, WN, MI, NI, 0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_glds_k<WM, WN, MI, NI, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_glds_k<WM, WN, MI, NI, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            } else {
                DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_k<WM, WN, MI, NI, BK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            }
            return DD_OK;
        });
        if (rc != DD_OK) return rc;
    }
    if constexpr (GLDS) {
        const bool pw = P.kh == 1 && P.kw == 1 && P.stride == 1 && P.pad_t == 0 && P.pad_l == 0 && P.cin % 64 == 0 &&
                        P.ho == P.H && P.wo == P.W;
        static const bool no_fm2 = getenv("DD_NO_FM2") != nullptr;                      // A/B switch
        const bool tapu = P.cin % 64 == 0 && P.kh <= 3 && P.kw <= 3 && P.kpad == P.kh * P.kw * P.cin && !no_fm2;
        if (pw) hipLaunchKernelGGL((conv_glds_k<WM, WN, MI, NI, 1>), dim3(gx, gy, splitk), dim3(WM * WN * 64), lds_bytes, s, P);
        else if (tapu) hipLaunchKernelGGL((conv_glds_k<WM, WN, MI, NI, 2>), dim3(gx, gy, splitk), dim3(WM * WN * 64), lds_bytes, s, P);
        else hipLaunchKernelGGL((conv_glds_k<WM, WN, MI, NI, 0>), dim3(gx, gy, splitk), dim3(WM * WN * 64), lds_bytes, s, P);
    } else
        hipLaunchKernelGGL((conv_mfma_k<WM, WN, MI, NI, BK>), dim3(gx, gy, splitk), dim3(WM * WN * 64), lds_bytes, s, P);
    DD_LAUNCH_CHECK();
    if (splitk > 1) {
        const long long total = (long long)P.m * (P.cout_pad >> 2);
        DD_REQUIRE(total < (1LL << 31), DD_E_CAPACITY, "split-K finish of %lld items exceeds 32-bit indexing", total);
        hipLaunchKernelGGL(conv_splitk_finish_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P);
        DD_LAUNCH_CHECK();
    }
    return DD_OK;
}

// SSD head layer with the decode in its epilogue: 128 pixels x one anchor (96 channels) per block, weights in the
// per-anchor layout nets.py packs for it (P.w / P.bias already point at that copy).
int launch_ssd_head_dec(hipStream_t s, ConvP &P) {
    constexpr int WM = 2, WN = 2, MI = 4, NI = 3, BM = WM * MI * 16, BN = WN * NI * 16;
    static_assert(BN == 96, "one anchor per tile");
    DD_REQUIRE(P.kh == 1 && P.kw == 1 && P.stride == 1 && P.pad_t == 0 && P.pad_l == 0 && P.cin % 64 == 0 && P.kpad == P.cin &&
               P.ho == P.H && P.wo == P.W && 4 + P.p[0] <= BN, DD_E_ARG, "ssd head decode: layer shape");
    P.splitk = 1; P.slab = nullptr;
    constexpr size_t lds_bytes = (size_t)2 * (BM + BN) * 64 * sizeof(_Float16);
    static_assert(lds_bytes >= (size_t)BM * (BN + 4) * sizeof(float) && lds_bytes <= 65536, "staged tile fits the operand buffers");
    const int gx = dd_ceil_div(P.m, BM), gy = P.p[5];           // channel tiles = anchors per pixel
    hipLaunchKernelGGL((conv_glds_k<WM, WN, MI, NI, 1, 1>), dim3(gx, gy, 1), dim3(WM * WN * 64), lds_bytes, s, P);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int launch_yolo_head_dec(hipStream_t s, ConvP &P) {
    constexpr int WM = 2, WN = 2, MI = 4, NI = 3, BM = WM * MI * 16, BN = WN * NI * 16;
    DD_REQUIRE(P.kh == 1 && P.kw == 1 && P.stride == 1 && P.pad_t == 0 && P.pad_l == 0 && P.cin % 64 == 0 && P.kpad == P.cin &&
               P.ho == P.H && P.wo == P.W && P.p[0] >= 6 && P.p[0] <= BN && P.cout % P.p[0] == 0, DD_E_ARG, "yolo head decode: layer shape");
    P.splitk = 1; P.slab = nullptr;
    constexpr size_t lds_bytes = (size_t)2 * (BM + BN) * 64 * sizeof(_Float16);
    const int gx = dd_ceil_div(P.m, BM), gy = P.cout / P.p[0];    // channel tiles = anchors per pixel
    hipLaunchKernelGGL((conv_glds_k<WM, WN, MI, NI, 1, 2>), dim3(gx, gy, 1), dim3(WM * WN * 64), lds_bytes, s, P);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// Tile of th x tw output pixels (<= 512, i.e. 32 pixel fragments) whose input patch fits `max_patch`
// pixels (0 = no limit); tiles are balanced so the last row / column of tiles is not a sliver.
void spatial_tile(int ho, int wo, int stride, int max_patch, ConvP &P) {
    int tiles_x = dd_ceil_div(wo, 32);
    int tw = dd_ceil_div(wo, tiles_x);
    int th = std::max(1, std::min(ho, 512 / tw));
    while (max_patch && th > 1 && ((th - 1) * stride + 3) * ((tw - 1) * stride + 3) > max_patch) --th;
    int tiles_y = dd_ceil_div(ho, th);
    th = dd_ceil_div(ho, tiles_y);
    P.th = th; P.tw = tw; P.tiles_x = tiles_x; P.tiles_y = tiles_y;
}

// Can the pooled 3x3 layer P run as conv3x3_pool_rows_k on nimg images?  (P.p[0..1] = pooled height, width.)
bool pool_rows_fusable(const ConvP &P, int nimg) {
    static const bool tiled = getenv("DD_POOL_TILED") && atoi(getenv("DD_POOL_TILED")) != 0;
    return !tiled && nimg >= 160 && P.W == 32 && P.wo == 32 && P.H == P.ho && P.H % 2 == 0 && P.p[0] == P.H / 2 - 1 && P.p[1] == 15 &&
           P.pad_t == 1 && P.pad_l == 1 && P.cin == 32 && P.cout == 32 && P.act == ACT_ELU;
}

int launch_conv3x3_rw(hipStream_t s, ConvP &P, int nimg, bool pool, int device) {
    if (pool) {                                                  // 8 pooled rows per tile = 17 conv rows, full width
        P.tw = P.wo; P.th = 17; P.tiles_x = 1; P.tiles_y = dd_ceil_div(P.p[0], 8);
        DD_REQUIRE(P.wo == 32 && P.act == ACT_ELU && !P.res && !P.out2, DD_E_ARG, "conv3x3_rw: fused pooling needs a 32-wide ELU layer");
        // one wave per image, rows streamed (conv3x3_pool_rows_k); DD_POOL_TILED=1 keeps the tiled kernel (same bits)
        // (measured at 3840 / 1024 / 256 / 64 images: rows 136 / 41 / 17 / 9.3 us, tiled 251 / 75 / 23 / 9.0 us; below
        // ~160 images the tiled kernel's 16 waves per image win, profiles/r02_pool_rows_sweep.txt)
        const bool stem = P.src8 != nullptr;                     // the caller folded the network's first layer in (pool_rows_fusable)
        DD_REQUIRE(!stem || pool_rows_fusable(P, nimg), DD_E_ARG, "conv3x3_rw: first layer folded into a launch that cannot take it");
        if (pool_rows_fusable(P, nimg)) {
            const size_t ring_bytes = (size_t)4 * pr_wave_halves(stem) * sizeof(_Float16);
            static DevOnce once;
            const int rc = once.run(device, [&]() -> int {
                DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_pool_rows_k<ACT_ELU, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * pr_wave_halves(false) * sizeof(_Float16))));
                DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_pool_rows_k<ACT_ELU, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * pr_wave_halves(true) * sizeof(_Float16))));
                return DD_OK;
            });
            if (rc != DD_OK) return rc;
            static const int force_split = getenv("DD_POOL_SPLIT") ? atoi(getenv("DD_POOL_SPLIT")) : 0;
            static const int dbg = getenv("DD_PR_DBG") ? atoi(getenv("DD_PR_DBG")) : 0;
            const int split = force_split > 0 ? force_split : nimg >= 1536 ? 1 : nimg >= 768 ? 2 : 4;
            const int n_units = nimg * split;
            const int grid = std::min(dd_ceil_div(n_units, 4), 2 * 256);
            if (stem) hipLaunchKernelGGL((conv3x3_pool_rows_k<ACT_ELU, true>), dim3((unsigned)grid), dim3(256), ring_bytes, s, P, n_units, split | (dbg << 8));
            else hipLaunchKernelGGL((conv3x3_pool_rows_k<ACT_ELU, false>), dim3((unsigned)grid), dim3(256), ring_bytes, s, P, n_units, split | (dbg << 8));
            DD_LAUNCH_CHECK();
            return DD_OK;
        }
    } else {
        spatial_tile(P.ho, P.wo, 1, RW_MAX_PATCH, P);
    }
    const int npix = (P.th + 2) * (P.tw + 2);
    DD_REQUIRE(npix <= RW_MAX_PATCH, DD_E_ARG, "conv3x3_rw: patch of %d pixels", npix);
    DD_REQUIRE(pool || !P.src8 || (P.tw == 32 && P.act == ACT_SILU && !P.res && !P.out2), DD_E_ARG, "conv3x3_rw: an input op was folded into a launch that cannot take it");
    const size_t lds_bytes = (size_t)4 * ((npix * 16 + 255) & ~255) + (pool ? (size_t)P.th * P.tw * 32 * sizeof(_Float16) : 0);
    const int total = nimg * P.tiles_x * P.tiles_y;
    const int grid = std::min(total, 2 * 256);                   // persistent: 2 blocks per CU (register-bound), multiple of 8
    if (pool) {
        static DevOnce once;
        const int rc = once.run(device, [&]() -> int {
            DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_rw_k<2, 32, ACT_ELU, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            return DD_OK;
        });
        if (rc != DD_OK) return rc;
    }
    if (pool) {
        hipLaunchKernelGGL((conv3x3_rw_k<2, 32, ACT_ELU, true>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, total);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
#define DD_RW(TW_, ACT_, EF_) hipLaunchKernelGGL((conv3x3_rw_k<2, TW_, ACT_, false, EF_>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, total)
    const int ef = (!P.res && !P.out2) ? 0 : (P.res && P.out2) ? 1 : -1;
    if (P.tw == 32 && P.act == ACT_ELU && ef == 0) DD_RW(32, ACT_ELU, 0);
    else if (P.tw == 32 && P.act == ACT_SILU && P.src8 && ef == 0)
        hipLaunchKernelGGL((conv3x3_rw_k<2, 32, ACT_SILU, false, 0, true>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, total);
    else if (P.tw == 32 && P.act == ACT_SILU) DD_RW(32, ACT_SILU, -1);
    else if (P.tw == 15 && P.act == ACT_ELU && ef == 0) DD_RW(15, ACT_ELU, 0);
    else if (P.tw == 15 && P.act == ACT_NONE && ef == 1) DD_RW(15, ACT_NONE, 1);
    else DD_RW(0, -1, -1);
#undef DD_RW
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// Residual unit (layer A then layer B reading only A's output) as one launch of res_unit_rows_k?
bool res_unit_fusable(const ConvP &A, const ConvP &B, int nimg) {
    static const bool off = getenv("DD_RES_UNIT_UNFUSED") && atoi(getenv("DD_RES_UNIT_UNFUSED")) != 0;
    // one wave per image: both layers of a unit take 56-60 us whatever the batch up to 2048 images; the two tiled launches
    // take 34 / 52 / 58 / 77 us at 160 / 320 / 480 / 640 images (profiles/r02_res_unit_sweep.txt)
    static const int min_img = getenv("DD_RES_UNIT_MIN") ? atoi(getenv("DD_RES_UNIT_MIN")) : 512;
    auto plain = [](const ConvP &P) {
        return P.kh == 3 && P.kw == 3 && P.stride == 1 && P.pad_t == 1 && P.pad_l == 1 && P.cin == 32 && P.cout == 32 && P.cout_pad == 32 &&
               P.epi == EPI_F16 && P.W == 15 && P.wo == 15 && P.H == P.ho && P.splitk <= 1;
    };
    return !off && nimg >= min_img && plain(A) && plain(B) && A.act == ACT_ELU && !A.res && !A.out2 && !A.coff_out &&
           B.act == ACT_NONE && B.res && B.out2 && B.in == static_cast<const _Float16 *>(A.out) && !B.coff_in && B.cs_in == A.cs_out && B.H == A.H;
}

int launch_res_unit(hipStream_t s, const ConvP &A, const ConvP &B, int nimg, int device) {
    const bool raw_sep = !(B.res == A.in && B.cs_res == A.cs_in && B.coff_res == A.coff_in);
    const size_t lds_bytes = (size_t)4 * ru_wave_halves(raw_sep) * sizeof(_Float16);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&res_unit_rows_k<ACT_ELU, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * ru_wave_halves(true) * sizeof(_Float16))));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int grid = std::min(dd_ceil_div(nimg, 4), 2 * 256);
    if (raw_sep) hipLaunchKernelGGL((res_unit_rows_k<ACT_ELU, true>), dim3((unsigned)grid), dim3(256), lds_bytes, s, A, B, nimg);
    else hipLaunchKernelGGL((res_unit_rows_k<ACT_ELU, false>), dim3((unsigned)grid), dim3(256), lds_bytes, s, A, B, nimg);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// Two residual units in a row (A1, B1 then A2, B2; each pair res_unit_fusable) as one launch of res_pair_rows_k?  The caller
// knows from the program (op flag 2 on B1) that B1's two outputs are read by the second unit only.
bool res_pair_fusable(const ConvP &A1, const ConvP &B1, const ConvP &A2, const ConvP &B2, int nimg) {
    static const bool off = getenv("DD_RES_PAIR_OFF") && atoi(getenv("DD_RES_PAIR_OFF")) != 0;
    static const int min_img = getenv("DD_RES_PAIR_MIN") ? atoi(getenv("DD_RES_PAIR_MIN")) : 1024;    // four image pairs per CU
    return !off && nimg >= min_img && A1.H == 31 && A1.W == 15 &&
           B1.res == A1.in && B1.cs_res == A1.cs_in && B1.coff_res == A1.coff_in &&                  // first unit: skip tensor = its input
           A2.in == static_cast<const _Float16 *>(B1.out2) && A2.cs_in == B1.cs_out2 && !A2.coff_in && !B1.coff_out2 &&
           B2.res == static_cast<const _Float16 *>(B1.out) && B2.cs_res == B1.cs_out && B2.coff_res == B1.coff_out &&
           A2.H == A1.H && A2.W == A1.W && A1.act == ACT_ELU && A2.act == ACT_ELU;
}

int launch_res_pair(hipStream_t s, const ConvP &A1, const ConvP &B1, const ConvP &A2, const ConvP &B2, int nimg, int device) {
    constexpr size_t lds_bytes = (size_t)4 * rp_pair_halves() * sizeof(_Float16);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&res_pair_rows_k<ACT_ELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int grid = std::min(dd_ceil_div(nimg, 4), 256);          // one block (four wave pairs, 128 KiB of rings) per CU
    hipLaunchKernelGGL((res_pair_rows_k<ACT_ELU>), dim3((unsigned)grid), dim3(512), lds_bytes, s, A1, B1, A2, B2, nimg);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

bool s2_rows_eligible(const ConvP &P, int nimg, int max_batch) {
    static const bool off = getenv("DD_S2_ROWS_OFF") && atoi(getenv("DD_S2_ROWS_OFF")) != 0;
    static const int min_img = getenv("DD_S2_ROWS_MIN") ? atoi(getenv("DD_S2_ROWS_MIN")) : 512;
    return !off && nimg >= min_img && P.kh == 3 && P.kw == 3 && P.stride == 2 && P.pad_t == 1 && P.pad_l == 1 && P.cin == 32 && P.cout == 64 &&
           P.cout_pad == 64 && P.kpad >= 288 && P.epi == EPI_F16 && P.W == 15 && P.H == 31 && P.ho == 16 && P.wo == 8 && P.splitk <= 1 &&
           P.act == ACT_ELU && !P.res && !P.out2 && (long long)dd_ceil_div(max_batch * P.ho * P.wo, 64) >= 256;
}

int launch_conv3x3_s2_rows(hipStream_t s, const ConvP &P, int nimg, int device) {
    constexpr size_t lds_bytes = (size_t)4 * S2_SLOTS * S2_PITCH * sizeof(_Float16);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_s2_rows_k<ACT_ELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int grid = std::min(dd_ceil_div(nimg, 4), 2 * 256);
    hipLaunchKernelGGL((conv3x3_s2_rows_k<ACT_ELU>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, nimg);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// YOLOv5's 3x3 64 -> 64 layers (80 x 80 maps, SiLU, with or without the bottleneck's shortcut) as 8-column strips of conv3x3_c64_rows_k
bool c64_strips_eligible(const ConvP &P, int nimg, int max_batch) {
    static const bool off = getenv("DD_C64_STRIPS_OFF") && atoi(getenv("DD_C64_STRIPS_OFF")) != 0;
    static const int min_items = getenv("DD_C64_STRIPS_MIN") ? atoi(getenv("DD_C64_STRIPS_MIN")) : 1024;      // two items per wave slot of the chip
    return !off && P.kh == 3 && P.kw == 3 && P.stride == 1 && P.pad_t == 1 && P.pad_l == 1 && P.cin == 64 && P.cout == 64 &&
           P.cout_pad == 64 && P.kpad == 576 && P.epi == EPI_F16 && P.W > 8 && P.W % 8 == 0 && P.wo == P.W && P.H == P.ho && P.H % 2 == 0 &&
           P.splitk <= 1 && P.act == ACT_SILU && !P.out2 && (!P.res || (P.cs_res % 8 == 0 && P.coff_res % 8 == 0)) &&
           P.cs_in % 8 == 0 && P.coff_in % 8 == 0 && (long long)nimg * (P.W >> 3) >= min_items &&
           (long long)dd_ceil_div(max_batch * P.ho * P.wo, 64) >= 256;    // launch_conv would not split K for this engine: same summation order
}

int launch_conv3x3_c64_strips(hipStream_t s, const ConvP &P, int nimg, int device) {
    constexpr size_t lds_bytes = (size_t)4 * c64_wave_halves() * sizeof(_Float16);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_c64_rows_k<ACT_SILU, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_c64_rows_k<ACT_SILU, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const long long items = (long long)nimg * (P.W >> 3);
    DD_REQUIRE(items * P.H * 8 < (1LL << 31), DD_E_CAPACITY, "conv3x3_c64 strips: %lld items exceed 32-bit pixel indexing", items);
    const int grid = (int)std::min<long long>((items + 1) / 2, 2 * 256);      // a block: two items x two channel halves
    if (P.res) hipLaunchKernelGGL((conv3x3_c64_rows_k<ACT_SILU, 2, true>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, (int)items);
    else hipLaunchKernelGGL((conv3x3_c64_rows_k<ACT_SILU, 0, true>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, (int)items);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

bool c64_rows_eligible(const ConvP &P, int nimg, int max_batch) {
    static const bool off = getenv("DD_C64_ROWS_OFF") && atoi(getenv("DD_C64_ROWS_OFF")) != 0;
    static const int min_img = getenv("DD_C64_ROWS_MIN") ? atoi(getenv("DD_C64_ROWS_MIN")) : 512;
    return !off && nimg >= min_img && P.kh == 3 && P.kw == 3 && P.stride == 1 && P.pad_t == 1 && P.pad_l == 1 && P.cin == 64 && P.cout == 64 &&
           P.cout_pad == 64 && P.kpad == 576 && P.epi == EPI_F16 && P.W == 8 && P.wo == 8 && P.H == P.ho && P.H % 2 == 0 && P.splitk <= 1 &&
           ((P.act == ACT_ELU && !P.res && !P.out2) || (P.act == ACT_NONE && P.res && P.out2 && P.cs_res % 8 == 0 && P.coff_res % 8 == 0)) &&
           (long long)dd_ceil_div(max_batch * P.ho * P.wo, 64) >= 256;    // launch_conv would not split K for this engine: same summation order
}

int launch_conv3x3_c64_rows(hipStream_t s, const ConvP &P, int nimg, int device) {
    constexpr size_t lds_bytes = (size_t)4 * c64_wave_halves() * sizeof(_Float16);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_c64_rows_k<ACT_ELU, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_c64_rows_k<ACT_NONE, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int grid = std::min(dd_ceil_div(nimg, 2), 2 * 256);      // a block: two images x two channel halves
    if (P.res) hipLaunchKernelGGL((conv3x3_c64_rows_k<ACT_NONE, 1>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, nimg);
    else hipLaunchKernelGGL((conv3x3_c64_rows_k<ACT_ELU, 0>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, nimg);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// First layer (S) + first MobileNet block (P, reading only S's output) as one launch of ssd_front_k?
bool ssd_front_fusable(const ConvP &S, const ConvP &P, int nimg) {
    static const bool off = getenv("DD_SSD_FRONT_UNFUSED") && atoi(getenv("DD_SSD_FRONT_UNFUSED")) != 0;
    // whole forward at 8 / 16 / 32 / 48 / 96 / 192 frames: fused 305 / 366 / 507 / 627 / 969 / 1662 us, two launches 293 / 376 / 515 / 652 / 1024 / 1740 us
    static const int min_img = getenv("DD_SSD_FRONT_MIN") ? atoi(getenv("DD_SSD_FRONT_MIN")) : 16;
    return !off && nimg >= min_img && S.stride == 2 && S.cout == 32 && S.cout_pad == 32 && S.act == ACT_RELU6 && (S.W * 3) % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(S.src8) & 3) == 0 && S.pad_t >= 0 && S.pad_l >= 0 && S.pad_t <= 1 && S.pad_l <= 1 &&
           P.cin == 32 && P.cout == 64 && P.cout_pad == 64 && P.kpad == 32 && P.stride == 1 && P.pad_t == 1 && P.pad_l == 1 &&
           P.act == ACT_RELU6 && P.dw_act == ACT_RELU6 && P.in == static_cast<const _Float16 *>(S.out) && !P.coff_in && P.cs_in == 32 &&
           P.H == S.ho && P.W == S.wo && P.ho == P.H && P.wo == P.W && !P.res && !P.out2;
}

int launch_ssd_front(hipStream_t s, const ConvP &S, const ConvP &P, int nimg, int device) {
    const size_t lds_bytes = (size_t)4 * sf_wave_halves() * sizeof(_Float16);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ssd_front_k<ACT_RELU6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int strips = dd_ceil_div(P.wo, SF_SW);
    static const int force_parts = getenv("DD_SSD_FRONT_PARTS") ? atoi(getenv("DD_SSD_FRONT_PARTS")) : 0;
    // row parts per strip: 2048 wave slots (two per SIMD) are filled in rounds; a part costs its rows + 2 (the two stem rows above it)
    int parts = 1;
    long long best = -1;
    for (int p = 1; p <= 8; ++p) {
        const long long rounds = ((long long)nimg * strips * p + 2047) / 2048, cost = rounds * (dd_ceil_div(P.ho, p) + 2);
        if (best < 0 || cost < best) { best = cost; parts = p; }
    }
    if (force_parts > 0) parts = force_parts;
    const int n_tasks = nimg * strips * parts;
    const int grid = std::min(dd_ceil_div(n_tasks, 4), 2 * 256);
    hipLaunchKernelGGL((ssd_front_k<ACT_RELU6>), dim3((unsigned)grid), dim3(256), lds_bytes, s, S, P, n_tasks, parts);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int launch_stem(hipStream_t s, ConvP &P, int nimg) {
    spatial_tile(P.ho, P.wo, P.stride, 0, P);
    // 4-byte loads for the patch fill need 4-byte aligned image rows (300 x 300 and 64 x 32 frames: yes)
    P.p[2] = (P.W * 3) % 4 == 0 && ((size_t)P.H * P.W * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(P.src8) & 3) == 0;
    const int pw3 = ((P.tw - 1) * P.stride + 3) * 3;
    const size_t lds_bytes = (size_t)((P.th - 1) * P.stride + 3) * (P.p[2] ? ((pw3 + 6) & ~3) : pw3) * sizeof(_Float16);
    const dim3 grid((unsigned)(nimg * P.tiles_x * P.tiles_y));
    if (P.stride == 1 && P.act == ACT_ELU) hipLaunchKernelGGL((stem_conv3_k<1, ACT_ELU>), grid, dim3(256), lds_bytes, s, P);
    else if (P.stride == 2 && P.act == ACT_RELU6) hipLaunchKernelGGL((stem_conv3_k<2, ACT_RELU6>), grid, dim3(256), lds_bytes, s, P);
    else if (P.stride == 1) hipLaunchKernelGGL((stem_conv3_k<1, -1>), grid, dim3(256), lds_bytes, s, P);
    else hipLaunchKernelGGL((stem_conv3_k<2, -1>), grid, dim3(256), lds_bytes, s, P);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

bool dwpw_rows_eligible(const ConvP &P, int nimg) {
    static const bool off = getenv("DD_DWPW_ROWS_OFF") && atoi(getenv("DD_DWPW_ROWS_OFF")) != 0;
    static const int min_img = getenv("DD_DWPW_ROWS_MIN") ? atoi(getenv("DD_DWPW_ROWS_MIN")) : 96;     // 64 frames: 745 vs 741 us for the whole forward; 384: 2767 vs 2859
    return !off && nimg >= min_img && P.cin == 128 && P.cout == 128 && P.cout_pad == 128 && P.kpad == 128 && P.stride == 1 && P.pad_t == 1 && P.pad_l == 1 &&
           P.act == ACT_RELU6 && P.dw_act == ACT_RELU6 && P.ho == P.H && P.wo == P.W && !P.res && !P.out2 && P.cs_in % 8 == 0 && P.coff_in % 8 == 0;
}

int launch_dwpw_rows(hipStream_t s, const ConvP &P, int nimg, int device) {
    const size_t lds_bytes = (size_t)(4 * dr_wave_halves() + 9 * 128) * sizeof(_Float16) + 2 * 128 * sizeof(float);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dwpw_rows_k<ACT_RELU6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int strips = dd_ceil_div(P.W, DR_SW);
    static const int force_parts = getenv("DD_DWPW_ROWS_PARTS") ? atoi(getenv("DD_DWPW_ROWS_PARTS")) : 0;
    int parts = 1;                                               // row parts per strip, as in launch_ssd_front
    long long best = -1;
    for (int p = 1; p <= 8; ++p) {
        const long long rounds = ((long long)nimg * strips * p + 2047) / 2048, cost = rounds * (dd_ceil_div(P.H, p) + 3);
        if (best < 0 || cost < best) { best = cost; parts = p; }
    }
    if (force_parts > 0) parts = force_parts;
    const int n_tasks = nimg * strips * parts;
    const int grid = std::min(dd_ceil_div(n_tasks, 4), 2 * 256);
    hipLaunchKernelGGL((dwpw_rows_k<ACT_RELU6>), dim3((unsigned)grid), dim3(256), lds_bytes, s, P, n_tasks, parts);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

template <int WM, int WN, int MI, int CIN, int STRIDE>
int launch_dwpw(hipStream_t s, ConvP &P, int device) {
    constexpr int BM = WM * MI * 16, BN = WN * 64;
    constexpr size_t lds_bytes = (size_t)(BM + BN) * CIN * sizeof(_Float16) + BM * sizeof(int);
    const int n_tiles = dd_ceil_div(P.total_quads, BM / 4), gy = dd_ceil_div(P.cout_pad, BN);
    const bool relu6 = P.act == ACT_RELU6 && P.dw_act == ACT_RELU6;
    if (lds_bytes > 65536) {
        static DevOnce once;
        const int rc = once.run(device, [&]() -> int {
            DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dwpw_k<WM, WN, MI, CIN, STRIDE, ACT_RELU6, ACT_RELU6>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dwpw_k<WM, WN, MI, CIN, STRIDE, -1, -1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            return DD_OK;
        });
        if (rc != DD_OK) return rc;
    }
    // persistent: exactly as many blocks as stay resident (registers and LDS; a surplus block would only start when a
    // resident one has walked all its tiles)
    static std::atomic<int> per_cu_cache[64];                       // zero-initialised; threads that race compute the same number
    int per_cu = per_cu_cache[device & 63].load(std::memory_order_relaxed);
    if (per_cu == 0) {
        int nb = 0;
        DD_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&dwpw_k<WM, WN, MI, CIN, STRIDE, ACT_RELU6, ACT_RELU6>),
                                                            256, lds_bytes));
        per_cu = std::max(1, std::min(8, nb));
        per_cu_cache[device & 63].store(per_cu, std::memory_order_relaxed);
    }
    const dim3 grid((unsigned)std::min(n_tiles, std::max(8, per_cu * 256 / gy / 8 * 8)), (unsigned)gy);
    if (relu6) hipLaunchKernelGGL((dwpw_k<WM, WN, MI, CIN, STRIDE, ACT_RELU6, ACT_RELU6>), grid, dim3(256), lds_bytes, s, P, n_tiles);
    else hipLaunchKernelGGL((dwpw_k<WM, WN, MI, CIN, STRIDE, -1, -1>), grid, dim3(256), lds_bytes, s, P, n_tiles);
    DD_LAUNCH_CHECK();
    return DD_OK;
}


// 1x1 / stride 1 / unpadded layer with >= 16 K pixels and K = 256 or 512: weight-stationary persistent kernel.
bool ws_eligible(const ConvP &P) {
    // From 160 images per launch (DD_WS=1: always, DD_WS=0: never).  Same-box A/B of the whole SSD forward, conv_glds_k -> conv_ws_k
    // for its eligible layers, bit-identical: 48 / 64 frames +0.6 / +0.4 %, 96 / 128 / 160 / 192 / 256 frames -2.9 / -1.5 / -5.7 /
    // -2.3 / -2.9 %; end to end at 256 frames per launch (4 worker groups) 68.8 -> 72.4 k frames/s.  At the 96-frame launches of an
    // earlier default the persistent blocks lost to the other groups' kernels they cannot share a CU with (-1.5 % end to end).
    static const int mode = getenv("DD_WS") ? atoi(getenv("DD_WS")) : -1;
    const int nimg = P.m / std::max(1, P.ho * P.wo);
    return (mode > 0 || (mode < 0 && nimg >= 160)) && P.kh == 1 && P.kw == 1 && P.stride == 1 && P.pad_t == 0 && P.pad_l == 0 && P.ho == P.H && P.wo == P.W &&
           (P.cin == 256 || P.cin == 512) && P.kpad == P.cin && P.epi == EPI_F16 && !P.res && !P.out2 && P.m >= 16384 &&
           P.cout_pad % 128 == 0 && (P.act == ACT_RELU6 || P.act == ACT_NONE || P.act == ACT_SILU);
}

template <int KS>
int launch_conv_ws(hipStream_t s, ConvP &P, int device) {
    constexpr int NW = 4, D = 8;                                 // 128-channel slices, 64 KiB ring: two blocks per CU (measured faster than
                                                                 // one 8-wave block with a 128 KiB ring: 43.9 vs 46.8 us, 19x19x512 -> 512 at 192 frames)
    const int n_slices = P.cout_pad / (32 * NW);
    P.splitk = 1;
    static const int dbg_mode = getenv("DD_WS_MODE") ? atoi(getenv("DD_WS_MODE")) : 0;
    P.p[5] = dbg_mode;
    constexpr size_t lds_bytes = (size_t)D * WS_BM * 64 * sizeof(_Float16);
    DD_REQUIRE(n_slices >= 1 && (256 * 4 / NW / 4) % n_slices == 0, DD_E_ARG, "conv_ws: %d channel slices", n_slices);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_k<KS, NW, D, ACT_RELU6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_k<KS, NW, D, ACT_SILU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_k<KS, NW, D, ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const dim3 grid(2 * 256 * 4 / NW);                           // persistent: 8 waves per CU
    // DD_WSR=1: register-staged fills (conv_wsr_k), =2: that with 64 channels per wave (weights spill into AGPRs, one block per CU).
    // Same bits; measured at 384 frames: 19x19x512 94.5 (LDS-DMA ring) / 93.8 / 88.6 us, 38x38x256 119 / 119 / 126 us -- neither the
    // fills' issue cost nor the LDS read bandwidth is what holds the loop at 0.37 MFMA-busy, so the ring stays the default.
    static const int wsr = getenv("DD_WSR") ? atoi(getenv("DD_WSR")) : 0;
    if (wsr == 2 && P.cout_pad % 256 == 0) {                     // 64 channels per wave, one block per CU: half the LDS reads per MFMA
        constexpr size_t lds_r = (size_t)4 * WS_BM * 64 * sizeof(_Float16);
        const int ns = P.cout_pad / 256;
        if (P.act == ACT_RELU6) hipLaunchKernelGGL((conv_wsr_k<KS, NW, ACT_RELU6, 2>), dim3(256), dim3(NW * 64), lds_r, s, P, ns);
        else if (P.act == ACT_SILU) hipLaunchKernelGGL((conv_wsr_k<KS, NW, ACT_SILU, 2>), dim3(256), dim3(NW * 64), lds_r, s, P, ns);
        else hipLaunchKernelGGL((conv_wsr_k<KS, NW, ACT_NONE, 2>), dim3(256), dim3(NW * 64), lds_r, s, P, ns);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    if (wsr) {
        constexpr size_t lds_r = (size_t)4 * WS_BM * 64 * sizeof(_Float16);
        const dim3 grid_r(2 * 256 * 4 / NW);
        if (P.act == ACT_RELU6) hipLaunchKernelGGL((conv_wsr_k<KS, NW, ACT_RELU6>), grid_r, dim3(NW * 64), lds_r, s, P, n_slices);
        else if (P.act == ACT_SILU) hipLaunchKernelGGL((conv_wsr_k<KS, NW, ACT_SILU>), grid_r, dim3(NW * 64), lds_r, s, P, n_slices);
        else hipLaunchKernelGGL((conv_wsr_k<KS, NW, ACT_NONE>), grid_r, dim3(NW * 64), lds_r, s, P, n_slices);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    static const int spb = getenv("DD_WS_SPB") ? atoi(getenv("DD_WS_SPB")) : 1;     // stages per barrier; 2 measured null (19x19x512 93.6 / 91.8 vs 94.5 / 92.0 us, same bits)
    if (spb == 2) {
        static DevOnce once2;
        const int rc2 = once2.run(device, [&]() -> int {
            DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_k<KS, NW, D, ACT_RELU6, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_k<KS, NW, D, ACT_SILU, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_k<KS, NW, D, ACT_NONE, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            return DD_OK;
        });
        if (rc2 != DD_OK) return rc2;
        if (P.act == ACT_RELU6) hipLaunchKernelGGL((conv_ws_k<KS, NW, D, ACT_RELU6, 2>), grid, dim3(NW * 64), lds_bytes, s, P, n_slices);
        else if (P.act == ACT_SILU) hipLaunchKernelGGL((conv_ws_k<KS, NW, D, ACT_SILU, 2>), grid, dim3(NW * 64), lds_bytes, s, P, n_slices);
        else hipLaunchKernelGGL((conv_ws_k<KS, NW, D, ACT_NONE, 2>), grid, dim3(NW * 64), lds_bytes, s, P, n_slices);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    if (P.act == ACT_RELU6) hipLaunchKernelGGL((conv_ws_k<KS, NW, D, ACT_RELU6>), grid, dim3(NW * 64), lds_bytes, s, P, n_slices);
    else if (P.act == ACT_SILU) hipLaunchKernelGGL((conv_ws_k<KS, NW, D, ACT_SILU>), grid, dim3(NW * 64), lds_bytes, s, P, n_slices);
    else hipLaunchKernelGGL((conv_ws_k<KS, NW, D, ACT_NONE>), grid, dim3(NW * 64), lds_bytes, s, P, n_slices);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// Pointwise layer P (conv_ws_k material) followed by the depthwise layer Q that alone reads its output: one launch of conv_ws_dw_k?
bool ws_dw_fusable(const ConvP &P, const DwP &Q, int nimg) {
    static const bool off = getenv("DD_WS_DW_OFF") && atoi(getenv("DD_WS_DW_OFF")) != 0;
    if (off || !ws_eligible(P) || P.act != ACT_RELU6 || Q.act != ACT_RELU6 || Q.stride != 1 || Q.pad_t != 1 || Q.pad_l != 1) return false;
    if (Q.in != static_cast<const _Float16 *>(P.out) || Q.coff_in || Q.cs_in != P.cs_out || P.coff_out || Q.H != P.ho || Q.W != P.wo ||
        Q.ho != Q.H || Q.wo != Q.W || Q.c != P.cout || P.cout != P.cout_pad || (Q.W != 19 && Q.W != 10) || (long long)nimg * Q.H * Q.W >= (1 << 23))
        return false;
    const int n_slices = P.cout_pad / 128, n_workers = (512 / 8 / n_slices) * 8;
    if (n_slices < 1 || (512 / 8) % n_slices) return false;
    const int per = dd_ceil_div(nimg, n_workers);                // whole frames per worker: only when that balances (256 frames on 128 workers: 2 each)
    return (double)nimg / ((double)per * n_workers) >= 0.85;
}

template <int KS, int WMAP>
int launch_conv_ws_dw(hipStream_t s, ConvP &P, const DwP &Q, int nimg, int device) {
    constexpr int NW = 4, D = 4;
    const int n_slices = P.cout_pad / (32 * NW);
    P.splitk = 1;
    constexpr size_t lds_bytes = ((size_t)D * WS_BM * 64 + NW * 4 * WSD_SLOTS * 8 + 8 + 9 * 128) * sizeof(_Float16);
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_ws_dw_k<KS, WMAP, ACT_RELU6, ACT_RELU6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    static const int dbg_mode = getenv("DD_WS_MODE") ? atoi(getenv("DD_WS_MODE")) : 0;
    P.p[5] = dbg_mode;
    hipLaunchKernelGGL((conv_ws_dw_k<KS, WMAP, ACT_RELU6, ACT_RELU6>), dim3(512), dim3(NW * 64), lds_bytes, s, P, Q, n_slices, nimg);
    DD_LAUNCH_CHECK();
    return DD_OK;
}


template <int STRIDE>
int launch_dwpw_big(hipStream_t s, ConvP &P, int device, int nimg) {
    const size_t in_bytes = (size_t)nimg * P.H * P.W * P.cs_in * sizeof(_Float16);
    DD_REQUIRE(in_bytes < (1ull << 31), DD_E_CAPACITY, "dwpw_big: input of %zu bytes exceeds the 2 GiB buffer-resource window", in_bytes);
    P.p[0] = (int)in_bytes;
    P.total_quads = nimg * P.ho * ((P.wo + 1) / 2);              // pairs
    const size_t lds_bytes = (size_t)(2 * DWB_BM + 2 * DWB_BN) * 64 * sizeof(_Float16) + (size_t)P.cin * (18 + 4) + DWB_BM * sizeof(int);
    const bool relu6 = P.act == ACT_RELU6 && P.dw_act == ACT_RELU6;
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dwpw_big_k<STRIDE, ACT_RELU6, ACT_RELU6>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&dwpw_big_k<STRIDE, -1, -1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const dim3 grid((unsigned)(dd_ceil_div(P.total_quads, DWB_BM / 2) * (P.cout_pad / DWB_BN)));
    if (relu6) hipLaunchKernelGGL((dwpw_big_k<STRIDE, ACT_RELU6, ACT_RELU6>), grid, dim3(512), lds_bytes, s, P);
    else hipLaunchKernelGGL((dwpw_big_k<STRIDE, -1, -1>), grid, dim3(512), lds_bytes, s, P);
    DD_LAUNCH_CHECK();
    return DD_OK;
}


// MARS conv4_x on the crop-resident weight-stationary kernels (csrc/mars_tail.hip).  They run at EVERY batch size (their K halves
// meet at the end: not conv_glds_k's summation order), so a crop's feature does not depend on the launch shape.  DD_MARS_WS=0: off.
bool mars_ws_on() {
    static const bool on = !(getenv("DD_MARS_WS") && atoi(getenv("DD_MARS_WS")) == 0);
    return on;
}
bool mars_ws_s1_eligible(const ConvP &P) {
    return mars_ws_on() && P.kh == 3 && P.kw == 3 && P.stride == 1 && P.pad_t == 1 && P.pad_l == 1 && P.H == 8 && P.W == 4 && P.ho == 8 && P.wo == 4 &&
           P.cin == 128 && P.cout == 128 && P.cout_pad == 128 && P.kpad >= 1152 && P.epi == EPI_F16 && P.cs_in % 8 == 0 && P.coff_in % 8 == 0 &&
           P.cs_out % 8 == 0 && P.coff_out % 8 == 0 &&
           ((P.act == ACT_ELU && !P.res && !P.out2) || (P.act == ACT_NONE && P.res && P.cs_res % 8 == 0 && P.coff_res % 8 == 0 && (!P.out2 || (P.cs_out2 % 8 == 0 && P.coff_out2 % 8 == 0))));
}
bool mars_ws_s2_eligible(const ConvP &P) {                        // the 3x3 stride-2 layer of a widening block; the caller checks the projection behind it
    return mars_ws_on() && P.kh == 3 && P.kw == 3 && P.stride == 2 && P.pad_t == 0 && P.pad_l == 0 && P.H == 16 && P.W == 8 && P.ho == 8 && P.wo == 4 &&
           P.cin == 64 && P.cout == 128 && P.cout_pad == 128 && P.kpad >= 576 && P.epi == EPI_F16 && P.act == ACT_ELU && !P.res && !P.out2 &&
           P.cs_in % 8 == 0 && P.coff_in % 8 == 0 && P.cs_out % 8 == 0 && P.coff_out % 8 == 0;
}
bool mars_proj_follows(const dd_net *net, int i, const ConvP &P) {   // op i + 1: 1x1 stride 2, 64 -> 128, no activation, same map as op i's input
    const int32_t *q = net->prog.data() + net->ops_off + (size_t)(i + 1) * OP_WORDS;
    if (q[0] != OP_CONV || q[5] != 1 || q[6] != 1 || q[7] != 2 || q[8] != 0 || q[9] != 0 || q[10] != 64 || q[11] != 128 || q[12] != 128 ||
        q[14] != ACT_NONE || q[15] != EPI_F16 || q[3] >= 0 || q[4] >= 0 || q[13] < 64 || q[1] < 0 || q[2] < 0) return false;
    const TensorDesc &qs = net->tensors[q[1]], &qd = net->tensors[q[2]];
    return qs.h == 16 && qs.w == 8 && qd.h == 8 && qd.w == 4 && qs.cs % 8 == 0 && qs.coff % 8 == 0 && qd.cs % 8 == 0 && qd.coff % 8 == 0 && qd.dtype == DT_F16;
}
// MARS conv3_x blocks as one launch each (csrc/mars_pair.hip), from DD_MARS_PAIR_MIN crops (default 256; the same bits as the layer-by-layer
// kernels, so a crop's feature does not depend on which ran).  Ops i .. i + n - 1 must be: [3x3 stride-2 32 -> 64 ELU on a 31x15 map; 1x1
// stride-2 projection of another 31x15 tensor;] or [3x3 64 -> 64 ELU on 16x8;] then 3x3 64 -> 64, no activation, residual = the projection /
// a 16x8x64 tensor, second output with its affine.  Returns the number of ops (3 or 2) and fills Q, or 0.
int mars_pair_match(const dd_net *net, int i, int nimg, MarsPairP &Q, bool &first) {
    void *const *bufs = net->bufs.data();
    static const bool off = getenv("DD_MARS_PAIR") && atoi(getenv("DD_MARS_PAIR")) == 0;
    static const int min_crops = getenv("DD_MARS_PAIR_MIN") ? atoi(getenv("DD_MARS_PAIR_MIN")) : 256;
    if (off || nimg < min_crops) return 0;
    auto op = [&](int k) { return net->prog.data() + net->ops_off + (size_t)k * OP_WORDS; };
    auto W = [&](const int32_t *o, int word) { return net->d_weights + (size_t)(uint32_t)o[word]; };
    auto conv3 = [&](const int32_t *o, int stride, int cin, int act) {
        return o[0] == OP_CONV && o[5] == 3 && o[6] == 3 && o[7] == stride && o[10] == cin && o[11] == 64 && o[12] == 64 && o[14] == act && o[15] == EPI_F16 &&
               o[8] == 1 && o[9] == 1 && o[13] >= 9 * cin && o[1] >= 0 && o[2] >= 0 && !o[29];
    };
    auto t16x8 = [&](int t, int c) { const TensorDesc &d = net->tensors[t]; return d.h == 16 && d.w == 8 && d.c == c && d.cs % 8 == 0 && d.coff % 8 == 0 && d.dtype == DT_F16; };
    auto t31x15 = [&](int t) { const TensorDesc &d = net->tensors[t]; return d.h == 31 && d.w == 15 && d.c == 32 && d.cs % 8 == 0 && d.coff % 8 == 0 && d.dtype == DT_F16; };
    memset(&Q, 0, sizeof(Q));
    const int32_t *a = op(i);
    int nb;                                                       // index of the block's second layer
    if (i + 2 < net->n_ops && a[30] == 3 && conv3(a, 2, 32, ACT_ELU) && a[3] < 0 && a[4] < 0 && t31x15(a[1]) && t16x8(a[2], 64)) {
        const int32_t *p = op(i + 1);
        if (!(p[30] == 4 && p[0] == OP_CONV && p[5] == 1 && p[6] == 1 && p[7] == 2 && p[8] == 0 && p[9] == 0 && p[10] == 32 && p[11] == 64 && p[12] == 64 &&
              p[14] == ACT_NONE && p[15] == EPI_F16 && p[3] < 0 && p[4] < 0 && p[13] >= 32 && p[1] >= 0 && p[2] >= 0 && t31x15(p[1]) && t16x8(p[2], 64))) return 0;
        first = true; nb = i + 2;
        const TensorDesc &ti = net->tensors[a[1]], &tr = net->tensors[p[1]];
        Q.in = reinterpret_cast<const _Float16 *>(bufs[ti.buf]); Q.cs_in = ti.cs; Q.coff_in = ti.coff;
        Q.in2 = reinterpret_cast<const _Float16 *>(bufs[tr.buf]); Q.cs_in2 = tr.cs; Q.coff_in2 = tr.coff;
        Q.wp = reinterpret_cast<const _Float16 *>(W(p, 16)); Q.kpad_p = p[13]; Q.bias_p = reinterpret_cast<const float *>(W(p, 17));
        const int32_t *b = op(nb);
        if (b[3] != p[2]) return 0;                               // the second layer adds the projection
    } else if (i + 1 < net->n_ops && a[30] == 4 && conv3(a, 1, 64, ACT_ELU) && a[3] < 0 && a[4] < 0 && t16x8(a[1], 64) && t16x8(a[2], 64)) {
        first = false; nb = i + 1;
        const TensorDesc &ti = net->tensors[a[1]];
        Q.in = reinterpret_cast<const _Float16 *>(bufs[ti.buf]); Q.cs_in = ti.cs; Q.coff_in = ti.coff;
        const int32_t *b = op(nb);
        if (b[3] < 0 || !t16x8(b[3], 64)) return 0;
        const TensorDesc &tr = net->tensors[b[3]];
        Q.res = reinterpret_cast<const _Float16 *>(bufs[tr.buf]); Q.cs_res = tr.cs; Q.coff_res = tr.coff;
    } else return 0;
    const int32_t *b = op(nb);
    if (!(conv3(b, 1, 64, ACT_NONE) && b[1] == a[2] && b[3] >= 0 && b[4] >= 0 && b[19] && t16x8(b[2], 64) && t16x8(b[4], 64))) return 0;
    Q.wa = reinterpret_cast<const _Float16 *>(W(a, 16)); Q.kpad_a = a[13]; Q.bias_a = reinterpret_cast<const float *>(W(a, 17));
    Q.wb = reinterpret_cast<const _Float16 *>(W(b, 16)); Q.kpad_b = b[13]; Q.bias_b = reinterpret_cast<const float *>(W(b, 17));
    const TensorDesc &to = net->tensors[b[2]], &t2 = net->tensors[b[4]];
    Q.out = reinterpret_cast<_Float16 *>(bufs[to.buf]); Q.cs_out = to.cs; Q.coff_out = to.coff;
    Q.out2 = reinterpret_cast<_Float16 *>(bufs[t2.buf]); Q.cs_out2 = t2.cs; Q.coff_out2 = t2.coff;
    Q.aff2 = reinterpret_cast<const float *>(W(b, 18)); Q.cout_pad = b[12];
    Q.zero = net->d_zero; Q.n_img = nimg;
    return nb - i + 1;
}

MarsWsP mars_ws_params(const ConvP &P, int nimg) {
    MarsWsP Q;
    memset(&Q, 0, sizeof(Q));
    Q.in = P.in; Q.cs_in = P.cs_in; Q.coff_in = P.coff_in;
    Q.w = P.w; Q.kpad = P.kpad; Q.bias = P.bias;
    Q.res = P.res; Q.cs_res = P.cs_res; Q.coff_res = P.coff_res;
    Q.out = static_cast<_Float16 *>(P.out); Q.cs_out = P.cs_out; Q.coff_out = P.coff_out;
    Q.out2 = P.out2; Q.cs_out2 = P.cs_out2; Q.coff_out2 = P.coff_out2; Q.aff2 = P.aff2; Q.cout_pad = P.cout_pad;
    Q.zero = P.zero; Q.n_img = nimg;
    return Q;
}

}  // namespace

extern "C" {

// Program layout (int32 words): [magic 'DDN1'][n_tensors][n_bufs][n_ops][in_h][in_w][out_tensor][rsvd]
// then n_tensors * 8 words (buf,h,w,c,cs,coff,dtype,0), n_bufs * 2 words (elements per image, dtype),
// n_ops * 48 words.  See deepdish_amd/nets.py (Program.serialize) for the field order.
static int net_create(dd_ctx *ctx, const int32_t *program_host, int n_words, const void *weights_host,
                      int64_t n_weight_bytes, int max_batch, dd_net **out, bool share);
int dd_net_destroy(dd_net *n);

int dd_net_create(dd_ctx *ctx, const int32_t *program_host, int n_words, const void *weights_host,
                  int64_t n_weight_bytes, int max_batch, dd_net **out) {
    return net_create(ctx, program_host, n_words, weights_host, n_weight_bytes, max_batch, out, false);
}

// The same engine with its activation buffers overlaid by lifetime: a buffer is live from the first op that writes it to the last op
// that reads it (the output tensor: to the end), and buffers whose lifetimes do not meet share device memory (first fit, in op
// order).  For the engines a pipeline drives -- nobody reads an intermediate tensor after the forward (dd_net_read of anything but
// the output is DD_E_STATE) -- and for f16 programs only: the uint8 programs' bordered tensors keep their zero-point borders
// precisely because nothing else is ever written there.
int dd_net_create_shared(dd_ctx *ctx, const int32_t *program_host, int n_words, const void *weights_host,
                         int64_t n_weight_bytes, int max_batch, dd_net **out) {
    return net_create(ctx, program_host, n_words, weights_host, n_weight_bytes, max_batch, out, true);
}

static int net_create(dd_ctx *ctx, const int32_t *program_host, int n_words, const void *weights_host,
                      int64_t n_weight_bytes, int max_batch, dd_net **out, bool share) {
    DD_REQUIRE(ctx && program_host && weights_host && out && max_batch > 0 && n_words >= 8, DD_E_ARG,
               "dd_net_create: bad argument");
    DD_REQUIRE(program_host[0] == 0x314E4444, DD_E_ARG, "dd_net_create: bad program magic");
    const int nt = program_host[1], nb = program_host[2], no = program_host[3];
    DD_REQUIRE(n_words == 8 + nt * TENSOR_WORDS + nb * 2 + no * OP_WORDS, DD_E_ARG, "dd_net_create: program size mismatch");
    dd_net *n = new dd_net();
    struct Guard { dd_net *n; ~Guard() { if (n) (void)dd_net_destroy(n); } } guard{n};      // every early return below frees what was built so far
    n->ctx = ctx;
    n->max_batch = max_batch;
    n->prog.assign(program_host, program_host + n_words);
    n->in_h = program_host[4];
    n->in_w = program_host[5];
    n->out_tensor = program_host[6];
    const int32_t *p = program_host + 8;
    for (int i = 0; i < nt; ++i, p += TENSOR_WORDS) n->tensors.push_back(TensorDesc{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]});   // p[7] = 1: uint8 tensor in the bordered 16-channel-plane layout (csrc/netsq.hip)
    DD_HIP(hipSetDevice(ctx->device));
    if (share) {
        // lifetimes from the op list (it follows the buffer table): [first writer, last reader]; -1 = never written (the input of a view-only chain)
        const int32_t *ops = p + (size_t)nb * 2;
        std::vector<int> first(nb, -1), last(nb, -1);
        for (int t = 0; t < nt; ++t) DD_REQUIRE(n->tensors[t].buf >= 0 && n->tensors[t].buf < nb, DD_E_ARG, "dd_net_create_shared: tensor %d names buffer %d of %d", t, n->tensors[t].buf, nb);
        for (int i = 0; i < no; ++i)
            DD_REQUIRE((ops + (size_t)i * OP_WORDS)[0] < 16, DD_E_ARG, "dd_net_create_shared: uint8 programs keep one buffer per tensor (their borders are set once)");
        // An op flagged "only the next op reads my output" (word 30) may run inside that op's launch -- a first layer folded into the
        // pooled layer behind it, a residual unit held back until its second layer, a pair of units until the fourth -- so every buffer
        // touched anywhere in such a chain of ops is live over the whole chain.
        std::vector<int> gs(no), ge(no);
        for (int i = 0; i < no;) {
            int j = i;
            while (j + 1 < no && (ops + (size_t)j * OP_WORDS)[30] != 0) ++j;
            for (int k = i; k <= j; ++k) { gs[k] = i; ge[k] = j; }
            i = j + 1;
        }
        auto touch2 = [&](int t, int i) {
            if (t < 0 || t >= nt) return;
            const int b = n->tensors[t].buf;
            first[b] = first[b] < 0 ? gs[i] : std::min(first[b], gs[i]);
            last[b] = std::max(last[b], ge[i]);
        };
        for (int i = 0; i < no; ++i) {
            const int32_t *o = ops + (size_t)i * OP_WORDS;
            touch2(o[1], i); touch2(o[2], i); touch2(o[3], i); touch2(o[4], i);
        }
        if (n->out_tensor >= 0) last[n->tensors[n->out_tensor].buf] = no;
        std::vector<size_t> bytes(nb), off(nb, 0);
        size_t arena = 0;
        std::vector<int> order(nb);
        for (int b = 0; b < nb; ++b) { order[b] = b; bytes[b] = ((size_t)p[2 * b] * max_batch * dtype_size(p[2 * b + 1] & 0xff) + 256 + 255) & ~(size_t)255; }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return first[a] < first[b]; });
        std::vector<int> placed;
        for (int b : order) {
            if (first[b] < 0) { first[b] = 0; last[b] = no; }      // untouched by any op: keep it apart for the whole forward
            size_t at = 0;
            for (bool moved = true; moved;) {                      // lowest offset that overlaps no placed buffer alive at the same time
                moved = false;
                for (int q : placed)
                    if (!(last[q] < first[b] || last[b] < first[q]) && at < off[q] + bytes[q] && off[q] < at + bytes[b]) { at = off[q] + bytes[q]; moved = true; }
            }
            off[b] = at;
            arena = std::max(arena, at + bytes[b]);
            placed.push_back(b);
        }
        DD_HIP(hipMalloc(&n->arena, arena));
        DD_HIP(hipMemset(n->arena, 0, arena));
        n->arena_bytes = (int64_t)arena;
        for (int i = 0; i < nb; ++i, p += 2) {
            n->buf_elems.push_back(p[0]);
            n->buf_dtype.push_back(p[1] & 0xff);
            n->bufs.push_back(static_cast<char *>(n->arena) + off[i]);
        }
    } else
    for (int i = 0; i < nb; ++i, p += 2) {
        n->buf_elems.push_back(p[0]);
        n->buf_dtype.push_back(p[1] & 0xff);
        void *d = nullptr;
        const size_t bytes = (size_t)p[0] * max_batch * dtype_size(p[1] & 0xff) + 256;
        DD_HIP(hipMalloc(&d, bytes));
        DD_HIP(hipMemset(d, (p[1] >> 8) & 0xff, bytes));       // uint8 tensors: the zero point, which their borders keep (nobody writes them)
        n->bufs.push_back(d);
    }
    n->n_ops = no;
    n->ops_off = (int)(p - program_host);
    {   // the split-K slab, once, for the largest dense layer at max_batch (launch_conv: same formula): its pointer is baked into
        // captured graphs, so it must never be reallocated by a forward
        size_t need = 0;
        for (int i = 0; i < no; ++i) {
            const int32_t *o = p + (size_t)i * OP_WORDS;
            if (o[0] != OP_CONV) continue;
            const int bk = o[28] == 32 ? 32 : 64, sk = conv_splitk(o[26], o[27], o[12], o[13] / bk, max_batch);
            if (sk > 1) need = std::max(need, (size_t)sk * max_batch * o[26] * o[27] * o[12] * sizeof(float));
        }
        if (need) { const int rc = n->slab.reserve(need); if (rc != DD_OK) return rc; }
    }
    n->weight_bytes = n_weight_bytes;
    DD_HIP(hipMalloc(&n->d_zero, 256));
    DD_HIP(hipMemset(n->d_zero, 0, 256));
    n->use_glds = getenv("DD_NO_GLDS") == nullptr;
    n->use_rw = getenv("DD_NO_RW") == nullptr;
    n->tile_mode = getenv("DD_TILE_MODE") ? atoi(getenv("DD_TILE_MODE")) : 0;
    DD_HIP(hipMalloc(&n->d_weights, (size_t)n_weight_bytes + 256));
    DD_HIP(hipMemcpy(n->d_weights, weights_host, (size_t)n_weight_bytes, hipMemcpyHostToDevice));
    guard.n = nullptr;
    *out = n;
    return DD_OK;
}

int dd_net_destroy(dd_net *n) {
    if (!n) return DD_OK;
    if (n->arena) (void)hipFree(n->arena);
    else for (void *b : n->bufs) (void)hipFree(b);
    for (hipEvent_t e : n->events) (void)hipEventDestroy(e);
    net_drop_graphs(n);
    n->slab.release();
    (void)hipFree(n->d_zero);
    (void)hipFree(n->d_weights);
    for (void *q : {(void *)n->d_anchors, (void *)n->dec_boxes, (void *)n->dec_score, (void *)n->dec_keys, (void *)n->dec_cls}) (void)hipFree(q);
    delete n;
    return DD_OK;
}

int dd_net_activation_bytes(dd_net *n, int64_t *out_host) {
    DD_REQUIRE(n && out_host, DD_E_ARG, "dd_net_activation_bytes: NULL argument");
    if (n->arena) { *out_host = n->arena_bytes; return DD_OK; }
    int64_t tot = 0;
    for (size_t i = 0; i < n->bufs.size(); ++i) tot += n->buf_elems[i] * n->max_batch * (int64_t)dtype_size(n->buf_dtype[i]) + 256;
    *out_host = tot;
    return DD_OK;
}

int dd_net_last_batch(dd_net *n, int *out_host) {
    DD_REQUIRE(n && out_host, DD_E_ARG, "dd_net_last_batch: NULL argument");
    *out_host = n->last_batch;
    return DD_OK;
}

int dd_net_input_size(dd_net *n, int *h_host, int *w_host) {
    DD_REQUIRE(n && h_host && w_host, DD_E_ARG, "dd_net_input_size: NULL argument");
    *h_host = n->in_h; *w_host = n->in_w;
    return DD_OK;
}

int dd_net_max_batch(dd_net *n, int *out_host) {
    DD_REQUIRE(n && out_host, DD_E_ARG, "dd_net_max_batch: NULL argument");
    *out_host = n->max_batch;
    return DD_OK;
}

int dd_net_output(dd_net *n, int tensor, void **dev_ptr, int *h, int *w, int *c, int *cs, int *dtype) {
    DD_REQUIRE(n, DD_E_ARG, "dd_net_output: NULL net");
    const int t = tensor < 0 ? n->out_tensor : tensor;
    DD_REQUIRE(t >= 0 && t < (int)n->tensors.size(), DD_E_ARG, "dd_net_output: tensor %d out of range", t);
    const TensorDesc &d = n->tensors[t];
    if (dev_ptr) *dev_ptr = static_cast<char *>(n->bufs[d.buf]) + (size_t)d.coff * dtype_size(d.dtype);
    if (h) *h = d.h;
    if (w) *w = d.w;
    if (c) *c = d.c;
    if (cs) *cs = d.cs;
    if (dtype) *dtype = d.dtype;
    return DD_OK;
}

// Per-op device timing (HIP events on the launch stream) for bench.py's roofline line.
int dd_net_profile(dd_net *n, int enable) {
    DD_REQUIRE(n, DD_E_ARG, "dd_net_profile: NULL net");
    DD_DEVICE(n->ctx);
    if (enable && n->events.empty()) {
        n->events.resize(n->n_ops + 2);                 // + an empty bracket to price the event record itself
        for (auto &e : n->events) DD_HIP(hipEventCreate(&e));
    }
    n->profile = enable != 0;
    return DD_OK;
}

int dd_net_profile_read(dd_net *n, float *ms_host, int cap, int *n_ops_host) {
    DD_REQUIRE(n && ms_host && n_ops_host, DD_E_ARG, "dd_net_profile_read: NULL argument");
    DD_DEVICE(n->ctx);
    DD_REQUIRE(n->profile && !n->events.empty(), DD_E_STATE, "dd_net_profile_read: profiling is off");
    DD_REQUIRE(cap >= n->n_ops, DD_E_ARG, "dd_net_profile_read: cap %d < %d ops", cap, n->n_ops);
    DD_HIP(hipEventSynchronize(n->events[n->n_ops + 1]));
    float empty = 0.f;                                        // two back-to-back records: what a bracket costs by itself
    DD_HIP(hipEventElapsedTime(&empty, n->events[n->n_ops], n->events[n->n_ops + 1]));
    for (int i = 0; i < n->n_ops; ++i) {
        DD_HIP(hipEventElapsedTime(&ms_host[i], n->events[i], n->events[i + 1]));
        ms_host[i] = ms_host[i] > empty ? ms_host[i] - empty : 0.f;
    }
    *n_ops_host = n->n_ops;
    return DD_OK;
}

int dd_net_op_launches(dd_net *n, int32_t *codes_host, int cap, int *n_ops_host) {
    DD_REQUIRE(n && codes_host && n_ops_host, DD_E_ARG, "dd_net_op_launches: NULL argument");
    DD_REQUIRE(cap >= n->n_ops, DD_E_ARG, "dd_net_op_launches: cap %d < %d ops", cap, n->n_ops);
    for (int i = 0; i < n->n_ops; ++i) codes_host[i] = i < (int)n->op_launch.size() ? n->op_launch[i] : OPK_DEFAULT;
    *n_ops_host = n->n_ops;
    return DD_OK;
}

int dd_net_read(dd_net *n, int tensor, int n_img, void *dst, int dst_on_device, void *stream) {
    DD_REQUIRE(n && dst && n_img >= 0 && n_img <= n->max_batch, DD_E_ARG, "dd_net_read: bad argument");
    DD_DEVICE(n->ctx);
    const int t = tensor < 0 ? n->out_tensor : tensor;
    DD_REQUIRE(t >= 0 && t < (int)n->tensors.size(), DD_E_ARG, "dd_net_read: tensor %d out of range", t);
    const TensorDesc &d = n->tensors[t];
    DD_REQUIRE(!n->arena || t == n->out_tensor, DD_E_STATE, "dd_net_read: tensor %d of an engine whose buffers share memory by lifetime "
               "(dd_net_create_shared): only the output tensor outlives the forward", t);
    DD_REQUIRE(d.coff == 0, DD_E_ARG, "dd_net_read: tensor %d is a channel slice", t);
    // A tensor whose producing op ran inside the NEXT op's launch in the last forward (conv1_1 in conv3x3_pool_rows_k<STEM>,
    // conv0 in ssd_front_k, the first layer of a residual unit, a pointwise layer in conv_ws_dw_k; batch dependent) was
    // never written: reading it would return stale or uninitialised data.
    for (int i = 0; i < n->n_ops && i < (int)n->op_launch.size(); ++i) {
        const int32_t *o = n->prog.data() + n->ops_off + (size_t)i * OP_WORDS;
        DD_REQUIRE(!(o[2] == t && n->op_launch[i] == OPK_SSD_HEAD_DEC), DD_E_STATE,
                   "dd_net_read: tensor %d (the SSD head matrix) was not written: the head layers decoded in their epilogue "
                   "(dd_net_ssd_decode); read dd_net_ssd_decoded instead", t);
        DD_REQUIRE(!(o[2] == t && n->op_launch[i] == OPK_YOLO_HEAD_DEC), DD_E_STATE,
                   "dd_net_read: tensor %d (the Detect matrix) was not written: the head layers reduced their rows in their epilogue "
                   "(dd_net_yolo_decode); read dd_net_yolo_decoded instead", t);
        DD_REQUIRE(!((o[2] == t || o[4] == t) && n->op_launch[i] == OPK_FOLDED), DD_E_STATE,
                   "dd_net_read: tensor %d was not written by the last forward (op %d ran inside the next op's launch at this batch "
                   "size and its output stayed on chip); run a smaller batch or a program compiled without the fusion flags", t, i);
    }
    hipStream_t s = dd_pick_stream(n->ctx, stream);
    const size_t bytes = d.pad ? (size_t)n_img * (d.h + 2) * (d.w + 2) * d.cs                     // bordered uint8 layout, as it lies
                               : (size_t)n_img * d.h * d.w * d.cs * dtype_size(d.dtype);
    if (!bytes) return DD_OK;
    DD_HIP(hipMemcpyAsync(dst, n->bufs[d.buf], bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    if (!dst_on_device) DD_HIP(hipStreamSynchronize(s));
    return DD_OK;
}

static int net_run_ops(dd_net *net, const uint8_t *input, int nimg, hipStream_t s);

int dd_net_use_graph(dd_net *net, int enable) {
    DD_REQUIRE(net, DD_E_ARG, "dd_net_use_graph: NULL net");
    DD_DEVICE(net->ctx);
    net->use_graph = enable != 0;
    if (!net->use_graph) net_drop_graphs(net);
    return DD_OK;
}

// SSD detector: run the first stage of TFLite_Detection_PostProcess (tools/ssd_mobilenet.py:103: best class, anchor decode,
// sigmoid, score threshold) inside the head layers' epilogues.  anchors_host f32 [n_anchors][4] (yc, xc, h, w).
int dd_net_ssd_decode(dd_net *net, const float *anchors_host, int n_anchors, float score_thr, int enable) {
    DD_REQUIRE(net, DD_E_ARG, "dd_net_ssd_decode: NULL net");
    DD_DEVICE(net->ctx);
    net_drop_graphs(net);
    if (!enable) { net->ssd_dec = false; return DD_OK; }
    DD_REQUIRE(anchors_host && n_anchors > 0, DD_E_ARG, "dd_net_ssd_decode: anchors missing");
    int heads = 0;
    for (int i = 0; i < net->n_ops; ++i) {
        const int32_t *o = net->prog.data() + net->ops_off + (size_t)i * OP_WORDS;
        if (o[0] == OP_CONV && o[15] == EPI_SSD_HEAD) {
            DD_REQUIRE(o[18] && o[21] == n_anchors && 4 + o[20] <= 96, DD_E_ARG,
                       "dd_net_ssd_decode: head op %d has no per-anchor weight copy, or %d anchors / %d classes do not fit", i, o[21], o[20]);
            ++heads;
        }
        if (o[0] == 20) {                                      // OP_QSSD_DECODE (uint8 programs decode in an op of their own)
            DD_REQUIRE(o[21] == n_anchors, DD_E_ARG, "dd_net_ssd_decode: the program decodes %d anchors, %d given", o[21], n_anchors);
            ++heads;
        }
    }
    DD_REQUIRE(heads > 0, DD_E_ARG, "dd_net_ssd_decode: the program has no SSD head");
    if (net->dec_anchors != n_anchors) {
        for (void *q : {(void *)net->d_anchors, (void *)net->dec_boxes, (void *)net->dec_score, (void *)net->dec_keys, (void *)net->dec_cls}) (void)hipFree(q);
        net->d_anchors = net->dec_boxes = net->dec_score = net->dec_keys = nullptr; net->dec_cls = nullptr; net->dec_anchors = 0;
        const size_t per = (size_t)net->max_batch * n_anchors;
        DD_HIP(hipMalloc(&net->d_anchors, (size_t)n_anchors * 4 * sizeof(float)));
        DD_HIP(hipMalloc(&net->dec_boxes, per * 4 * sizeof(float)));
        DD_HIP(hipMalloc(&net->dec_score, per * sizeof(float)));
        DD_HIP(hipMalloc(&net->dec_keys, per * sizeof(float)));
        DD_HIP(hipMalloc(&net->dec_cls, per * sizeof(int)));
        net->dec_anchors = n_anchors;
    }
    DD_HIP(hipMemcpy(net->d_anchors, anchors_host, (size_t)n_anchors * 4 * sizeof(float), hipMemcpyHostToDevice));
    net->dec_thr = score_thr;
    net->ssd_dec = true;
    return DD_OK;
}

// YOLOv5 detector: tools/yolov5.py:120-128 (cls *= obj, argmax, confidence) inside the Detect layers' epilogues: per row the
// decoded box (x, y, w, h normalised), the confidence and the class instead of the [rows][5 + C] matrix.
int dd_net_yolo_decode(dd_net *net, int enable) {
    DD_REQUIRE(net, DD_E_ARG, "dd_net_yolo_decode: NULL net");
    DD_DEVICE(net->ctx);
    net_drop_graphs(net);
    if (!enable) { net->yolo_dec = false; return DD_OK; }
    DD_REQUIRE(!net->ssd_dec, DD_E_STATE, "dd_net_yolo_decode: the SSD decode is on for this network");
    int heads = 0, rows = 0;
    for (int i = 0; i < net->n_ops; ++i) {
        const int32_t *o = net->prog.data() + net->ops_off + (size_t)i * OP_WORDS;
        if (o[0] == OP_CONV && o[15] == EPI_YOLO) {
            DD_REQUIRE(o[18] && !o[19] && o[20] >= 6 && o[20] <= 96, DD_E_ARG,
                       "dd_net_yolo_decode: head op %d has no per-anchor weight copy, or %d columns per row do not fit", i, o[20]);
            DD_REQUIRE(rows == 0 || rows == o[21], DD_E_ARG, "dd_net_yolo_decode: heads disagree on the rows per image");
            rows = o[21];
            ++heads;
        }
    }
    DD_REQUIRE(heads > 0 && rows > 0, DD_E_ARG, "dd_net_yolo_decode: the program has no YOLOv5 Detect head");
    if (net->dec_anchors != rows || !net->dec_boxes) {
        for (void *q : {(void *)net->d_anchors, (void *)net->dec_boxes, (void *)net->dec_score, (void *)net->dec_keys, (void *)net->dec_cls}) (void)hipFree(q);
        net->d_anchors = net->dec_boxes = net->dec_score = net->dec_keys = nullptr; net->dec_cls = nullptr; net->dec_anchors = 0;
        const size_t per = (size_t)net->max_batch * rows;
        DD_HIP(hipMalloc(&net->dec_boxes, per * 4 * sizeof(float)));
        DD_HIP(hipMalloc(&net->dec_score, per * sizeof(float)));
        DD_HIP(hipMalloc(&net->dec_cls, per * sizeof(int)));
        net->dec_anchors = rows;
    }
    net->yolo_dec = true;
    return DD_OK;
}

// Device pointers to what the last forward's Detect heads wrote: boxes f32 [n][rows][4] (x, y, w, h as the matrix's first four
// columns), conf f32 [n][rows], classes int32 [n][rows] -- what yolo_conf_k makes of the matrix.
int dd_net_yolo_decoded(dd_net *net, float **boxes, float **conf, int **classes, int *rows) {
    DD_REQUIRE(net && net->yolo_dec, DD_E_STATE, "dd_net_yolo_decoded: dd_net_yolo_decode is off");
    if (boxes) *boxes = net->dec_boxes;
    if (conf) *conf = net->dec_score;
    if (classes) *classes = net->dec_cls;
    if (rows) *rows = net->dec_anchors;
    return DD_OK;
}

int dd_net_yolo_decoded_read(dd_net *net, int n, float *boxes_host, float *conf_host, int *classes_host) {
    DD_REQUIRE(net && net->yolo_dec && n >= 0 && n <= net->max_batch, DD_E_STATE, "dd_net_yolo_decoded_read: decode is off or n out of range");
    DD_DEVICE(net->ctx);
    DD_HIP(hipStreamSynchronize(net->ctx->stream));
    const size_t per = (size_t)n * net->dec_anchors;
    if (boxes_host) DD_HIP(hipMemcpy(boxes_host, net->dec_boxes, per * 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (conf_host) DD_HIP(hipMemcpy(conf_host, net->dec_score, per * sizeof(float), hipMemcpyDeviceToHost));
    if (classes_host) DD_HIP(hipMemcpy(classes_host, net->dec_cls, per * sizeof(int), hipMemcpyDeviceToHost));
    return DD_OK;
}


// Device arrays the last forward decoded into: boxes f32 [n][n_anchors][4] (ymin, xmin, ymax, xmax), scores f32, classes
// int32 (class id - 1), keys f32 (score, or -1 below the threshold) [n][n_anchors] -- what ssd_decode_k makes of the head matrix.
int dd_net_ssd_decoded(dd_net *net, float **boxes, float **scores, int **classes, float **keys) {
    DD_REQUIRE(net && net->ssd_dec, DD_E_STATE, "dd_net_ssd_decoded: dd_net_ssd_decode is off");
    if (boxes) *boxes = net->dec_boxes;
    if (scores) *scores = net->dec_score;
    if (classes) *classes = net->dec_cls;
    if (keys) *keys = net->dec_keys;
    return DD_OK;
}

int dd_net_ssd_decoded_read(dd_net *net, int n, float *boxes_host, float *scores_host, int *classes_host, float *keys_host) {
    DD_REQUIRE(net && net->ssd_dec && n >= 0 && n <= net->max_batch, DD_E_STATE, "dd_net_ssd_decoded_read: decode is off or n out of range");
    DD_DEVICE(net->ctx);
    hipStream_t s = net->ctx->stream;
    DD_HIP(hipStreamSynchronize(s));
    const size_t per = (size_t)n * net->dec_anchors;
    if (!per) return DD_OK;
    if (boxes_host) DD_HIP(hipMemcpy(boxes_host, net->dec_boxes, per * 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (scores_host) DD_HIP(hipMemcpy(scores_host, net->dec_score, per * sizeof(float), hipMemcpyDeviceToHost));
    if (classes_host) DD_HIP(hipMemcpy(classes_host, net->dec_cls, per * sizeof(int), hipMemcpyDeviceToHost));
    if (keys_host) DD_HIP(hipMemcpy(keys_host, net->dec_keys, per * sizeof(float), hipMemcpyDeviceToHost));
    return DD_OK;
}

int dd_net_forward(dd_net *net, const uint8_t *input, int nimg, void *stream) {
    DD_REQUIRE(net && input && nimg >= 0, DD_E_ARG, "dd_net_forward: bad argument");
    DD_DEVICE(net->ctx);
    DD_REQUIRE(nimg <= net->max_batch, DD_E_CAPACITY, "dd_net_forward: batch %d > max_batch %d", nimg, net->max_batch);
    if (nimg == 0) return DD_OK;
    net->last_batch = nimg;
    hipStream_t s = dd_pick_stream(net->ctx, stream);
    auto eager = [&]() -> int {
        const int rc = net_run_ops(net, input, nimg, s);
        if (net->slab_moved) { net->slab_moved = false; net_drop_graphs(net); }     // (cannot happen since dd_net_create reserves it)
        return rc;
    };
    if (!net->use_graph || net->profile) return eager();
    if (net->graphs.size() > 256) net_drop_graphs(net);           // a caller that keeps moving its input buffer
    dd_net::GraphEntry &g = net->graphs[std::make_pair(static_cast<const void *>(input), nimg)];
    if (g.exec) { DD_HIP(hipGraphLaunch(g.exec, s)); return DD_OK; }
    if (g.calls++ == 0) return eager();                           // first sight of this key: eager (allocations, attributes)
    DD_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    const int rc = net_run_ops(net, input, nimg, s);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != DD_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    DD_HIP(e);
    g.graph = graph;
    DD_HIP(hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
    DD_HIP(hipGraphLaunch(g.exec, s));
    return DD_OK;
}

static int net_run_ops(dd_net *net, const uint8_t *input, int nimg, hipStream_t s) {
    auto base = [&](int t) -> char * {
        const TensorDesc &d = net->tensors[t];
        return static_cast<char *>(net->bufs[d.buf]);
    };
    net->op_launch.assign((size_t)net->n_ops, OPK_DEFAULT);
    ConvP stem_p;                                                 // a first layer waiting to be folded into the next op's launch
    bool stem_pending = false;
    bool input_pending = false;                                   // a space-to-depth input op waiting to be folded into the 3x3 layer behind it
    float input_mean = 0.f, input_scale = 1.f;
    int input_op = -1;
    ConvP unit_a;                                                 // first 3x3 layer of a residual unit, likewise
    bool unit_pending = false;
    ConvP pw_p;                                                   // pointwise layer whose only reader is the next (depthwise) op
    bool pw_pending = false;
    bool proj_done = false;                                       // the previous op's launch also ran this op (a 1x1 stride-2 projection)
    MarsPairP pair64;                                             // a conv3_x block whose ops run as one launch at its last op (mars_pair64_k)
    bool pair64_first = false;
    int pair_left = 0;
    ConvP pair_a, pair_b;                                         // a whole residual unit held back: it may run with the next unit (res_pair_rows_k)
    bool pair_pending = false;
    int pair_op = -1;
    auto flush_pair = [&]() -> int {                              // ... or on its own after all
        pair_pending = false;
        net->op_launch[pair_op] = OPK_RES_UNIT;
        return launch_res_unit(s, pair_a, pair_b, nimg, net->ctx->device);
    };
    auto run_ws = [&](ConvP &P) { return P.cin == 256 ? launch_conv_ws<4>(s, P, net->ctx->device) : launch_conv_ws<8>(s, P, net->ctx->device); };
    int first_op = 0;
    bool mid_on = false;
    {
        // uint8 SSD: the first three launches (first layer, MobileNet blocks 1 and 2) over chunks of frames that REUSE the image slots 0 .. chunk - 1
        // of the two tensors between them, so that 0.72 + 1.44 MB per frame of producer -> consumer traffic can stay in the 256 MB Infinity
        // Cache instead of going to HBM and back (DD_Q_FRONT_CHUNK=n frames; 0 = off).  Same kernels, same bits: block 2 writes its own
        // output where the whole-batch launch would.
        static const int chunk_env = getenv("DD_Q_FRONT_CHUNK") ? atoi(getenv("DD_Q_FRONT_CHUNK")) : 0;
        auto opw = [&](int k) { return net->prog.data() + net->ops_off + (size_t)k * OP_WORDS; };
        // uint8 SSD: the same three ops as ONE launch, the two tensors between them in LDS rings (csrc/netsq_front.hip).  DD_Q_FRONT=0: the three
        // launches; DD_Q_FRONT_MIN: frames per forward from which the row pipeline runs (a few frames spread better as three wide launches).  Read on
        // every forward: a test flips them between two forwards of one engine.
        const char *front_env = getenv("DD_Q_FRONT"), *front_min_env = getenv("DD_Q_FRONT_MIN");
        const int front_min = front_min_env ? atoi(front_min_env) : 24;
        {   // blocks 3 + 4 as one launch (csrc/netsq_mid.hip), taken in the op loop below: DD_Q_MID=0 the two launches; same batch threshold
            const char *mid_env = getenv("DD_Q_MID");
            mid_on = !(mid_env && atoi(mid_env) == 0) && nimg >= front_min;
        }
        if (!(front_env && atoi(front_env) == 0) && chunk_env <= 0 && nimg >= front_min && net->n_ops > 3 && opw(0)[0] == 16 && opw(1)[0] == 19 && opw(2)[0] == 19) {
            if (net->profile) for (int k = 0; k < 3; ++k) DD_HIP(hipEventRecord(net->events[k], s));
            int ran = 0;
            const int rc = netq_run_front(net, opw(0), opw(1), opw(2), input, nimg, s, &ran);
            if (rc != DD_OK) return rc;
            if (ran) { first_op = 3; net->op_launch[0] = OPK_FOLDED; net->op_launch[1] = OPK_FOLDED; net->op_launch[2] = OPK_Q_FRONT; }
        }
        if (first_op == 0 && chunk_env > 0 && nimg > chunk_env && net->n_ops > 3 && opw(0)[0] == 16 && opw(1)[0] == 19 && opw(2)[0] == 19 && opw(1)[1] == opw(0)[2] && opw(2)[1] == opw(1)[2]) {
            const TensorDesc &t2 = net->tensors[opw(2)[2]];
            const size_t img_out = (size_t)(t2.h + 2) * (t2.w + 2) * t2.cs;          // bordered uint8 layout: bytes per image
            void *&out_buf = net->bufs[t2.buf];
            void *const out_base = out_buf;
            if (net->profile) for (int k = 0; k < 3; ++k) DD_HIP(hipEventRecord(net->events[k], s));
            for (int c0 = 0; c0 < nimg; c0 += chunk_env) {
                const int nc = std::min(chunk_env, nimg - c0);
                out_buf = static_cast<char *>(out_base) + (size_t)c0 * img_out;
                for (int k = 0; k < 3; ++k) {
                    int handled = 0;
                    const int rc = netq_run_op(net, k, opw(k), input + (size_t)c0 * net->in_h * net->in_w * 3, nc, s, &handled);
                    if (rc != DD_OK || !handled) { out_buf = out_base; return rc != DD_OK ? rc : DD_E_ARG; }
                }
            }
            out_buf = out_base;
            first_op = 3;
        }
    }
    for (int i = first_op; i < net->n_ops; ++i) {
        if (net->profile) DD_HIP(hipEventRecord(net->events[i], s));
        const int32_t *o = net->prog.data() + net->ops_off + (size_t)i * OP_WORDS;
        const float *of = reinterpret_cast<const float *>(o);
        const int kind = o[0], src = o[1], dst = o[2], res = o[3], dst2 = o[4];
        const TensorDesc *ts = src >= 0 ? &net->tensors[src] : nullptr;
        const TensorDesc *td = dst >= 0 ? &net->tensors[dst] : nullptr;
        DD_REQUIRE(!input_pending || (kind == OP_CONV && i == input_op + 1), DD_E_STATE, "dd_net_forward: input op %d was folded into an op that did not take it", input_op);
        if (mid_on && kind == 19 && i + 1 < net->n_ops && net->prog[net->ops_off + (size_t)(i + 1) * OP_WORDS] == 19) {   // two uint8 block ops: blocks 3 + 4 as one launch?
            const int32_t *o4 = net->prog.data() + net->ops_off + (size_t)(i + 1) * OP_WORDS;
            if (net->profile) DD_HIP(hipEventRecord(net->events[i + 1], s));
            int ran = 0;
            const int rc = netq_run_mid(net, o, o4, nimg, s, &ran);
            if (rc != DD_OK) return rc;
            if (ran) { net->op_launch[i] = OPK_FOLDED; net->op_launch[i + 1] = OPK_Q_MID; ++i; continue; }
        }
        if (pair_pending && kind != OP_CONV) { const int rc = flush_pair(); if (rc != DD_OK) return rc; }
        if (pw_pending && kind != OP_DWCONV) {
            pw_pending = false; net->op_launch[i - 1] = OPK_CONV_WS;
            const int rc = run_ws(pw_p);
            if (rc != DD_OK) return rc;
        }
        if (unit_pending && kind != OP_CONV) {                        // (a program that sets the flag wrongly still computes the right thing)
            unit_pending = false; net->op_launch[i - 1] = OPK_DEFAULT;
            const int rc = launch_conv3x3_rw(s, unit_a, nimg, false, net->ctx->device);
            if (rc != DD_OK) return rc;
        }
        if (stem_pending && !(kind == OP_CONV && o[29]) && kind != OP_DWPW) {   // not followed by the layer it was meant for: run it on its own
            stem_pending = false; net->op_launch[i - 1] = OPK_DEFAULT;
            const int rc = launch_stem(s, stem_p, nimg);
            if (rc != DD_OK) return rc;
        }
        switch (kind) {
            case OP_INPUT: {
                const int s2d = o[5];
                const int m = nimg * td->h * td->w;
                if (s2d && o[30] && !o[6] && td->cs == 32 && !td->coff && i + 1 < net->n_ops && net->use_rw) {   // o[30]: only the next op reads it
                    const int32_t *q = net->prog.data() + net->ops_off + (size_t)(i + 1) * OP_WORDS;
                    static const bool off = getenv("DD_FOCUS_UNFUSED") && atoi(getenv("DD_FOCUS_UNFUSED")) != 0;
                    if (!off && q[0] == OP_CONV && q[1] == dst && q[5] == 3 && q[6] == 3 && q[7] == 1 && q[8] == 1 && q[9] == 1 && q[10] == 32 &&
                        q[12] == 32 && q[14] == ACT_SILU && q[15] == EPI_F16 && q[3] < 0 && q[4] < 0 && !q[29] && td->w % 32 == 0) {
                        input_pending = true; input_mean = of[32]; input_scale = of[33]; input_op = i;
                        net->op_launch[i] = OPK_FOLDED;
                        break;
                    }
                }
                if (s2d && td->cs >= 16 && td->cs % 8 == 0) {
                    const int cpp = td->cs / 8;
                    DD_REQUIRE((long long)m * cpp < (1ll << 31), DD_E_CAPACITY, "dd_net_forward: %d x %d input chunks", m, cpp);
                    hipLaunchKernelGGL(input_s2d_chunks_k, dim3(dd_ceil_div(m * cpp, 256)), dim3(256), 0, s, input, net->in_h, net->in_w, o[6],
                                       of[32], of[33], m * cpp, cpp, reinterpret_cast<_Float16 *>(base(dst)) + td->coff, td->cs);
                } else
                hipLaunchKernelGGL(input_k, dim3(dd_ceil_div(m, 256)), dim3(256), 0, s, input, net->in_h, net->in_w, o[6],
                                   of[32], of[33], s2d, m, reinterpret_cast<_Float16 *>(base(dst)) + td->coff, td->cs);
                DD_LAUNCH_CHECK();
                break;
            }
            case OP_CONV: {
                // (the early exits below skip the held-back launches' flush further down: a program that puts a held residual unit, stem or
                // pointwise layer directly in front of a folded op is refused instead of silently never writing it)
#define DD_NO_PENDING() DD_REQUIRE(!unit_pending && !pair_pending && !stem_pending && !pw_pending && !input_pending, DD_E_STATE, \
                                   "dd_net_forward: op %d is folded into a neighbouring launch while an earlier op's launch is still held back", i)
                if (proj_done) { DD_NO_PENDING(); proj_done = false; net->op_launch[i] = OPK_FOLDED_PREV; break; }
                if (pair_left > 0) {                               // inside a conv3_x block that runs as one launch: at its last op
                    DD_NO_PENDING();
                    if (--pair_left > 0) { net->op_launch[i] = OPK_FOLDED; break; }
                    const int rc = mars_pair64_launch(s, net->ctx->device, pair64, pair64_first);
                    if (rc != DD_OK) return rc;
                    net->op_launch[i] = OPK_MARS_PAIR;
                    break;
                }
                if (o[30] == 3 || o[30] == 4) {
                    const int n_ops = mars_pair_match(net, i, nimg, pair64, pair64_first);
                    if (n_ops) { DD_NO_PENDING(); pair_left = n_ops - 1; net->op_launch[i] = OPK_FOLDED; break; }
                }
#undef DD_NO_PENDING
                ConvP P;
                memset(&P, 0, sizeof(P));
                P.in = reinterpret_cast<const _Float16 *>(base(src)); P.H = ts->h; P.W = ts->w; P.cs_in = ts->cs;
                P.coff_in = ts->coff; P.cin = o[10];
                P.kh = o[5]; P.kw = o[6]; P.stride = o[7]; P.pad_t = o[8]; P.pad_l = o[9];
                P.cout = o[11]; P.cout_pad = o[12]; P.kpad = o[13]; P.act = o[14]; P.epi = o[15];
                P.w = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)o[16]);
                P.bias = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[17]);
                P.ho = o[26]; P.wo = o[27]; P.m = nimg * P.ho * P.wo;      // conv grid (dst may be a head matrix)
                P.out = base(dst); P.cs_out = td->cs; P.coff_out = td->coff;
                if (res >= 0) {
                    const TensorDesc &tr = net->tensors[res];
                    P.res = reinterpret_cast<const _Float16 *>(base(res)); P.cs_res = tr.cs; P.coff_res = tr.coff;
                }
                if (dst2 >= 0) {
                    const TensorDesc &t2 = net->tensors[dst2];
                    P.out2 = reinterpret_cast<_Float16 *>(base(dst2)); P.cs_out2 = t2.cs; P.coff_out2 = t2.coff;
                    P.aff2 = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[18]);
                }
                if (dst2 < 0 && o[19]) {                       // EPI_F32 with a post-activation affine (MARS fc1 + "ball")
                    P.aff2 = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[18]);
                    P.post_aff = 1;
                }
                for (int q = 0; q < 6; ++q) P.p[q] = o[20 + q];
                for (int q = 0; q < 8; ++q) P.f[q] = of[32 + q];
                int rc;
                DD_REQUIRE(!o[29] || (net->use_rw && P.kh == 3 && P.stride == 1 && P.cin == 32 && P.cout_pad == 32), DD_E_ARG,
                           "dd_net_forward: fused pooling is only built for the 3x3 32->32 kernel");
                if (unit_pending) {                            // the previous op was a residual unit's first layer
                    unit_pending = false;
                    if (res_unit_fusable(unit_a, P, nimg)) {
                        P.zero = net->d_zero;
                        if (pair_pending) {                        // ... and the unit before is waiting for this one
                            if (res_pair_fusable(pair_a, pair_b, unit_a, P, nimg)) {
                                pair_pending = false;
                                rc = launch_res_pair(s, pair_a, pair_b, unit_a, P, nimg, net->ctx->device);
                                if (rc != DD_OK) return rc;
                                net->op_launch[i] = OPK_RES_PAIR;  // ops i - 3 .. i - 1 stay "in the next launch"
                                break;
                            }
                            if ((rc = flush_pair()) != DD_OK) return rc;
                        }
                        // o[30] == 2: this unit's outputs are read by the next residual unit only -- hold it back, the two may run as one launch
                        if (o[30] == 2 && i + 2 < net->n_ops) {
                            pair_a = unit_a; pair_b = P; pair_pending = true; pair_op = i;
                            net->op_launch[i] = OPK_FOLDED;
                            break;
                        }
                        rc = launch_res_unit(s, unit_a, P, nimg, net->ctx->device);
                        if (rc != DD_OK) return rc;
                        net->op_launch[i] = OPK_RES_UNIT;
                        break;
                    }
                    if (pair_pending && (rc = flush_pair()) != DD_OK) return rc;
                    net->op_launch[i - 1] = OPK_DEFAULT;
                    rc = launch_conv3x3_rw(s, unit_a, nimg, false, net->ctx->device);
                    if (rc != DD_OK) return rc;
                }
                if (pair_pending) {                            // a held unit waits only for the first layer of the unit that reads it
                    const bool next_a = net->use_rw && P.kh == 3 && P.kw == 3 && P.stride == 1 && P.cin == 32 && P.cout_pad == 32 && P.epi == EPI_F16 &&
                                        P.pad_t == 1 && P.pad_l == 1 && o[30] && !o[29] && i + 1 < net->n_ops && nimg >= 160 && !P.res && !P.out2 &&
                                        P.in == static_cast<const _Float16 *>(pair_b.out2);
                    if (!next_a && (rc = flush_pair()) != DD_OK) return rc;
                }
                if (P.epi == EPI_YOLO && net->yolo_dec && o[18] && !o[19]) {      // o[18] / o[31]: the per-anchor copy of weights / bias
                    DD_REQUIRE(P.p[1] == net->dec_anchors, DD_E_ARG, "dd_net_forward: head of %d rows, decode set up for %d", P.p[1], net->dec_anchors);
                    P.w = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)o[18]);
                    P.bias = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[31]);
                    P.zero = net->d_zero;
                    P.dec_boxes = net->dec_boxes; P.dec_score = net->dec_score; P.dec_cls = net->dec_cls;
                    rc = launch_yolo_head_dec(s, P);
                    if (rc != DD_OK) return rc;
                    net->op_launch[i] = OPK_YOLO_HEAD_DEC;
                    break;
                }
                if (P.epi == EPI_SSD_HEAD && net->ssd_dec && o[18] && !o[19]) {   // o[18] / o[31]: the per-anchor copy of weights / bias
                    DD_REQUIRE(P.p[1] == net->dec_anchors, DD_E_ARG, "dd_net_forward: head of %d anchors, decode set up for %d", P.p[1], net->dec_anchors);
                    P.w = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)o[18]);
                    P.bias = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[31]);
                    P.zero = net->d_zero;
                    P.anchors = net->d_anchors; P.dec_boxes = net->dec_boxes; P.dec_score = net->dec_score; P.dec_keys = net->dec_keys;
                    P.dec_cls = net->dec_cls; P.dec_thr = net->dec_thr;
                    rc = launch_ssd_head_dec(s, P);
                    if (rc != DD_OK) return rc;
                    net->op_launch[i] = OPK_SSD_HEAD_DEC;
                    break;
                }
                const bool bk32 = o[28] == 32;                 // shallow K (<= 96): one or few 32-wide steps
                const bool glds = !bk32 && net->use_glds;          // K >= 97: direct-to-LDS fills, any Cin % 8 == 0
                P.zero = net->d_zero;
                if (net->use_rw && P.kh == 3 && P.kw == 3 && P.stride == 1 && P.cin == 32 && P.cout_pad == 32 &&
                    P.epi == EPI_F16 && P.pad_t == 1 && P.pad_l == 1) {
                    // whole filter in registers, input patch staged once (see conv3x3_rw_k)
                    if (input_pending) { input_pending = false; P.src8 = input; P.in_mean = input_mean; P.in_scale = input_scale; }
                    if (o[29]) { P.p[0] = td->h; P.p[1] = td->w; }      // fused 3x3/2 max pool: dst is the pooled tensor
                    if (stem_pending) {
                        stem_pending = false;
                        if (o[29] && pool_rows_fusable(P, nimg) && stem_p.out == static_cast<const void *>(P.in) && !P.coff_in) {
                            P.src8 = stem_p.src8; P.in_mean = stem_p.in_mean; P.in_scale = stem_p.in_scale;
                            P.dw_w = stem_p.w; P.dw_bias = stem_p.bias; P.dw_act = stem_p.act;
                        } else {
                            net->op_launch[i - 1] = OPK_DEFAULT;
                            rc = launch_stem(s, stem_p, nimg);
                            if (rc != DD_OK) return rc;
                        }
                    }
                    if (o[30] && !o[29] && i + 1 < net->n_ops && nimg >= 160 && !P.res && !P.out2) {
                        if (pair_pending && P.in != static_cast<const _Float16 *>(pair_b.out2) && (rc = flush_pair()) != DD_OK) return rc;
                        unit_a = P; unit_pending = true;       // o[30]: only the next op reads this layer's output
                        net->op_launch[i] = OPK_FOLDED;
                        break;
                    }
                    if (o[29] && pool_rows_fusable(P, nimg)) net->op_launch[i] = P.src8 ? OPK_POOL_ROWS_STEM : OPK_POOL_ROWS;
                    rc = launch_conv3x3_rw(s, P, nimg, o[29] != 0, net->ctx->device);
                } else if (mars_ws_s1_eligible(P)) {
                    const MarsWsP Q = mars_ws_params(P, nimg);
                    net->op_launch[i] = OPK_MARS_WS;
                    rc = mars_ws128_launch(s, net->ctx->device, Q, P.res ? MARS_WS_S1_RES : MARS_WS_S1, P.act, P.out2 != nullptr);
                } else if (o[30] == 3 && i + 1 < net->n_ops && mars_ws_s2_eligible(P) && mars_proj_follows(net, i, P)) {
                    // 3x3 stride-2 layer of a widening residual block + the 1x1 stride-2 projection of the block's raw input (the next op): one launch
                    const int32_t *q = net->prog.data() + net->ops_off + (size_t)(i + 1) * OP_WORDS;
                    const TensorDesc &qs = net->tensors[q[1]], &qd = net->tensors[q[2]];
                    MarsWsP Q = mars_ws_params(P, nimg);
                    Q.in2 = reinterpret_cast<const _Float16 *>(base(q[1])); Q.cs_in2 = qs.cs; Q.coff_in2 = qs.coff;
                    Q.w2 = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)q[16]); Q.kpad2 = q[13];
                    Q.bias2 = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)q[17]);
                    Q.out2 = reinterpret_cast<_Float16 *>(base(q[2])); Q.cs_out2 = qd.cs; Q.coff_out2 = qd.coff;
                    net->op_launch[i] = OPK_MARS_WS;
                    rc = mars_ws128_launch(s, net->ctx->device, Q, MARS_WS_S2_PROJ, P.act, false);
                    proj_done = true;
                } else if (s2_rows_eligible(P, nimg, net->max_batch)) {
                    net->op_launch[i] = OPK_S2_ROWS;
                    rc = launch_conv3x3_s2_rows(s, P, nimg, net->ctx->device);
                } else if (c64_rows_eligible(P, nimg, net->max_batch)) {
                    net->op_launch[i] = OPK_C64_ROWS;
                    rc = launch_conv3x3_c64_rows(s, P, nimg, net->ctx->device);
                } else if (c64_strips_eligible(P, nimg, net->max_batch)) {
                    net->op_launch[i] = OPK_C64_STRIPS;
                    rc = launch_conv3x3_c64_strips(s, P, nimg, net->ctx->device);
                } else if (ws_eligible(P) && o[30] && i + 1 < net->n_ops && P.act == ACT_RELU6) {
                    pw_p = P; pw_pending = true; net->op_launch[i] = OPK_FOLDED;     // o[30]: only the next (depthwise) op reads this output
                    break;
                } else if (ws_eligible(P)) {
                    net->op_launch[i] = OPK_CONV_WS;
                    rc = P.cin == 256 ? launch_conv_ws<4>(s, P, net->ctx->device) : launch_conv_ws<8>(s, P, net->ctx->device);
                } else if (P.cout_pad <= 32) {
                    // 32 output channels: 128 pixels per block (each wave 32 px x 32 ch) once there are enough pixels
                    rc = bk32 ? launch_conv<4, 1, 1, 2, 32, false>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved)
                       : (glds && P.m >= 16384) ? launch_conv<4, 1, 2, 2, 64, true>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved)
                       : glds ? launch_conv<4, 1, 1, 2, 64, true>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved)
                              : launch_conv<4, 1, 1, 2, 64, false>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved);
                } else if (glds && net->tile_mode != 1 && P.m >= 16384 && P.cout_pad >= 128) {
                    // plenty of pixels: 128 x 128 with 8 waves -- a third less L2->LDS traffic per FLOP than 64 x 128
                    // (24.3 us vs 26.6 us for 19x19x512 -> 512 at 64 frames; at 10x10 it halves the block count and loses).
                    // A K-capped run shows ~40 % of such a launch is per-round cost (prologue, epilogue, a partly filled
                    // last round of the 512 resident blocks), so 192 x 128 is taken when it saves rounds:
                    // 724 -> 484 tiles for 19x19x512 at 64 frames is one round instead of two (21.3 -> 18.0 us).
                    const int gy128 = dd_ceil_div(P.cout_pad, 128);
                    const int c128 = dd_ceil_div(dd_ceil_div(P.m, 128) * gy128, 512) * 128;
                    const int c192 = dd_ceil_div(dd_ceil_div(P.m, 192) * gy128, 512) * 192;
                    if (P.epi == EPI_F16 && c192 <= c128 && net->tile_mode != 2)
                        rc = launch_conv<4, 2, 3, 4, 64, true>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved);
                    else
                        rc = launch_conv<4, 2, 2, 4, 64, true>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved);
                } else if (glds && net->tile_mode != 1 && P.m >= 16384 && P.cout_pad == 64) {
                    rc = launch_conv<4, 2, 2, 2, 64, true>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved);      // 128 x 64, 8 waves
                } else if (glds && net->tile_mode != 1 && P.m >= 4096 && P.cout_pad >= 128) {
                    // 64 pixels x 128 channels: each staged pixel row feeds twice the MFMAs; measured 31 us vs 38 us
                    // for 19x19x512 -> 512 at 64 frames (128 x 64 gave nothing, 128 x 128 was 2.5x slower: 2 blocks/CU)
                    rc = launch_conv<2, 2, 2, 4, 64, true>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved);
                } else {
                    rc = bk32 ? launch_conv<2, 2, 2, 2, 32, false>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved)
                       : glds ? launch_conv<2, 2, 2, 2, 64, true>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved)
                              : launch_conv<2, 2, 2, 2, 64, false>(s, P, net->slab, net->max_batch, net->ctx->device, &net->slab_moved);
                }
                if (rc != DD_OK) return rc;
                break;
            }
            case OP_STEM: {
                ConvP P;
                memset(&P, 0, sizeof(P));
                P.src8 = input; P.H = net->in_h; P.W = net->in_w; P.in_mean = of[32]; P.in_scale = of[33];
                P.zero = net->d_zero;
                P.kh = P.kw = 3; P.stride = o[7]; P.pad_t = o[8]; P.pad_l = o[9];
                P.cout = o[11]; P.cout_pad = o[12]; P.act = o[14]; P.epi = EPI_F16; P.splitk = 1;
                DD_REQUIRE(P.cout_pad == 32 && (P.stride == 1 || P.stride == 2), DD_E_ARG, "dd_net_forward: stem needs 32 output channels, stride 1 or 2");
                P.w = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)o[16]);
                P.bias = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[17]);
                P.ho = td->h; P.wo = td->w; P.m = nimg * P.ho * P.wo;
                P.out = base(dst); P.cs_out = td->cs; P.coff_out = td->coff;
                // o[30]: the next op is the pooled 3x3 layer reading this tensor and nothing else does -- with enough images
                // both run as one launch (conv3x3_pool_rows_k<STEM>) and this tensor is never written
                static const bool unfused = getenv("DD_STEM_UNFUSED") && atoi(getenv("DD_STEM_UNFUSED")) != 0;
                if (o[30] && !unfused && i + 1 < net->n_ops && P.stride == 1 && P.pad_t == 1 && P.pad_l == 1 && P.cout == 32 && P.act == ACT_ELU &&
                    P.W == 32 && (reinterpret_cast<uintptr_t>(input) & 3) == 0) {
                    stem_p = P; stem_pending = true; net->op_launch[i] = OPK_FOLDED;
                    break;
                }
                if (o[30] && i + 1 < net->n_ops && P.stride == 2) {       // followed by the first MobileNet block (ssd_front_k)
                    stem_p = P; stem_p.zero = net->d_zero; stem_pending = true; net->op_launch[i] = OPK_FOLDED;
                    break;
                }
                int rc = launch_stem(s, P, nimg);
                if (rc != DD_OK) return rc;
                break;
            }
            case OP_DWPW: {
                ConvP P;
                memset(&P, 0, sizeof(P));
                P.in = reinterpret_cast<const _Float16 *>(base(src)); P.H = ts->h; P.W = ts->w; P.cs_in = ts->cs;
                P.coff_in = ts->coff; P.cin = o[10];
                P.kh = P.kw = 1; P.stride = o[7]; P.pad_t = o[8]; P.pad_l = o[9];
                P.cout = o[11]; P.cout_pad = o[12]; P.kpad = o[13]; P.act = o[14]; P.epi = EPI_F16; P.splitk = 1;
                P.w = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)o[16]);
                P.bias = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[17]);
                P.dw_w = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)o[20]);
                P.dw_bias = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[21]);
                P.dw_act = o[22]; P.zero = net->d_zero;
                P.ho = td->h; P.wo = td->w; P.m = nimg * P.ho * P.wo;
                P.total_quads = nimg * P.ho * ((P.wo + 3) / 4);
                P.out = base(dst); P.cs_out = td->cs; P.coff_out = td->coff;
                int rc;
                if (stem_pending) {
                    stem_pending = false;
                    if (ssd_front_fusable(stem_p, P, nimg)) {
                        rc = launch_ssd_front(s, stem_p, P, nimg, net->ctx->device);
                        if (rc != DD_OK) return rc;
                        net->op_launch[i] = OPK_SSD_FRONT;
                        break;
                    }
                    net->op_launch[i - 1] = OPK_DEFAULT;
                    rc = launch_stem(s, stem_p, nimg);
                    if (rc != DD_OK) return rc;
                }
                if (P.cin == 32 && P.cout_pad == 64 && P.stride == 1) rc = launch_dwpw<4, 1, 4, 32, 1>(s, P, net->ctx->device);
                else if (P.cin == 64 && P.cout_pad == 128 && P.stride == 2) rc = launch_dwpw<2, 2, 4, 64, 2>(s, P, net->ctx->device);
                else if (dwpw_rows_eligible(P, nimg)) { net->op_launch[i] = OPK_DWPW_ROWS; rc = launch_dwpw_rows(s, P, nimg, net->ctx->device); }
                else if (P.cin == 128 && P.cout_pad == 128 && P.stride == 1) rc = launch_dwpw<2, 2, 2, 128, 1>(s, P, net->ctx->device);
                else if (P.cin == 128 && P.cout_pad == 256 && P.stride == 2) rc = launch_dwpw<1, 4, 4, 128, 2>(s, P, net->ctx->device);
                else if (P.cin >= 256 && P.cin % 64 == 0 && P.cin <= 1024 && P.cout_pad % DWB_BN == 0 && P.kpad == P.cin)
                    rc = P.stride == 1 ? launch_dwpw_big<1>(s, P, net->ctx->device, nimg) : launch_dwpw_big<2>(s, P, net->ctx->device, nimg);
                else DD_REQUIRE(false, DD_E_ARG, "dd_net_forward: no fused dw+pw kernel for %d -> %d stride %d", P.cin, P.cout_pad, P.stride);
                if (rc != DD_OK) return rc;
                break;
            }
            case OP_DWCONV: {
                DwP P;
                P.in = reinterpret_cast<const _Float16 *>(base(src)); P.H = ts->h; P.W = ts->w; P.cs_in = ts->cs; P.coff_in = ts->coff;
                P.w = reinterpret_cast<const _Float16 *>(net->d_weights + (size_t)(uint32_t)o[16]);
                P.bias = reinterpret_cast<const float *>(net->d_weights + (size_t)(uint32_t)o[17]);
                P.stride = o[7]; P.pad_t = o[8]; P.pad_l = o[9]; P.ho = td->h; P.wo = td->w; P.c = o[12];
                P.m = nimg * td->h * td->w; P.act = o[14];
                P.out = reinterpret_cast<_Float16 *>(base(dst)); P.cs_out = td->cs; P.coff_out = td->coff;
                P.zero = net->d_zero;
                if (pw_pending) {
                    pw_pending = false;
                    if (ws_dw_fusable(pw_p, P, nimg)) {
                        const int rc = P.W == 10 ? (pw_p.cin == 256 ? launch_conv_ws_dw<4, 10>(s, pw_p, P, nimg, net->ctx->device) : launch_conv_ws_dw<8, 10>(s, pw_p, P, nimg, net->ctx->device))
                                                 : (pw_p.cin == 256 ? launch_conv_ws_dw<4, 19>(s, pw_p, P, nimg, net->ctx->device) : launch_conv_ws_dw<8, 19>(s, pw_p, P, nimg, net->ctx->device));
                        if (rc != DD_OK) return rc;
                        net->op_launch[i] = OPK_WS_DW;
                        break;
                    }
                    net->op_launch[i - 1] = OPK_CONV_WS;
                    const int rc = run_ws(pw_p);
                    if (rc != DD_OK) return rc;
                }
                static const bool two_rows = getenv("DD_DW_ONE_ROW") == nullptr;              // A/B switch
                const int ty = (P.stride == 1 && two_rows) ? 2 : 1;
                const long long total = (long long)nimg * ((P.ho + ty - 1) / ty) * ((P.wo + 3) / 4) * (P.c >> 3);
                DD_REQUIRE(total < (1LL << 31), DD_E_CAPACITY, "dd_net_forward: depthwise layer of %lld items exceeds 32-bit indexing", total);
                DD_REQUIRE(P.stride == 1 || P.stride == 2, DD_E_ARG, "dd_net_forward: depthwise stride %d", P.stride);
                const dim3 grid((unsigned)((total + 255) / 256));
#define DD_DW(S_, A_) hipLaunchKernelGGL((dwconv3_k<S_, A_>), grid, dim3(256), 0, s, P)
#define DD_DW2(A_) hipLaunchKernelGGL((dwconv3_k<1, A_, 2>), grid, dim3(256), 0, s, P)
                if (ty == 2) { if (P.act == ACT_RELU6) DD_DW2(ACT_RELU6); else if (P.act == ACT_SILU) DD_DW2(ACT_SILU); else DD_DW2(-1); }
                else if (P.act == ACT_RELU6) { if (P.stride == 1) DD_DW(1, ACT_RELU6); else DD_DW(2, ACT_RELU6); }
                else if (P.act == ACT_SILU) { if (P.stride == 1) DD_DW(1, ACT_SILU); else DD_DW(2, ACT_SILU); }
                else { if (P.stride == 1) DD_DW(1, -1); else DD_DW(2, -1); }
#undef DD_DW2
#undef DD_DW
                DD_LAUNCH_CHECK();
                break;
            }
            case OP_MAXPOOL: {
                PoolP P;
                P.in = reinterpret_cast<const _Float16 *>(base(src)); P.H = ts->h; P.W = ts->w; P.cs_in = ts->cs; P.coff_in = ts->coff;
                P.k = o[5]; P.stride = o[7]; P.pad = o[8]; P.ho = td->h; P.wo = td->w; P.c = o[12];
                P.m = nimg * td->h * td->w;
                P.out = reinterpret_cast<_Float16 *>(base(dst)); P.cs_out = td->cs; P.coff_out = td->coff;
                if (o[6] > 1) {                                  // a cascade of o[6] stride-1 pools into consecutive channel slices (nets.py pool_cascade)
                    const size_t lds = (size_t)2 * ts->h * ts->w * 16 * PC_GROUPS;
                    DD_REQUIRE(P.stride == 1 && P.pad == P.k / 2 && (P.k & 1) && td->h == ts->h && td->w == ts->w && lds <= 64 * 1024 && ts->h * ts->w <= 4096 && P.coff_out + o[6] * P.c <= td->cs &&      /* (the kernel's p / W by float reciprocal is exact below 4 096 pixels) */
                               (P.c >> 3) % PC_GROUPS == 0 && (long long)nimg * (P.c >> 3) < (1LL << 31), DD_E_ARG, "dd_net_forward: pool cascade %d: shape", i);
                    hipLaunchKernelGGL(pool_cascade_k, dim3((unsigned)(nimg * ((P.c >> 3) / PC_GROUPS))), dim3(256), lds, s, P, o[6]);
                    DD_LAUNCH_CHECK();
                    break;
                }
                const long long total = (long long)P.m * (P.c >> 3);
                DD_REQUIRE(total < (1LL << 31), DD_E_CAPACITY, "dd_net_forward: pooling layer of %lld items exceeds 32-bit indexing", total);
                hipLaunchKernelGGL(maxpool_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P);
                DD_LAUNCH_CHECK();
                break;
            }
            case OP_UPSAMPLE: {
                const int m = nimg * td->h * td->w;
                const long long total = (long long)m * (o[12] >> 3);
                DD_REQUIRE(total < (1LL << 31), DD_E_CAPACITY, "dd_net_forward: upsample layer of %lld items exceeds 32-bit indexing", total);
                hipLaunchKernelGGL(upsample2_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                                   reinterpret_cast<const _Float16 *>(base(src)), ts->h, ts->w, ts->cs, ts->coff, o[12], m,
                                   reinterpret_cast<_Float16 *>(base(dst)), td->cs, td->coff);
                DD_LAUNCH_CHECK();
                break;
            }
            case OP_L2NORM: {
                hipLaunchKernelGGL(l2norm_k, dim3(dd_ceil_div(nimg, 4)), dim3(256), 0, s,
                                   reinterpret_cast<const float *>(base(src)) + ts->coff, nimg, ts->c, of[32],
                                   reinterpret_cast<float *>(base(dst)) + td->coff);
                DD_LAUNCH_CHECK();
                break;
            }
            default: {
                int handled = 0;
                const int rc = netq_run_op(net, i, o, input, nimg, s, &handled);       // csrc/netsq.hip: the uint8 programs' ops
                if (rc != DD_OK) return rc;
                DD_REQUIRE(handled, DD_E_ARG, "dd_net_forward: unknown op kind %d at %d", kind, i);
            }
        }
    }
    if (pair_pending) { const int rc = flush_pair(); if (rc != DD_OK) return rc; }
    if (net->profile) {
        DD_HIP(hipEventRecord(net->events[net->n_ops], s));
        DD_HIP(hipEventRecord(net->events[net->n_ops + 1], s));
    }
    return DD_OK;
}

}  // extern "C"
