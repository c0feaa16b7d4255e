// Device-side arithmetic shared by the uint8 executors (csrc/netsq.hip, csrc/netsq_front.hip): vector types, the requantisation
// parameters of one layer and the requantise-and-pack forms.  See the header comment of csrc/netsq.hip for the arithmetic.
#pragma once
#include "common.h"

namespace {

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef short s2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) int cint;      // read-only data at a wave-uniform address: s_load

enum { OP_QCONV0 = 16, OP_QCONV = 17, OP_QDW = 18, OP_QDWPW = 19, OP_QSSD_DECODE = 20 };
enum { QEPI_Q16 = 0, QEPI_ROWS = 1 };

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

__device__ __forceinline__ int q_requant_relu(int x, int M, long long C, int sh32) {      // e >= 1: z = (x M + C) >> (32 + sh32)
    const long long t = (long long)x * M + C;
    return ((int)(t >> 32)) >> sh32;
}
__device__ __forceinline__ int q_clamp(int z, int lo, int hi) {                           // lo <= hi: one v_med3_i32 (min(max()) is two instructions)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(z), "s"(lo), "v"(hi));     // one scalar operand per vector instruction (constant bus)
    return r;
}

// Four requantised values -> four bytes under the layer's own clamp (4 v_med3 + 3 v_lshl_or).
__device__ __forceinline__ unsigned q_pack4(int z0, int z1, int z2, int z3, int lo, int hi) {
    return (unsigned)q_clamp(z0, lo, hi) | (unsigned)q_clamp(z1, lo, hi) << 8 | (unsigned)q_clamp(z2, lo, hi) << 16 | (unsigned)q_clamp(z3, lo, hi) << 24;
}

// Shift right, saturate to 0 .. 255 and pack, two values per instruction (gfx950: v_ashr_pk_u8_i32 D, S0, S1, S2 writes sat_u8(S0 >> S2),
// sat_u8(S1 >> S2) into the half of D that op_sel[3] names and keeps the other half): four values -> four bytes in TWO instructions, where
// v_cvt_pk_i16_i32 x2, v_pk_ashrrev_i16 x2, v_sat_pk_u8_i16 x2 were six.  The operands are v_mad_i64_i32 results (ordinary vector results: an
// MFMA destination read by inline assembly gets no wait states from hipcc, see csrc/image.hip).
__device__ __forceinline__ unsigned q_ashr_sat_pk4(int h0, int h1, int h2, int h3, int sh) {
    unsigned r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(h0), "v"(h1), "s"(sh));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(h2), "v"(h3), "s"(sh));
    return r;
}

// Four accumulators -> four requantised bytes (ReLU-type layers, e >= 1): z = (x M + C) >> (32 + sh), clamped.  SAT: 0 = the layer's own clamp
// (v_med3 + shifts), 1 / 2 = the clamp is the byte range (ReLU6 at scale 6/255 makes it): the high words of the four 64-bit sums go through
// q_ashr_sat_pk4 -- 4 + 2 instructions per four values (round 5: 14, round 6 with packed 16-bit shifts and the SDWA pack: 11); these kernels
// are bound by instruction issue.
template <int SAT>
__device__ __forceinline__ unsigned q_requant_pack4(int x0, int x1, int x2, int x3, int M, long long C0, long long C1, long long C2, long long C3, int sh32, int lo, int hi) {
    if constexpr (SAT >= 1) {
        const int h0 = (int)(((long long)x0 * M + C0) >> 32), h1 = (int)(((long long)x1 * M + C1) >> 32);
        const int h2 = (int)(((long long)x2 * M + C2) >> 32), h3 = (int)(((long long)x3 * M + C3) >> 32);
        return q_ashr_sat_pk4(h0, h1, h2, h3, sh32);
    } else {
        return q_pack4(q_requant_relu(x0, M, C0, sh32), q_requant_relu(x1, M, C1, sh32), q_requant_relu(x2, M, C2, sh32), q_requant_relu(x3, M, C3, sh32), lo, hi);
    }
}

// The same with the bytes in the form the tensors store (a - 128, the matrix instructions' signed operands) and no XOR behind the pack: the caller's
// 64-bit addend carries the - 128 (q_signed_c: C - 128 * 2^(32 + sh), exact -- a multiple of the shift's unit) and v_ashr_pk_i8_i32 saturates to
// -128 .. 127 = clamp(z, 0, 255) - 128.  4 + 2 instructions per four bytes.  SAT = 0 (the layer's own clamp): q_requant_pack4 and the XOR, C unchanged.
template <int SAT>
__device__ __forceinline__ long long q_signed_c(long long C, int sh32) {
    return SAT >= 1 ? C - (128ll << (32 + sh32)) : C;
}
template <int SAT>
__device__ __forceinline__ unsigned q_requant_pack4s(int x0, int x1, int x2, int x3, int M, long long C0, long long C1, long long C2, long long C3, int sh32, int lo, int hi) {
    if constexpr (SAT >= 1) {
        const int h0 = (int)(((long long)x0 * M + C0) >> 32), h1 = (int)(((long long)x1 * M + C1) >> 32);
        const int h2 = (int)(((long long)x2 * M + C2) >> 32), h3 = (int)(((long long)x3 * M + C3) >> 32);
        unsigned r;
        asm("v_ashr_pk_i8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(h0), "v"(h1), "s"(sh32));
        asm("v_ashr_pk_i8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(h2), "v"(h3), "s"(sh32));
        return r;
    } else {
        return 0x80808080u ^ q_requant_pack4<0>(x0, x1, x2, x3, M, C0, C1, C2, C3, sh32, lo, hi);
    }
}

// The same for a layer without activation (the SSD heads), clamp = the byte range, e >= 1: the literal two roundings
//     y = (x M + 2^30) >> 31,   z = ((y + 2^(e-1) + (y >> 31)) >> e) + zo  =  (y + [2^(e-1) + (zo << e)] + (y >> 31)) >> e
// with the bias in the 64-bit addend (C = cbias * M + 2^30 per channel), shift and clamp in v_ashr_pk_u8_i32: 4 instructions per value
// + 2 per four (q_requant's statement of the same arithmetic: 11 per value).
__device__ __forceinline__ unsigned q_requant_linear_pack4(int x0, int x1, int x2, int x3, int M, long long C0, long long C1, long long C2, long long C3, int e, int k1) {
    int z[4];
    const int x[4] = {x0, x1, x2, x3};
    const long long C[4] = {C0, C1, C2, C3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = (int)(((long long)x[i] * M + C[i]) >> 31);
        z[i] = y + k1 + (y >> 31);                           // the shift by e: in the pack
    }
    unsigned r;                                              // (e may differ between the fragments of a wave's heads: a vector operand)
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(z[0]), "v"(z[1]), "v"(e));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(z[2]), "v"(z[3]), "v"(e));
    return r;
}

__device__ __forceinline__ int sdot4(int a, int b, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sdot4(a, b, c, false);
#else
    return c;
#endif
}

inline QReq make_req(const int32_t *o) {
    QReq R;
    R.M = o[32]; R.e = o[33];
    R.zo = o[40]; R.lo = o[36]; R.hi = o[37]; R.linear = o[41];
    R.C = (1ll << 30) + (R.e > 0 ? (1ll << (30 + R.e)) : 0) + ((long long)R.zo << (31 + R.e));
    return R;
}

}  // namespace
