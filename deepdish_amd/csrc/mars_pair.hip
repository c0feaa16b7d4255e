// MARS re-ID encoder, conv3_x (tools/freeze_model.py:118-141 upstream: residual blocks conv3_1 / conv3_3 on 16x8x64 maps): two layers per
// launch, the tensor between them in LDS.
//
// Layer by layer these 64-channel layers are bound by HBM: 16 KB in, 16 KB out (+ 16 KB residual, + 16 KB second output) per crop and layer
// for 9.4 MFLOP -- conv3x3_c64_rows_k ran them at 3-4 TB/s of algorithmic traffic.  A launch here runs TWO layers: waves 0-3 of a workgroup
// (stage A) compute the first layer of crop c into an LDS tile while waves 4-7 (stage B) compute the second layer of crop c - 1 from the tile
// stage A left one step earlier -- a two-stage pipeline over the workgroup's crops, one barrier per crop.  Each wave keeps the whole filter of its
// 32 output channels in registers for the launch (144 VGPRs: conv3x3_c64_rows_k's arrangement) and takes four of the crop's eight 16-pixel
// fragments; waves w and w + 4 share a SIMD, so every matrix pipe is fed by one wave of each stage.
//   FIRST = true  (conv3_1): stage A = the block's 3x3 stride-2 layer 32 -> 64 on the pre-activation (31x15) + ELU, AND the 1x1 stride-2
//                 projection of the raw block input (kept as the f16 tile stage B adds); stage B = the 3x3 64 -> 64 layer + skip,
//                 second output ELU(scale v + shift) (the next block's pre-activation).
//   FIRST = false (conv3_3): stage A = 3x3 64 -> 64 + ELU; stage B = 3x3 64 -> 64 + the residual tensor (LDS-DMA'd) + second output.
// Summation order, rounding points (the tile between the layers is f16, as the tensor it replaces) and epilogue expressions are those of
// conv3x3_s2_rows_k / conv_mfma_k / conv3x3_c64_rows_k: the same bits as the layer-by-layer path (tests/test_gpu_properties.py).
//
// LDS tiles (16-byte chunks = 8 channels of a pixel): 64-channel maps as [18 rows: zero, 16 map rows, zero][8 planes][8 columns] -- a row is
// one contiguous 1 KiB LDS-DMA and a fragment read (two map rows x 8 pixels x 4 planes) is bank-conflict free; the 31x15x32 input of the
// stride-2 layer as [33 rows][even columns | odd columns][4 planes][8]; column padding by lanes that read a zero region (at the bank phase of
// the address they replace).
#include "mars_tail.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float elu(float v) {                  // as apply_act(ACT_ELU) in nets.hip
    float e = __builtin_amdgcn_exp2f(v * 1.44269504088896340736f);
    e = __builtin_amdgcn_fmed3f(e, 0.f, 1.f);
    return __builtin_amdgcn_fmed3f(v, e - 1.f, 3.0e38f);
}

__device__ __forceinline__ void lds_fill16(const _Float16 *g, char *lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(g, lds_wave_base, 16, 0, 0);
#endif
}

__device__ __forceinline__ int frag_row(int a, int fr) { return (fr >> 2) * 8 + (a & 1) * 4 + (fr & 3); }

constexpr int PZ_B = 3072;                                       // zero region: 256 (bank phase) + largest tap offset (2 rows + a k slice) + 16
constexpr int PT64_B = 18 * 1024;                                // 64-channel tile with its zero rows
constexpr int PT32_B = 33 * 1024;                                // 31x15x32 input of the stride-2 layer
constexpr int PPLAIN_B = 16 * 1024;                              // residual / projection tile (no padding rows)
constexpr int PRAW_B = 8 * 1024;                                 // even pixels of the raw block input (projection operand)
constexpr int pair_off_in(bool) { return PZ_B; }
constexpr int pair_off_h(bool first) { return PZ_B + 2 * (first ? PT32_B : PT64_B); }
constexpr int pair_off_res(bool first) { return pair_off_h(first) + 2 * PT64_B; }
constexpr int pair_off_raw(bool first) { return pair_off_res(first) + (first ? 2 : 3) * PPLAIN_B; }
constexpr int pair_off_const(bool first) { return pair_off_raw(first) + (first ? 2 * PRAW_B : 0); }
constexpr int pair_lds_bytes(bool first) { return pair_off_const(first) + 5 * 64 * 4; }

template <bool FIRST>
__global__ __launch_bounds__(512, 2) void mars_pair64_k(const MarsPairP P) {
    constexpr int OFF_IN = pair_off_in(FIRST), OFF_H = pair_off_h(FIRST), OFF_RES = pair_off_res(FIRST), OFF_RAW = pair_off_raw(FIRST),
                  OFF_C = pair_off_const(FIRST);
    constexpr int TIN_B = FIRST ? PT32_B : PT64_B, NRES = FIRST ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stage = wave >> 2, half = wave & 1, q = (wave >> 1) & 1;      // stage A / B; output channels 32 half ..; fragments 4 q .. 4 q + 3
    const int fr = lane & 15, fq = lane >> 4, j = fr >> 3, x = fr & 7;

    for (int i = tid * 16; i < OFF_C; i += 512 * 16) *reinterpret_cast<u4 *>(smem + i) = u4{0u, 0u, 0u, 0u};
    float *cst = reinterpret_cast<float *>(smem + OFF_C);      // [0..63] bias A, [64..127] bias B, [128..191] scale, [192..255] shift, [256..319] bias of the projection
    if (tid < 64) {
        cst[tid] = P.bias_a[tid];
        cst[64 + tid] = P.bias_b[tid];
        cst[128 + tid] = P.aff2[tid];
        cst[192 + tid] = P.aff2[P.cout_pad + tid];
        cst[256 + tid] = (FIRST && P.bias_p) ? P.bias_p[tid] : 0.f;
    }
    __syncthreads();

    // ---- the wave's filter: stage A of FIRST: 9 taps x 1 k slice (+ the projection's slice); everything else 9 taps x 2 k slices
    h8 wf[9][2][2];
    h8 wp[2];
    const bool a32 = FIRST && stage == 0;
    {
        const _Float16 *w = stage == 0 ? P.wa : P.wb;
        const int kpad = stage == 0 ? P.kpad_a : P.kpad_b;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const size_t row = (size_t)(half * 32 + frag_row(a, fr)) * kpad;
                    wf[t][ks][a] = a32 ? *reinterpret_cast<const h8 *>(w + row + t * 32 + fq * 8)          // (ks unused: both hold the slice)
                                       : *reinterpret_cast<const h8 *>(w + row + t * 64 + ks * 32 + fq * 8);
                }
#pragma unroll
        for (int a = 0; a < 2; ++a)
            wp[a] = FIRST ? *reinterpret_cast<const h8 *>(P.wp + (size_t)(half * 32 + frag_row(a, fr)) * P.kpad_p + fq * 8) : wf[0][0][a];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) asm volatile("" ::"v"(wf[t][ks][a]));      // loaded before the loop (see mars_tail.hip)
    asm volatile("" ::"v"(wp[0]), "v"(wp[1]));
#endif

    const int n0 = blockIdx.x, nstep = gridDim.x;
    const int Kc = (P.n_img - n0 + nstep - 1) / nstep;            // crops of this workgroup: n0 + c * nstep

    // ---- LDS-DMA: running per-lane source pointers (they stay on the last crop when the crops run out)
    const _Float16 *src[5];
    size_t stp[5];
    int n_dma;
    if constexpr (FIRST) {
        // rows wave, wave + 8, wave + 16, wave + 24 (row 31 does not exist: the duplicate of row 30 lands in its own slot again) of the
        // 31x15x32 pre-activation: lane -> (even | odd columns, plane, index); column 15 is the zero line
        const int par = lane >> 5, pl = (lane >> 3) & 3, col = 2 * (lane & 7) + par;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int r = wave + 8 * h < 31 ? wave + 8 * h : 30;
            src[h] = col < 15 ? P.in + ((size_t)n0 * 31 + r) * 15 * P.cs_in + (size_t)col * P.cs_in + P.coff_in + pl * 8 : P.zero;
            stp[h] = col < 15 ? (size_t)nstep * 31 * 15 * P.cs_in : 0;
        }
        // even pixels of the raw input: piece `wave` = output rows 2 wave, 2 wave + 1
        const int jr = lane >> 5, xx = lane & 7;
        src[4] = P.in2 + ((size_t)n0 * 31 + 2 * (2 * wave + jr)) * 15 * P.cs_in2 + (size_t)(2 * xx) * P.cs_in2 + P.coff_in2 + pl * 8;
        stp[4] = (size_t)nstep * 31 * 15 * P.cs_in2;
        n_dma = 5;
    } else {
        const int pl = lane >> 3, xx = lane & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            src[h] = P.in + (((size_t)n0 * 16 + wave + 8 * h) * 8 + xx) * P.cs_in + P.coff_in + pl * 8;
            stp[h] = (size_t)nstep * 128 * P.cs_in;
            src[2 + h] = P.res + (((size_t)n0 * 16 + wave + 8 * h) * 8 + xx) * P.cs_res + P.coff_res + pl * 8;
            stp[2 + h] = (size_t)nstep * 128 * P.cs_res;
        }
        src[4] = P.zero; stp[4] = 0;
        n_dma = 4;
    }
    (void)n_dma;
    int dq = 0;
    auto issue = [&]() {                                          // crop dq: input tile slot dq & 1, residual slot dq % 3
        char *T = smem + OFF_IN + (dq & 1) * TIN_B;
        if constexpr (FIRST) {
#pragma unroll
            for (int h = 0; h < 4; ++h) lds_fill16(src[h], T + ((wave + 8 * h < 31 ? wave + 8 * h : 30) + 1) * 1024);
            lds_fill16(src[4], smem + OFF_RAW + (dq & 1) * PRAW_B + wave * 1024);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) lds_fill16(src[h], T + (wave + 8 * h + 1) * 1024);
            const int rs = dq % 3;
#pragma unroll
            for (int h = 0; h < 2; ++h) lds_fill16(src[2 + h], smem + OFF_RES + rs * PPLAIN_B + (wave + 8 * h) * 1024);
        }
        ++dq;
        const bool more = dq < Kc;
#pragma unroll
        for (int h = 0; h < 5; ++h) src[h] += more ? stp[h] : 0;
    };

    const int co = half * 32 + fq * 8;                            // this lane's eight output channels
    const unsigned lo64 = (unsigned)((j * 64 + fq * 8 + x) * 16);             // in an 18-row tile: fragment row j, plane fq of k slice 0, column x
    const unsigned lo32 = (unsigned)((2 * j * 64 + fq * 8 + x) * 16);         // in the 33-row stride-2 tile: input row 2 j of the fragment, index x
    const f4 ba0 = *reinterpret_cast<const f4 *>(cst + (stage ? 64 : 0) + co), ba1 = *reinterpret_cast<const f4 *>(cst + (stage ? 64 : 0) + co + 4);

    issue();                                                      // crop 0
    for (int c = 0; c <= Kc; ++c) {
#if defined(__HIP_DEVICE_COMPILE__)
        // this wave's pieces of crop c (requested at the top of step c - 1, or just now) have landed; younger: the stores of stage B's epilogues
        // in step c - 1 (two per fragment)
        // (a stage B wave had no crop, hence no stores, in steps 0 and Kc + 1 .. : c < 2 waits for everything)
        if (stage == 0 || c < 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // crop c's tiles are complete, stage A's tile of crop c - 1 too; the slots of crop c - 1's inputs are free
        asm volatile("" ::: "memory");
#endif
        issue();                                                  // crop c + 1
        const int cc = stage == 0 ? c : c - 1;                    // the crop this wave works on
        if (cc < 0 || cc >= Kc) continue;
        const size_t n = (size_t)(n0 + cc * nstep);
#pragma unroll 1
        for (int f = 0; f < 4; ++f) {
            const int F = 4 * q + f;                              // fragment: map rows 2 F, 2 F + 1
            f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
            h8 X[4];
            if (a32) {
                // stride 2 from the 31x15x32 tile: tile row of input row r is r + 1; output row o reads input rows 2 o - 1 .. 2 o + 1, i.e. tile
                // rows 2 o + dy; column 2 x - 1 + dx: odd index x - 1 (dx = 0; x = 0 -> zero lane), even index x (dx = 1), odd index x (dx = 2)
                const unsigned tb = (unsigned)(OFF_IN + (cc & 1) * PT32_B + 4 * F * 1024) + lo32;
                const unsigned bL = x == 0 ? ((tb + 512u - 16u) & 255u) : tb + 512u - 16u, bC = tb, bR = tb + 512u;
                const unsigned pb = (unsigned)(OFF_RAW + (cc & 1) * PRAW_B) + (unsigned)(((2 * F + j) * 32 + fq * 8 + x) * 16);
                f4 accp[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#if defined(__HIP_DEVICE_COMPILE__)
#define DD_RD(u_) do { if constexpr ((u_) < 9) { constexpr int dy_ = (u_) / 3, dx_ = (u_) - dy_ * 3;                                               \
                           asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(X[(u_) & 3]) : "v"(dx_ == 0 ? bL : dx_ == 1 ? bC : bR), "n"(dy_ * 1024) : "memory"); } \
                       else asm volatile("ds_read_b128 %0, %1" : "=&v"(X[(u_) & 3]) : "v"(pb) : "memory"); } while (0)
#define DD_UNIT(u_) do { if constexpr ((u_) + 3 < 10) DD_RD(((u_) + 3) % 10);                                                                      \
                         asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(X[(u_) & 3]) : "n"((u_) + 3 < 10 ? 3 : 9 - (u_)));                           \
                         if constexpr ((u_) < 9) { _Pragma("unroll") for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(u_) % 9][0][a], X[(u_) & 3], acc[a], 0, 0, 0); } \
                         else { _Pragma("unroll") for (int a = 0; a < 2; ++a) accp[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[a], X[(u_) & 3], accp[a], 0, 0, 0); } } while (0)
                DD_RD(0); DD_RD(1); DD_RD(2);
                DD_UNIT(0); DD_UNIT(1); DD_UNIT(2); DD_UNIT(3); DD_UNIT(4); DD_UNIT(5); DD_UNIT(6); DD_UNIT(7); DD_UNIT(8); DD_UNIT(9);
#undef DD_UNIT
#undef DD_RD
#endif
                // the projection's tile for stage B: f16(sum + bias), no activation (conv_epilogue_f16x8 with ACT_NONE)
                const f4 c0 = *reinterpret_cast<const f4 *>(cst + 256 + co), c1 = *reinterpret_cast<const f4 *>(cst + 256 + co + 4);
                h8 sk;
#pragma unroll
                for (int i = 0; i < 4; ++i) { sk[i] = (_Float16)(accp[0][i] + c0[i]); sk[4 + i] = (_Float16)(accp[1][i] + c1[i]); }
                *reinterpret_cast<h8 *>(smem + OFF_RES + (cc & 1) * PPLAIN_B + ((2 * F + j) * 64 + (half * 4 + fq) * 8 + x) * 16) = sk;
            } else {
                // stride 1 from an 18-row 64-channel tile: tap dy of map row y reads tile row y + dy, column x + dx - 1 (zero lanes at the edges)
                const unsigned tb = (unsigned)((stage == 0 ? OFF_IN + (cc & 1) * PT64_B : OFF_H + (cc & 1) * PT64_B) + 2 * F * 1024) + lo64;
                const unsigned bL = x == 0 ? ((tb - 16u) & 255u) : tb - 16u, bC = tb, bR = x == 7 ? ((tb + 16u) & 255u) : tb + 16u;
#if defined(__HIP_DEVICE_COMPILE__)
#define DD_RD(u_) do { constexpr int t_ = (u_) >> 1, ks_ = (u_) & 1, dy_ = t_ / 3, dx_ = t_ - dy_ * 3;                                             \
                       asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(X[(u_) & 3]) : "v"(dx_ == 0 ? bL : dx_ == 1 ? bC : bR), "n"(dy_ * 1024 + ks_ * 512) : "memory"); } while (0)
#define DD_UNIT(u_) do { if constexpr ((u_) + 3 < 18) DD_RD(((u_) + 3) % 18);                                                                      \
                         asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(X[(u_) & 3]) : "n"((u_) + 3 < 18 ? 3 : 17 - (u_)));                          \
                         _Pragma("unroll") for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(u_) >> 1][(u_) & 1][a], X[(u_) & 3], acc[a], 0, 0, 0); } while (0)
                DD_RD(0); DD_RD(1); DD_RD(2);
                DD_UNIT(0); DD_UNIT(1); DD_UNIT(2); DD_UNIT(3); DD_UNIT(4); DD_UNIT(5); DD_UNIT(6); DD_UNIT(7); DD_UNIT(8);
                DD_UNIT(9); DD_UNIT(10); DD_UNIT(11); DD_UNIT(12); DD_UNIT(13); DD_UNIT(14); DD_UNIT(15); DD_UNIT(16); DD_UNIT(17);
#undef DD_UNIT
#undef DD_RD
#endif
            }
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = acc[0][i] + ba0[i]; v[4 + i] = acc[1][i] + ba1[i]; }
            if (stage == 0) {                                     // first layer: ELU, f16, into stage B's tile (map row y -> tile row y + 1)
                h8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (_Float16)elu(v[i]);
                *reinterpret_cast<h8 *>(smem + OFF_H + (cc & 1) * PT64_B + ((2 * F + j + 1) * 64 + (half * 4 + fq) * 8 + x) * 16) = o;
            } else {                                              // second layer: + residual, f16 output, second output ELU(scale v + shift)
                const int rs = FIRST ? (cc & 1) : cc % 3;
                const h8 rv = *reinterpret_cast<const h8 *>(smem + OFF_RES + rs * PPLAIN_B + ((2 * F + j) * 64 + (half * 4 + fq) * 8 + x) * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += (float)rv[i];
                const size_t m = (n * 16 + 2 * F + j) * 8 + x;
                h8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[i];
                *reinterpret_cast<h8 *>(P.out + m * P.cs_out + P.coff_out + co) = o;
                const f4 s0 = *reinterpret_cast<const f4 *>(cst + 128 + co), s1 = *reinterpret_cast<const f4 *>(cst + 128 + co + 4);
                const f4 t0 = *reinterpret_cast<const f4 *>(cst + 192 + co), t1 = *reinterpret_cast<const f4 *>(cst + 192 + co + 4);
                h8 o2;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o2[i] = (_Float16)elu(s0[i] * v[i] + t0[i]); o2[4 + i] = (_Float16)elu(s1[i] * v[4 + i] + t1[i]); }
                *reinterpret_cast<h8 *>(P.out2 + m * P.cs_out2 + P.coff_out2 + co) = o2;
            }
        }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <bool FIRST>
int launch_pair(hipStream_t s, int device, const MarsPairP &P) {
    constexpr int lds = pair_lds_bytes(FIRST);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mars_pair64_k<FIRST>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int n_cu = dd_cu_count(device);
    const int grid = P.n_img < n_cu ? P.n_img : n_cu;             // one workgroup per CU
    hipLaunchKernelGGL((mars_pair64_k<FIRST>), dim3(grid), dim3(512), lds, s, P);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace

int mars_pair64_launch(hipStream_t s, int device, const MarsPairP &P, bool first) {
    DD_REQUIRE(P.n_img > 0 && P.in && P.wa && P.wb && P.bias_a && P.bias_b && P.out && P.out2 && P.aff2 && P.zero && (first ? (P.in2 && P.wp) : P.res != nullptr), DD_E_ARG,
               "mars_pair64: bad argument");
    return first ? launch_pair<true>(s, device, P) : launch_pair<false>(s, device, P);
}
