// MARS re-ID encoder, conv4_x (tools/freeze_model.py:118-141 upstream: residual blocks conv4_1 / conv4_3 on 8x4x128 maps):
// crop-resident, weight-stationary convolutions.
//
// The layers of this stage have M = 32 pixels per crop, N = 128 output channels and K = 1152 (3x3x128): a 295 KB filter.  A tiled
// GEMM (conv_glds_k) re-fetches filter panels for every pixel tile and ran these layers at 0.24-0.32 of the f16 MFMA peak.  Here a
// workgroup of 8 waves IS the filter: wave (cg, kh) keeps the A fragments of output channels 32 cg .. 32 cg + 31 for the input-channel
// half kh of every tap in registers for the whole launch (9 taps x 2 k slices x 2 fragments = 144 VGPRs), and the crops stream
// through it: a crop's input tile (8 KB) comes into LDS by LDS-DMA (one map row per wave instruction, LEAD crops ahead, counted
// vmcnt), every B fragment read from it feeds two MFMAs (LDS at half its bandwidth when the matrix pipe is full), the two K halves
// meet through LDS -- each wave of a pair hands the partial sums of ONE of the two 16-pixel fragments to its partner and finishes
// the other -- and the epilogue of crop c runs behind the barrier of crop c + 1, so one barrier per crop orders everything:
// the DMA'd tile, the exchanged partial sums and the reuse of both.
//
// Tile layouts (16-byte chunks = 8 channels of one pixel), chosen so that a fragment read is bank-conflict free and a map row is
// one contiguous 1 KiB LDS-DMA:
//   stride 1: [10 rows: zero, 8 map rows, zero][16 planes][4 columns], row pitch 72 chunks; the left / right padding columns do not
//             exist: the lanes of column 0 (tap dx = 0) and column 3 (dx = 2) point into a zero region instead;
//   stride 2: [17 rows: 16 map rows, zero][even columns | odd columns][8 planes][4], row pitch 68 chunks (TF SAME for an even size and
//             stride 2 pads only below / right); the lanes of output column 3 mask the right tap;
//   projection (1x1 stride 2 of the block's raw input): [8 planes][8 rows][4 columns] of the even pixels.
#include "mars_tail.h"
#include <vector>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_ELU = 2 };

__device__ __forceinline__ float elu(float v) {                  // as apply_act(ACT_ELU) in nets.hip: max(v, exp(min(v, 0)) - 1)
    float e = __builtin_amdgcn_exp2f(v * 1.44269504088896340736f);
    e = __builtin_amdgcn_fmed3f(e, 0.f, 1.f);
    return __builtin_amdgcn_fmed3f(v, e - 1.f, 3.0e38f);
}

__device__ __forceinline__ void lds_fill16(const _Float16 *g, char *lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(g, lds_wave_base, 16, 0, 0);
#endif
}

// row of the [32 x K] filter slice a lane loads for A fragment a: the lane that owns pixel fr of the result then holds output
// channels 8 fq .. 8 fq + 7 in acc[0][0..3], acc[1][0..3] (the order conv_glds_k stores plain f16 outputs in)
__device__ __forceinline__ int frag_row(int a, int fr) { return (fr >> 2) * 8 + (a & 1) * 4 + (fr & 3); }

// -DDD_MARS_STAMPS (scripts/experiments/mars_stamps.sh; never in the product build): s_memtime stamps per step and wave, summed per phase
#ifdef DD_MARS_STAMPS
__device__ unsigned long long dd_mars_stamps[256 * 8 * 8];
#define DD_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_last; st_last = t_; } while (0)
#else
#define DD_STAMP(k) do { } while (0)
#endif

constexpr int WS_LEAD = 2, WS_NB = WS_LEAD + 2, WS_NBR = WS_LEAD + 3;
constexpr int ws_tile_bytes(int mode) { return mode == MARS_WS_S2_PROJ ? 17 * 68 * 16 : 10 * 72 * 16; }
constexpr int ws_zero_bytes(int mode) { return mode == MARS_WS_S2_PROJ ? 0 : 7680; }          // 256 (a lane's bank phase) + largest stride-1 tap offset + 16, rounded up
constexpr int WS_RES_BYTES = 8 * 72 * 16, WS_PRJ_BYTES = 4096;
constexpr int ws_xw_bytes(int mode) { return mode == MARS_WS_S2_PROJ ? 4096 : 2048; }
constexpr int ws_off_tiles(int mode) { return ws_zero_bytes(mode); }
constexpr int ws_off_aux(int mode) { return ws_off_tiles(mode) + WS_NB * ws_tile_bytes(mode); }
constexpr int ws_off_xchg(int mode) {
    return ws_off_aux(mode) + (mode == MARS_WS_S1_RES ? WS_NBR * WS_RES_BYTES : mode == MARS_WS_S2_PROJ ? WS_NB * WS_PRJ_BYTES : 0);
}
constexpr int ws_off_const(int mode) { return ws_off_xchg(mode) + 2 * 8 * ws_xw_bytes(mode); }
constexpr int ws_lds_bytes(int mode) { return ws_off_const(mode) + 3 * 128 * 4; }

template <int MODE, int ACT, bool OUT2>
__global__ __launch_bounds__(512, 2) void mars_ws128_k(const MarsWsP P) {
    constexpr bool S2 = MODE == MARS_WS_S2_PROJ, RES = MODE == MARS_WS_S1_RES;
    constexpr int LEAD = WS_LEAD, NB = WS_NB, NBR = WS_NBR;
    constexpr int TILE_B = ws_tile_bytes(MODE), XW_B = ws_xw_bytes(MODE);
    constexpr int OFF_T = ws_off_tiles(MODE), OFF_A = ws_off_aux(MODE), OFF_X = ws_off_xchg(MODE), OFF_C = ws_off_const(MODE);
    constexpr int KS = S2 ? 1 : 2, CIN = S2 ? 64 : 128;
    [[maybe_unused]] constexpr int N_DMA = S2 ? 3 : RES ? 2 : 1;
    [[maybe_unused]] constexpr int N_ST = S2 ? 2 : (RES && OUT2) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kh = wave >> 2;                      // waves w and w + 4 (one SIMD) are the two K halves of a channel group
    const int fr = lane & 15, fq = lane >> 4, yy = fr >> 2, x = fr & 3;

    // zero region, tile padding rows, constants
    for (int i = tid * 16; i < OFF_A; i += 512 * 16) *reinterpret_cast<u4 *>(smem + i) = u4{0u, 0u, 0u, 0u};
    float *cst = reinterpret_cast<float *>(smem + OFF_C);
    if (tid < 128) {
        cst[tid] = P.bias[tid];
        if constexpr (S2) cst[128 + tid] = P.bias2 ? P.bias2[tid] : 0.f;
        if constexpr (RES && OUT2) { cst[128 + tid] = P.aff2[tid]; cst[256 + tid] = P.aff2[P.cout_pad + tid]; }
    }
    __syncthreads();

    h8 wf[9][KS][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
                wf[t][ks][a] = *reinterpret_cast<const h8 *>(P.w + (size_t)(cg * 32 + frag_row(a, fr)) * P.kpad + t * CIN + kh * (CIN / 2) + ks * 32 + fq * 8);
    h8 wp[2];
    if constexpr (S2) {
#pragma unroll
        for (int a = 0; a < 2; ++a) wp[a] = *reinterpret_cast<const h8 *>(P.w2 + (size_t)(cg * 32 + frag_row(a, fr)) * P.kpad2 + kh * 32 + fq * 8);
    }

    // The filter has arrived before the loop: hipcc cannot see the counted waits below (inline assembly), so with these loads still
    // on its scoreboard it would put an s_waitcnt vmcnt(0) in front of the first MFMA of EVERY step -- and drain the look-ahead DMAs.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) asm volatile("" ::"v"(wf[t][ks][a]));
    if constexpr (S2) asm volatile("" ::"v"(wp[0]), "v"(wp[1]));
#endif

    const int n0 = blockIdx.x, nstep = gridDim.x;
    const int Kc = (P.n_img - n0 + nstep - 1) / nstep;            // crops of this workgroup: n0 + c * nstep

    // ---- LDS-DMA of the next crop (dq): running per-lane source pointers, advanced by a wave-uniform stride.  Past the last crop the
    // pointers stay on it: the same bytes go once more into a tile slot nobody reads, and every step issues the same number of operations.
    int dq = 0, dq_slot = 0, dq_rslot = 0;
    const _Float16 *src_a, *src_b, *src_c;                        // stride 1: input row, residual row, -; stride 2: input rows w and w + 8, projection piece
    size_t step_a, step_b, step_c;
    if constexpr (!S2) {
        src_a = P.in + (size_t)n0 * 32 * P.cs_in + ((wave * 4 + (lane & 3)) * P.cs_in + P.coff_in + (lane >> 2) * 8);
        src_b = RES ? P.res + (size_t)n0 * 32 * P.cs_res + ((wave * 4 + (lane & 3)) * P.cs_res + P.coff_res + (lane >> 2) * 8) : P.zero;
        src_c = P.zero;
        step_a = (size_t)nstep * 32 * P.cs_in; step_b = (size_t)nstep * 32 * P.cs_res; step_c = 0;
    } else {
        const int par = lane >> 5, pl = (lane >> 2) & 7, col = 2 * (lane & 3) + par;
        src_a = P.in + (size_t)n0 * 128 * P.cs_in + ((wave * 8 + col) * P.cs_in + P.coff_in + pl * 8);
        src_b = src_a + (size_t)64 * P.cs_in;
        const int ch = (wave & 3) * 64 + lane;                    // chunk of the projection tile: [plane][row][column]
        src_c = P.in2 + (size_t)n0 * 128 * P.cs_in2 + ((2 * ((ch >> 2) & 7) * 8 + 2 * (ch & 3)) * P.cs_in2 + P.coff_in2 + (ch >> 5) * 8);
        step_a = step_b = (size_t)nstep * 128 * P.cs_in; step_c = (size_t)nstep * 128 * P.cs_in2;
    }
    auto issue = [&]() {
        char *T = smem + OFF_T + dq_slot * TILE_B;
        if constexpr (!S2) {
            lds_fill16(src_a, T + (wave + 1) * 72 * 16);
            if constexpr (RES) lds_fill16(src_b, smem + OFF_A + dq_rslot * WS_RES_BYTES + wave * 72 * 16);
        } else {
            lds_fill16(src_a, T + wave * 68 * 16);
            lds_fill16(src_b, T + (wave + 8) * 68 * 16);
            lds_fill16(src_c, smem + OFF_A + dq_slot * WS_PRJ_BYTES + (wave & 3) * 1024);
        }
        ++dq;
        const bool more = dq < Kc;
        src_a += more ? step_a : 0;
        if constexpr (RES || S2) src_b += more ? step_b : 0;
        if constexpr (S2) src_c += more ? step_c : 0;
        dq_slot = dq_slot + 1 == NB ? 0 : dq_slot + 1;
        dq_rslot = dq_rslot + 1 == NBR ? 0 : dq_rslot + 1;
    };

    // per-lane constants of the fragment reads (byte offsets inside a tile)
    const unsigned lo = S2 ? (unsigned)((2 * yy * 68 + (kh * 4 + fq) * 4 + x) * 16) : (unsigned)((yy * 72 + (kh * 8 + fq) * 4 + x) * 16);
    const unsigned lo_prj = (unsigned)(((kh * 4 + fq) * 32 + yy * 4 + x) * 16);
    const unsigned lo_res = (unsigned)(((4 * kh + yy) * 72 + (cg * 4 + fq) * 4 + x) * 16);
    const int co = cg * 32 + fq * 8;                              // this lane's eight output channels
    const f4 b0 = *reinterpret_cast<const f4 *>(cst + co), b1 = *reinterpret_cast<const f4 *>(cst + co + 4);

    f4 keep[S2 ? 4 : 2];
    int slot = 0, rslot_prev = 0, rslot = 0;                     // tile slot of crop c; residual slots of crops c - 1 and c

    auto epilogue = [&](int cprev) {                              // crop cprev = c - 1: this wave finishes pixel fragment kh
        const char *Xp = smem + OFF_X + ((cprev & 1) * 8 + (wave ^ 4)) * XW_B + lane * 16;
        const size_t n = (size_t)(n0 + cprev * nstep);
        const size_t m = (n * 8 + 4 * kh + yy) * 4 + x;
        float v[8];
        {
            const f4 o0 = *reinterpret_cast<const f4 *>(Xp), o1 = *reinterpret_cast<const f4 *>(Xp + 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = keep[0][i] + o0[i] + b0[i]; v[4 + i] = keep[1][i] + o1[i] + b1[i]; }
        }
        if constexpr (ACT == ACT_ELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = elu(v[i]);
        }
        if constexpr (RES) {
            const h8 rv = *reinterpret_cast<const h8 *>(smem + OFF_A + rslot_prev * WS_RES_BYTES + lo_res);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)rv[i];
        }
        h8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[i];
        *reinterpret_cast<h8 *>(P.out + m * P.cs_out + P.coff_out + co) = o;
        if constexpr (RES && OUT2) {
            const f4 s0 = *reinterpret_cast<const f4 *>(cst + 128 + co), s1 = *reinterpret_cast<const f4 *>(cst + 128 + co + 4);
            const f4 t0 = *reinterpret_cast<const f4 *>(cst + 256 + co), t1 = *reinterpret_cast<const f4 *>(cst + 256 + co + 4);
            h8 o2;
#pragma unroll
            for (int i = 0; i < 4; ++i) { o2[i] = (_Float16)elu(s0[i] * v[i] + t0[i]); o2[4 + i] = (_Float16)elu(s1[i] * v[4 + i] + t1[i]); }
            *reinterpret_cast<h8 *>(P.out2 + m * P.cs_out2 + P.coff_out2 + co) = o2;
        }
        if constexpr (S2) {                                       // the projection: no activation
            const f4 o0 = *reinterpret_cast<const f4 *>(Xp + 2048), o1 = *reinterpret_cast<const f4 *>(Xp + 3072);
            const f4 c0 = *reinterpret_cast<const f4 *>(cst + 128 + co), c1 = *reinterpret_cast<const f4 *>(cst + 128 + co + 4);
            h8 o2;
#pragma unroll
            for (int i = 0; i < 4; ++i) { o2[i] = (_Float16)(keep[2][i] + o0[i] + c0[i]); o2[4 + i] = (_Float16)(keep[3][i] + o1[i] + c1[i]); }
            *reinterpret_cast<h8 *>(P.out2 + m * P.cs_out2 + P.coff_out2 + co) = o2;
        }
    };

#ifdef DD_MARS_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_begin = st_last;
#endif
    static_assert(LEAD == 2, "the counted waits below are written for two crops of look-ahead");
    for (int ci = 0; ci < LEAD; ++ci) issue();
    DD_STAMP(0);                                                  // prologue DMAs
    // Step c in two phases, the two K halves in antiphase: phase 1 -- half 0 multiplies crop c while half 1 requests crop c + 2 and
    // finishes crop c - 1; phase 2 -- the roles swap.  Every SIMD holds one wave of each half, so its matrix pipe is fed by one wave
    // while the other issues the vector, LDS and memory work of its epilogue: run alike (both multiply, then both finish) the waves of
    // a SIMD contend for the pipe and then leave it idle (stamps: 4 250 cycles per step for 2 304 of matrix work).
    for (int c = 0; c <= Kc; ++c) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Barrier 1: crop c's tile is complete.  This wave requested its piece two non-multiplying phases ago; younger and allowed to fly:
        // the stores of that phase's epilogue, the requests for crop c + 1 and the stores of the epilogue after them.
        if (kh == 0 || c == 0) {                                  // half 1 waited at the end of its phase 1 (below): its compute phase ends at this barrier
            if (c <= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N_DMA) : "memory");
            else if (c == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N_DMA + N_ST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N_DMA + 2 * N_ST) : "memory");
        } else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DD_STAMP(2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#endif
        DD_STAMP(3);
        auto compute = [&](f4 (&kp)[S2 ? 4 : 2]) {
            const unsigned tb = (unsigned)(OFF_T + slot * TILE_B) + lo;
            f4 acc[2][2] = {{f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}};
            if constexpr (!S2) {
                // tile row of map row y is y + 1, so tap dy of output row 4 p + yy reads tile row 4 p + yy + dy; the column of tap dx
                // is x + dx - 1: the lanes of x = 0 (dx = 0) and x = 3 (dx = 2) read the zero region, at the same immediate offsets
                // (at the bank phase of the address they replace -- addr mod 256 -- so that a read stays conflict free: with all of them
                // on one zero chunk, six taps of nine took two LDS passes per lane group and the LDS, not the matrix pipe, set the pace)
                const unsigned bs[3] = {x == 0 ? ((tb - 16u) & 255u) : tb - 16u, tb, x == 3 ? ((tb + 16u) & 255u) : tb + 16u};
                // 18 units of (tap, pixel fragment): two fragment reads (the tap's two k slices) feed four MFMAs.  Four rotating operand
                // sets, three units (six reads) in flight.  The reads are inline assembly with hand-counted waits: hipcc waits for a
                // ds_read result with lgkmcnt(0) -- for every read in flight, however many are younger -- which made the look-ahead
                // one unit deep whatever the source said (stamps: 1 690 cycles per crop and wave for 1 152 cycles of matrix work).
                h8 X[4][2];
#if defined(__HIP_DEVICE_COMPILE__)
#define DD_RD(u_) do { constexpr int t_ = (u_) >> 1, p_ = (u_) & 1, dy_ = t_ / 3, dx_ = t_ - dy_ * 3;                                      \
                       asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                                       \
                                    : "=&v"(X[(u_) & 3][0]), "=&v"(X[(u_) & 3][1]) : "v"(bs[dx_]), "n"(((4 * p_ + dy_) * 72) * 16), "n"(((4 * p_ + dy_) * 72 + 16) * 16) : "memory"); } while (0)
#define DD_UNIT(u_) do { if constexpr ((u_) + 3 < 18) DD_RD(((u_) + 3) % 18);                                                            \
                         asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(X[(u_) & 3][0]), "+v"(X[(u_) & 3][1]) : "n"((u_) + 3 < 18 ? 6 : 2 * (17 - (u_))));          \
                         _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int a = 0; a < 2; ++a)                  \
                             acc[(u_) & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(u_) >> 1][ks][a], X[(u_) & 3][ks], acc[(u_) & 1][a], 0, 0, 0); } while (0)
                DD_RD(0); DD_RD(1); DD_RD(2);
                DD_UNIT(0); DD_UNIT(1); DD_UNIT(2); DD_UNIT(3); DD_UNIT(4); DD_UNIT(5); DD_UNIT(6); DD_UNIT(7); DD_UNIT(8);
                DD_UNIT(9); DD_UNIT(10); DD_UNIT(11); DD_UNIT(12); DD_UNIT(13); DD_UNIT(14); DD_UNIT(15); DD_UNIT(16); DD_UNIT(17);
#undef DD_UNIT
#undef DD_RD
#endif
            } else {
                // output pixel (4 p + yy, x) reads input rows 8 p + 2 yy + dy and columns 2 x + dx: even columns for dx = 0 / 2 (index x,
                // x + 1), odd ones for dx = 1; column 8 does not exist: the lanes of x = 3 zero their dx = 2 operands
                f4 accp[2][2] = {{f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}};
                // 20 units: (tap, pixel fragment) x 18, then the projection's two fragments; one read feeds two MFMAs; four rotating
                // operand registers, three reads in flight, hand-counted waits (see the stride-1 form)
                h8 X[4];
                const unsigned pb = (unsigned)(OFF_A + slot * WS_PRJ_BYTES) + lo_prj;
                const h8 hz = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
#if defined(__HIP_DEVICE_COMPILE__)
#define DD_RD(u_) do { if constexpr ((u_) < 18) { constexpr int t_ = (u_) >> 1, p_ = (u_) & 1, dy_ = t_ / 3, dx_ = t_ - dy_ * 3;                    \
                           asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(X[(u_) & 3]) : "v"(tb), "n"(((8 * p_ + dy_) * 68 + (dx_ == 1 ? 32 : 0) + (dx_ == 2 ? 1 : 0)) * 16) : "memory"); } \
                       else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(X[(u_) & 3]) : "v"(pb), "n"(((u_) - 18) * 256) : "memory"); } while (0)
#define DD_UNIT(u_) do { if constexpr ((u_) + 3 < 20) DD_RD(((u_) + 3) % 20);                                                            \
                         asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(X[(u_) & 3]) : "n"((u_) + 3 < 20 ? 3 : 19 - (u_)));               \
                         if constexpr ((u_) < 18) { const h8 xv_ = (((u_) >> 1) % 3 == 2 && x == 3) ? hz : X[(u_) & 3];                    \
                             _Pragma("unroll") for (int a = 0; a < 2; ++a) acc[(u_) & 1][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[(u_) >> 1][0][a], xv_, acc[(u_) & 1][a], 0, 0, 0); } \
                         else { _Pragma("unroll") for (int a = 0; a < 2; ++a) accp[(u_) - 18][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[a], X[(u_) & 3], accp[(u_) - 18][a], 0, 0, 0); } } while (0)
                DD_RD(0); DD_RD(1); DD_RD(2);
                DD_UNIT(0); DD_UNIT(1); DD_UNIT(2); DD_UNIT(3); DD_UNIT(4); DD_UNIT(5); DD_UNIT(6); DD_UNIT(7); DD_UNIT(8); DD_UNIT(9);
                DD_UNIT(10); DD_UNIT(11); DD_UNIT(12); DD_UNIT(13); DD_UNIT(14); DD_UNIT(15); DD_UNIT(16); DD_UNIT(17); DD_UNIT(18); DD_UNIT(19);
#undef DD_UNIT
#undef DD_RD
#endif
                char *Xm = smem + OFF_X + ((c & 1) * 8 + wave) * XW_B + lane * 16;
                *reinterpret_cast<f4 *>(Xm + 2048) = kh ? accp[0][0] : accp[1][0];
                *reinterpret_cast<f4 *>(Xm + 3072) = kh ? accp[0][1] : accp[1][1];
                kp[2] = kh ? accp[1][0] : accp[0][0];
                kp[3] = kh ? accp[1][1] : accp[0][1];
            }
            // hand the other fragment's partial sums to the partner, keep this wave's own
            char *Xm = smem + OFF_X + ((c & 1) * 8 + wave) * XW_B + lane * 16;
            *reinterpret_cast<f4 *>(Xm) = kh ? acc[0][0] : acc[1][0];
            *reinterpret_cast<f4 *>(Xm + 1024) = kh ? acc[0][1] : acc[1][1];
            kp[0] = kh ? acc[1][0] : acc[0][0];
            kp[1] = kh ? acc[1][1] : acc[0][1];
        };
        if (kh == 0) {
            f4 kn[S2 ? 4 : 2];
            if (c < Kc) compute(kn);
            DD_STAMP(5);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // barrier 2: the halves swap roles (half 0's partial sums of crop c are out)
            asm volatile("" ::: "memory");
#endif
            DD_STAMP(3);
            issue();                                              // crop c + 2
            DD_STAMP(1);
            if (c > 0) epilogue(c - 1);
            DD_STAMP(4);
#pragma unroll
            for (int i = 0; i < (S2 ? 4 : 2); ++i) keep[i] = kn[i];
        } else {
            issue();
            DD_STAMP(1);
            if (c > 0) epilogue(c - 1);
            DD_STAMP(4);
#if defined(__HIP_DEVICE_COMPILE__)
            // this wave's piece of crop c + 1 (requested one step ago) has landed: nothing of it is waited for behind the multiplications.
            // Younger, as at barrier 1: the stores of that phase's epilogue, the requests of this phase and the stores of this epilogue
            if (c == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N_DMA) : "memory");
            else if (c == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N_DMA + N_ST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(N_DMA + 2 * N_ST) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#endif
            DD_STAMP(3);
            if (c < Kc) compute(keep);
            DD_STAMP(5);
        }
        slot = slot + 1 == NB ? 0 : slot + 1;
        rslot_prev = rslot;
        rslot = rslot + 1 == NBR ? 0 : rslot + 1;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the look-ahead DMAs past the last crop
#endif
#ifdef DD_MARS_STAMPS
    if (lane == 0) {
        unsigned long long *d = dd_mars_stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int k = 0; k < 6; ++k) d[k] = st_acc[k];
        d[6] = __builtin_amdgcn_s_memtime() - st_begin;
        d[7] = (unsigned long long)Kc;
    }
#endif
}

template <int MODE, int ACT, bool OUT2>
int launch_one(hipStream_t s, int device, const MarsWsP &P) {
    constexpr int lds = ws_lds_bytes(MODE);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mars_ws128_k<MODE, ACT, OUT2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const int n_cu = dd_cu_count(device);
    const int grid = P.n_img < n_cu ? P.n_img : n_cu;             // one workgroup per CU (the device's count, not a literal), each the whole filter
    hipLaunchKernelGGL((mars_ws128_k<MODE, ACT, OUT2>), dim3(grid), dim3(512), lds, s, P);
    DD_LAUNCH_CHECK();
#ifdef DD_MARS_STAMPS
    {
        static std::vector<unsigned long long> h(256 * 8 * 8);
        DD_HIP(hipStreamSynchronize(s));
        DD_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(dd_mars_stamps), h.size() * 8));
        // per K half: mean cycles per step of each phase over the workgroups' waves
        for (int half = 0; half < 2; ++half) {
            double a[7] = {0, 0, 0, 0, 0, 0, 0};
            int nw = 0;
            for (int b = 0; b < grid; ++b)
                for (int w = half * 4; w < half * 4 + 4; ++w, ++nw) {
                    const unsigned long long *d = h.data() + ((size_t)b * 8 + w) * 8;
                    for (int k = 0; k < 7; ++k) a[k] += (double)d[k] / (double)(k == 0 || k == 6 ? 1 : d[7] + 1);
                }
            fprintf(stderr, "mars_ws128_k<%d,%d,%d> K half %d: prologue DMAs %.0f | per step: issue %.0f, wait %.0f, barrier %.0f, epilogue %.0f, compute %.0f | loop total %.0f cycles, %d steps\n",
                    MODE, ACT, (int)OUT2, half, a[0] / nw, a[1] / nw, a[2] / nw, a[3] / nw, a[4] / nw, a[5] / nw, a[6] / nw, (int)h[7] + 1);
        }
    }
#endif
    return DD_OK;
}

}  // namespace

int mars_ws128_launch(hipStream_t s, int device, const MarsWsP &P, int mode, int act, bool out2) {
    DD_REQUIRE(P.n_img > 0 && P.in && P.w && P.bias && P.out && P.zero, DD_E_ARG, "mars_ws128: bad argument");
    DD_REQUIRE((long long)P.n_img * 128 * (P.cs_in > P.cs_out ? P.cs_in : P.cs_out) < (1ll << 40), DD_E_CAPACITY, "mars_ws128: batch of %d", P.n_img);
    if (mode == MARS_WS_S2_PROJ) {
        DD_REQUIRE(P.in2 && P.w2 && P.out2 && act == ACT_ELU, DD_E_ARG, "mars_ws128: stride-2 form needs the projection operands and ELU");
        return launch_one<MARS_WS_S2_PROJ, ACT_ELU, false>(s, device, P);
    }
    if (mode == MARS_WS_S1) {
        DD_REQUIRE(!P.res && !P.out2, DD_E_ARG, "mars_ws128: plain form with a residual or a second output");
        return act == ACT_ELU ? launch_one<MARS_WS_S1, ACT_ELU, false>(s, device, P) : launch_one<MARS_WS_S1, ACT_NONE, false>(s, device, P);
    }
    DD_REQUIRE(mode == MARS_WS_S1_RES && P.res && act == ACT_NONE && (!out2 || (P.out2 && P.aff2)), DD_E_ARG, "mars_ws128: residual form");
    return out2 ? launch_one<MARS_WS_S1_RES, ACT_NONE, true>(s, device, P) : launch_one<MARS_WS_S1_RES, ACT_NONE, false>(s, device, P);
}
