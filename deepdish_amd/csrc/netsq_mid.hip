// uint8 SSD-MobileNet-v1, MobileNet blocks 3 and 4 as ONE row pipeline: depthwise 3x3 + pointwise 128 -> 128 at 75 x 75, then depthwise 3x3
// stride 2 + pointwise 128 -> 256 down to 38 x 38.  The two launches it replaces (q_dwpw_k<128,128> and <128,256> in csrc/netsq.hip) write block
// 3's 75 x 75 x 128 tensor to HBM and read it straight back (2.2 GB per 1 536 frames) and run at 56 % of what their instruction counts allow
// (64-pixel tiles, two barriers and a ring refill per tile); the row form of csrc/netsq_front.hip runs at ~85 %.  Same arithmetic, same packed
// filters, same bits as the two launches (tests/test_gpu_quant.py runs both forms against oracle/nets_quant.py); the replaced interface is the
// middle of `interpreter.invoke()` (tools/ssd_mobilenet.py:100-109 upstream).
//
// A workgroup of sixteen waves (one per CU, four waves per SIMD) walks a contiguous range of block-4 output rows (frame-major); per output row r
// ("tick" t) two phases with one barrier behind each, every phase a mix of stages of neighbouring ticks so that producer and consumer never share
// a phase:
//     X(t):  pointwise 3 of rows 2t-2, 2t-1: opnd3 -> ring3                  pointwise 4 of row t-2: opnd4 -> HBM
//     Y(t):  depthwise 3 of rows 2t, 2t+1: ring_in -> opnd3                  depthwise 4 of row t-1: ring3 -> opnd4
// ring_in = six rows of block 2's tensor as they lie in HBM ([8 planes][77][16] bytes, border columns and border rows included: a padding tap is
// a plain read), filled by LDS-DMA two rows a tick, requested at the head of Y(t-1) and waited for at the end of X(t); ring3 = three block-3 rows in
// the same layout (border columns and the padding rows above / below the frame hold its zero point); opnd3 / opnd4 = the pointwise stages' MFMA
// operand tiles [k group][pixel][16].  Work per wave: a row of 75 pixels is five 16-pixel fragments, one of 38 is three, and
//     depthwise 3: wave (row, plane) -> five fragments             pointwise 3: wave (row, channel fragment) -> five fragments x two k slices
//     depthwise 4: wave (plane, fragments {0, 1} | {2})            pointwise 4: wave = channel fragment -> three fragments x two k slices.
// Every filter a wave needs stays in its registers for the whole launch.  The lo parts of the split filters (zero in all but one plane / fragment
// of a tensor, see csrc/netsq.hip) are added behind a stage's stream from re-read operands.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "net_priv.h"
#include "netsq_dev.h"

namespace {

constexpr int M_S3 = 75, M_S4 = 38;                              // block 2 / 3 maps, block 4 map (square)
constexpr int M_PP = (M_S3 + 2) * 16;                            // plane pitch of a ring row (bytes)
constexpr int M_RB = 8 * M_PP;                                   // ring row: 128 channels
constexpr int M_NRI = 6, M_NR3 = 3;
constexpr int M_OP3 = 8 * 160 * 16, M_OP4 = 8 * 48 * 16;        // operand tiles: two rows of 80 / one row of 48 pixel slots, eight k groups
constexpr int M_LDS = M_NRI * M_RB + M_NR3 * M_RB + M_OP3 + M_OP4;
constexpr int M_PPO = (M_S4 + 2) * 16;                           // plane pitch of the output tensor

struct QMidP {
    const uint8_t *in;                                            // Q16 [n][77][8][77][16]: block 2's output
    uint8_t *out;                                                 // Q16 [n][40][16][40][16]
    const uint2 *dwa3; const int *dcb3; const i4v *w3, *w3l; const int *cb3; int zp3;      // block 3 (zp3: stored zero-point byte of its output)
    const uint2 *dwa4; const int *dcb4; const i4v *w4, *w4l; const int *cb4;               // block 4
    QReq Rd3, Rp3, Rd4, Rp4;
};

template <int SAT, bool SPLIT>
__global__ __launch_bounds__(1024) void q_mid_k(const QMidP P, const int rows_total, const int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const ring_in = smem;
    uint8_t *const ring3 = ring_in + M_NRI * M_RB;
    uint8_t *const opnd3 = ring3 + M_NR3 * M_RB;
    uint8_t *const opnd4 = opnd3 + M_OP3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int w8 = wave & 7, wj = wave >> 3;                       // plane / channel fragment of blocks 3's stages and the row of the pair
    const int g_begin = blockIdx.x * rows_per_block, g_end = min(rows_total, g_begin + rows_per_block);
    if (g_begin >= g_end) return;

    // ---- the wave's filters and constants, once
    unsigned dmask[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dmask[d] = (fr >> 2) == d ? 0xffu << (8 * (fr & 3)) : 0u;
    int tap_dx[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) tap_dx[ks] = (min(4 * ks + fq, 8) % 3) * 16;
    const bool row_up0 = fq == 3, row_up1 = fq >= 2;               // the lane's tap of k step 0 / 1 lies in the later of the step's two rows
    auto build_a = [&](const unsigned w, i4v (&A)[3]) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const unsigned rh = __builtin_amdgcn_perm(w, w, 0x01010101u * (unsigned)ks);
#pragma unroll
            for (int d = 0; d < 4; ++d) A[ks][d] = (int)(rh & dmask[d]);
        }
    };
    i4v A3h[3], A4h[3];
    const uint2 ab3 = P.dwa3[w8 * 64 + lane], ab4 = P.dwa4[w8 * 64 + lane];
    build_a(ab3.x, A3h);
    build_a(ab4.x, A4h);
    const unsigned lom3 = SPLIT ? (unsigned)__builtin_amdgcn_readfirstlane((int)(ab3.y >> 24)) : 0u;     // k steps of the plane with a lo part (netsq.pack_dw_mfma)
    const unsigned lom4 = SPLIT ? (unsigned)__builtin_amdgcn_readfirstlane((int)(ab4.y >> 24)) : 0u;
    const i4v dcb3 = *reinterpret_cast<const i4v *>(P.dcb3 + 16 * w8 + 4 * fq);
    const i4v dcb4 = *reinterpret_cast<const i4v *>(P.dcb4 + 16 * w8 + 4 * fq);
    // pointwise filters: fragment 4 mg + m holds channels 64 mg + 16 g + 4 m + r; two k slices of 64 input channels each
    i4v W3[2], W4[2];
    unsigned nz3 = 0, nz4 = 0;                                     // k slices in which the wave's fragment has a lo part at all
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        W3[kc] = P.w3[(w8 * 2 + kc) * 64 + lane];
        W4[kc] = P.w4[(wave * 2 + kc) * 64 + lane];
        if constexpr (SPLIT) {
            const i4v a = P.w3l[(w8 * 2 + kc) * 64 + lane], b = P.w4l[(wave * 2 + kc) * 64 + lane];
            if (__builtin_amdgcn_ballot_w64((a[0] | a[1] | a[2] | a[3]) != 0) != 0ull) nz3 |= 1u << kc;
            if (__builtin_amdgcn_ballot_w64((b[0] | b[1] | b[2] | b[3]) != 0) != 0ull) nz4 |= 1u << kc;
        }
    }
    const i4v cb3 = *reinterpret_cast<const i4v *>(P.cb3 + 64 * (w8 >> 2) + 16 * fq + 4 * (w8 & 3));
    const i4v cb4 = *reinterpret_cast<const i4v *>(P.cb4 + 64 * (wave >> 2) + 16 * fq + 4 * (wave & 3));
    const int Md3 = P.Rd3.M, shd3 = P.Rd3.e - 1, lod3 = P.Rd3.lo, hid3 = P.Rd3.hi;
    const int Mp3 = P.Rp3.M, shp3 = P.Rp3.e - 1, lop3 = P.Rp3.lo, hip3 = P.Rp3.hi;
    const int Md4 = P.Rd4.M, shd4 = P.Rd4.e - 1, lod4 = P.Rd4.lo, hid4 = P.Rd4.hi;
    const int Mp4 = P.Rp4.M, shp4 = P.Rp4.e - 1, lop4 = P.Rp4.lo, hip4 = P.Rp4.hi;
    // (the stored bytes are a - 128: the addends carry it, see q_requant_pack4s)
    const long long Cd3 = q_signed_c<SAT>(P.Rd3.C, shd3), Cp3 = q_signed_c<SAT>(P.Rp3.C, shp3), Cd4 = q_signed_c<SAT>(P.Rd4.C, shd4), Cp4 = q_signed_c<SAT>(P.Rp4.C, shp4);
    const u4v z3v = {(unsigned)P.zp3 * 0x01010101u, (unsigned)P.zp3 * 0x01010101u, (unsigned)P.zp3 * 0x01010101u, (unsigned)P.zp3 * 0x01010101u};

    // ring3: every byte the zero point (the border columns keep it: pointwise 3 writes interiors only)
    for (int i = tid * 16; i < M_NR3 * M_RB; i += 1024 * 16) *reinterpret_cast<u4v *>(ring3 + i) = z3v;
    __syncthreads();

    // One row of the source tensor, as it lies in HBM, into its ring slot by LDS-DMA (1 KB per wave instruction, lane-linear; the statement is
    // assembly for the reason given at q_dwpw_k's glds16: hipcc would make every later ds_read wait for it).  Rows v = -1 .. 75 (the border rows
    // of the bordered tensor are its rows 0 and 76); ten pieces a row, dealt over the waves.
    auto glds16 = [&](const uint8_t *g, const uint8_t *l) {
        unsigned keep;
        const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)l;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    };
    auto request_rows = [&](int n, int v0, int nrows) {            // rows v0 .. v0 + nrows - 1 of frame n (those inside -1 .. 75)
        for (int p = wave; p < 10 * nrows; p += 16) {
            const int rr = p / 10, piece = p - rr * 10, v = v0 + rr;
            if (v < -1 || v > M_S3) continue;
            const unsigned off = (unsigned)piece * 1024u + (unsigned)lane * 16u;
            const uint8_t *src = P.in + ((size_t)n * (M_S3 + 2) + (size_t)(v + 1)) * M_RB + off;
            if (off < (unsigned)M_RB) glds16(src, ring_in + ((v + 6) % M_NRI) * M_RB + piece * 1024);
        }
    };

    int g = g_begin;
    while (g < g_end) {
        const int n = g / M_S4, r_lo = g - n * M_S4, r_hi = min(M_S4 - 1, r_lo + (g_end - g) - 1);
        const int y_lo = 2 * r_lo - 1, y_hi = 2 * r_hi + 1;                  // block-3 rows this segment needs (row -1 / 75: padding)

        // ---- depthwise 3x3 of one plane over NF fragments of one output row: ring rows at s0, s1, s2 (byte offsets of the plane in the three
        //      slots), pixel x of fragment f at column byte lane_col + f * frag_pitch; the packed bytes go to tile + f * 256
        auto dw_run = [&](auto nf_tag, const uint8_t *ring, int s0, int s1, int s2, int lane_col, int frag_pitch, const i4v (&Ah)[3], const unsigned al_w,
                          const unsigned lom, const i4v cb, int Md, long long Cd, int shd, int lod, int hid, uint8_t *tile) __attribute__((always_inline)) {
            constexpr int NF = decltype(nf_tag)::value;
            constexpr int DW_W = NF >= 5 ? 4 : 3 * NF;                 // rolling window of operands in flight (four waves per SIMD cover the rest; 128 registers)
            const uint8_t *const a0 = ring + (row_up0 ? s1 : s0) + tap_dx[0] + lane_col;
            const uint8_t *const a1 = ring + (row_up1 ? s2 : s1) + tap_dx[1] + lane_col;
            const uint8_t *const a2 = ring + s2 + tap_dx[2] + lane_col;
            i4v acc[NF], b[DW_W];
            auto opnd_at = [&](int j) __attribute__((always_inline)) { const int ks = j / NF, f = j - ks * NF; return *reinterpret_cast<const i4v *>((ks == 0 ? a0 : ks == 1 ? a1 : a2) + f * frag_pitch); };
#pragma unroll
            for (int j = 0; j < DW_W; ++j) b[j] = opnd_at(j);
#pragma unroll
            for (int j = 0; j < 3 * NF; ++j) {
                const int ks = j / NF, f = j - ks * NF;
                if (ks == 0) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah[0], b[j % DW_W], cb, 0, 0, 0);
                else acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah[ks], b[j % DW_W], acc[f], 0, 0, 0);
                if (j + DW_W < 3 * NF) b[j % DW_W] = opnd_at(j + DW_W);
            }
            if (lom) {                                                 // (few planes: only a tensor's extreme weights overflow int8)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    if (lom & (1u << ks)) {
                        unsigned lw = al_w;
                        asm volatile("" : "+v"(lw));                // (opaque: hipcc would hoist the rare path's operands out of the tick loop and keep them in registers for good)
                        const unsigned rl = __builtin_amdgcn_perm(lw, lw, 0x01010101u * (unsigned)ks);
                        i4v Al;
#pragma unroll
                        for (int d = 0; d < 4; ++d) Al[d] = (int)(rl & dmask[d]);
#pragma unroll
                        for (int f = 0; f < NF; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Al, opnd_at(ks * NF + f), acc[f], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int f = 0; f < NF; ++f)
                *reinterpret_cast<unsigned *>(tile + f * 256) = q_requant_pack4s<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Md, Cd, Cd, Cd, Cd, shd, lod, hid);
        };
        // ---- depthwise 3: block-3 row y = 2t + wj, plane w8, five fragments: ring_in -> opnd3
        auto dw3_stage = [&](int t) __attribute__((always_inline)) {
            const int y = 2 * t + wj;
            if (y < max(y_lo, 0) || y > min(y_hi, M_S3 - 1)) return;
            const int s0 = ((y + 5) % M_NRI) * M_RB + w8 * M_PP, s1 = ((y + 6) % M_NRI) * M_RB + w8 * M_PP, s2 = ((y + 7) % M_NRI) * M_RB + w8 * M_PP;   // source rows y - 1 .. y + 1
            dw_run(std::integral_constant<int, 5>{}, ring_in, s0, s1, s2, fr * 16, 256, A3h, ab3.y, lom3, dcb3, Md3, Cd3, shd3, lod3, hid3,
                   opnd3 + ((w8 * 160 + wj * 80 + fr) * 16 + 4 * fq));
        };
        // ---- depthwise 4 (stride 2): block-4 row r = t - 1, plane w8; waves 0-7 fragments 0, 1, waves 8-15 fragment 2: ring3 -> opnd4
        auto dw4_stage = [&](int t) __attribute__((always_inline)) {
            const int r = t - 1;
            if (r < r_lo || r > r_hi) return;
            const int y = 2 * r - 1;                                // block-3 rows y .. y + 2, columns 2 x - 1 .. 2 x + 1 (bordered: 2 x .. 2 x + 2)
            const int s0 = ((y + 3) % M_NR3) * M_RB + w8 * M_PP, s1 = ((y + 4) % M_NR3) * M_RB + w8 * M_PP, s2 = ((y + 5) % M_NR3) * M_RB + w8 * M_PP;
            uint8_t *const tile = opnd4 + ((w8 * 48 + fr) * 16 + 4 * fq);
            if (wj == 0) dw_run(std::integral_constant<int, 2>{}, ring3, s0, s1, s2, fr * 32, 512, A4h, ab4.y, lom4, dcb4, Md4, Cd4, shd4, lod4, hid4, tile);
            else dw_run(std::integral_constant<int, 1>{}, ring3 + 2 * 512, s0, s1, s2, fr * 32, 512, A4h, ab4.y, lom4, dcb4, Md4, Cd4, shd4, lod4, hid4, tile + 2 * 256);
        };
        // ---- pointwise 3: block-3 row y = 2t - 2 + wj, channel fragment w8, five fragments x two k slices: opnd3 -> ring3
        auto pw3_stage = [&](int t) __attribute__((always_inline)) {
            const int y = 2 * t - 2 + wj;
            if (y < y_lo || y > y_hi) return;
            uint8_t *const dst = ring3 + ((y + 3) % M_NR3) * M_RB;
            if (y < 0 || y >= M_S3) {                               // a padding row above / below the frame: this wave's plane of the slot
                for (int i = lane * 16; i < M_PP; i += 1024) *reinterpret_cast<u4v *>(dst + w8 * M_PP + i) = z3v;
                return;
            }
            const uint8_t *const bp = opnd3 + (fq * 160 + wj * 80 + fr) * 16;
            i4v b[5], acc[5];                                       // (one k slice of operands at a time: 128 registers per wave)
#pragma unroll
            for (int f = 0; f < 5; ++f) b[f] = *reinterpret_cast<const i4v *>(bp + f * 256);
#pragma unroll
            for (int f = 0; f < 5; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W3[0], b[f], cb3, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 5; ++f) b[f] = *reinterpret_cast<const i4v *>(bp + (4 * 160 * 16) + f * 256);
#pragma unroll
            for (int f = 0; f < 5; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W3[1], b[f], acc[f], 0, 0, 0);
            if (nz3) {                                              // (one fragment and k slice of the tensor)
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    if (nz3 & (1u << kc)) {
                        int li = lane;
                        asm volatile("" : "+v"(li));                // (opaque: see dw_run)
                        const i4v wl = P.w3l[(w8 * 2 + kc) * 64 + li];
#pragma unroll
                        for (int f = 0; f < 5; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wl, *reinterpret_cast<const i4v *>(bp + kc * (4 * 160 * 16) + f * 256), acc[f], 0, 0, 0);
                    }
                }
            }
            uint8_t *const d4 = dst + (4 * (w8 >> 2) + fq) * M_PP + 4 * (w8 & 3) + (fr + 1) * 16;
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                const unsigned o = q_requant_pack4s<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Mp3, Cp3, Cp3, Cp3, Cp3, shp3, lop3, hip3);
                if (f < 4 || fr < 11) *reinterpret_cast<unsigned *>(d4 + f * 256) = o;      // (pixels 64 .. 74 of 75 in the last fragment)
            }
        };
        // ---- pointwise 4: block-4 row r = t - 2, channel fragment `wave` of 16, three fragments x two k slices: opnd4 -> HBM
        auto pw4_stage = [&](int t) __attribute__((always_inline)) {
            const int r = t - 2;
            if (r < r_lo || r > r_hi) return;
            const uint8_t *const bp = opnd4 + (fq * 48 + fr) * 16;
            i4v b[3], acc[3];
#pragma unroll
            for (int f = 0; f < 3; ++f) b[f] = *reinterpret_cast<const i4v *>(bp + f * 256);
#pragma unroll
            for (int f = 0; f < 3; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W4[0], b[f], cb4, 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 3; ++f) b[f] = *reinterpret_cast<const i4v *>(bp + (4 * 48 * 16) + f * 256);
#pragma unroll
            for (int f = 0; f < 3; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(W4[1], b[f], acc[f], 0, 0, 0);
            if (nz4) {
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    if (nz4 & (1u << kc)) {
                        int li = lane;
                        asm volatile("" : "+v"(li));
                        const i4v wl = P.w4l[(wave * 2 + kc) * 64 + li];
#pragma unroll
                        for (int f = 0; f < 3; ++f) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wl, *reinterpret_cast<const i4v *>(bp + kc * (4 * 48 * 16) + f * 256), acc[f], 0, 0, 0);
                    }
                }
            }
            uint8_t *const dst = P.out + ((size_t)((n * (M_S4 + 2) + r + 1) * 16 + 4 * (wave >> 2) + fq) * M_PPO + 4 * (wave & 3) + (fr + 1) * 16);
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const unsigned o = q_requant_pack4s<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Mp4, Cp4, Cp4, Cp4, Cp4, shp4, lop4, hip4);
                if (f < 2 || fr < 6) *reinterpret_cast<unsigned *>(dst + f * 256) = o;      // (pixels 32 .. 37 of 38 in the last fragment)
            }
        };

        const int t0 = r_lo - 1;
        request_rows(n, 2 * t0 - 1, 4);                              // what depthwise 3 of the first tick reads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = t0; t <= r_hi + 2; ++t) {
            pw3_stage(t);
            pw4_stage(t);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the rows requested in Y(t - 1) have landed (and this wave's stores are out)
            __builtin_amdgcn_s_barrier();
            if (t + 1 <= r_hi) request_rows(n, 2 * t + 3, 2);        // the rows the next tick adds: in flight through this phase and the next X
            dw3_stage(t);
            dw4_stage(t);
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's tile traffic is done (bare barrier: the requests stay in flight)
            __builtin_amdgcn_s_barrier();
        }
        g += r_hi - r_lo + 1;
    }
}

}  // namespace

// Two consecutive block ops of a uint8 SSD-MobileNet-v1 program (o3 = block 3, 128 -> 128 stride 1 at 75 x 75; o4 = block 4, 128 -> 256 stride 2) as one
// launch.  *ran = 0: not these shapes / quantisation parameters -- the caller runs the two launches.
int netq_run_mid(dd_net *net, const int32_t *o3, const int32_t *o4, int nimg, hipStream_t s, int *ran) {
    *ran = 0;
    if (o3[0] != OP_QDWPW || o4[0] != OP_QDWPW || o4[1] != o3[2] || o3[1] < 0) return DD_OK;
    const TensorDesc &ti = net->tensors[o3[1]], &t3 = net->tensors[o3[2]], &t4 = net->tensors[o4[2]];
    if (ti.h != M_S3 || ti.w != M_S3 || ti.cs != 128 || t3.h != M_S3 || t3.w != M_S3 || t3.cs != 128 || t4.h != M_S4 || t4.w != M_S4 || t4.cs != 256 ||
        !ti.pad || !t3.pad || !t4.pad) return DD_OK;
    if (o3[7] != 1 || o3[8] != 1 || o3[9] != 1 || o3[10] != 128 || o3[11] != 128 || o4[7] != 2 || o4[8] != 1 || o4[9] != 1 || o4[10] != 128 || o4[11] != 256) return DD_OK;
    const bool split = o3[18] != 0;                                // both pointwise filters split into hi + lo parts (no row sums), or neither (zero points of 128)
    if ((o4[18] != 0) != split || o3[38] != 0 || o4[38] != 0 || o3[47] != 0 || o4[47] != 0) return DD_OK;
    if (!o3[17] || !o3[20] || !o3[21] || !o4[17] || !o4[20] || !o4[21]) return DD_OK;
    char *W = net->d_weights;
    QMidP P;
    memset(&P, 0, sizeof(P));
    P.in = static_cast<const uint8_t *>(net->bufs[ti.buf]);
    P.out = static_cast<uint8_t *>(net->bufs[t4.buf]);
    auto blob = [&](int32_t off) { return W + (size_t)(uint32_t)off; };
    auto dwreq = [&](const int32_t *o) { int32_t d[48] = {0}; d[32] = o[22]; d[33] = o[23]; d[36] = o[24]; d[37] = o[25]; d[40] = o[28]; return make_req(d); };
    P.dwa3 = reinterpret_cast<const uint2 *>(blob(o3[20])); P.dcb3 = reinterpret_cast<const int *>(blob(o3[21]));
    P.w3 = reinterpret_cast<const i4v *>(blob(o3[16])); P.w3l = split ? reinterpret_cast<const i4v *>(blob(o3[18])) : nullptr;
    P.cb3 = reinterpret_cast<const int *>(blob(o3[17])); P.Rd3 = dwreq(o3); P.Rp3 = make_req(o3); P.zp3 = (P.Rp3.zo ^ 0x80) & 0xff;
    P.dwa4 = reinterpret_cast<const uint2 *>(blob(o4[20])); P.dcb4 = reinterpret_cast<const int *>(blob(o4[21]));
    P.w4 = reinterpret_cast<const i4v *>(blob(o4[16])); P.w4l = split ? reinterpret_cast<const i4v *>(blob(o4[18])) : nullptr;
    P.cb4 = reinterpret_cast<const int *>(blob(o4[17])); P.Rd4 = dwreq(o4); P.Rp4 = make_req(o4);
    const QReq *R[4] = {&P.Rd3, &P.Rp3, &P.Rd4, &P.Rp4};
    bool byte_clamp = true, small_shift = true;
    for (const QReq *r : R) {
        if (r->linear || r->e < 1) return DD_OK;
        byte_clamp = byte_clamp && r->lo == 0 && r->hi == 255;
        small_shift = small_shift && r->e <= 8;
    }
    const int sat = byte_clamp ? (small_shift ? 2 : 1) : 0;
    void (*kern)(const QMidP, const int, const int) =
        split ? (sat == 2 ? &q_mid_k<2, true> : sat == 1 ? &q_mid_k<1, true> : &q_mid_k<0, true>)
              : (sat == 2 ? &q_mid_k<2, false> : sat == 1 ? &q_mid_k<1, false> : &q_mid_k<0, false>);
    static DevOnce once;
    const int rc = once.run(net->ctx->device, [&]() -> int {
        for (const void *f : {reinterpret_cast<const void *>(&q_mid_k<2, true>), reinterpret_cast<const void *>(&q_mid_k<1, true>), reinterpret_cast<const void *>(&q_mid_k<0, true>),
                              reinterpret_cast<const void *>(&q_mid_k<2, false>), reinterpret_cast<const void *>(&q_mid_k<1, false>), reinterpret_cast<const void *>(&q_mid_k<0, false>)})
            DD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, M_LDS));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    // a segment costs three ticks beyond its rows: ranges of at least a quarter of a frame, one workgroup per CU
    const int n_cu = dd_cu_count(net->ctx->device);
    const int rows_total = nimg * M_S4;
    const int blocks = std::max(1, std::min(n_cu, rows_total / 10));
    const int rpb = dd_ceil_div(rows_total, blocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)dd_ceil_div(rows_total, rpb)), dim3(1024), M_LDS, s, P, rows_total, rpb);
    DD_LAUNCH_CHECK();
    *ran = 1;
    return DD_OK;
}
