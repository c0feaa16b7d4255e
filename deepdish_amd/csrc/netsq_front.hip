// uint8 SSD-MobileNet-v1 front end as ONE row pipeline: first layer (3x3 stride 2, 3 -> 32) -> MobileNet block 1 (depthwise 3x3 +
// pointwise 32 -> 64) -> block 2 (depthwise 3x3 stride 2 + pointwise 64 -> 128) in one launch.  The three launches it replaces
// (q_conv0_k, q_dwpw_k x 2 in csrc/netsq.hip) write the 150 x 150 x 32 and 150 x 150 x 64 tensors to HBM and read them straight back:
// 4.08 GB of the forward's 9.8 GB per 768 frames.  Here only the 300 x 300 x 3 frame is read and block 2's 75 x 75 x 128 tensor is
// written (0.76 GB per 768 frames); everything between them lives in LDS rings of three or four image rows.
// Same arithmetic, same packed filters, same bits as the three launches (tests/test_gpu_quant.py runs both forms against
// oracle/nets_quant.py); the replaced interface is the front of `interpreter.invoke()` (tools/ssd_mobilenet.py:100-109 upstream).
//
// A workgroup of four waves walks a contiguous range of block-2 output rows (frame-major); per output row r ("tick" t) it runs two phases
// with one barrier behind each, every phase a mix of stages of neighbouring ticks so that producer and consumer never share a phase:
//     X(t):  first layer rows 2t+2, 2t+3 -> ring0          pointwise 1 of rows 2t-1, 2t: opnd1 -> ring1       pointwise 2 of row t-2: opnd2 -> HBM
//     Y(t):  depthwise 1 of rows 2t+1, 2t+2: ring0 -> opnd1                              depthwise 2 of row t-1: ring1 -> opnd2
// ring0 = first-layer rows (4 slots of [2 planes][152][16] bytes, slot = row & 3), ring1 = block-1 rows (3 slots of [4 planes][152][16],
// slot = row % 3), both in the bordered 16-channel-plane layout of the HBM tensors (border columns and the padding rows hold the tensor's
// zero point), opnd1 / opnd2 = the pointwise stages' MFMA operand tiles [k group][pixel][16].  Work per wave and phase is the same for
// all four waves: a row of 150 pixels is ten 16-pixel fragments, one of 75 is five, and
//     first layer: wave (row, half) -> five fragments x two channel fragments       depthwise 1: wave (row, plane) -> ten fragments
//     pointwise 1: wave (row, channel-fragment pair) -> ten fragments                 depthwise 2: wave = plane -> five fragments
//     pointwise 2: wave = channel-fragment pair -> five fragments.
// Every filter a wave needs stays in its registers for the whole launch (two waves per SIMD, 256 registers each: two workgroups per CU,
// 72.5 KB of LDS each).  The next tick's frame bytes are requested at the head of phase Y and used at the head of phase X.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "net_priv.h"
#include "netsq_dev.h"

namespace {

constexpr int F_S0 = 300, F_S1 = 150, F_S2 = 75;                 // frame, first layer / block 1, block 2 (square)
constexpr int F_PP = (F_S1 + 2) * 16;                            // plane pitch of a ring row (bytes)
constexpr int F_RB0 = 2 * F_PP, F_RB1 = 4 * F_PP;                // ring row: 32 / 64 channels
constexpr int F_NR0 = 4, F_NR1 = 3;
constexpr int F_OP1 = 4 * 320 * 16, F_OP2 = 4 * 80 * 16;        // operand tiles: two rows of 160 / one row of 80 pixel slots, four k groups
constexpr int F_LDS = F_NR0 * F_RB0 + F_NR1 * F_RB1 + F_OP1 + F_OP2;
constexpr int F_PPO = (F_S2 + 2) * 16;                           // plane pitch of the output tensor

struct QFrontP {
    const uint8_t *src; long long src_bytes;                      // u8 [n][300][300][3]
    uint8_t *out;                                                 // Q16 [n][77][8][77][16]
    const i4v *w0, *w0l; const int *cb0; int in_zp, zp0;          // first layer: filter (hi / lo parts), constants, input zero point, stored zero-point byte of its output
    const uint2 *dwa1; const int *dcb1; const i4v *w1; const int *cb1; int zp1;     // block 1 (zp1: stored zero-point byte of its output)
    const uint2 *dwa2; const int *dcb2; const i4v *w2, *w2l; const int *cb2;        // block 2
    QReq R0, Rd1, Rp1, Rd2, Rp2;
    unsigned long long *dbg;                                       // DD_Q_STAMPS=1: per wave, cycles spent in each stage (diagnostic launches only)
};

template <int SAT, bool SPLIT>
__global__ __launch_bounds__(256, 2) void q_front_k(const QFrontP P, const int rows_total, const int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const ring0 = smem;
    uint8_t *const ring1 = ring0 + F_NR0 * F_RB0;
    uint8_t *const opnd1 = ring1 + F_NR1 * F_RB1;
    uint8_t *const opnd2 = opnd1 + F_OP1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int wj = wave >> 1, wh = wave & 1;                       // (row of the pair, half of the row) / (row, plane) / channel-fragment pair
    const int g_begin = blockIdx.x * rows_per_block, g_end = min(rows_total, g_begin + rows_per_block);
    if (g_begin >= g_end) return;

    // ---- the wave's filters and constants, once
    const i4v wa = P.w0[lane], wb = P.w0[64 + lane];
    i4v wal = {0, 0, 0, 0}, wbl = wal;
    if constexpr (SPLIT) { wal = P.w0l[lane]; wbl = P.w0l[64 + lane]; }
    // first layer: fragment m's row 4g + r was packed with channel 8g + 4m + r: this lane holds channels 8 fq .. 8 fq + 7
    const i4v cb0a = *reinterpret_cast<const i4v *>(P.cb0 + 8 * fq), cb0b = *reinterpret_cast<const i4v *>(P.cb0 + 8 * fq + 4);
    // depthwise lane constants (as q_dwpw_k): which byte of the 16 is this lane's diagonal element; which window column its tap of k step ks is
    unsigned dmask[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dmask[d] = (fr >> 2) == d ? 0xffu << (8 * (fr & 3)) : 0u;
    int tap_dx[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) tap_dx[ks] = (min(4 * ks + fq, 8) % 3) * 16;
    const bool row_up0 = fq == 3, row_up1 = fq >= 2;               // the lane's tap of k step 0 / 1 lies in the later of the step's two rows
    auto build_a = [&](const unsigned w, i4v (&A)[3]) {               // (the hi parts stay in registers; a lo part, needed in few k steps of few planes, is built where it is used)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const unsigned rh = __builtin_amdgcn_perm(w, w, 0x01010101u * (unsigned)ks);
#pragma unroll
            for (int d = 0; d < 4; ++d) A[ks][d] = (int)(rh & dmask[d]);
        }
    };
    i4v A1h[3], A2h[3];
    const uint2 ab1 = P.dwa1[wh * 64 + lane], ab2 = P.dwa2[wave * 64 + lane];
    build_a(ab1.x, A1h);                                           // depthwise 1: plane wh
    build_a(ab2.x, A2h);                                           // depthwise 2: plane wave
    // Only a tensor's extreme weights overflow int8, so the lo part of a split filter is zero nearly everywhere: which k steps of the wave's
    // depthwise planes have one at all (netsq.pack_dw_mfma keeps the mask in byte 3 of the lo word), which of its first-layer and pointwise
    // fragments.  The MFMAs of the others are skipped (wave-uniform branches): a product with zeros, the same bits.
    const unsigned lom1 = SPLIT ? (unsigned)__builtin_amdgcn_readfirstlane((int)(ab1.y >> 24)) : 0u;
    const unsigned lom2 = SPLIT ? (unsigned)__builtin_amdgcn_readfirstlane((int)(ab2.y >> 24)) : 0u;
    auto any_nz = [&](const i4v v) { return __builtin_amdgcn_ballot_w64((v[0] | v[1] | v[2] | v[3]) != 0) != 0ull; };
    const i4v dcb1 = *reinterpret_cast<const i4v *>(P.dcb1 + 16 * wh + 4 * fq);
    const i4v dcb2 = *reinterpret_cast<const i4v *>(P.dcb2 + 16 * wave + 4 * fq);
    // pointwise 1: channel fragments 2 wh, 2 wh + 1 (fragment m's row 4g + r = channel 16 g + 4 m + r); pointwise 2: fragments 2 wave, 2 wave + 1
    i4v W1[2], cb1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { W1[q] = P.w1[(2 * wh + q) * 64 + lane]; cb1[q] = *reinterpret_cast<const i4v *>(P.cb1 + 16 * fq + 4 * (2 * wh + q)); }
    i4v W2[2], W2l[2], cb2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int mm = 2 * wave + q;                                // fragment 4 mg + m holds channels 64 mg + 16 g + 4 m + r
        W2[q] = P.w2[mm * 64 + lane];
        W2l[q] = i4v{0, 0, 0, 0};
        if constexpr (SPLIT) W2l[q] = P.w2l[mm * 64 + lane];
        cb2[q] = *reinterpret_cast<const i4v *>(P.cb2 + 64 * (mm >> 2) + 16 * fq + 4 * (mm & 3));
    }
    const bool nz2a = SPLIT && any_nz(W2l[0]), nz2b = SPLIT && any_nz(W2l[1]);
    const bool nz0a = SPLIT && any_nz(wal), nz0b = SPLIT && any_nz(wbl);
    const int M0 = P.R0.M, sh0 = P.R0.e - 1, lo0 = P.R0.lo, hi0 = P.R0.hi;
    const int Md1 = P.Rd1.M, shd1 = P.Rd1.e - 1, lod1 = P.Rd1.lo, hid1 = P.Rd1.hi;
    const int Mp1 = P.Rp1.M, shp1 = P.Rp1.e - 1, lop1 = P.Rp1.lo, hip1 = P.Rp1.hi;
    const int Md2 = P.Rd2.M, shd2 = P.Rd2.e - 1, lod2 = P.Rd2.lo, hid2 = P.Rd2.hi;
    const int Mp2 = P.Rp2.M, shp2 = P.Rp2.e - 1, lop2 = P.Rp2.lo, hip2 = P.Rp2.hi;
    // (the stored bytes are a - 128: the addends carry it, see q_requant_pack4s)
    const long long C0 = q_signed_c<SAT>(P.R0.C, sh0), Cd1 = q_signed_c<SAT>(P.Rd1.C, shd1), Cp1 = q_signed_c<SAT>(P.Rp1.C, shp1), Cd2 = q_signed_c<SAT>(P.Rd2.C, shd2), Cp2 = q_signed_c<SAT>(P.Rp2.C, shp2);
    const unsigned zin4 = (unsigned)P.in_zp * 0x01010101u;
    const u4v z0v = {(unsigned)P.zp0 * 0x01010101u, (unsigned)P.zp0 * 0x01010101u, (unsigned)P.zp0 * 0x01010101u, (unsigned)P.zp0 * 0x01010101u};
    const u4v z1v = {(unsigned)P.zp1 * 0x01010101u, (unsigned)P.zp1 * 0x01010101u, (unsigned)P.zp1 * 0x01010101u, (unsigned)P.zp1 * 0x01010101u};

    // rings: every byte the zero point (the border columns keep it: the stages write interiors only); operand tiles: zeros
    for (int i = tid * 16; i < F_NR0 * F_RB0; i += 256 * 16) *reinterpret_cast<u4v *>(ring0 + i) = z0v;
    for (int i = tid * 16; i < F_NR1 * F_RB1; i += 256 * 16) *reinterpret_cast<u4v *>(ring1 + i) = z1v;
    for (int i = tid * 16; i < F_OP1 + F_OP2; i += 256 * 16) *reinterpret_cast<u4v *>(opnd1 + i) = u4v{0, 0, 0, 0};
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(P.src), 0, (int)P.src_bytes, 0x00020000);
    // (a row's last fragment is partly filled: pixels 144 .. 149 of 150, 64 .. 74 of 75 -- the only lane-dependent store conditions of the kernel)
    const bool ok6 = wh == 0 || fr < 6;
    const int al_o = 2 * (fr & 1);                                  // byte offset of the lane's window inside its first dword (6 x mod 4)
    unsigned win[5][3];                                             // the lane's three dwords of filter row fq at its pixel of the wave's five fragments
#pragma unroll
    for (int i = 0; i < 5; ++i) { win[i][0] = 0; win[i][1] = 0; win[i][2] = 0; }

    unsigned long long st[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = P.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define F_STAMP(k) do { if (P.dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st[k] += now_ - tprev; tprev = now_; } } while (0)
    int g = g_begin;
    while (g < g_end) {
        const int n = g / F_S2, r_lo = g - n * F_S2, r_hi = min(F_S2 - 1, r_lo + (g_end - g) - 1);
        const int c_lo = 2 * r_lo - 1, c_hi = min(2 * r_hi + 3, F_S1);       // first-layer rows this segment needs (row -1 / 150: padding)
        const int y_lo = 2 * r_lo, y_hi = 2 * r_hi + 2;                       // block-1 rows (row 150: padding)

        // frame bytes of the first-layer row of tick t1 (range-checked buffer loads: a window past the end of the batch reads zeros)
        auto prefetch = [&](int t1) __attribute__((always_inline)) {
            const int c = 2 * t1 + 2 + wj;
            if (c < max(c_lo, 0) || c > min(c_hi, F_S1 - 1)) return;
            const int row = min(2 * c + min(fq, 2), F_S0 - 1);
            const int a = ((n * F_S0 + row) * F_S0 + 2 * (80 * wh + fr)) * 3;
            const int a4 = a & ~3;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int k = 0; k < 3; ++k) win[i][k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, a4 + 96 * i + 4 * k, 0, 0);
        };
        // ---- first layer: row c = 2t + 2 + wj, fragments 5 wh .. 5 wh + 4 -> ring0 (one body per combination of fragments with a lo part)
        auto conv0_t = [&](auto nza_tag, auto nzb_tag, int c, uint8_t *const dst) __attribute__((always_inline)) {
            constexpr bool NZA = decltype(nza_tag)::value, NZB = decltype(nzb_tag)::value;
            const bool below = 2 * c + fq >= F_S0 && fq < 3;        // filter row 2 of the last output row lies under the frame
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                unsigned d0 = __builtin_amdgcn_alignbyte(win[i][1], win[i][0], al_o);
                unsigned d1 = __builtin_amdgcn_alignbyte(win[i][2], win[i][1], al_o);
                unsigned d2 = __builtin_amdgcn_alignbyte(0u, win[i][2], al_o);
                if (below) { d0 = zin4; d1 = zin4; d2 = zin4; }
                const int x = 80 * wh + 16 * i + fr;
                if (i == 4) {                                       // (wh = 1: pixel 149's third column lies right of the frame: bytes 6 .. 8)
                    if (x == F_S1 - 1) { d1 = (d1 & 0x0000ffffu) | (zin4 & 0xffff0000u); d2 = (d2 & 0xffffff00u) | (zin4 & 0xffu); }
                }
                i4v b;
                b[0] = (int)(d0 ^ 0x80808080u); b[1] = (int)(d1 ^ 0x80808080u); b[2] = (int)(d2 ^ 0x80808080u); b[3] = 0;
                i4v acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wa, b, cb0a, 0, 0, 0);
                i4v acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wb, b, cb0b, 0, 0, 0);
                if constexpr (NZA) acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wal, b, acc0, 0, 0, 0);
                if constexpr (NZB) acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(wbl, b, acc1, 0, 0, 0);
                const unsigned lo = q_requant_pack4s<SAT>(acc0[0], acc0[1], acc0[2], acc0[3], M0, C0, C0, C0, C0, sh0, lo0, hi0);
                const unsigned hi = q_requant_pack4s<SAT>(acc1[0], acc1[1], acc1[2], acc1[3], M0, C0, C0, C0, C0, sh0, lo0, hi0);
                if (i < 4 || ok6) *reinterpret_cast<uint2 *>(dst + (fq >> 1) * F_PP + (x + 1) * 16 + (fq & 1) * 8) = make_uint2(lo, hi);
            }
        };
        auto conv0_stage = [&](int t) __attribute__((always_inline)) {
            const int c = 2 * t + 2 + wj;
            if (c < c_lo || c > c_hi) return;
            uint8_t *const dst = ring0 + ((c + 4) & 3) * F_RB0;
            if (c < 0 || c >= F_S1) {                              // a padding row: this wave's half of the slot
                for (int i = lane * 16; i < F_RB0 / 2; i += 1024) *reinterpret_cast<u4v *>(dst + wh * (F_RB0 / 2) + i) = z0v;
                return;
            }
            if (nz0a) { if (nz0b) conv0_t(std::true_type{}, std::true_type{}, c, dst); else conv0_t(std::true_type{}, std::false_type{}, c, dst); }
            else { if (nz0b) conv0_t(std::false_type{}, std::true_type{}, c, dst); else conv0_t(std::false_type{}, std::false_type{}, c, dst); }
        };
        // ---- depthwise 3x3 of one plane over NF fragments of one output row: ring rows at s0, s1, s2 (byte offsets of the plane in the three slots),
        //      pixel x of fragment f at column byte XS * (16 f + fr) + col0; the packed bytes go to tile + f * 256 (+ dup)
        auto dw_run_t = [&](auto lom_tag, const uint8_t *ring, int s0, int s1, int s2, int lane_col, int frag_pitch, const i4v (&Ah)[3], const unsigned al_w, const i4v cb,
                            int Md, long long Cd, int shd, int lod, int hid, uint8_t *tile, int dup_off) __attribute__((always_inline)) {
            constexpr int NF = 5;
            constexpr unsigned LOM = decltype(lom_tag)::value;
            const uint8_t *const a0 = ring + (row_up0 ? s1 : s0) + tap_dx[0] + lane_col;
            const uint8_t *const a1 = ring + (row_up1 ? s2 : s1) + tap_dx[1] + lane_col;
            const uint8_t *const a2 = ring + s2 + tap_dx[2] + lane_col;
            // the 15 (k step, fragment) operands as one sequence, DW_W of them in flight (a rolling window instead of a second buffer for all five
            // fragments of the next k step: 32 registers instead of 40 -- the kernel sits at the 256-register line)
            constexpr int DW_W = 8;
            i4v acc[NF], b[DW_W];
            auto opnd_at = [&](int j) { const int ks = j / NF, f = j - ks * NF; return *reinterpret_cast<const i4v *>((ks == 0 ? a0 : ks == 1 ? a1 : a2) + f * frag_pitch); };
#pragma unroll
            for (int j = 0; j < DW_W; ++j) b[j] = opnd_at(j);
#pragma unroll
            for (int j = 0; j < 3 * NF; ++j) {
                const int ks = j / NF, f = j - ks * NF;
                if (ks == 0) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah[0], b[j % DW_W], cb, 0, 0, 0);
                else acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah[ks], b[j % DW_W], acc[f], 0, 0, 0);
                if (SPLIT && ((LOM >> ks) & 1u)) {                  // (LOM is a literal, the loop unrolled: no branch is emitted)
                    const unsigned rl = __builtin_amdgcn_perm(al_w, al_w, 0x01010101u * (unsigned)ks);
                    i4v Al;
#pragma unroll
                    for (int d = 0; d < 4; ++d) Al[d] = (int)(rl & dmask[d]);
                    acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Al, b[j % DW_W], acc[f], 0, 0, 0);
                }
                if (j + DW_W < 3 * NF) b[j % DW_W] = opnd_at(j + DW_W);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const unsigned packed = q_requant_pack4s<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Md, Cd, Cd, Cd, Cd, shd, lod, hid);
                *reinterpret_cast<unsigned *>(tile + f * 256) = packed;
                if (dup_off) *reinterpret_cast<unsigned *>(tile + f * 256 + dup_off) = packed;
            }
        };
        // (one straight-line body per lo mask that occurs in practice: none, one k step, all -- a branch inside the k loop cost more than it saved)
        auto dw_run = [&](const unsigned lom, const uint8_t *ring, int s0, int s1, int s2, int lane_col, int frag_pitch, const i4v (&Ah)[3], const unsigned al_w, const i4v cb,
                          int Md, long long Cd, int shd, int lod, int hid, uint8_t *tile, int dup_off) __attribute__((always_inline)) {
            switch (lom) {
                case 0: dw_run_t(std::integral_constant<unsigned, 0u>{}, ring, s0, s1, s2, lane_col, frag_pitch, Ah, al_w, cb, Md, Cd, shd, lod, hid, tile, dup_off); break;
                case 1: dw_run_t(std::integral_constant<unsigned, 1u>{}, ring, s0, s1, s2, lane_col, frag_pitch, Ah, al_w, cb, Md, Cd, shd, lod, hid, tile, dup_off); break;
                case 2: dw_run_t(std::integral_constant<unsigned, 2u>{}, ring, s0, s1, s2, lane_col, frag_pitch, Ah, al_w, cb, Md, Cd, shd, lod, hid, tile, dup_off); break;
                case 4: dw_run_t(std::integral_constant<unsigned, 4u>{}, ring, s0, s1, s2, lane_col, frag_pitch, Ah, al_w, cb, Md, Cd, shd, lod, hid, tile, dup_off); break;
                default: dw_run_t(std::integral_constant<unsigned, 7u>{}, ring, s0, s1, s2, lane_col, frag_pitch, Ah, al_w, cb, Md, Cd, shd, lod, hid, tile, dup_off); break;
            }
        };
        // ---- depthwise 1: block-1 row y = 2t + 1 + wj, plane wh, ten fragments: ring0 -> opnd1 (32 channels fill half of the 64-byte k slice: with
        //      the split filter the bytes go to BOTH halves, whose filter halves are the hi and lo parts -- see q_dwpw_k's P.dup)
        auto dw1_stage = [&](int t) __attribute__((always_inline)) {
            const int y = 2 * t + 1 + wj;
            if (y < y_lo || y > min(y_hi, F_S1 - 1)) return;
            const int s0 = ((y + 3) & 3) * F_RB0 + wh * F_PP, s1 = ((y + 4) & 3) * F_RB0 + wh * F_PP, s2 = ((y + 5) & 3) * F_RB0 + wh * F_PP;   // first-layer rows y - 1 .. y + 1
            uint8_t *const tile = opnd1 + ((wh * 320 + wj * 160 + fr) * 16 + 4 * fq);
#pragma unroll
            for (int h = 0; h < 2; ++h)
                dw_run(lom1, ring0 + h * 5 * 256, s0, s1, s2, fr * 16, 256, A1h, ab1.y, dcb1, Md1, Cd1, shd1, lod1, hid1, tile + h * 5 * 256, SPLIT ? 2 * 320 * 16 : 0);
        };
        // ---- depthwise 2 (stride 2): block-2 row r = t - 1, plane wave, five fragments: ring1 -> opnd2
        auto dw2_stage = [&](int t) __attribute__((always_inline)) {
            const int r = t - 1;
            if (r < r_lo || r > r_hi) return;
            const int y = 2 * r;                                    // block-1 rows y .. y + 2, columns 2 x .. 2 x + 2 (bordered: + 1)
            const int s0 = (y % 3) * F_RB1 + wave * F_PP, s1 = ((y + 1) % 3) * F_RB1 + wave * F_PP, s2 = ((y + 2) % 3) * F_RB1 + wave * F_PP;
            uint8_t *const tile = opnd2 + ((wave * 80 + fr) * 16 + 4 * fq);
            dw_run(lom2, ring1, s0, s1, s2, fr * 32 + 16, 512, A2h, ab2.y, dcb2, Md2, Cd2, shd2, lod2, hid2, tile, 0);
        };
        // ---- pointwise 1: block-1 row y = 2t - 1 + wj, ten fragments, channel fragments 2 wh, 2 wh + 1 (bytes 8 wh .. 8 wh + 7 of a pixel's plane slot): opnd1 -> ring1
        auto pw1_stage = [&](int t) __attribute__((always_inline)) {
            const int y = 2 * t - 1 + wj;
            if (y < y_lo || y > y_hi) return;
            uint8_t *const dst = ring1 + (y % 3) * F_RB1;
            if (y >= F_S1) {                                        // the padding row under the frame
                for (int i = lane * 16; i < F_RB1 / 2; i += 1024) *reinterpret_cast<u4v *>(dst + wh * (F_RB1 / 2) + i) = z1v;
                return;
            }
            const uint8_t *const bp = opnd1 + (fq * 320 + wj * 160 + fr) * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                i4v b[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) b[i] = *reinterpret_cast<const i4v *>(bp + (5 * h + i) * 256);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const i4v a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W1[0], b[i], cb1[0], 0, 0, 0);
                    const i4v a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W1[1], b[i], cb1[1], 0, 0, 0);
                    const unsigned o0 = q_requant_pack4s<SAT>(a0[0], a0[1], a0[2], a0[3], Mp1, Cp1, Cp1, Cp1, Cp1, shp1, lop1, hip1);
                    const unsigned o1 = q_requant_pack4s<SAT>(a1[0], a1[1], a1[2], a1[3], Mp1, Cp1, Cp1, Cp1, Cp1, shp1, lop1, hip1);
                    const int x = 16 * (5 * h + i) + fr;
                    if (5 * h + i < 9 || fr < 6) *reinterpret_cast<uint2 *>(dst + fq * F_PP + (x + 1) * 16 + 8 * wh) = make_uint2(o0, o1);
                }
            }
        };
        // ---- pointwise 2: block-2 row r = t - 2, channel fragments 2 wave, 2 wave + 1, five fragments: opnd2 -> HBM
        auto pw2_t = [&](auto nza_tag, auto nzb_tag, int r) __attribute__((always_inline)) {
            constexpr bool NZA = decltype(nza_tag)::value, NZB = decltype(nzb_tag)::value;
            const uint8_t *const bp = opnd2 + (fq * 80 + fr) * 16;
            i4v b[5];
#pragma unroll
            for (int f = 0; f < 5; ++f) b[f] = *reinterpret_cast<const i4v *>(bp + f * 256);
            uint8_t *const dst = P.out + ((size_t)((n * (F_S2 + 2) + r + 1) * 8 + 4 * (wave >> 1) + fq) * F_PPO + 8 * (wave & 1));
#pragma unroll
            for (int f = 0; f < 5; ++f) {
                i4v a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W2[0], b[f], cb2[0], 0, 0, 0);
                i4v a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W2[1], b[f], cb2[1], 0, 0, 0);
                if constexpr (NZA) a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W2l[0], b[f], a0, 0, 0, 0);
                if constexpr (NZB) a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(W2l[1], b[f], a1, 0, 0, 0);
                const unsigned o0 = q_requant_pack4s<SAT>(a0[0], a0[1], a0[2], a0[3], Mp2, Cp2, Cp2, Cp2, Cp2, shp2, lop2, hip2);
                const unsigned o1 = q_requant_pack4s<SAT>(a1[0], a1[1], a1[2], a1[3], Mp2, Cp2, Cp2, Cp2, Cp2, shp2, lop2, hip2);
                const int x = 16 * f + fr;
                if (f < 4 || fr < 11) *reinterpret_cast<uint2 *>(dst + (x + 1) * 16) = make_uint2(o0, o1);
            }
        };
        auto pw2_stage = [&](int t) __attribute__((always_inline)) {
            const int r = t - 2;
            if (r < r_lo || r > r_hi) return;
            if (nz2a) { if (nz2b) pw2_t(std::true_type{}, std::true_type{}, r); else pw2_t(std::true_type{}, std::false_type{}, r); }
            else { if (nz2b) pw2_t(std::false_type{}, std::true_type{}, r); else pw2_t(std::false_type{}, std::false_type{}, r); }
        };

        prefetch(r_lo - 2);
        for (int t = r_lo - 2; t <= r_hi + 2; ++t) {
            conv0_stage(t);
            F_STAMP(0);
            pw1_stage(t);
            F_STAMP(1);
            pw2_stage(t);
            F_STAMP(2);
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): this wave's ring and tile traffic is done
            __builtin_amdgcn_s_barrier();                            // (bare: the output stores and nothing else stay in flight)
            F_STAMP(3);
            prefetch(t + 1);
            dw1_stage(t);
            F_STAMP(4);
            dw2_stage(t);
            F_STAMP(5);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
            F_STAMP(6);
        }
        g += r_hi - r_lo + 1;
    }
#undef F_STAMP
    if (P.dbg && lane == 0) for (int k = 0; k < 7; ++k) P.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + k] = st[k];
}

}  // namespace

// The three ops at the head of a uint8 SSD-MobileNet-v1 program (o0 = first layer, o1 / o2 = blocks 1 / 2) as one launch.  *ran = 0: the
// program, the geometry or the quantisation parameters are not what the row pipeline is built for -- the caller runs the three launches.
int netq_run_front(dd_net *net, const int32_t *o0, const int32_t *o1, const int32_t *o2, const uint8_t *input, int nimg, hipStream_t s, int *ran) {
    *ran = 0;
    if (o0[0] != OP_QCONV0 || o1[0] != OP_QDWPW || o2[0] != OP_QDWPW || o1[1] != o0[2] || o2[1] != o1[2]) return DD_OK;
    const TensorDesc &t0 = net->tensors[o0[2]], &t1 = net->tensors[o1[2]], &t2 = net->tensors[o2[2]];
    if (net->in_h != F_S0 || net->in_w != F_S0 || t0.h != F_S1 || t0.w != F_S1 || t0.cs != 32 || t1.h != F_S1 || t1.w != F_S1 || t1.cs != 64 ||
        t2.h != F_S2 || t2.w != F_S2 || t2.cs != 128 || !t0.pad || !t1.pad || !t2.pad) return DD_OK;
    // first layer 3x3 stride 2 without padding above / left; block 1 stride 1 padded by one; block 2 stride 2 without padding above / left
    if (o0[7] != 2 || o0[8] != 0 || o0[9] != 0 || o1[7] != 1 || o1[8] != 1 || o1[9] != 1 || o1[10] != 32 || o1[11] != 64 ||
        o2[7] != 2 || o2[8] != 0 || o2[9] != 0 || o2[10] != 64 || o2[11] != 128) return DD_OK;
    // filters: all three split into hi + lo parts (no row sums), or none (weight zero points of 128: no row sums either)
    const bool split = o0[18] != 0;
    if ((o1[47] != 0) != split || (o2[18] != 0) != split || o0[38] != 0 || o1[38] != 0 || o2[38] != 0 || (!split && o1[18] != 0)) return DD_OK;
    if (!o0[17] || !o1[17] || !o1[21] || !o2[17] || !o2[21] || !o1[20] || !o2[20]) return DD_OK;
    char *W = net->d_weights;
    QFrontP P;
    memset(&P, 0, sizeof(P));
    P.src = input; P.src_bytes = (long long)nimg * F_S0 * F_S0 * 3;
    P.out = static_cast<uint8_t *>(net->bufs[t2.buf]);
    auto blob = [&](int32_t off) { return W + (size_t)(uint32_t)off; };
    P.w0 = reinterpret_cast<const i4v *>(blob(o0[16])); P.w0l = split ? reinterpret_cast<const i4v *>(blob(o0[18])) : nullptr;
    P.cb0 = reinterpret_cast<const int *>(blob(o0[17])); P.in_zp = o0[39]; P.R0 = make_req(o0); P.zp0 = (P.R0.zo ^ 0x80) & 0xff;
    auto dwreq = [&](const int32_t *o) { int32_t d[48] = {0}; d[32] = o[22]; d[33] = o[23]; d[36] = o[24]; d[37] = o[25]; d[40] = o[28]; return make_req(d); };
    P.dwa1 = reinterpret_cast<const uint2 *>(blob(o1[20])); P.dcb1 = reinterpret_cast<const int *>(blob(o1[21]));
    P.w1 = reinterpret_cast<const i4v *>(blob(o1[16])); P.cb1 = reinterpret_cast<const int *>(blob(o1[17]));
    P.Rd1 = dwreq(o1); P.Rp1 = make_req(o1); P.zp1 = (P.Rp1.zo ^ 0x80) & 0xff;
    P.dwa2 = reinterpret_cast<const uint2 *>(blob(o2[20])); P.dcb2 = reinterpret_cast<const int *>(blob(o2[21]));
    P.w2 = reinterpret_cast<const i4v *>(blob(o2[16])); P.w2l = split ? reinterpret_cast<const i4v *>(blob(o2[18])) : nullptr;
    P.cb2 = reinterpret_cast<const int *>(blob(o2[17]));
    P.Rd2 = dwreq(o2); P.Rp2 = make_req(o2);
    const QReq *R[5] = {&P.R0, &P.Rd1, &P.Rp1, &P.Rd2, &P.Rp2};
    bool byte_clamp = true, small_shift = true;
    for (const QReq *r : R) {
        if (r->linear || r->e < 1) return DD_OK;
        byte_clamp = byte_clamp && r->lo == 0 && r->hi == 255;
        small_shift = small_shift && r->e <= 8;
    }
    if ((reinterpret_cast<uintptr_t>(input) & 3) != 0 || P.src_bytes >= (1ll << 31) - 16) return DD_OK;
    const int sat = byte_clamp ? (small_shift ? 2 : 1) : 0;
    void (*kern)(const QFrontP, const int, const int) =
        split ? (sat == 2 ? &q_front_k<2, true> : sat == 1 ? &q_front_k<1, true> : &q_front_k<0, true>)
              : (sat == 2 ? &q_front_k<2, false> : sat == 1 ? &q_front_k<1, false> : &q_front_k<0, false>);
    static DevOnce once;
    const int rc = once.run(net->ctx->device, [&]() -> int {
        for (const void *f : {reinterpret_cast<const void *>(&q_front_k<2, true>), reinterpret_cast<const void *>(&q_front_k<1, true>), reinterpret_cast<const void *>(&q_front_k<0, true>),
                              reinterpret_cast<const void *>(&q_front_k<2, false>), reinterpret_cast<const void *>(&q_front_k<1, false>), reinterpret_cast<const void *>(&q_front_k<0, false>)})
            DD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    // a segment costs four ticks beyond its rows: ranges of at least a fifth of a frame, two workgroups per CU
    const int n_cu = dd_cu_count(net->ctx->device);
    const int rows_total = nimg * F_S2;
    const int blocks = std::max(1, std::min(2 * n_cu, rows_total / 15));
    const int rpb = dd_ceil_div(rows_total, blocks);
    const unsigned grid = (unsigned)dd_ceil_div(rows_total, rpb);
    static const bool stamps = getenv("DD_Q_STAMPS") && atoi(getenv("DD_Q_STAMPS")) != 0;
    const size_t n_st = (size_t)grid * 4 * 8;
    if (stamps) { DD_HIP(hipMalloc(&P.dbg, n_st * 8)); DD_HIP(hipMemsetAsync(P.dbg, 0, n_st * 8, s)); }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), F_LDS, s, P, rows_total, rpb);
    DD_LAUNCH_CHECK();
    if (stamps) {                                                   // diagnostic: where the waves of this launch spent their cycles
        std::vector<unsigned long long> h(n_st);
        DD_HIP(hipStreamSynchronize(s));
        DD_HIP(hipMemcpy(h.data(), P.dbg, n_st * 8, hipMemcpyDeviceToHost));
        DD_HIP(hipFree(P.dbg));
        double sum[7] = {0, 0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < n_st / 8; ++w) for (int k = 0; k < 7; ++k) sum[k] += (double)h[w * 8 + k];
        const double nw = (double)(n_st / 8) * rpb;
        fprintf(stderr, "q_front_k %u blocks of %d rows: cycles per wave and output row: first layer %.0f  pointwise 1 %.0f  pointwise 2 %.0f  barrier %.0f | depthwise 1 %.0f  depthwise 2 %.0f  barrier %.0f\n",
                grid, rpb, sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw, sum[5] / nw, sum[6] / nw);
    }
    *ran = 1;
    return DD_OK;
}
