// RECORDED EXPERIMENT, not part of the library (round 6): bit-identical to q_dwpw_k<512,512> and exactly as fast -- 2 560 / 2 553 / 2 551 us
// against 2 556 / 2 537 / 2 549 us for the SSD forward of 768 frames, same box (profiles/r06_ab_requant.txt).  It was built into the library as
// csrc/netsq_b512.hip with `int netq_run_b512(const NetqB512 &, int nimg, hipStream_t, int device, int *ran)` declared in csrc/net_priv.h and called
// from csrc/netsq.hip's OP_QDWPW case for cin = cout = 512, stride 1.  What it shows: with one barrier per tile, no barrier around the row ring and
// the two stages of a wave on different tiles, a tile still costs 14-15 k cycles -- the block is at its instruction sum, not waiting for its barriers.
//
// uint8 SSD-MobileNet-v1, the 512 -> 512 MobileNet blocks (7-11: depthwise 3x3 stride 1 + pointwise at 19 x 19): q_dwpw_k's arithmetic and packed
// operands (csrc/netsq.hip), same bits, ONE barrier per 64-pixel tile instead of two.
//
// q_dwpw_k's tile is depthwise stage | barrier | pointwise stage | barrier: the eight waves of the workgroup (two per SIMD -- the wave's 64 output
// channels x K = 512 of filter are 128 of its 256 registers) move in lockstep, both waves of a SIMD read operands, then both multiply, then both
// requantise, and the stage stamps showed the depthwise stage at 60 % of its instruction sum with 1.7 k of a tile's 14.2 k cycles in the barriers.
// Here the two stages of a wave belong to DIFFERENT tiles -- in step t a wave runs the depthwise stage of tile t + 1 and the pointwise stage of tile
// t -- so the waves of a SIMD drift apart and fill each other's gaps, as the two workgroups per CU of q_front_k do:
//   * the pointwise operand tile [32 planes][64 pixels][16] is double-buffered (2 x 32 KB);
//   * the input rows live in a PRIVATE ring per wave: a wave's depthwise stage reads only its own four planes (1 344 B of a 10 752-B row at
//     19 x 19), so it requests them itself (LDS-DMA, two instructions per row) when its own depthwise stage has left the slots they replace,
//     waits for them with a counted vmcnt before its next depthwise stage, and no barrier guards the ring; rows of tile t + 2 land behind the
//     pointwise stage of tile t;
//   * row sums and per-pixel geometry are triple-buffered (written a step ahead of their first reader, read for two steps).
// LDS: 2 x 32 KB + 8 waves x NR rows x 1 344 B (NR = 7: 73.5 KB) + 10 KB of tables = 147 KB, one workgroup per CU.
// The replaced interface is the middle of `interpreter.invoke()` (tools/ssd_mobilenet.py:100-109 upstream).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include "common.h"
#include "net_priv.h"
#include "netsq_dev.h"

namespace {

constexpr int B_C16 = 32, B_KC = 8, B_NW = 8, B_NT = 512, B_QT = 64, B_MW = 4, B_CPW = 4;
constexpr int B_OPND = B_C16 * B_QT * 16;                         // one operand tile (bytes)

struct QB512P {
    const uint8_t *in; int H, W;
    int off_y, off_x, ho, wo, hw, tiles_per_frame;
    uint8_t *out;
    const uint2 *dw_a;               // [32 planes][64 lanes]: .x bytes 0..2 = hi parts of the lane's tap in k steps 0..2, .y = lo parts + the plane's lo mask (byte 3)
    const long long *dw_cq;          // [512]: the depthwise requantisation's 64-bit addend per channel
    const i4v *w;                    // [8 groups of 64 channels][4][8 k slices][64 lanes]
    const int *cbias;                // [512]
    int zwc, NR;
    unsigned wo_magic, nr_magic, tpf_magic;
    unsigned long long *dbg;         // DD_Q_STAMPS=1
    QReq Rd, Rp;
};

template <int SAT, bool ROWSUM>
__global__ __launch_bounds__(B_NT, 2) void q_b512_k(const QB512P P, const int n_tiles, const int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int PP = (P.W + 2) * 16, RB = B_C16 * PP, RBW = B_CPW * PP;      // plane pitch; row pitch in HBM; row pitch of a wave's ring (its four planes)
    uint8_t *const opnd = smem;                                       // [2][32][64][16]
    int *const rowsum = reinterpret_cast<int *>(opnd + 2 * B_OPND);   // [3][64]
    i4v *const pinfo = reinterpret_cast<i4v *>(rowsum + 3 * B_QT);    // [3][64]: ring offsets of the pixel's window (rows 0..2, first column; wave-independent), offset of its output
    int *const cbl = reinterpret_cast<int *>(pinfo + 3 * B_QT);       // [512]: the pointwise layer's per-channel constants
    long long *const dwq_l = reinterpret_cast<long long *>(cbl + 512);   // [512]: the depthwise addends
    uint8_t *const ringw = reinterpret_cast<uint8_t *>(dwq_l + 512) + (size_t)wave * P.NR * RBW;     // this wave's ring: [NR][4 planes][W + 2][16]

    // the wave's pointwise filter (channels 64 wave .. 64 wave + 63; fragment m's row 4g + r = channel 64 wave + 16 g + 4 m + r), once
    i4v Wr[B_MW][B_KC];
#pragma unroll
    for (int m = 0; m < B_MW; ++m)
#pragma unroll
        for (int kc = 0; kc < B_KC; ++kc) Wr[m][kc] = P.w[((size_t)(wave * B_MW + m) * B_KC + kc) * 64 + lane];
    for (int i = tid; i < 512; i += B_NT) { cbl[i] = P.cbias[i]; dwq_l[i] = P.dw_cq[i]; }
    for (int i = tid; i < 3 * B_QT; i += B_NT) rowsum[i] = 0;
    unsigned dmask[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dmask[d] = (fr >> 2) == d ? 0xffu << (8 * (fr & 3)) : 0u;
    int tap_dx[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) tap_dx[ks] = (min(4 * ks + fq, 8) % 3) * 16;
    const bool row_up0 = fq == 3, row_up1 = fq >= 2;
    const uint2 ab_first = P.dw_a[(wave * B_CPW) * 64 + lane];      // the table word of the wave's first plane (the others come from L2 a plane ahead)

    const int t_begin = blockIdx.x * tiles_per_block, t_end = min(n_tiles, t_begin + tiles_per_block);
    if (t_begin >= t_end) return;
    const int Md = P.Rd.M, shd = P.Rd.e - 1, Mp = P.Rp.M, shp = P.Rp.e - 1;
    const long long Cp = P.Rp.C;
    const int lod = P.Rd.lo, hid = P.Rd.hi, lop = P.Rp.lo, hip_ = P.Rp.hi;
    const int out_row = B_C16 * ((P.wo + 2) * 16);                 // bytes of one bordered output row (all planes)

    auto tile_rows = [&](int t, int &n, int &q0, int &q1, int &ga, int &gb) {
        n = (int)__umulhi((unsigned)t, P.tpf_magic);
        q0 = (t - n * P.tiles_per_frame) * B_QT;
        q1 = min(q0 + B_QT, P.hw) - 1;
        const int y0 = (int)__umulhi((unsigned)q0, P.wo_magic), y1 = (int)__umulhi((unsigned)q1, P.wo_magic);
        ga = n * (P.H + 2) + y0 + P.off_y;
        gb = n * (P.H + 2) + y1 + P.off_y + 2;
    };
    // threads 0 .. 63: where pixel tid of tile t reads (offsets inside a wave's ring) and writes
    auto geometry = [&](int t, int buf) {
        int n, q0, q1, ga, gb;
        tile_rows(t, n, q0, q1, ga, gb);
        const int q = q0 + tid;
        const int qc = min(q, q1);
        const int y = (int)__umulhi((unsigned)qc, P.wo_magic), x = qc - __mul24(y, P.wo);
        const int y0 = (int)__umulhi((unsigned)q0, P.wo_magic);
        const int sa = ga - (int)__umulhi((unsigned)ga, P.nr_magic) * P.NR;
        int s0 = sa + (y - y0);
        s0 = s0 >= P.NR ? s0 - P.NR : s0;
        const int s1 = s0 + 1 == P.NR ? 0 : s0 + 1, s2 = s1 + 1 == P.NR ? 0 : s1 + 1;
        const int col = (x + P.off_x) * 16;
        const unsigned po = q <= q1 ? (unsigned)__mul24(n * (P.ho + 2) + y + 1, out_row) + (unsigned)(x + 1) * 16u : 0xffffffffu;
        pinfo[buf * B_QT + tid] = i4v{__mul24(s0, RBW) + col, __mul24(s1, RBW) + col, __mul24(s2, RBW) + col, (int)po};
    };
    // The wave's four planes of rows [lo, hi] from HBM straight into its ring (LDS-DMA; assembly, not the builtin: hipcc would make every later
    // ds_read wait for an LDS-DMA in flight, and the requests are to fly through the pointwise stage -- see q_dwpw_k).  RBW bytes of a row lie
    // together in HBM and in the slot: one instruction moves 1 KB lane-linearly, the second the rest (its upper lanes masked).
    auto glds16 = [&](const uint8_t *g, const uint8_t *l) {
        unsigned keep;
        const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)l;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    };
    auto request_rows = [&](int lo, int hi) {
        for (int row = lo; row <= hi; ++row) {
            const int slot = row - (int)__umulhi((unsigned)row, P.nr_magic) * P.NR;
            const uint8_t *src = P.in + (size_t)row * RB + (size_t)wave * RBW + lane * 16;
            uint8_t *dst = ringw + slot * RBW;
            for (int cb = 0; cb < RBW; cb += 1024)
                if (cb + lane * 16 < RBW) glds16(src + cb, dst + cb);
        }
    };

    // ---- depthwise stage of tile (geometry buffer gbuf): the wave's four planes x the four pixel fragments, as ONE stream of (plane, k step,
    //      fragment) operands with a rolling window of DW_W in flight (q_dwpw_k's form); bytes into operand tile obuf, row sums into slot rsb
    auto dw_tile = [&](int gbuf, int obuf, int rsb) __attribute__((always_inline)) {
        int rs[4] = {0, 0, 0, 0};
        int tapoff[3][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const i4v pi = pinfo[gbuf * B_QT + 16 * f + fr];
            tapoff[0][f] = (row_up0 ? pi[1] : pi[0]) + tap_dx[0];
            tapoff[1][f] = (row_up1 ? pi[2] : pi[1]) + tap_dx[1];
            tapoff[2][f] = pi[2] + tap_dx[2];
        }
        uint8_t *const ob = opnd + obuf * B_OPND;
        constexpr int DW_W = 6;
        auto opnd_at = [&](int pofs, int j) __attribute__((always_inline)) {
            return *reinterpret_cast<const i4v *>(ringw + tapoff[j / 4][j % 4] + pofs);
        };
        i4v b[DW_W];
        int cg = wave * B_CPW;
        uint2 ab = ab_first;
#pragma unroll
        for (int j = 0; j < DW_W; ++j) b[j] = opnd_at(0, j);
#pragma unroll 1
        for (int ci = 0; ci < B_CPW; ++ci) {
            const int pofs = ci * PP;
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
            // the next plane's head (the last round re-requests its own: nobody reads them); its table word comes from L2 a plane ahead
            const int cin = min(ci + 1, B_CPW - 1);
            const uint2 abn = P.dw_a[(wave * B_CPW + cin) * 64 + lane];
            long long Cq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) Cq[r] = dwq_l[cg * 16 + 4 * fq + r];
            const unsigned lom = (unsigned)__builtin_amdgcn_readfirstlane((int)(ab.y >> 24));
            if (lom) {                                               // (few planes: only a tensor's extreme weights overflow int8)
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {
                    if (lom & (1u << ks)) {
                        const unsigned rl = __builtin_amdgcn_perm(ab.y, ab.y, 0x01010101u * (unsigned)ks);
                        i4v Al;
#pragma unroll
                        for (int d = 0; d < 4; ++d) Al[d] = (int)(rl & dmask[d]);
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Al, *reinterpret_cast<const i4v *>(ringw + tapoff[ks][f] + pofs), acc[f], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < DW_W; ++j) b[j] = opnd_at(cin * PP, j);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                unsigned packed = q_requant_pack4<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Md, Cq[0], Cq[1], Cq[2], Cq[3], shd, lod, hid);
                packed ^= 0x80808080u;
                if (ROWSUM) rs[f] = sdot4((int)packed, 0x01010101, rs[f]);
                *reinterpret_cast<unsigned *>(ob + ((size_t)cg * B_QT + 16 * f + fr) * 16 + 4 * fq) = packed;
            }
            cg += 1;
            ab = abn;
        }
        if (ROWSUM) {
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                int v = rs[f];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (fq == 0) atomicAdd(&rowsum[rsb * B_QT + 16 * f + fr], v);
            }
        }
    };
    // ---- pointwise stage of the wave's 64 channels over the nf fragments of tile (gbuf, obuf, rsb); returns the stores it issued
    auto pw_tile = [&](int nf, int gbuf, int obuf, int rsb) __attribute__((always_inline)) {
        const uint8_t *const ob = opnd + obuf * B_OPND;
        constexpr int KB = 4;                                         // K slices requested at a time
        i4v b[KB];
        {
            const uint8_t *bp = ob + ((size_t)fq * B_QT + fr) * 16;
#pragma unroll
            for (int kc = 0; kc < KB; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bp + (size_t)kc * 4 * B_QT * 16);
        }
        for (int f = 0; f < nf; ++f) {
            i4v acc[B_MW];
#pragma unroll
            for (int m = 0; m < B_MW; ++m) acc[m] = *reinterpret_cast<const i4v *>(cbl + 64 * wave + 16 * fq + 4 * m);
            const uint8_t *bp = ob + ((size_t)fq * B_QT + 16 * f + fr) * 16;
            const uint8_t *bn = ob + ((size_t)fq * B_QT + 16 * min(f + 1, nf - 1) + fr) * 16;      // (the last fragment re-requests its own slices: nobody reads them)
            // the eight K slices through a window of KB registers sets: slice kc + KB is requested as soon as the MFMAs of slice kc are issued (and
            // the next fragment's first slices behind the last ones), so a request has the MFMAs of KB - 1 slices to land in -- requested block by
            // block, every fragment waited out two LDS round trips
#pragma unroll
            for (int kc = 0; kc < B_KC; ++kc) {
#pragma unroll
                for (int m = 0; m < B_MW; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wr[m][kc], b[kc % KB], acc[m], 0, 0, 0);
                if (kc + KB < B_KC) b[kc % KB] = *reinterpret_cast<const i4v *>(bp + (size_t)(kc + KB) * 4 * B_QT * 16);
                else b[kc % KB] = *reinterpret_cast<const i4v *>(bn + (size_t)(kc + KB - B_KC) * 4 * B_QT * 16);
            }
            const int rsv = ROWSUM ? rowsum[rsb * B_QT + 16 * f + fr] * P.zwc : 0;
            const unsigned po = (unsigned)pinfo[gbuf * B_QT + 16 * f + fr][3];
            unsigned o[B_MW];
#pragma unroll
            const long long Cr = Cp + (long long)rsv * Mp;          // (x + rsv) M + C = x M + (C + rsv M): one 64-bit multiply-add per pixel instead of sixteen adds
#pragma unroll
            for (int m = 0; m < B_MW; ++m)
                o[m] = 0x80808080u ^ q_requant_pack4<SAT>(acc[m][0], acc[m][1], acc[m][2], acc[m][3], Mp, Cr, Cr, Cr, Cr, shp, lop, hip_);
            if (po != 0xffffffffu) *reinterpret_cast<u4v *>(P.out + po + (size_t)(4 * wave + fq) * ((P.wo + 2) * 16)) = u4v{o[0], o[1], o[2], o[3]};
        }
    };
    auto wait_vm = [&](int ns) {                                     // this wave's row requests have landed: only the ns stores issued after them may be outstanding
        if (ns >= 4) __builtin_amdgcn_s_waitcnt(0x0f74);
        else if (ns == 3) __builtin_amdgcn_s_waitcnt(0x0f73);
        else if (ns == 2) __builtin_amdgcn_s_waitcnt(0x0f72);
        else if (ns == 1) __builtin_amdgcn_s_waitcnt(0x0f71);
        else __builtin_amdgcn_s_waitcnt(0x0f70);
    };

    unsigned long long st[4] = {0, 0, 0, 0}, tprev = P.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
#define B_STAMP(k) do { if (P.dbg) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st[k] += now_ - tprev; tprev = now_; } } while (0)
    // prologue: the first tile's rows, the first two tiles' geometry, the first tile's depthwise stage
    int n, q0, q1, ga, gb;
    tile_rows(t_begin, n, q0, q1, ga, gb);
    request_rows(ga, gb);
    int loaded_hi = gb;
    if (tid < B_QT) { geometry(t_begin, t_begin % 3); if (t_begin + 1 < t_end) geometry(t_begin + 1, (t_begin + 1) % 3); }
    __builtin_amdgcn_s_waitcnt(0x0f70);                              // vmcnt(0): the first tile's rows are in (the requests are assembly: hipcc does not count them)
    __syncthreads();
    dw_tile(t_begin % 3, t_begin & 1, t_begin % 3);
    if (t_begin + 1 < t_end) {
        int n2, q02, q12, ga2, gb2;
        tile_rows(t_begin + 1, n2, q02, q12, ga2, gb2);
        request_rows(max(loaded_hi + 1, ga2), gb2);
        loaded_hi = max(loaded_hi, gb2);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    int ns_prev = 0;                                                  // stores issued after this wave's last row requests
    for (int t = t_begin; t < t_end; ++t) {
        const int b3 = t % 3, b3n = b3 == 2 ? 0 : b3 + 1, b3nn = b3n == 2 ? 0 : b3n + 1;
        if (t + 1 < t_end) {
            wait_vm(ns_prev);                                         // the rows of tile t + 1 are in this wave's ring
            dw_tile(b3n, (t + 1) & 1, b3n);
            if (t + 2 < t_end) {                                      // its own depthwise stage has left the slots the rows of tile t + 2 replace
                int n2, q02, q12, ga2, gb2;
                tile_rows(t + 2, n2, q02, q12, ga2, gb2);
                request_rows(max(loaded_hi + 1, ga2), gb2);
                loaded_hi = max(loaded_hi, gb2);
            }
        }
        B_STAMP(0);
        if (tid < B_QT) {
            if (ROWSUM) rowsum[b3nn * B_QT + tid] = 0;
            if (t + 2 < t_end) geometry(t + 2, b3nn);
        }
        const int nf = (q1 - q0) / 16 + 1;
        pw_tile(nf, b3, t & 1, b3);
        ns_prev = nf;
        B_STAMP(1);
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): this wave's LDS traffic of the step is done
        __builtin_amdgcn_s_barrier();                                // tile t + 1's operand tile and row sums are complete; tile t's are free
        B_STAMP(2);
        if (t + 1 < t_end) { int n1, ga1, gb1; tile_rows(t + 1, n1, q0, q1, ga1, gb1); }
    }
#undef B_STAMP
    if (P.dbg && lane == 0) for (int k = 0; k < 3; ++k) P.dbg[((size_t)blockIdx.x * B_NW + wave) * 4 + k] = st[k];
}

}  // namespace

int netq_run_b512(const NetqB512 &a, int nimg, hipStream_t s, int device, int *ran) {
    *ran = 0;
    static const int on = getenv("DD_Q_B512") ? atoi(getenv("DD_Q_B512")) : 1;
    if (!on || a.stride != 1 || a.cin != 512 || a.cout != 512) return DD_OK;
    QB512P P;
    memset(&P, 0, sizeof(P));
    P.in = a.in; P.H = a.H; P.W = a.W; P.off_y = a.off_y; P.off_x = a.off_x; P.ho = a.ho; P.wo = a.wo; P.hw = a.ho * a.wo;
    P.out = a.out; P.dw_a = static_cast<const uint2 *>(a.dw_a); P.dw_cq = static_cast<const long long *>(a.dw_cq);
    P.w = static_cast<const i4v *>(a.w); P.cbias = static_cast<const int *>(a.cbias); P.zwc = a.zwc;
    auto req = [](int M, int e, long long C, int lo, int hi) { QReq R; R.M = M; R.e = e; R.C = C; R.zo = 0; R.lo = lo; R.hi = hi; R.linear = 0; return R; };
    P.Rd = req(a.rd_M, a.rd_e, a.rd_C, a.rd_lo, a.rd_hi);
    P.Rp = req(a.rp_M, a.rp_e, a.rp_C, a.rp_lo, a.rp_hi);
    if (a.split_pw || P.Rd.e < 1 || P.Rp.e < 1 || P.hw < 1) return DD_OK;
    P.tiles_per_frame = dd_ceil_div(P.hw, B_QT);
    // ring rows: the widest row span of one tile (two frames: the step into the next frame included)
    int span = 0;
    for (int k = 0; k < P.tiles_per_frame; ++k) {
        const int q0 = k * B_QT, q1 = std::min(q0 + B_QT, P.hw) - 1;
        span = std::max(span, q1 / P.wo - q0 / P.wo + 3);
    }
    const int PP = (P.W + 2) * 16, RBW = B_CPW * PP;
    const size_t fixed = (size_t)2 * B_OPND + 3 * B_QT * 4 + 3 * B_QT * 16 + 512 * 4 + 512 * 8;
    const int nr_max = (int)((160 * 1024 - fixed) / ((size_t)B_NW * RBW));
    if (nr_max < span) return DD_OK;                                 // (a wider map: q_dwpw_k)
    P.NR = span;
    const size_t lds = fixed + (size_t)B_NW * P.NR * RBW;
    P.wo_magic = (unsigned)((1ull << 32) / (unsigned)P.wo) + 1u;
    P.nr_magic = (unsigned)((1ull << 32) / (unsigned)P.NR) + 1u;
    P.tpf_magic = (unsigned)((1ull << 32) / (unsigned)P.tiles_per_frame) + 1u;
    const long long lim24 = 1ll << 23;                              // the kernel's 24-bit multiplies
    const long long n_tiles_ll = (long long)nimg * P.tiles_per_frame;
    const bool ok = n_tiles_ll * P.tiles_per_frame < (1ll << 32) && (long long)P.hw * P.wo < (1ll << 32) && (long long)(nimg + 1) * (P.H + 2) * P.NR < (1ll << 32) &&
                    RBW < lim24 && (long long)B_C16 * (P.wo + 2) * 16 < lim24 && (long long)(nimg + 1) * (P.ho + 2) < lim24 && P.hw < lim24 &&
                    (double)(nimg + 1) * (P.H + 2) * B_C16 * PP < 4294967296.0 && (double)nimg * (P.ho + 2) * (P.wo + 2) * 512 < 4294967296.0 &&
                    P.off_y >= 0 && P.off_x >= 0 && P.ho + 1 + P.off_y <= P.H + 1 && P.wo + 1 + P.off_x <= P.W + 1;
    if (!ok) return DD_OK;
    const bool rsum = P.zwc != 0;
    const int sat = P.Rd.lo == 0 && P.Rd.hi == 255 && P.Rp.lo == 0 && P.Rp.hi == 255 ? (P.Rd.e <= 8 && P.Rp.e <= 8 ? 2 : 1) : 0;
#define DD_BK(S_, R_) q_b512_k<S_, R_>
    void (*kern)(const QB512P, const int, const int) =
        sat == 2 ? (rsum ? &DD_BK(2, true) : &DD_BK(2, false)) : sat == 1 ? (rsum ? &DD_BK(1, true) : &DD_BK(1, false)) : (rsum ? &DD_BK(0, true) : &DD_BK(0, false));
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        for (const void *f : {reinterpret_cast<const void *>(&DD_BK(2, true)), reinterpret_cast<const void *>(&DD_BK(2, false)), reinterpret_cast<const void *>(&DD_BK(1, true)),
                              reinterpret_cast<const void *>(&DD_BK(1, false)), reinterpret_cast<const void *>(&DD_BK(0, true)), reinterpret_cast<const void *>(&DD_BK(0, false))})
            DD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return DD_OK;
    });
#undef DD_BK
    if (rc != DD_OK) return rc;
    const int n_tiles = (int)n_tiles_ll;
    const int blocks = std::min(n_tiles, dd_cu_count(device));
    const int tpb = dd_ceil_div(n_tiles, blocks);
    const dim3 grid((unsigned)dd_ceil_div(n_tiles, tpb));
    static const bool stamps = getenv("DD_Q_STAMPS") && atoi(getenv("DD_Q_STAMPS")) != 0;
    const size_t n_st = (size_t)grid.x * B_NW * 4;
    if (stamps) { DD_HIP(hipMalloc(&P.dbg, n_st * 8)); DD_HIP(hipMemsetAsync(P.dbg, 0, n_st * 8, s)); }
    hipLaunchKernelGGL(kern, grid, dim3(B_NT), lds, s, P, n_tiles, tpb);
    DD_LAUNCH_CHECK();
    if (stamps) {
        std::vector<unsigned long long> h(n_st);
        DD_HIP(hipStreamSynchronize(s));
        DD_HIP(hipMemcpy(h.data(), P.dbg, n_st * 8, hipMemcpyDeviceToHost));
        DD_HIP(hipFree(P.dbg));
        double sum[3] = {0, 0, 0};
        for (size_t w = 0; w < n_st / 4; ++w) for (int k = 0; k < 3; ++k) sum[k] += (double)h[w * 4 + k];
        const double nw = (double)(n_st / 4) * tpb;
        fprintf(stderr, "q_b512_k %d tiles/block: cycles per wave and tile: depthwise (tile t + 1) + row requests %.0f  pointwise (tile t) %.0f  barrier %.0f\n",
                tpb, sum[0] / nw, sum[1] / nw, sum[2] / nw);
    }
    *ran = 1;
    return DD_OK;
}
