// EXPERIMENT (round 6), not part of the library: measured slower than q_dwpw_k and taken out again.  uint8 SSD forward of 768 frames, same box:
// the 256-channel block 271 us as strips against 213 us as tiles, the 512-channel blocks 250 us against 124 us (that form also spilled 38 VGPRs
// at the 128-register line: 64 of them hold the wave's filter).  Same bits as q_dwpw_k in both.  Why: the two phases of a strip are strictly one
// after the other and short -- a wave's depthwise share is one plane x five fragments -- so every phase is a chain of LDS, matrix and vector
// latencies plus a barrier; the row pipelines that did pay (netsq_front.hip, netsq_mid.hip) put stages of neighbouring rows into one phase.
// To build it again: copy into deepdish_amd/csrc/, declare netq_run_strip in net_priv.h and call it from nets.hip's op loop for kind 19 ops.
//
// uint8 SSD-MobileNet-v1, the stride-1 MobileNet blocks on small maps (256 -> 256 at 38 x 38, 512 -> 512 at 19 x 19: blocks 5 and 7-11, six of the
// thirteen) as strips of whole rows walked by sixteen waves.  q_dwpw_k (csrc/netsq.hip) runs them as 64-pixel tiles with eight 256-register
// waves -- two per SIMD -- two barriers and a ring refill per tile, at ~60 % of what their instruction counts allow; the row
// kernels of this round (netsq_front.hip, netsq_mid.hip) reach 85-100 % with four waves per SIMD.  Same arithmetic, same packed filters, same
// bits (tests/test_gpu_quant.py runs both forms against oracle/nets_quant.py); the replaced interface is the middle of `interpreter.invoke()`
// (tools/ssd_mobilenet.py:100-109 upstream).
//
// A strip = RS consecutive output rows of one frame = at most 76 pixels = five 16-pixel fragments (four rows of 19, two of 38).  One workgroup
// of sixteen waves per CU walks a contiguous range of strips; per strip two phases, one barrier behind each:
//     Y:  depthwise 3x3 of the strip: inbuf -> opnd (and the pixels' operand row sums, for filters whose zero point is not 128)
//     X:  the NEXT strip's source rows are requested (LDS-DMA into inbuf: the depthwise stage has read it); pointwise: opnd -> HBM
// inbuf = the RS + 2 source rows of the strip exactly as they lie in HBM ([C / 16 planes][W + 2][16] bytes per row, border columns and border
// rows included: a padding tap is a plain read; the rows are one contiguous piece of the tensor); opnd = the pointwise stage's MFMA operand
// tile [k group][80 pixels][16].  A wave keeps the A fragments of its 32 (16) output channels for all of K in registers for the whole launch
// (64 / 16 VGPRs) and runs the depthwise stage of its two (one) planes; everything else it needs per strip is a handful of scalars.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "net_priv.h"
#include "netsq_dev.h"

namespace {

struct QStripP {
    const uint8_t *in; uint8_t *out;                              // Q16 [n][W + 2][C / 16][W + 2][16], both
    const uint2 *dwa; const int *dcb;                             // depthwise: operand table (netsq.pack_dw_mfma), constants [C]
    const i4v *w; const int *cb;                                  // pointwise: A fragments [C / 16][C / 64][64 lanes], constants [C]
    int zwc;                                                      // 128 - zw of the pointwise filter (times the operand row sum)
    QReq Rd, Rp;
};

template <int C, int W, int RS, bool ROWSUM, int SAT>
__global__ __launch_bounds__(1024) void q_strip_k(const QStripP P, const int strips_total, const int strips_per_block) {
    constexpr int C16 = C / 16, KC = C / 64, PPI = (W + 2) * 16, RB = C16 * PPI;          // planes, k slices, plane pitch, row pitch (source = destination geometry)
    constexpr int SPF = (W + RS - 1) / RS;                          // strips per frame
    constexpr int NF = 5, NPX = 80;                                 // fragments / pixel slots of a strip
    static_assert(RS * W <= NPX && C16 % 16 == 0, "strip geometry");
    constexpr int PPW = C16 / 16;                                   // depthwise planes per wave
    constexpr int MPW = C16 / 16;                                   // pointwise channel fragments per wave
    constexpr int INB = (RS + 2) * RB, OPB = C16 * NPX * 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *const inbuf = smem;
    uint8_t *const opnd = inbuf + INB;
    int *const rowsum = reinterpret_cast<int *>(opnd + OPB);        // [2][NPX]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int s_begin = blockIdx.x * strips_per_block, s_end = min(strips_total, s_begin + strips_per_block);
    if (s_begin >= s_end) return;

    // ---- the wave's filters and constants, once
    i4v Wr[MPW][KC], cbv[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
        const int mm = wave * MPW + m;                              // fragment 4 mg + m' holds channels 64 mg + 16 g + 4 m' + r
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) Wr[m][kc] = P.w[((size_t)mm * KC + kc) * 64 + lane];
        cbv[m] = *reinterpret_cast<const i4v *>(P.cb + 64 * (mm >> 2) + 16 * fq + 4 * (mm & 3));
    }
    uint2 ab[PPW];
    i4v dcbv[PPW];
    unsigned lom[PPW];
#pragma unroll
    for (int p = 0; p < PPW; ++p) {
        const int pl = wave * PPW + p;
        ab[p] = P.dwa[pl * 64 + lane];
        dcbv[p] = *reinterpret_cast<const i4v *>(P.dcb + 16 * pl + 4 * fq);
        lom[p] = (unsigned)__builtin_amdgcn_readfirstlane((int)(ab[p].y >> 24));      // k steps of the plane with a lo part (netsq.pack_dw_mfma)
    }
    unsigned dmask[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dmask[d] = (fr >> 2) == d ? 0xffu << (8 * (fr & 3)) : 0u;
    // the lane's tap of k step ks: tap t = min(4 ks + fq, 8) = (dy, dx) -> byte offset inside the strip's source rows
    int tapo[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { const int t = min(4 * ks + fq, 8); tapo[ks] = (t / 3) * RB + (t % 3) * 16; }
    const int Md = P.Rd.M, shd = P.Rd.e - 1, lod = P.Rd.lo, hid = P.Rd.hi;
    const int Mp = P.Rp.M, shp = P.Rp.e - 1, lop = P.Rp.lo, hip_ = P.Rp.hi;
    const long long Cd = P.Rd.C, Cp = P.Rp.C;

    auto glds16 = [&](const uint8_t *g, const uint8_t *l) {           // (assembly: see q_dwpw_k's glds16)
        unsigned keep;
        const unsigned dst = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)l;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    };
    // the source rows of strip s: rows y0 - 1 .. y0 + rows of frame n = bordered rows y0 .. y0 + rows + 1, one contiguous piece
    auto request_strip = [&](int s) {
        const int n = s / SPF, k = s - n * SPF, y0 = k * RS, rows = min(RS, W - y0);
        const unsigned nb = (unsigned)(rows + 2) * (unsigned)RB;
        const uint8_t *src = P.in + ((size_t)n * (W + 2) + (size_t)y0) * RB;
        for (unsigned p = (unsigned)wave; p * 1024u < nb; p += 16u) {
            const unsigned off = p * 1024u + (unsigned)lane * 16u;
            if (off < nb) glds16(src + off, inbuf + p * 1024u);
        }
    };
    if (ROWSUM) for (int i = tid; i < 2 * NPX; i += 1024) rowsum[i] = 0;
    request_strip(s_begin);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = s_begin; s < s_end; ++s) {
        const int n = s / SPF, k = s - n * SPF, y0 = k * RS, rows = min(RS, W - y0), npx = rows * W;
        const int par = s & 1;                                      // row sums: two sets, the other one is cleared for the next strip
        // ---------------- phase Y: depthwise stage of this wave's planes
        {
            int rs[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) rs[f] = 0;
#pragma unroll
            for (int p = 0; p < PPW; ++p) {
                const int pl = wave * PPW + p;
                // fragments in groups of G (the 512-channel block keeps 64 registers of filter: two fragments at a time, a window of three operands)
                constexpr int G = PPW > 1 ? 2 : 3;
#pragma unroll
                for (int g0 = 0; g0 < NF; g0 += G) {
                    constexpr int DW_W = G + 1;
                    const int ng = NF - g0 < G ? NF - g0 : G;        // (a literal after unrolling)
                    int base[G];
#pragma unroll
                    for (int f = 0; f < G; ++f) {
                        const int q = min(16 * (g0 + min(f, ng - 1)) + fr, npx - 1);     // (pixel slots past the strip read its last pixel; nothing of them is stored)
                        const int dr = q / W, x = q - dr * W;
                        base[f] = dr * RB + pl * PPI + x * 16;
                    }
                    i4v acc[G], b[DW_W];
                    auto opnd_at = [&](int j) __attribute__((always_inline)) { return *reinterpret_cast<const i4v *>(inbuf + base[j % ng] + tapo[j / ng]); };
#pragma unroll
                    for (int j = 0; j < DW_W; ++j) { if (j < 3 * ng) b[j] = opnd_at(j); }
#pragma unroll
                    for (int j = 0; j < 3 * G; ++j) {
                        if (j >= 3 * ng) continue;
                        const int ks = j / ng, f = j % ng;
                        const unsigned rh = __builtin_amdgcn_perm(ab[p].x, ab[p].x, 0x01010101u * (unsigned)ks);
                        i4v Ah;
#pragma unroll
                        for (int d = 0; d < 4; ++d) Ah[d] = (int)(rh & dmask[d]);
                        if (ks == 0) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah, b[j % DW_W], dcbv[p], 0, 0, 0);
                        else acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Ah, b[j % DW_W], acc[f], 0, 0, 0);
                        if (j + DW_W < 3 * ng) b[j % DW_W] = opnd_at(j + DW_W);
                    }
                    if (lom[p]) {                                      // (few planes: only a tensor's extreme weights overflow int8)
                        unsigned lw = ab[p].y;
                        asm volatile("" : "+v"(lw));                // (opaque: hipcc would hoist the rare path's operands out of the strip loop)
#pragma unroll
                        for (int ks = 0; ks < 3; ++ks) {
                            if (lom[p] & (1u << ks)) {
                                const unsigned rl = __builtin_amdgcn_perm(lw, lw, 0x01010101u * (unsigned)ks);
                                i4v Al;
#pragma unroll
                                for (int d = 0; d < 4; ++d) Al[d] = (int)(rl & dmask[d]);
#pragma unroll
                                for (int f = 0; f < G; ++f) { if (f < ng) acc[f] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Al, opnd_at(ks * ng + f), acc[f], 0, 0, 0); }
                            }
                        }
                    }
#pragma unroll
                    for (int f = 0; f < G; ++f) {
                        if (f >= ng) continue;
                        const unsigned packed = 0x80808080u ^ q_requant_pack4<SAT>(acc[f][0], acc[f][1], acc[f][2], acc[f][3], Md, Cd, Cd, Cd, Cd, shd, lod, hid);
                        if (ROWSUM) rs[g0 + f] = sdot4((int)packed, 0x01010101, rs[g0 + f]);
                        *reinterpret_cast<unsigned *>(opnd + ((size_t)pl * NPX + 16 * (g0 + f) + fr) * 16 + 4 * fq) = packed;
                    }
                }
            }
            if (ROWSUM) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    int v = rs[f];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (fq == 0) atomicAdd(&rowsum[par * NPX + 16 * f + fr], v);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): this wave's tile and row-sum traffic is done
        __builtin_amdgcn_s_barrier();
        // ---------------- phase X: the next strip's rows on their way; pointwise stage of this wave's channel fragments
        if (s + 1 < s_end) request_strip(s + 1);
        if (ROWSUM && tid < NPX) rowsum[(par ^ 1) * NPX + tid] = 0;
        {
            const int nf = (npx + 15) >> 4;
#pragma unroll 1
            for (int f = 0; f < nf; ++f) {
                const uint8_t *const bp = opnd + ((size_t)fq * NPX + 16 * f + fr) * 16;
                i4v acc[MPW];
#pragma unroll
                for (int m = 0; m < MPW; ++m) acc[m] = cbv[m];
#pragma unroll
                for (int k0 = 0; k0 < KC; k0 += 4) {
                    i4v b[4];
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) b[kc] = *reinterpret_cast<const i4v *>(bp + (size_t)(k0 + kc) * 4 * NPX * 16);
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                        for (int m = 0; m < MPW; ++m) acc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Wr[m][k0 + kc], b[kc], acc[m], 0, 0, 0);
                }
                const int rsv = ROWSUM ? rowsum[par * NPX + 16 * f + fr] * P.zwc : 0;
                const int q = 16 * f + fr;
                const int dr = min(q, npx - 1) / W, x = min(q, npx - 1) - dr * W;
#pragma unroll
                for (int m = 0; m < MPW; ++m) {
                    const int mm = wave * MPW + m;
                    const unsigned o = 0x80808080u ^ q_requant_pack4<SAT>(acc[m][0] + rsv, acc[m][1] + rsv, acc[m][2] + rsv, acc[m][3] + rsv, Mp, Cp, Cp, Cp, Cp, shp, lop, hip_);
                    if (q < npx)
                        *reinterpret_cast<unsigned *>(P.out + ((size_t)(n * (W + 2) + y0 + dr + 1) * C16 + 4 * (mm >> 2) + fq) * PPI + (size_t)(x + 1) * 16 + 4 * (mm & 3)) = o;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next strip's rows have landed (and this wave's stores are out)
        __builtin_amdgcn_s_barrier();
    }
}

template <int C, int W, int RS>
int launch_strip(dd_net *net, const QStripP &P, int nimg, hipStream_t s) {
    constexpr int C16 = C / 16, RB = C16 * (W + 2) * 16, LDS = (RS + 2) * RB + C16 * 80 * 16 + 2 * 80 * 4, SPF = (W + RS - 1) / RS;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    const bool rsum = P.zwc != 0;
    const int sat = P.Rd.lo == 0 && P.Rd.hi == 255 && P.Rp.lo == 0 && P.Rp.hi == 255 ? (P.Rd.e <= 8 && P.Rp.e <= 8 ? 2 : 1) : 0;
#define DD_SK(R_, S_) q_strip_k<C, W, RS, R_, S_>
    void (*kern)(const QStripP, const int, const int) =
        rsum ? (sat == 2 ? &DD_SK(true, 2) : sat == 1 ? &DD_SK(true, 1) : &DD_SK(true, 0)) : (sat == 2 ? &DD_SK(false, 2) : sat == 1 ? &DD_SK(false, 1) : &DD_SK(false, 0));
    static DevOnce once;
    const int rc = once.run(net->ctx->device, [&]() -> int {
        for (const void *f : {reinterpret_cast<const void *>(&DD_SK(true, 2)), reinterpret_cast<const void *>(&DD_SK(true, 1)), reinterpret_cast<const void *>(&DD_SK(true, 0)),
                              reinterpret_cast<const void *>(&DD_SK(false, 2)), reinterpret_cast<const void *>(&DD_SK(false, 1)), reinterpret_cast<const void *>(&DD_SK(false, 0))})
            DD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        return DD_OK;
    });
#undef DD_SK
    if (rc != DD_OK) return rc;
    const int strips = nimg * SPF;
    const int blocks = std::max(1, std::min(dd_cu_count(net->ctx->device), strips));
    const int spb = dd_ceil_div(strips, blocks);
    hipLaunchKernelGGL(kern, dim3((unsigned)dd_ceil_div(strips, spb)), dim3(1024), LDS, s, P, strips, spb);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace

// One stride-1 block op of a uint8 SSD-MobileNet-v1 program (256 -> 256 at 38 x 38 or 512 -> 512 at 19 x 19, unsplit pointwise filter) as strips of
// rows.  *ran = 0: not such a block -- the caller runs q_dwpw_k.
int netq_run_strip(dd_net *net, const int32_t *o, int nimg, hipStream_t s, int *ran) {
    *ran = 0;
    if (o[0] != OP_QDWPW || o[1] < 0 || o[7] != 1 || o[8] != 1 || o[9] != 1 || o[18] != 0 || o[47] != 0) return DD_OK;
    const TensorDesc &ti = net->tensors[o[1]], &to = net->tensors[o[2]];
    const int c = o[10];
    if (o[11] != c || ti.cs != c || to.cs != c || ti.h != ti.w || to.h != ti.h || to.w != ti.w || !ti.pad || !to.pad) return DD_OK;
    if (!((c == 512 && ti.w == 19) || (c == 256 && ti.w == 38))) return DD_OK;
    if (!o[17] || !o[20] || !o[21]) return DD_OK;
    char *Wt = net->d_weights;
    QStripP P;
    memset(&P, 0, sizeof(P));
    P.in = static_cast<const uint8_t *>(net->bufs[ti.buf]);
    P.out = static_cast<uint8_t *>(net->bufs[to.buf]);
    auto blob = [&](int32_t off) { return Wt + (size_t)(uint32_t)off; };
    P.dwa = reinterpret_cast<const uint2 *>(blob(o[20])); P.dcb = reinterpret_cast<const int *>(blob(o[21]));
    P.w = reinterpret_cast<const i4v *>(blob(o[16])); P.cb = reinterpret_cast<const int *>(blob(o[17]));
    P.zwc = o[38];
    { int32_t d[48] = {0}; d[32] = o[22]; d[33] = o[23]; d[36] = o[24]; d[37] = o[25]; d[40] = o[28]; P.Rd = make_req(d); }
    P.Rp = make_req(o);
    if (P.Rd.linear || P.Rp.linear || P.Rd.e < 1 || P.Rp.e < 1 || P.zwc < -128 || P.zwc > 128) return DD_OK;
    const int rc = c == 512 ? launch_strip<512, 19, 4>(net, P, nimg, s) : launch_strip<256, 38, 2>(net, P, nimg, s);
    if (rc != DD_OK) return rc;
    *ran = 1;
    return DD_OK;
}
