// Greedy score-ordered non-maximum suppression, bit-exact with the reference on the same boxes.
//
// mode 0: deep_sort/preprocessing.py:6-73  (tlwh f64, +1 pixel, inter / area_other > thr suppresses)
// mode 1: tools/ssd_mobilenet.py:59-98     (xyxy f64, +1 on the intersection only, IoU <= thr keeps)
// mode 2: TFLite detection_postprocess fast NMS (ymin,xmin,ymax,xmax f32, plain IoU > thr suppresses)
//
// k <= 64 (the per-frame case, ~25 boxes): ONE wave does everything in one launch -- rank by key,
// permute through LDS, then replay the greedy chain with the pivot box broadcast from LDS.
// k <= 4096: (A) rank sort (each candidate counts the candidates ahead of it, keys staged in LDS),
// (B) a 2-D grid builds the K x K/64 suppression bit-matrix with one wave ballot per 64 pairs,
// (C) a single wave replays the greedy scan 64 rows at a time: the chunk's rows of the bit-matrix
// are staged in LDS with fully pipelined loads, the diagonal word is resolved in registers, and the
// scan stops early once `max_keep` survivors exist (greedy prefix property).
// Ties in the key are resolved "higher original index first" (what a stable ascending argsort read
// from the back gives); the reference's own order under ties is unspecified (unstable sort).
#include "common.h"

namespace {

constexpr int MAXK = 4096;
typedef unsigned long long u64;

struct SBox { double a, b, c, d, area; };   // mode 0: x1,y1,x2,y2,area ; mode 1: x,y,w,h,area ; mode 2: ymin,xmin,ymax,xmax,area

__device__ __forceinline__ bool before(double ka, int ia, double kb, int ib) {
    return ka > kb || (ka == kb && ia > ib);
}

template <typename TB>
__device__ __forceinline__ SBox make_sbox(const TB *b, int mode) {
    SBox s;
    if (mode == 2) {
        s.a = b[0]; s.b = b[1]; s.c = b[2]; s.d = b[3];
        s.area = (double)(((float)b[2] - (float)b[0]) * ((float)b[3] - (float)b[1]));
    } else if (mode == 0) {
        s.a = b[0]; s.b = b[1]; s.c = b[2] + b[0]; s.d = b[3] + b[1];
        s.area = (s.c - s.a + 1) * (s.d - s.b + 1);         // preprocessing.py:43-48
    } else {
        s.a = b[0]; s.b = b[1]; s.c = b[2] - b[0]; s.d = b[3] - b[1];
        s.area = s.c * s.d;                                 // ssd_mobilenet.py:67-72
    }
    return s;
}

__device__ __forceinline__ bool suppresses(const SBox &pi, const SBox &pj, double thr, int mode) {
    if (mode == 2) {                                        // detection_postprocess.cc ComputeIntersectionOverUnion, f32
        const float ai = (float)pi.area, aj = (float)pj.area;
        if (ai <= 0.f || aj <= 0.f) return false;
        const float y0 = fmaxf((float)pi.a, (float)pj.a), x0 = fmaxf((float)pi.b, (float)pj.b);
        const float y1 = fminf((float)pi.c, (float)pj.c), x1 = fminf((float)pi.d, (float)pj.d);
        const float inter = fmaxf(y1 - y0, 0.f) * fmaxf(x1 - x0, 0.f);
        return inter / (ai + aj - inter) > (float)thr;
    }
    if (mode == 0) {
        const double xx1 = fmax(pi.a, pj.a), yy1 = fmax(pi.b, pj.b);
        const double xx2 = fmin(pi.c, pj.c), yy2 = fmin(pi.d, pj.d);
        const double w = fmax(0.0, xx2 - xx1 + 1), h = fmax(0.0, yy2 - yy1 + 1);
        return (w * h) / pj.area > thr;                     // preprocessing.py:59-71
    }
    const double xx1 = fmax(pi.a, pj.a), yy1 = fmax(pi.b, pj.b);
    const double xx2 = fmin(pi.a + pi.c, pj.a + pj.c), yy2 = fmin(pi.b + pi.d, pj.b + pj.d);
    const double w1 = fmax(0.0, xx2 - xx1 + 1), h1 = fmax(0.0, yy2 - yy1 + 1);
    const double inter = w1 * h1;
    const double ovr = inter / (pi.area + pj.area - inter);
    return !(ovr <= thr);                                   // ssd_mobilenet.py:85-91
}

// ---------------------------------------------------------------- k <= 64: one wave, one launch
// blockIdx.x selects an independent problem (batched form: offsets[p] .. offsets[p+1]).
template <typename TB>
__global__ __launch_bounds__(64) void nms_small_k(const TB *__restrict__ boxes, const TB *__restrict__ keys,
                                                  const int *__restrict__ offsets, int k_single, double thr, int mode,
                                                  int max_keep, int *__restrict__ out_idx, int *__restrict__ out_n) {
    __shared__ SBox sb[64];
    __shared__ int sid[64];
    const int lane = threadIdx.x;
    const int p = blockIdx.x;
    const int o0 = offsets ? offsets[p] : 0;
    const int k = offsets ? offsets[p + 1] - o0 : k_single;
    double key = -__builtin_inf();
    SBox mine = {0, 0, 0, 0, 0};
    if (lane < k) {
        key = (double)keys[o0 + lane];
        mine = make_sbox(boxes + (size_t)(o0 + lane) * 4, mode);
    }
    int rank = 0;
    for (int j = 0; j < k; ++j) {
        const double kj = __shfl(key, j, 64);
        rank += (lane < k && j != lane && before(kj, j, key, lane)) ? 1 : 0;
    }
    if (lane < k) { sb[rank] = mine; sid[rank] = lane; }
    __syncthreads();
    SBox me = {0, 0, 0, 0, 0};
    if (lane < k) me = sb[lane];                             // lane now owns sorted position `lane`
    u64 removed = 0, keep = 0;
    int n_keep = 0;
    for (int i = 0; i < k; ++i) {
        if ((removed >> i) & 1ull) continue;                // wave-uniform
        keep |= 1ull << i;
        if (++n_keep == max_keep) break;
        const SBox pi = sb[i];                              // LDS broadcast
        const bool s = lane > i && lane < k && suppresses(pi, me, thr, mode);
        removed |= __ballot(s);
    }
    if (lane < k && ((keep >> lane) & 1ull))
        out_idx[o0 + __popcll(keep & ((1ull << lane) - 1ull))] = sid[lane];
    if (lane == 0) out_n[p] = __popcll(keep);
}

// ---------------------------------------------------------------- general path
// (A) rank sort: block b ranks candidates [256 b, 256 b + 256) against all k keys staged in LDS
// (keys kept in their own precision; four f32 / two f64 keys per LDS read).
template <typename TB>
__global__ __launch_bounds__(256) void nms_rank_k(const TB *__restrict__ boxes, const TB *__restrict__ keys, int k, int mode,
                                                  SBox *__restrict__ sorted, int *__restrict__ sidx) {
    constexpr int V = 16 / sizeof(TB);                       // keys per 16-byte LDS read
    __shared__ __attribute__((aligned(16))) TB skey[MAXK + 4];
    boxes += (size_t)blockIdx.z * k * 4; keys += (size_t)blockIdx.z * k;     // blockIdx.z = image of a batch
    sorted += (size_t)blockIdx.z * k; sidx += (size_t)blockIdx.z * k;
    const int kp = (k + V - 1) / V * V;
    for (int i = threadIdx.x; i < kp; i += blockDim.x) skey[i] = i < k ? keys[i] : (TB)(-__builtin_inf());
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const TB ki = skey[i];
    int rank = 0;
    for (int j = 0; j < kp; j += V) {
        TB kj[V];
        *reinterpret_cast<uint4 *>(kj) = *reinterpret_cast<const uint4 *>(skey + j);
#pragma unroll
        for (int u = 0; u < V; ++u) rank += (kj[u] > ki || (kj[u] == ki && j + u > i && j + u < k)) ? 1 : 0;
    }
    sorted[rank] = make_sbox(boxes + (size_t)i * 4, mode);
    sidx[rank] = i;
}

// (A') f32 keys: bitonic sort of one problem per workgroup in LDS on the composite key
// (order-preserving bits of the score << 32 | original index), descending -- equal scores come out
// "higher original index first" exactly as the rank count above orders them.  n log^2 n
// compare-exchanges (67 K for the 1917 SSD anchors) instead of k^2 comparisons (3.7 M).
__global__ __launch_bounds__(1024) void nms_sort_f32_k(const float *__restrict__ boxes, const float *__restrict__ keys, int k,
                                                       int npad, int mode, SBox *__restrict__ sorted, int *__restrict__ sidx) {
    __shared__ u64 sk[MAXK];
    boxes += (size_t)blockIdx.x * k * 4; keys += (size_t)blockIdx.x * k;     // blockIdx.x = image of a batch
    sorted += (size_t)blockIdx.x * k; sidx += (size_t)blockIdx.x * k;
    for (int i = threadIdx.x; i < npad; i += 1024) {
        u64 c = 0ull;                                           // padding sorts last
        if (i < k) {
            const unsigned b = __float_as_uint(keys[i]);
            const unsigned m = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
            c = ((u64)m << 32) | (unsigned)i;
        }
        sk[i] = c;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (npad >> 1); t += 1024) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const u64 a = sk[lo], b = sk[hi];
                const bool desc = (lo & size) == 0;
                if (desc ? a < b : a > b) { sk[lo] = b; sk[hi] = a; }
            }
            __syncthreads();
        }
    for (int r = threadIdx.x; r < k; r += 1024) {
        const int idx = (int)(unsigned)sk[r];
        sorted[r] = make_sbox(boxes + (size_t)idx * 4, mode);
        sidx[r] = idx;
    }
}

// (B) grid (words, ceil(k/4)); block = 4 waves; wave handles row i, lanes cover the 64 columns of word w.
__global__ __launch_bounds__(256) void nms_mask_k(const SBox *__restrict__ sorted, int k, int words, double thr,
                                                  int mode, u64 *__restrict__ mask) {
    const int w = blockIdx.x;
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= k) return;
    sorted += (size_t)blockIdx.z * k; mask += (size_t)blockIdx.z * k * words;
    if (w * 64 + 63 <= i) {                                 // whole word lies on or below the diagonal
        if (lane == 0) mask[(size_t)i * words + w] = 0ull;
        return;
    }
    const int j = w * 64 + lane;
    bool s = false;
    if (j > i && j < k) s = suppresses(sorted[i], sorted[j], thr, mode);
    const u64 bits = __ballot(s);
    if (lane == 0) mask[(size_t)i * words + w] = bits;
}

// (C) one wave; lane l owns word l of the running "removed" set (words <= 64).
__global__ __launch_bounds__(64) void nms_scan_k(const u64 *__restrict__ mask, const int *__restrict__ sidx, int k,
                                                 int words, int max_keep, int *__restrict__ out_idx, int *__restrict__ out_n) {
    __shared__ u64 rows[64 * 64];                           // this chunk's 64 rows x `words` words
    const int lane = threadIdx.x;
    mask += (size_t)blockIdx.x * k * words; sidx += (size_t)blockIdx.x * k;   // blockIdx.x = image of a batch
    out_idx += (size_t)blockIdx.x * k; out_n += blockIdx.x;
    u64 removed = 0;
    int n_keep = 0;
    for (int c = 0; c < words; ++c) {
        const int row0 = c * 64;
        const int nrows = min(64, k - row0);
        // stage rows [row0, row0+nrows) x words [c, words): independent coalesced loads, one wait
        const int span = words - c;
        for (int t = lane; t < nrows * span; t += 64) {
            const int r = t / span, w = c + (t - r * span);
            rows[r * 64 + w] = mask[(size_t)(row0 + r) * words + w];
        }
        __syncthreads();
        const u64 diag = lane < nrows ? rows[lane * 64 + c] : 0ull;
        u64 rc = __shfl(removed, c, 64);
        u64 keep = 0;
        bool full = false;
        for (int b = 0; b < nrows; ++b) {
            if (!((rc >> b) & 1ull)) {
                keep |= 1ull << b;
                rc |= __shfl(diag, b, 64);
                if (max_keep > 0 && n_keep + __popcll(keep) >= max_keep) { full = true; break; }
            }
        }
        if (lane < nrows && ((keep >> lane) & 1ull))
            out_idx[n_keep + __popcll(keep & ((1ull << lane) - 1ull))] = sidx[row0 + lane];
        n_keep += __popcll(keep);
        if (full) break;                                    // wave-uniform
        if (lane > c && lane < words) {                     // fold the kept rows into my word
            u64 acc = 0, kk = keep;
            while (kk) {
                const int b = __ffsll((long long)kk) - 1;
                kk &= kk - 1;
                acc |= rows[b * 64 + lane];
            }
            removed |= acc;
        }
        __syncthreads();
    }
    if (lane == 0) *out_n = n_keep;
}

// (B+C fused) one workgroup per problem: for each chunk of 64 sorted rows, 16 waves build that
// chunk's rows of the suppression bit-matrix on the fly (skipping rows that are already suppressed;
// a wave owns whole 64-column words so its column box stays in registers and the pivot boxes are
// broadcast from LDS), then wave 0 resolves the chunk and folds the survivors into the running
// "removed" set.  With max_keep the loop usually ends after the first chunk, so only ~64 x k pairs
// are ever evaluated instead of k^2 / 2 (the SSD post-process needs 10 survivors of 1917 candidates).
__global__ __launch_bounds__(1024) void nms_lazy_k(const SBox *__restrict__ sorted, const int *__restrict__ sidx, int k,
                                                   int words, double thr, int mode, int max_keep,
                                                   int *__restrict__ out_idx, int *__restrict__ out_n) {
    __shared__ u64 rows[64 * 64];
    __shared__ u64 s_removed[64];
    __shared__ SBox srowbox[64];
    __shared__ float srowf[64][5];
    __shared__ int s_nkeep, s_done;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    sorted += (size_t)blockIdx.x * k; sidx += (size_t)blockIdx.x * k;
    out_idx += (size_t)blockIdx.x * k; out_n += blockIdx.x;
    if (tid < 64) s_removed[tid] = 0ull;
    if (tid == 0) { s_nkeep = 0; s_done = 0; }
    __syncthreads();
    for (int c = 0; c < words; ++c) {
        const int row0 = c * 64;
        const int nrows = min(64, k - row0);
        const u64 gone = s_removed[c];
        if (tid < nrows) {
            const SBox sb = sorted[row0 + tid];
            srowbox[tid] = sb;
            srowf[tid][0] = (float)sb.a; srowf[tid][1] = (float)sb.b; srowf[tid][2] = (float)sb.c; srowf[tid][3] = (float)sb.d;
            srowf[tid][4] = (float)sb.area;
        }
        __syncthreads();
        for (int w = c + wave; w < words; w += 16) {
            const int j = w * 64 + lane;
            SBox cj = {0, 0, 0, 0, 0};
            if (j < k) cj = sorted[j];
            if (mode == 2) {                                   // f32 boxes: narrow both sides once, not per pair
                const float ja = (float)cj.a, jb = (float)cj.b, jc = (float)cj.c, jd = (float)cj.d, jarea = (float)cj.area;
                const float thrf = (float)thr;
                for (int r = 0; r < nrows; ++r) {
                    u64 bits = 0ull;
                    if (!((gone >> r) & 1ull)) {
                        const float ia = srowf[r][0], ib = srowf[r][1], ic = srowf[r][2], id = srowf[r][3], iarea = srowf[r][4];
                        bool sup = false;
                        if (j > row0 + r && j < k && iarea > 0.f && jarea > 0.f) {
                            const float y0 = fmaxf(ia, ja), x0 = fmaxf(ib, jb), y1 = fminf(ic, jc), x1 = fminf(id, jd);
                            const float inter = fmaxf(y1 - y0, 0.f) * fmaxf(x1 - x0, 0.f);
                            sup = inter / (iarea + jarea - inter) > thrf;
                        }
                        bits = __ballot(sup);
                    }
                    if (lane == 0) rows[r * 64 + w] = bits;
                }
                continue;
            }
            for (int r = 0; r < nrows; ++r) {
                u64 bits = 0ull;
                if (!((gone >> r) & 1ull)) {                   // wave-uniform: suppressed rows never suppress
                    const bool s = j > row0 + r && j < k && suppresses(srowbox[r], cj, thr, mode);
                    bits = __ballot(s);
                }
                if (lane == 0) rows[r * 64 + w] = bits;
            }
        }
        __syncthreads();
        if (wave == 0) {
            const u64 diag = lane < nrows ? rows[lane * 64 + c] : 0ull;
            u64 rc = gone, keep = 0;
            const int n_keep = s_nkeep;
            bool full = false;
            for (int b = 0; b < nrows; ++b) {
                if (!((rc >> b) & 1ull)) {
                    keep |= 1ull << b;
                    rc |= __shfl(diag, b, 64);
                    if (max_keep > 0 && n_keep + __popcll(keep) >= max_keep) { full = true; break; }
                }
            }
            if (lane < nrows && ((keep >> lane) & 1ull))
                out_idx[n_keep + __popcll(keep & ((1ull << lane) - 1ull))] = sidx[row0 + lane];
            if (lane > c && lane < words) {
                u64 acc = 0, kk = keep;
                while (kk) {
                    const int b = __ffsll((long long)kk) - 1;
                    kk &= kk - 1;
                    acc |= rows[b * 64 + lane];
                }
                s_removed[lane] |= acc;
            }
            if (lane == 0) { s_nkeep = n_keep + __popcll(keep); s_done = full ? 1 : 0; }
        }
        __syncthreads();
        if (s_done) break;
    }
    if (tid == 0) *out_n = s_nkeep;
}


// The SSD post-process form (max_keep survivors of k candidates, max_keep << k): no sort at all.  Greedy score-ordered NMS is
// "keep the best live candidate, kill what it suppresses, repeat" -- max_keep rounds of a workgroup-wide arg-max over the live
// composite keys (order-preserving score bits << 32 | index: equal scores "higher index first", the order of nms_sort_f32_k) and one
// IoU per live candidate, against 67 K compare-exchanges of the full sort plus 64 x k pair tests of nms_lazy_k's first chunk.  (The
// model files' nms_score_threshold is 1e-8: practically every one of the 1917 anchors is a candidate, so compacting the candidates
// first -- this kernel's first form -- left the sort as long as it was.)  A thread keeps PER candidates (keys, boxes, areas) in
// registers; the boxes are staged in LDS once so that the round's pivot is one broadcast read.  One barrier per round: the
// per-wave maxima alternate between two LDS rows.  Same f32 expressions as `suppresses` mode 2, the same survivors in the same
// order as nms_f32_batched returns for keys = (score >= score_thr ? score : -1), up to the first below-threshold row.
template <int PER>
__global__ __launch_bounds__(256) void nms_greedy_f32_k(const float *__restrict__ boxes, const float *__restrict__ scores, int k,
                                                        float score_thr, float iou_thr, int max_keep, int *__restrict__ out_idx,
                                                        int *__restrict__ out_n) {
    extern __shared__ __attribute__((aligned(16))) float4 sbox[];         // [k]
    __shared__ u64 s_part[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    boxes += (size_t)blockIdx.x * k * 4; scores += (size_t)blockIdx.x * k;
    out_idx += (size_t)blockIdx.x * k; out_n += blockIdx.x;
    u64 key[PER];
    float4 bx[PER];
    float area[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + 256 * j;
        key[j] = 0ull;                                            // 0 = not a candidate / dead (a real key is never 0: only the bits of a NaN map to 0)
        bx[j] = float4{0.f, 0.f, 0.f, 0.f};
        if (i < k) {
            bx[j] = *reinterpret_cast<const float4 *>(boxes + (size_t)i * 4);
            sbox[i] = bx[j];
            const float sc = scores[i];
            if (sc >= score_thr) {                                // NaN: not a candidate (its key was -1 in the full sort as well)
                const unsigned b = __float_as_uint(sc);
                const unsigned mkey = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
                key[j] = ((u64)mkey << 32) | (unsigned)i;
            }
        }
        area[j] = (bx[j].z - bx[j].x) * (bx[j].w - bx[j].y);
    }
    int n_keep = 0;
    for (int it = 0; n_keep < max_keep; ++it) {
        u64 best = key[0];
#pragma unroll
        for (int j = 1; j < PER; ++j) best = key[j] > best ? key[j] : best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const u64 other = __shfl_xor(best, o, 64);
            best = other > best ? other : best;
        }
        if (lane == 0) s_part[it & 1][wave] = best;
        __syncthreads();                                          // (round 0: also orders the sbox writes above)
        u64 b = s_part[it & 1][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const u64 o = s_part[it & 1][w]; b = o > b ? o : b; }
        if (b == 0ull) break;                                     // nothing live any more (uniform)
        const int idx = (int)(unsigned)b;
        if (tid == 0) out_idx[n_keep] = idx;
        ++n_keep;
        const float4 pv = sbox[idx];
        const float ia = pv.x, ib = pv.y, ic = pv.z, id = pv.w;
        const float iarea = (ic - ia) * (id - ib);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (key[j] == b) key[j] = 0ull;                       // the survivor itself
            if (key[j] != 0ull && iarea > 0.f && area[j] > 0.f) {
                const float y0 = fmaxf(ia, bx[j].x), x0 = fmaxf(ib, bx[j].y), y1 = fminf(ic, bx[j].z), x1 = fminf(id, bx[j].w);
                const float inter = fmaxf(y1 - y0, 0.f) * fmaxf(x1 - x0, 0.f);
                if (inter / (iarea + area[j] - inter) > iou_thr) key[j] = 0ull;
            }
        }
    }
    if (tid == 0) *out_n = n_keep;
}

}  // namespace

namespace ddk {

size_t nms_scratch_bytes(int k) {
    const size_t words = (k + 63) / 64;
    return (size_t)k * sizeof(SBox) + (size_t)k * sizeof(int) + 64 + (size_t)k * words * sizeof(u64);
}

// max_keep <= 0: keep every survivor.
int nms_ex(hipStream_t s, const void *boxes, const void *keys, int k, double thr, int mode, int max_keep, int *out_idx,
           int *out_n, void *scratch, size_t scratch_bytes) {
    if (k <= 0) {
        DD_HIP(hipMemsetAsync(out_n, 0, sizeof(int), s));
        return DD_OK;
    }
    DD_REQUIRE(k <= MAXK, DD_E_CAPACITY, "dd_nms: k=%d exceeds the single-pass capacity %d", k, MAXK);
    const bool f32 = mode == 2;
    if (k <= 64) {
        if (f32)
            hipLaunchKernelGGL(nms_small_k<float>, dim3(1), dim3(64), 0, s, static_cast<const float *>(boxes),
                               static_cast<const float *>(keys), (const int *)nullptr, k, thr, mode, max_keep, out_idx, out_n);
        else
            hipLaunchKernelGGL(nms_small_k<double>, dim3(1), dim3(64), 0, s, static_cast<const double *>(boxes),
                               static_cast<const double *>(keys), (const int *)nullptr, k, thr, mode, max_keep, out_idx, out_n);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    DD_REQUIRE(scratch && scratch_bytes >= nms_scratch_bytes(k), DD_E_ARG, "dd_nms: scratch too small");
    const int words = (k + 63) / 64;
    char *p = static_cast<char *>(scratch);
    SBox *sorted = reinterpret_cast<SBox *>(p);
    p += (size_t)k * sizeof(SBox);
    int *sidx = reinterpret_cast<int *>(p);
    p += ((size_t)k * sizeof(int) + 63) / 64 * 64;
    u64 *mask = reinterpret_cast<u64 *>(p);
    int npad = 128;
    while (npad < k) npad <<= 1;
    if (f32)
        hipLaunchKernelGGL(nms_sort_f32_k, dim3(1), dim3(1024), 0, s, static_cast<const float *>(boxes),
                           static_cast<const float *>(keys), k, npad, mode, sorted, sidx);
    else
        hipLaunchKernelGGL(nms_rank_k<double>, dim3(dd_ceil_div(k, 256)), dim3(256), 0, s, static_cast<const double *>(boxes),
                           static_cast<const double *>(keys), k, mode, sorted, sidx);
    DD_LAUNCH_CHECK();
    if (max_keep > 0 || k <= 1024) {
        hipLaunchKernelGGL(nms_lazy_k, dim3(1), dim3(1024), 0, s, sorted, sidx, k, words, thr, mode, max_keep, out_idx, out_n);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    hipLaunchKernelGGL(nms_mask_k, dim3(words, dd_ceil_div(k, 4)), dim3(256), 0, s, sorted, k, words, thr, mode, mask);
    DD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_k, dim3(1), dim3(64), 0, s, mask, sidx, k, words, max_keep, out_idx, out_n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// `batch` independent f32 problems of the same size k (64 < k <= 4096): boxes [batch][k][4], keys
// [batch][k], out_idx [batch][k], out_n [batch]; scratch >= batch * nms_scratch_bytes(k).
int nms_f32_batched(hipStream_t s, const float *boxes, const float *keys, int k, float thr, int max_keep, int *out_idx,
                    int *out_n, void *scratch, size_t scratch_bytes, int batch) {
    DD_REQUIRE(k > 64 && k <= MAXK && batch > 0, DD_E_ARG, "nms_f32_batched: k=%d batch=%d", k, batch);
    DD_REQUIRE(scratch && scratch_bytes >= (size_t)batch * nms_scratch_bytes(k), DD_E_ARG, "nms_f32_batched: scratch too small");
    const int words = (k + 63) / 64;
    char *p = static_cast<char *>(scratch);
    SBox *sorted = reinterpret_cast<SBox *>(p);
    p += (size_t)batch * k * sizeof(SBox);
    int *sidx = reinterpret_cast<int *>(p);
    p += ((size_t)batch * k * sizeof(int) + 63) / 64 * 64;
    u64 *mask = reinterpret_cast<u64 *>(p);
    int npad = 128;
    while (npad < k) npad <<= 1;
    hipLaunchKernelGGL(nms_sort_f32_k, dim3(batch), dim3(1024), 0, s, boxes, keys, k, npad, 2, sorted, sidx);
    DD_LAUNCH_CHECK();
    if (max_keep > 0 || k <= 1024) {
        hipLaunchKernelGGL(nms_lazy_k, dim3(batch), dim3(1024), 0, s, sorted, sidx, k, words, (double)thr, 2, max_keep, out_idx, out_n);
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    hipLaunchKernelGGL(nms_mask_k, dim3(words, dd_ceil_div(k, 4), batch), dim3(256), 0, s, sorted, k, words, (double)thr, 2, mask);
    DD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_k, dim3(batch), dim3(64), 0, s, mask, sidx, k, words, max_keep, out_idx, out_n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// The SSD post-process form: `batch` problems of k <= 4096 boxes each, candidates = scores >= score_thr, at most max_keep
// (1..64) survivors each; out_idx [batch][k], out_n [batch].  The survivors and their order are those nms_f32_batched returns
// for keys = (score >= score_thr ? score : -1) up to the first below-threshold row.  No scratch.
int nms_f32_select_batched(hipStream_t s, const float *boxes, const float *scores, int k, float score_thr, float iou_thr,
                           int max_keep, int *out_idx, int *out_n, int batch) {
    DD_REQUIRE(k > 0 && k <= MAXK && batch > 0 && max_keep > 0 && max_keep <= 64, DD_E_ARG,
               "nms_f32_select_batched: k=%d batch=%d max_keep=%d", k, batch, max_keep);
    const size_t lds = (size_t)k * sizeof(float4);
    if (k <= 2048) hipLaunchKernelGGL(nms_greedy_f32_k<8>, dim3(batch), dim3(256), lds, s, boxes, scores, k, score_thr, iou_thr, max_keep, out_idx, out_n);
    else {                                                  // (exercised by the 3 000-box case of scripts/ssd_post_cases.py)
        static DevOnce once;                                   // a per-device attribute: once per device, not on every launch
        int dev = 0;
        DD_HIP(hipGetDevice(&dev));
        const int rc = once.run(dev, [&]() -> int {
            DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&nms_greedy_f32_k<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MAXK * sizeof(float4))));
            return DD_OK;
        });
        if (rc != DD_OK) return rc;
        hipLaunchKernelGGL(nms_greedy_f32_k<16>, dim3(batch), dim3(256), lds, s, boxes, scores, k, score_thr, iou_thr, max_keep, out_idx, out_n);
    }
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int nms(hipStream_t s, const double *boxes, const double *keys, int k, double thr, int mode, int *out_idx,
        int *out_n, void *scratch, size_t scratch_bytes) {
    return nms_ex(s, boxes, keys, k, thr, mode, 0, out_idx, out_n, scratch, scratch_bytes);
}

int nms_f32(hipStream_t s, const float *boxes_yxyx, const float *keys, int k, float thr, int max_keep, int *out_idx,
            int *out_n, void *scratch, size_t scratch_bytes) {
    return nms_ex(s, boxes_yxyx, keys, k, (double)thr, 2, max_keep, out_idx, out_n, scratch, scratch_bytes);
}

// P independent small problems (each <= 64 boxes) in one launch: problem p owns rows
// [offsets[p], offsets[p+1]) of boxes/keys/out_idx; out_n[p] survivors.
int nms_batched_small(hipStream_t s, const double *boxes, const double *keys, const int *d_offsets, int n_problems,
                      double thr, int mode, int *out_idx, int *out_n) {
    if (n_problems <= 0) return DD_OK;
    hipLaunchKernelGGL(nms_small_k<double>, dim3(n_problems), dim3(64), 0, s, boxes, keys, d_offsets, 0, thr, mode, 0,
                       out_idx, out_n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace ddk
