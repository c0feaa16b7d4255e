// Greedy score-ordered non-maximum suppression in f64, bit-exact with the reference on the same
// boxes: (A) one workgroup bitonic-sorts (key, index) in LDS and writes the boxes in pick order,
// (B) a 2-D grid builds the K x K/64 suppression bit-matrix with one wave ballot per 64 pairs,
// (C) a single wave replays the greedy scan 64 rows at a time (diagonal word resolved in registers,
// off-diagonal words OR-ed in with independent, pipelined loads).
//
// mode 0: deep_sort/preprocessing.py:6-73  (tlwh, +1 pixel, inter / area_other > thr suppresses)
// mode 1: tools/ssd_mobilenet.py:59-98     (xyxy, +1 on the intersection only, IoU <= thr keeps)
// Ties in the key are resolved "higher original index first" (what a stable ascending argsort
// read from the back gives); the reference's own order under ties is unspecified (unstable sort).
#include "common.h"

namespace {

constexpr int MAXK = 4096;
typedef unsigned long long u64;

struct SBox { double a, b, c, d, area; };   // mode 0: x1,y1,x2,y2,area ; mode 1: x,y,w,h,area

__device__ __forceinline__ bool before(double ka, int ia, double kb, int ib) {
    return ka > kb || (ka == kb && ia > ib);
}

template <typename TB>
__global__ __launch_bounds__(1024) void nms_sort_k(const TB *__restrict__ boxes, const TB *__restrict__ keys,
                                                   int k, int mode, SBox *__restrict__ sorted, int *__restrict__ sidx) {
    __shared__ double skey[MAXK];
    __shared__ int sid[MAXK];
    int n = 1;
    while (n < k) n <<= 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        skey[i] = i < k ? (double)keys[i] : -__builtin_inf();
        sid[i] = i < k ? i : -1 - i;                       // padding sorts last (lowest key, lowest id)
    }
    __syncthreads();
    for (int sz = 2; sz <= n; sz <<= 1) {
        for (int j = sz >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const double ka = skey[i], kb = skey[p];
                    const int ia = sid[i], ib = sid[p];
                    const bool up = (i & sz) == 0;          // "up" = pick order (best first)
                    const bool swap = up ? before(kb, ib, ka, ia) : before(ka, ia, kb, ib);
                    if (swap) { skey[i] = kb; skey[p] = ka; sid[i] = ib; sid[p] = ia; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const int o = sid[i];
        const TB *b = boxes + (size_t)o * 4;
        SBox s;
        if (mode == 2) {                                    // f32 (ymin, xmin, ymax, xmax), TFLite fast NMS
            s.a = b[0]; s.b = b[1]; s.c = b[2]; s.d = b[3];
            s.area = (double)(((float)b[2] - (float)b[0]) * ((float)b[3] - (float)b[1]));
        } else if (mode == 0) {
            s.a = b[0]; s.b = b[1]; s.c = b[2] + b[0]; s.d = b[3] + b[1];
            s.area = (s.c - s.a + 1) * (s.d - s.b + 1);     // preprocessing.py:43-48
        } else {
            s.a = b[0]; s.b = b[1]; s.c = b[2] - b[0]; s.d = b[3] - b[1];
            s.area = s.c * s.d;                             // ssd_mobilenet.py:67-72
        }
        sorted[i] = s;
        sidx[i] = o;
    }
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

// grid (words, ceil(k/4)); block = 4 waves; wave handles row i, lanes cover the 64 columns of word w.
__global__ __launch_bounds__(256) void nms_mask_k(const SBox *__restrict__ sorted, int k, int words, double thr,
                                                  int mode, u64 *__restrict__ mask) {
    const int w = blockIdx.x;
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= k) return;
    const int j = w * 64 + lane;
    bool s = false;
    if (j > i && j < k) s = suppresses(sorted[i], sorted[j], thr, mode);
    const u64 bits = __ballot(s);
    if (lane == 0) mask[(size_t)i * words + w] = bits;
}

__global__ __launch_bounds__(64) void nms_scan_k(const u64 *__restrict__ mask, const int *__restrict__ sidx, int k,
                                                 int words, int *__restrict__ out_idx, int *__restrict__ out_n) {
    const int lane = threadIdx.x;
    u64 removed = 0;                                        // lane l owns word l (words <= 64)
    int n_keep = 0;
    for (int c = 0; c < words; ++c) {
        const int row0 = c * 64;
        const int rows = min(64, k - row0);
        const u64 diag = lane < rows ? mask[(size_t)(row0 + lane) * words + c] : 0ull;
        u64 rc = __shfl(removed, c, 64);
        u64 keep = 0;
        for (int b = 0; b < rows; ++b) {
            if (!((rc >> b) & 1ull)) {
                keep |= 1ull << b;
                rc |= __shfl(diag, b, 64);
            }
        }
        // emit the kept rows of this chunk in pick order
        if (lane < rows && ((keep >> lane) & 1ull)) {
            const int pos = n_keep + __popcll(keep & ((1ull << lane) - 1ull));
            out_idx[pos] = sidx[row0 + lane];
        }
        n_keep += __popcll(keep);
        // fold the kept rows into the later words (lane l > c only needs word l)
        if (c + 1 < words) {
            u64 kk = keep;
            while (kk) {
                u64 acc = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (kk) {
                        const int b = __ffsll((long long)kk) - 1;
                        kk &= kk - 1;
                        if (lane < words) acc |= mask[(size_t)(row0 + b) * words + lane];
                    }
                }
                removed |= acc;
            }
        }
    }
    if (lane == 0) *out_n = n_keep;
}

}  // namespace

namespace ddk {

size_t nms_scratch_bytes(int k) {
    const size_t words = (k + 63) / 64;
    return (size_t)k * sizeof(SBox) + (size_t)k * sizeof(int) + 64 + (size_t)k * words * sizeof(u64);
}

int nms(hipStream_t s, const double *boxes, const double *keys, int k, double thr, int mode, int *out_idx,
        int *out_n, void *scratch, size_t scratch_bytes) {
    if (k <= 0) {
        DD_HIP(hipMemsetAsync(out_n, 0, sizeof(int), s));
        return DD_OK;
    }
    DD_REQUIRE(k <= MAXK, DD_E_CAPACITY, "dd_nms: k=%d exceeds the single-pass capacity %d", k, MAXK);
    DD_REQUIRE(scratch_bytes >= nms_scratch_bytes(k), DD_E_ARG, "dd_nms: scratch too small");
    const int words = (k + 63) / 64;
    char *p = static_cast<char *>(scratch);
    SBox *sorted = reinterpret_cast<SBox *>(p);
    p += (size_t)k * sizeof(SBox);
    int *sidx = reinterpret_cast<int *>(p);
    p += ((size_t)k * sizeof(int) + 63) / 64 * 64;
    u64 *mask = reinterpret_cast<u64 *>(p);
    if (mode == 2)
        hipLaunchKernelGGL(nms_sort_k<float>, dim3(1), dim3(1024), 0, s, reinterpret_cast<const float *>(boxes),
                           reinterpret_cast<const float *>(keys), k, mode, sorted, sidx);
    else
        hipLaunchKernelGGL(nms_sort_k<double>, dim3(1), dim3(1024), 0, s, boxes, keys, k, mode, sorted, sidx);
    DD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_mask_k, dim3(words, dd_ceil_div(k, 4)), dim3(256), 0, s, sorted, k, words, thr, mode, mask);
    DD_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_k, dim3(1), dim3(64), 0, s, mask, sidx, k, words, out_idx, out_n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int nms_f32(hipStream_t s, const float *boxes_yxyx, const float *keys, int k, float thr, int *out_idx, int *out_n,
            void *scratch, size_t scratch_bytes) {
    return nms(s, reinterpret_cast<const double *>(boxes_yxyx), reinterpret_cast<const double *>(keys), k, (double)thr, 2,
               out_idx, out_n, scratch, scratch_bytes);
}

}  // namespace ddk
