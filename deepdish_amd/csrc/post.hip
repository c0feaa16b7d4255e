// Detector post-processing kernels.
//   * dd_ssd_postprocess : the TFLite_Detection_PostProcess custom op that sits inside the reference's
//     SSD .tflite graph (invoked at tools/ssd_mobilenet.py:103, outputs read at :107-109): anchor
//     decode (scales 10,10,5,5), sigmoid class scores, best class per anchor, class-agnostic fast NMS
//     in f32, top max_det.  TensorFlow Lite is a third-party dependency absent from this image; the
//     op is restated from its published behaviour -- parity unpinned.
//   * dd_yolov5_decode   : tools/yolov5.py:120-131.
//   * dd_counts_accumulate : int64 count vector (deepdish.py:1141-1145) kept on the device for the
//     end-of-run RCCL reduction.
#include <algorithm>
#include "common.h"
#include "ssd_dev.h"

namespace ddk {
int nms_f32(hipStream_t s, const float *boxes_yxyx, const float *keys, int k, float thr, int max_keep, int *out_idx,
            int *out_n, void *scratch, size_t scratch_bytes);
int nms_f32_select_batched(hipStream_t s, const float *boxes, const float *scores, int k, float score_thr, float iou_thr,
                           int max_keep, int *out_idx, int *out_n, int batch);
int nms_f32_batched(hipStream_t s, const float *boxes, const float *keys, int k, float thr, int max_keep, int *out_idx,
                    int *out_n, void *scratch, size_t scratch_bytes, int batch);
}

namespace {

// Sixteen lanes per anchor (four anchors per wave): the lanes sweep the class logits of their anchor in
// 64-byte pieces, a 16-lane butterfly picks the best class (lowest class index on ties, like a
// sequential `>` scan); the sigmoid is applied once.  (One wave per anchor left most lanes idle and made
// the kernel a chain of 120 K tiny waves for a 64-frame batch.)
__global__ __launch_bounds__(256) void ssd_decode_k(const float *__restrict__ raw, const float *__restrict__ anchors,
                                                    int n_anchors, int n_classes, float score_thr,
                                                    float *__restrict__ boxes, float *__restrict__ best_score,
                                                    int *__restrict__ best_cls, float *__restrict__ keys) {
    const int a = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    {   // blockIdx.y = image of a batch
        const size_t z = blockIdx.y;
        raw += z * n_anchors * (4 + n_classes);
        boxes += z * n_anchors * 4; best_score += z * n_anchors; best_cls += z * n_anchors; keys += z * n_anchors;
    }
    const bool live = a < n_anchors;                          // whole 16-lane groups are live or not; all lanes shuffle
    const float *r = raw + (size_t)(live ? a : 0) * (4 + n_classes);
    float best = -__builtin_inff();
    int bi = 0x7fffffff;
    // all loads of a sweep first (8 x 16 classes per pass): in a load-compare loop hipcc waits for every load before the next
    // one is issued (PMC: 76 % of the wave cycles parked on s_waitcnt) -- out-of-range slots re-read the lane's first class
    for (int c0 = 1 + sub; c0 < n_classes; c0 += 128) {       // class 0 = background
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r[4 + (c0 + 16 * i < n_classes ? c0 + 16 * i : c0)];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + 16 * i;
            if (c < n_classes && v[i] > best) { best = v[i]; bi = c - 1; }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (live && sub == 0) {
        const float rr[4] = {r[0], r[1], r[2], r[3]};
        const float an[4] = {anchors[a * 4 + 0], anchors[a * 4 + 1], anchors[a * 4 + 2], anchors[a * 4 + 3]};
        float bx[4];
        const float sc = ssddev::decode_anchor(rr, an, best, bx);
#pragma unroll
        for (int q = 0; q < 4; ++q) boxes[a * 4 + q] = bx[q];
        best_score[a] = sc;
        best_cls[a] = bi;
        keys[a] = sc >= score_thr ? sc : -1.f;
    }
}

__global__ void ssd_gather_k(const int *__restrict__ keep, const int *__restrict__ n_keep, const float *__restrict__ boxes,
                             const float *__restrict__ best_score, const int *__restrict__ best_cls, float score_thr,
                             int max_det, int k_stride, float *__restrict__ out_boxes, float *__restrict__ out_cls,
                             float *__restrict__ out_scores, int *__restrict__ out_count) {
    const int i = threadIdx.x;
    {   // blockIdx.x = image of a batch; k_stride = anchors per image
        const size_t z = blockIdx.x;
        keep += z * k_stride; n_keep += z; boxes += z * k_stride * 4; best_score += z * k_stride; best_cls += z * k_stride;
        out_boxes += z * max_det * 4; out_cls += z * max_det; out_scores += z * max_det; out_count += z;
    }
    const int n = min(*n_keep, max_det);
    bool ok = false;
    if (i < max_det) {
        int idx = 0;
        if (i < n) { idx = keep[i]; ok = best_score[idx] >= score_thr; }
#pragma unroll
        for (int q = 0; q < 4; ++q) out_boxes[i * 4 + q] = ok ? boxes[idx * 4 + q] : 0.f;
        out_cls[i] = ok ? (float)best_cls[idx] : 0.f;
        out_scores[i] = ok ? best_score[idx] : 0.f;
    }
    const unsigned long long b = __ballot(ok);
    if (i == 0) *out_count = __popcll(b);
}

// tools/ssd_mobilenet.py:111-150 (SSDMobileNet.predict after the four get_tensor calls) on the <= 64 rows the
// post-process op returns, one wave per image (lane 0 does the O(100) work): NaN scrub, score >= confidence,
// reorder + scale to pixels (f64), per-class nms_boxes with its own overlap formula (:59-98: +1 on the intersection
// extents only, areas without it, survivors ovr <= thr).  Classes are emitted in ascending id -- the reference
// walks a Python set; the order is irrelevant downstream (deep_sort's NMS re-sorts by score) -- and inside a
// class in pick order (descending score), exactly as nms_boxes returns them.
constexpr int FIN_MAX = 64;                // one lane per row of the op's output (max_detections of the model file)
// One wave per image, lane i = row i of the op's output (N <= 64 rows): the greedy loops run on wave reductions and
// broadcasts -- class by class in ascending id, inside a class the best remaining score first (HIGHEST row on ties: the reference
// reverses an ascending argsort, :73, which its NumPy 1.19 computes stably for <= 16 rows), every other row of the class tested against it in parallel.  (One lane walking all of it
// alone took 83 us per 384-image launch: a chain of dependent loads and branches.)  Same f64 expressions, same order of the
// emitted rows.
__device__ __forceinline__ double shfl_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl((int)(b & 0xffffffffLL), src, 64), hi = __shfl((int)(b >> 32), src, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__global__ __launch_bounds__(64) void ssd_finish_k(const float *__restrict__ boxes, const float *__restrict__ cls, const float *__restrict__ scores,
                                                   int max_det, double conf, double iou_thr, double img_w, double img_h, double *__restrict__ out_boxes,
                                                   int *__restrict__ out_cls, double *__restrict__ out_scores, int *__restrict__ out_n) {
    const int z = blockIdx.x, N = max_det, i = threadIdx.x;
    boxes += (size_t)z * N * 4; cls += (size_t)z * N; scores += (size_t)z * N;
    out_boxes += (size_t)z * N * 4; out_cls += (size_t)z * N; out_scores += (size_t)z * N;
    const bool in = i < N;
    float sc = in ? scores[i] : 0.f;
    const int cl = in ? (int)cls[i] : 0;
    float bf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bf[c] = in ? boxes[i * 4 + c] : 0.f;
    // :111-113 scores[np.where(np.isnan(boxes))] = 0 -- np.where yields (rows, cols) and BOTH index the score vector
    bool nan_row = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool nn = in && isnan(bf[c]);
        nan_row |= nn;
        if (__any(nn) && i == c && c < N) sc = 0.f;
    }
    if (nan_row) sc = 0.f;
    if (isnan(sc)) sc = 0.f;                                              // :115-116
    bool done = !(in && sc >= (float)conf);                               // :119 (rows below the confidence never take part)
    const double x1 = (double)bf[1] * img_w, y1 = (double)bf[0] * img_h;  // :121-127 reorder [1,0,3,2] * [w,h,w,h]
    const double x2 = (double)bf[3] * img_w, y2 = (double)bf[2] * img_h;
    int n = 0;
    for (;;) {
        int cmin = done ? 0x7fffffff : cl;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmin = min(cmin, __shfl_xor(cmin, o, 64));
        if (cmin == 0x7fffffff) break;                                    // wave-uniform
        bool cand = !done && cl == cmin;
        for (;;) {                                                        // greedy by descending score within the class
            float bs = cand ? sc : -__builtin_inff();
            int bi = cand ? i : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float os = __shfl_xor(bs, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (os > bs || (os == bs && oi > bi)) { bs = os; bi = oi; }      // equal scores: the higher row first (s.argsort()[::-1] of a stable sort)
            }
            if (bi < 0) break;
            if (i == bi) {
                out_boxes[n * 4 + 0] = x1; out_boxes[n * 4 + 1] = y1; out_boxes[n * 4 + 2] = x2; out_boxes[n * 4 + 3] = y2;
                out_cls[n] = cmin;
                out_scores[n] = (double)sc;
                done = true; cand = false;
            }
            ++n;
            const double bx1 = shfl_f64(x1, bi), by1 = shfl_f64(y1, bi), bx2 = shfl_f64(x2, bi), by2 = shfl_f64(y2, bi);
            if (cand) {
                const double x = bx1, y = by1, w = bx2 - bx1, h = by2 - by1;
                const double xj = x1, yj = y1, wj = x2 - x1, hj = y2 - y1;
                const double xx1 = fmax(x, xj), yy1 = fmax(y, yj);
                const double xx2 = fmin(x + w, xj + wj), yy2 = fmin(y + h, yj + hj);
                const double w1 = fmax(0.0, xx2 - xx1 + 1), h1 = fmax(0.0, yy2 - yy1 + 1);
                const double inter = w1 * h1;
                const double ovr = inter / (w * h + wj * hj - inter);
                if (!(ovr <= iou_thr)) { done = true; cand = false; }
            }
        }
    }
    if (i == 0) out_n[z] = n;
}

// tools/yolov5.py:120-131, first half: per-row confidence and class (x[..., 5:] *= x[..., 4:5]; np.argmax; take_along_axis).
// Half a wave per row: lane l reads columns l, l + 32, l + 64 of the row (coalesced 128-byte pieces; one thread per row walked
// its 340 bytes alone and the launch ran at 0.26 TB/s -- a third of the GPU time of the YOLOv5 pipeline), the best class is a
// 32-lane butterfly.  np.argmax semantics: the first maximum wins, and a NaN product is a maximum (the first NaN's index, confidence
// NaN -- which then fails `>= threshold` like upstream).
__global__ __launch_bounds__(256) void yolo_conf_k(const float *__restrict__ raw, int n_rows, int n_cls,
                                                   float *__restrict__ conf, int *__restrict__ cls) {
    const int lane = threadIdx.x & 31;
    const int n_cols = 5 + n_cls;
    const int groups = (gridDim.x * blockDim.x) >> 5;
    for (int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 5; r < n_rows; r += groups) {     // row uniform per half-wave
        const float *x = raw + (size_t)r * n_cols;
        const float obj = x[4];
        float best = -__builtin_inff();
        int bi = 0x7fffffff, nan_i = 0x7fffffff;
        for (int c0 = lane; c0 < n_cols; c0 += 96) {
            float v[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) v[q] = x[c0 + 32 * q < n_cols ? c0 + 32 * q : c0];      // all loads of the sweep first
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int col = c0 + 32 * q;
                if (col < 5 || col >= n_cols) continue;
                const float p = v[q] * obj;
                const int ci = col - 5;
                if (p != p) nan_i = min(nan_i, ci);
                if (bi == 0x7fffffff || p > best) { best = p; bi = ci; }          // the lane's own ascending scan: first value, then strict >
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64), on = __shfl_xor(nan_i, o, 64);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
            nan_i = min(nan_i, on);
        }
        if (lane == 0) {
            const bool has_nan = nan_i != 0x7fffffff;
            conf[r] = has_nan ? __builtin_nanf("") : best;
            cls[r] = has_nan ? nan_i : bi;
        }
    }
}

// second half: ordered compaction of rows with conf >= thr (ascending row order, like np.where)
__global__ __launch_bounds__(1024) void yolo_compact_k(const float *__restrict__ raw, const float *__restrict__ conf,
                                                       const int *__restrict__ cls, int n_rows, int row_floats, float thr,
                                                       float img_w, float img_h, float *__restrict__ out_boxes,
                                                       float *__restrict__ out_scores, int *__restrict__ out_cls,
                                                       int cap, int *__restrict__ out_n) {
    __shared__ int wave_cnt[16];
    __shared__ int base;
    {   // blockIdx.x = image of a batch: rows, per-row confidences and outputs of that image
        const size_t z = blockIdx.x;
        raw += z * n_rows * row_floats; conf += z * n_rows; cls += z * n_rows;      // row_floats: 5 + classes (the matrix) or 4 (boxes only)
        out_boxes += z * cap * 4; out_scores += z * cap; out_cls += z * cap; out_n += z;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < n_rows; r0 += 1024) {
        const int r = r0 + tid;
        const bool ok = r < n_rows && conf[r] >= thr;
        const unsigned long long b = __ballot(ok);
        if (lane == 0) wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int pos = off + __popcll(b & ((1ull << lane) - 1ull));
        if (ok && pos < cap) {
            const float *x = raw + (size_t)r * row_floats;
            const float x1 = x[0] - x[2] / 2, y1 = x[1] - x[3] / 2, x2 = x[0] + x[2] / 2, y2 = x[1] + x[3] / 2;
            out_boxes[pos * 4 + 0] = (float)((double)x1 * (double)img_w);
            out_boxes[pos * 4 + 1] = (float)((double)y1 * (double)img_h);
            out_boxes[pos * 4 + 2] = (float)((double)x2 * (double)img_w);
            out_boxes[pos * 4 + 3] = (float)((double)y2 * (double)img_h);
            out_scores[pos] = conf[r];
            out_cls[pos] = cls[r];
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wave_cnt[w];
            base += t;
        }
        __syncthreads();
    }
    if (tid == 0) *out_n = base;
}

// Rows of all images behind one another (image z starts at the sum of the counts before it): what the pipeline copies to
// the host is then proportional to what passed the threshold, not to a per-image capacity.  One block per image.
__global__ __launch_bounds__(256) void yolo_pack_k(const float *__restrict__ boxes, const float *__restrict__ scores, const int *__restrict__ cls,
                                                   const int *__restrict__ n_rows, int cap, float *__restrict__ packed) {
    __shared__ int part[256];
    const int z = blockIdx.x, tid = threadIdx.x;
    int acc = 0;
    for (int i = tid; i < z; i += 256) acc += min(n_rows[i], cap);
    part[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) part[tid] += part[tid + o]; __syncthreads(); }
    const int base = part[0], n = min(n_rows[z], cap);
    for (int i = tid; i < n; i += 256) {
        float *o = packed + (size_t)(base + i) * 6;
        const float *b = boxes + ((size_t)z * cap + i) * 4;
        o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
        o[4] = scores[(size_t)z * cap + i];
        o[5] = __int_as_float(cls[(size_t)z * cap + i]);
    }
}

__global__ void counts_add_k(long long *__restrict__ acc, const long long *__restrict__ add, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += add[i];
}

}  // namespace

namespace ddk {

// DD_NMS_SELECT=0: the op's NMS as a full sort of all anchors + nms_lazy_k (rounds 1-4) instead of nms_greedy_f32_k -- same rows, for A/B runs
static bool nms_select_on() {
    static const bool on = [] { const char *e = getenv("DD_NMS_SELECT"); return !(e && e[0] == '0'); }();
    return on;
}

size_t ssd_post_scratch_bytes(int n_anchors, int batch) {
    const size_t per = (size_t)n_anchors;
    const size_t head = (size_t)batch * per * (4 * 4 + 4 + 4 + 4 + 4) + (size_t)batch * 4 + 512;
    const size_t nms_bytes = (size_t)batch * (per * 64 + per * ((n_anchors + 63) / 64) * 8 + 1024);
    return head + nms_bytes + 512;
}

// raw [batch][n_anchors][4+n_classes] -> boxes [batch][max_det][4], classes, scores [batch][max_det], count [batch]
int ssd_postprocess(hipStream_t s, const float *raw, const float *anchors, int n_anchors, int n_classes, int max_det,
                    float score_thr, float iou_thr, float *boxes, float *classes, float *scores, int *count, int batch,
                    void *scratch, size_t scratch_bytes) {
    DD_REQUIRE(n_anchors > 64 && n_anchors <= 4096 && n_classes > 1 && max_det > 0 && max_det <= 64 && batch > 0, DD_E_ARG,
               "ssd_postprocess: bad shape (anchors %d, classes %d, max_det %d, batch %d)", n_anchors, n_classes, max_det, batch);
    DD_REQUIRE(scratch_bytes >= ssd_post_scratch_bytes(n_anchors, batch), DD_E_ARG, "ssd_postprocess: scratch too small");
    const size_t per = (size_t)n_anchors * batch;
    char *p = static_cast<char *>(scratch);
    float *d_boxes = reinterpret_cast<float *>(p);
    float *d_score = d_boxes + per * 4;
    int *d_cls = reinterpret_cast<int *>(d_score + per);
    float *d_keys = reinterpret_cast<float *>(d_cls + per);
    int *d_keep = reinterpret_cast<int *>(d_keys + per);
    int *d_nkeep = d_keep + per;
    const size_t head = ((per * 32 + (size_t)batch * 4) + 255) / 256 * 256;
    char *d_nms = p + head;
    hipLaunchKernelGGL(ssd_decode_k, dim3(dd_ceil_div(n_anchors, 16), batch), dim3(256), 0, s, raw, anchors, n_anchors, n_classes,
                       score_thr, d_boxes, d_score, d_cls, d_keys);
    DD_LAUNCH_CHECK();
    int rc;
    if (nms_select_on())
        rc = nms_f32_select_batched(s, d_boxes, d_score, n_anchors, score_thr, iou_thr, max_det, d_keep, d_nkeep, batch);
    else
        rc = nms_f32_batched(s, d_boxes, d_keys, n_anchors, iou_thr, max_det, d_keep, d_nkeep, d_nms, scratch_bytes - head, batch);
    if (rc != DD_OK) return rc;
    hipLaunchKernelGGL(ssd_gather_k, dim3(batch), dim3(64), 0, s, d_keep, d_nkeep, d_boxes, d_score, d_cls, score_thr, max_det,
                       n_anchors, boxes, classes, scores, count);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// The same op from the decoded per-anchor arrays (dd_net_ssd_decode: the detector's head GEMMs decode in their epilogue and
// the raw head matrix does not exist): d_boxes [batch][n_anchors][4], d_score / d_cls / d_keys [batch][n_anchors].
size_t ssd_post_decoded_scratch_bytes(int n_anchors, int batch) {
    return (size_t)batch * n_anchors * 4 + (size_t)batch * 4 + 512 + (size_t)batch * nms_scratch_bytes(n_anchors) + 512;
}

int ssd_postprocess_decoded(hipStream_t s, const float *d_boxes, const float *d_score, const int *d_cls, const float *d_keys,
                            int n_anchors, int max_det, float score_thr, float iou_thr, float *boxes, float *classes, float *scores,
                            int *count, int batch, void *scratch, size_t scratch_bytes) {
    DD_REQUIRE(n_anchors > 64 && n_anchors <= 4096 && max_det > 0 && max_det <= 64 && batch > 0, DD_E_ARG,
               "ssd_postprocess_decoded: bad shape (anchors %d, max_det %d, batch %d)", n_anchors, max_det, batch);
    DD_REQUIRE(scratch_bytes >= ssd_post_decoded_scratch_bytes(n_anchors, batch), DD_E_ARG, "ssd_postprocess_decoded: scratch too small");
    const size_t per = (size_t)n_anchors * batch;
    char *p = static_cast<char *>(scratch);
    int *d_keep = reinterpret_cast<int *>(p);
    int *d_nkeep = d_keep + per;
    const size_t head = ((per * 4 + (size_t)batch * 4) + 255) / 256 * 256;
    int rc;
    if (nms_select_on())
        rc = nms_f32_select_batched(s, d_boxes, d_score, n_anchors, score_thr, iou_thr, max_det, d_keep, d_nkeep, batch);
    else
        rc = nms_f32_batched(s, d_boxes, d_keys, n_anchors, iou_thr, max_det, d_keep, d_nkeep, p + head, scratch_bytes - head, batch);
    if (rc != DD_OK) return rc;
    hipLaunchKernelGGL(ssd_gather_k, dim3(batch), dim3(64), 0, s, d_keep, d_nkeep, d_boxes, d_score, d_cls, score_thr, max_det,
                       n_anchors, boxes, classes, scores, count);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// tools/yolov5.py:120-131 for `batch` images: raw f32 [batch][n_rows][5 + n_cls] -> per image the rows with
// conf >= thr in ascending row order (out_boxes f32 [batch][cap][4] xyxy pixels, out_scores, out_cls, out_n [batch] --
// out_n counts every passing row, also those beyond cap).  scratch: batch * n_rows * 8 bytes.
int yolov5_decode(hipStream_t s, const float *raw, int n_rows, int n_cls, float thr, float img_w, float img_h, float *out_boxes,
                  float *out_scores, int *out_cls, int cap, int *out_n, int batch, void *scratch) {
    float *conf = static_cast<float *>(scratch);
    int *cls = reinterpret_cast<int *>(conf + (size_t)batch * n_rows);
    const long long total = (long long)batch * n_rows;
    DD_REQUIRE(total < (1LL << 31), DD_E_CAPACITY, "yolov5_decode: %lld rows exceed 32-bit indexing", total);
    hipLaunchKernelGGL(yolo_conf_k, dim3((unsigned)std::min<long long>((total + 7) / 8, 256 * 32)), dim3(256), 0, s, raw, (int)total, n_cls, conf, cls);
    DD_LAUNCH_CHECK();
    hipLaunchKernelGGL(yolo_compact_k, dim3(batch), dim3(1024), 0, s, raw, conf, cls, n_rows, 5 + n_cls, thr, img_w, img_h, out_boxes,
                       out_scores, out_cls, cap, out_n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// The second half alone, on what the Detect heads wrote when they reduced their rows themselves (dd_net_yolo_decode): boxes f32
// [batch][n_rows][4] (x, y, w, h), conf, cls [batch][n_rows].
int yolov5_select(hipStream_t s, const float *boxes4, const float *conf, const int *cls, int n_rows, float thr, float img_w, float img_h,
                  float *out_boxes, float *out_scores, int *out_cls, int cap, int *out_n, int batch) {
    hipLaunchKernelGGL(yolo_compact_k, dim3(batch), dim3(1024), 0, s, boxes4, conf, cls, n_rows, 4, thr, img_w, img_h, out_boxes,
                       out_scores, out_cls, cap, out_n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// per-image [cap] outputs of yolov5_decode -> packed rows {x1, y1, x2, y2, score, class bits} of all images in image order
int yolov5_pack(hipStream_t s, const float *boxes, const float *scores, const int *cls, const int *n_rows, int cap, int batch, float *packed) {
    hipLaunchKernelGGL(yolo_pack_k, dim3(batch), dim3(256), 0, s, boxes, scores, cls, n_rows, cap, packed);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int ssd_finish(hipStream_t s, const float *boxes, const float *cls, const float *scores, int batch, int max_det, double conf,
               double iou_thr, double img_w, double img_h, double *out_boxes, int *out_cls, double *out_scores, int *out_n) {
    DD_REQUIRE(batch > 0 && max_det > 0 && max_det <= FIN_MAX, DD_E_ARG, "ssd_finish: batch %d, max_det %d (<= %d)", batch, max_det, FIN_MAX);
    hipLaunchKernelGGL(ssd_finish_k, dim3(batch), dim3(64), 0, s, boxes, cls, scores, max_det, conf, iou_thr, img_w, img_h,
                       out_boxes, out_cls, out_scores, out_n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace ddk

extern "C" {

int dd_ssd_detections(dd_ctx *ctx, const float *boxes, const float *classes, const float *scores, int batch, int max_det,
                      double confidence, double iou_thr, double img_w, double img_h, double *out_boxes, int *out_cls,
                      double *out_scores, int *out_n, void *stream) {
    DD_REQUIRE(ctx && boxes && classes && scores && out_boxes && out_cls && out_scores && out_n, DD_E_ARG,
               "dd_ssd_detections: NULL argument");
    DD_DEVICE(ctx);
    return ddk::ssd_finish(dd_pick_stream(ctx, stream), boxes, classes, scores, batch, max_det, confidence, iou_thr, img_w, img_h,
                           out_boxes, out_cls, out_scores, out_n);
}

int dd_ssd_postprocess(dd_ctx *ctx, const float *raw, const float *anchors, int n_anchors, int n_classes, int max_det,
                       float score_thr, float iou_thr, float *boxes, float *classes, float *scores, int *count,
                       void *stream) {
    DD_REQUIRE(ctx && raw && anchors && boxes && classes && scores && count, DD_E_ARG, "dd_ssd_postprocess: NULL argument");
    DD_DEVICE(ctx);
    int rc;
    const size_t need = ddk::ssd_post_scratch_bytes(n_anchors, 1);
    if ((rc = ctx->scratch[2].reserve(need)) != DD_OK) return rc;
    return ddk::ssd_postprocess(dd_pick_stream(ctx, stream), raw, anchors, n_anchors, n_classes, max_det, score_thr, iou_thr,
                                boxes, classes, scores, count, 1, ctx->scratch[2].p, ctx->scratch[2].cap);
}

// First stage of the op alone (what the detector's head layers do in their epilogue when dd_net_ssd_decode is on), for `batch`
// images: raw f32 [batch][n_anchors][4 + n_classes] -> boxes f32 [batch][n_anchors][4], scores, classes (id - 1), keys.
int dd_ssd_decode(dd_ctx *ctx, const float *raw, const float *anchors, int n_anchors, int n_classes, float score_thr, float *boxes,
                  float *scores, int *classes, float *keys, int batch, void *stream) {
    DD_REQUIRE(ctx && raw && anchors && boxes && scores && classes && keys && n_anchors > 0 && n_classes > 1 && batch > 0, DD_E_ARG,
               "dd_ssd_decode: bad argument");
    DD_DEVICE(ctx);
    hipLaunchKernelGGL(ssd_decode_k, dim3(dd_ceil_div(n_anchors, 16), batch), dim3(256), 0, dd_pick_stream(ctx, stream), raw, anchors,
                       n_anchors, n_classes, score_thr, boxes, scores, classes, keys);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// dd_ssd_postprocess from decoded per-anchor arrays (dd_net_ssd_decoded / dd_ssd_decode), `batch` images at once.
int dd_ssd_postprocess_decoded(dd_ctx *ctx, const float *dec_boxes, const float *dec_scores, const int *dec_classes, const float *dec_keys,
                               int n_anchors, int max_det, float score_thr, float iou_thr, float *boxes, float *classes, float *scores,
                               int *count, int batch, void *stream) {
    DD_REQUIRE(ctx && dec_boxes && dec_scores && dec_classes && dec_keys && boxes && classes && scores && count, DD_E_ARG,
               "dd_ssd_postprocess_decoded: NULL argument");
    DD_DEVICE(ctx);
    int rc;
    const size_t need = ddk::ssd_post_decoded_scratch_bytes(n_anchors, batch);
    if ((rc = ctx->scratch[2].reserve(need)) != DD_OK) return rc;
    return ddk::ssd_postprocess_decoded(dd_pick_stream(ctx, stream), dec_boxes, dec_scores, dec_classes, dec_keys, n_anchors, max_det,
                                        score_thr, iou_thr, boxes, classes, scores, count, batch, ctx->scratch[2].p, ctx->scratch[2].cap);
}

int dd_yolov5_decode(dd_ctx *ctx, const float *raw, int n_rows, int n_cls, float thr, float img_w, float img_h,
                     float *out_boxes, float *out_scores, int *out_cls, int cap, int *out_n, void *stream) {
    DD_REQUIRE(ctx && raw && out_boxes && out_scores && out_cls && out_n && n_rows >= 0 && n_cls > 0 && cap >= 0,
               DD_E_ARG, "dd_yolov5_decode: bad argument");
    DD_DEVICE(ctx);
    hipStream_t s = dd_pick_stream(ctx, stream);
    if (n_rows == 0) { DD_HIP(hipMemsetAsync(out_n, 0, sizeof(int), s)); return DD_OK; }
    int rc;
    if ((rc = ctx->scratch[2].reserve((size_t)n_rows * 8 + 256)) != DD_OK) return rc;
    return ddk::yolov5_decode(s, raw, n_rows, n_cls, thr, img_w, img_h, out_boxes, out_scores, out_cls, cap, out_n, 1,
                              ctx->scratch[2].p);
}

int dd_counts_accumulate(dd_ctx *ctx, int64_t *acc, const int64_t *counts_host, int n, void *stream) {
    DD_REQUIRE(ctx && acc && counts_host && n > 0, DD_E_ARG, "dd_counts_accumulate: bad argument");
    DD_DEVICE(ctx);
    hipStream_t s = dd_pick_stream(ctx, stream);
    int rc;
    if ((rc = ctx->scratch[1].reserve((size_t)n * 8)) != DD_OK) return rc;
    DD_HIP(hipMemcpyAsync(ctx->scratch[1].p, counts_host, (size_t)n * 8, hipMemcpyHostToDevice, s));
    DD_HIP(hipStreamSynchronize(s));
    hipLaunchKernelGGL(counts_add_k, dim3(dd_ceil_div(n, 64)), dim3(64), 0, s, reinterpret_cast<long long *>(acc),
                       ctx->scratch[1].as<long long>(), n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // extern "C"
