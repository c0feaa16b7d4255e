// Shared host-side plumbing for libdeepdish_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <vector>
#include "../../include/deepdish_hip.h"

void dd_set_error(const char *fmt, ...);

#define DD_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            dd_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                         hipGetErrorString(e_));                                  \
            return DD_E_HIP;                                                      \
        }                                                                         \
    } while (0)

#define DD_REQUIRE(cond, code, ...)                                               \
    do {                                                                          \
        if (!(cond)) {                                                            \
            dd_set_error(__VA_ARGS__);                                            \
            return (code);                                                        \
        }                                                                         \
    } while (0)

#define DD_LAUNCH_CHECK() DD_HIP(hipGetLastError())

// Every entry point that launches or allocates selects its context's device first: a fresh host thread starts on
// device 0 whatever device the handle was created on.
#define DD_DEVICE(ctx) DD_HIP(hipSetDevice((ctx)->device))

// Growable device scratch buffer (never shrinks; growth happens outside graph capture).
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return DD_OK;
        if (p) { DD_HIP(hipFree(p)); p = nullptr; cap = 0; }
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        DD_HIP(hipMalloc(&p, want));
        cap = want;
        return DD_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return DD_OK;
        if (p) { DD_HIP(hipHostFree(p)); p = nullptr; cap = 0; }
        size_t want = bytes < 4096 ? 4096 : bytes + bytes / 4;
        DD_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return DD_OK;
    }
    void release() { if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; } }
    template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// One-time per-device set-up inside a launcher (hipFuncSetAttribute is a per-device property): run(device) returns
// true to exactly one caller per device, which does the set-up while the others wait for it.
#include <atomic>
#include <mutex>
struct DevOnce {
    std::atomic<uint64_t> done{0};
    std::mutex mu;
    template <class F> int run(int device, F &&f) {
        const uint64_t bit = 1ull << (device & 63);
        if (done.load(std::memory_order_acquire) & bit) return DD_OK;
        std::lock_guard<std::mutex> lk(mu);
        if (done.load(std::memory_order_relaxed) & bit) return DD_OK;
        const int rc = f();
        if (rc == DD_OK) done.fetch_or(bit, std::memory_order_release);
        return rc;
    }
};

struct dd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf scratch[4];     // general per-call scratch (index by purpose inside one call)
    PinBuf pin[2];
};

static inline hipStream_t dd_pick_stream(dd_ctx *ctx, void *stream) {
    return stream ? reinterpret_cast<hipStream_t>(stream) : ctx->stream;
}

static inline int dd_ceil_div(int a, int b) { return (a + b - 1) / b; }

// Compute units of a device (cached; 256 on MI355X): what the persistent one-workgroup-per-CU launches size their grids by.
static inline int dd_cu_count(int device) {
    static std::atomic<int> cache[64];
    int n = cache[device & 63].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
        cache[device & 63].store(n, std::memory_order_relaxed);
    }
    return n;
}

// XCD-aware block order (MI355X: 8 XCDs, each with a private 4 MiB L2; workgroups are dealt round-robin
// over them).  Maps the hardware's linear workgroup id to a logical id such that every XCD works on one
// contiguous 1/8 of the logical range: neighbouring tiles (shared halo rows, shared weight panels) then
// hit in the same L2 instead of being fetched by up to 8 of them.  Bijective for any count.
__device__ __forceinline__ unsigned dd_xcd_remap(unsigned id, unsigned n) {
    const unsigned q = n >> 3, r = n & 7u, xcd = id & 7u, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ---- kernel launchers shared between the flat C ABI and the tracker / pipeline handles ----
namespace ddk {
int kf_initiate(hipStream_t s, double *means, double *covs, const int *slots, const double *xyah, int n);
int kf_predict(hipStream_t s, double *means, double *covs, const int *slots, int n);
int kf_project(hipStream_t s, const double *means, const double *covs, const int *slots, int n,
               double *pm, double *pc);
int kf_update(hipStream_t s, double *means, double *covs, const int *slots, const double *xyah, int n);
int kf_gate(hipStream_t s, const double *means, const double *covs, const int *slots, int n,
            const double *xyah, int n_det, int only_position, double *out_d2);
int iou_cost(hipStream_t s, const double *tlwh_t, const int *tsu, int n_t, const double *tlwh_d, int n_d,
             double *out);
int normalize_rows(hipStream_t s, const float *in, float *out, int n);   // rows of 128 f32
// gallery rows are pre-normalised; target t owns rows [row_start[t], row_start[t]+row_count[t])
int cosine_nn_cost(hipStream_t s, const float *gallery_n, const long long *row_start, const int *row_count,
                   int n_t, const float *feats_n, int n_d, double *out, int ld_out);
int nms(hipStream_t s, const double *boxes, const double *keys, int k, double thr, int mode,
        int *out_idx, int *out_n, void *scratch, size_t scratch_bytes);
size_t nms_scratch_bytes(int k);
int nms_ex(hipStream_t s, const void *boxes, const void *keys, int k, double thr, int mode, int max_keep, int *out_idx,
           int *out_n, void *scratch, size_t scratch_bytes);
int nms_batched_small(hipStream_t s, const double *boxes, const double *keys, const int *d_offsets, int n_problems,
                      double thr, int mode, int *out_idx, int *out_n);
// tracker fused kernels
int crop_box_host(const int64_t *b, int ph, int pw, int H, int W, int *sx, int *sy, int *cw, int *ch);
int crop_box_host_f64(const double *b, int ph, int pw, int H, int W, int *sx, int *sy, int *cw, int *ch);
// d_boxes: device array of {sx, sy, cw, ch, frame, 0, 0, 0} int32 records
int crop_resize(hipStream_t s, const uint8_t *frames, int H, int W, const void *d_boxes, int n, int oh, int ow, uint8_t *out);
int resize_lanczos(hipStream_t s, int device, const uint8_t *src, int H, int W, int src_c, int swap_rb, uint8_t *dst,
                   int h, int w, uint8_t *tmp, int batch);
size_t ssd_post_scratch_bytes(int n_anchors, int batch);
int ssd_postprocess(hipStream_t s, const float *raw, const float *anchors, int n_anchors, int n_classes, int max_det,
                    float score_thr, float iou_thr, float *boxes, float *classes, float *scores, int *count, int batch,
                    void *scratch, size_t scratch_bytes);
void pyset_difference_order(const std::vector<int> &a, const std::vector<int> &b, std::vector<int> &out);
int lsap(const double *cost, int nr, int nc, int *rows, int *cols);   // host; returns pair count or -1
int gather_state(hipStream_t s, const double *means, const double *covs, const int *slots, int n,
                 double *out_means, double *out_covs);
}  // namespace ddk
