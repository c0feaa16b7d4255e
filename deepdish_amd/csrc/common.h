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
// tracker fused kernels
int lsap(const double *cost, int nr, int nc, int *rows, int *cols);   // host; returns pair count or -1
int gather_state(hipStream_t s, const double *means, const double *covs, const int *slots, int n,
                 double *out_means, double *out_covs);
}  // namespace ddk
