// Flat C ABI: context management and the per-function entry points of include/deepdish_hip.h
// (the handle-based entry points live next to their kernels: tracker.hip, nets.hip, pipeline.hip).
#include <cstdlib>
#include <mutex>
#include "common.h"

static thread_local char g_err[1024] = "";

void dd_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

// Small host->device staging copy for the flat entry points (tests / parity, not the hot path):
// synchronous so that caller memory and the pinned buffer can be reused immediately.
int upload(dd_ctx *ctx, hipStream_t s, void *dst, const void *src_host, size_t bytes) {
    if (!bytes) return DD_OK;
    DD_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, s));
    DD_HIP(hipStreamSynchronize(s));
    return DD_OK;
}

}  // namespace

extern "C" {

const char *dd_last_error(void) { return g_err; }

int dd_version(void) { return 100; }

int dd_ctx_create(int device, dd_ctx **out) {
    DD_REQUIRE(out, DD_E_ARG, "dd_ctx_create: NULL out");
    int ndev = 0;
    DD_HIP(hipGetDeviceCount(&ndev));
    DD_REQUIRE(device >= 0 && device < ndev, DD_E_ARG, "dd_ctx_create: device %d of %d", device, ndev);
    DD_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    DD_HIP(hipGetDeviceProperties(&prop, device));
    DD_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, DD_E_STATE,
               "dd_ctx_create: this library is built for gfx950 only, device is %s", prop.gcnArchName);
    dd_ctx *c = new dd_ctx();
    c->device = device;
    // DD_STREAM_PRIO=1 (experiment, off by default): the context's stream (the short kernels of a step the host waits for: NMS, crops,
    // association, Kalman updates) gets the highest stream priority and a pipeline's detector stream the lowest.  Measured on this
    // stack: no effect -- a 10 us NMS kernel still queues behind whatever part of the detector's train is resident (same-box A/B,
    // profiles/r03_ab_runs.txt) -- and a stream created with a priority makes a legacy-stream hipMemcpy of another thread fail while
    // it is being captured ("would make the legacy stream depend on a capturing blocking stream"), so latency-mode graphs need it off.
    {
        int least = 0, greatest = 0;
        const char *e = getenv("DD_STREAM_PRIO");
        DD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (e && atoi(e) == 1 && greatest != least) DD_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
        else DD_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    *out = c;
    return DD_OK;
}

int dd_ctx_destroy(dd_ctx *ctx) {
    if (!ctx) return DD_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->scratch) b.release();
    for (auto &b : ctx->pin) b.release();
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return DD_OK;
}

int dd_ctx_stream(dd_ctx *ctx, void **out_stream) {
    DD_REQUIRE(ctx && out_stream, DD_E_ARG, "dd_ctx_stream: NULL argument");
    *out_stream = reinterpret_cast<void *>(ctx->stream);
    return DD_OK;
}

int dd_ctx_sync(dd_ctx *ctx) {
    DD_REQUIRE(ctx, DD_E_ARG, "dd_ctx_sync: NULL ctx");
    DD_DEVICE(ctx);
    DD_HIP(hipStreamSynchronize(ctx->stream));
    return DD_OK;
}

int dd_kf_initiate(dd_ctx *ctx, double *means, double *covs, const int *slots, const double *xyah, int n,
                   void *stream) {
    DD_REQUIRE(ctx && means && covs && xyah && n >= 0, DD_E_ARG, "dd_kf_initiate: bad argument");
    DD_DEVICE(ctx);
    return ddk::kf_initiate(dd_pick_stream(ctx, stream), means, covs, slots, xyah, n);
}

int dd_kf_predict(dd_ctx *ctx, double *means, double *covs, const int *slots, int n, void *stream) {
    DD_REQUIRE(ctx && means && covs && n >= 0, DD_E_ARG, "dd_kf_predict: bad argument");
    DD_DEVICE(ctx);
    return ddk::kf_predict(dd_pick_stream(ctx, stream), means, covs, slots, n);
}

int dd_kf_project(dd_ctx *ctx, const double *means, const double *covs, const int *slots, int n,
                  double *proj_mean, double *proj_cov, void *stream) {
    DD_REQUIRE(ctx && means && covs && proj_mean && proj_cov && n >= 0, DD_E_ARG, "dd_kf_project: bad argument");
    DD_DEVICE(ctx);
    return ddk::kf_project(dd_pick_stream(ctx, stream), means, covs, slots, n, proj_mean, proj_cov);
}

int dd_kf_update(dd_ctx *ctx, double *means, double *covs, const int *slots, const double *xyah, int n,
                 void *stream) {
    DD_REQUIRE(ctx && means && covs && xyah && n >= 0, DD_E_ARG, "dd_kf_update: bad argument");
    DD_DEVICE(ctx);
    return ddk::kf_update(dd_pick_stream(ctx, stream), means, covs, slots, xyah, n);
}

int dd_kf_gate(dd_ctx *ctx, const double *means, const double *covs, const int *slots, int n,
               const double *xyah, int n_det, int only_position, double *out_d2, void *stream) {
    DD_REQUIRE(ctx && means && covs && n >= 0 && n_det >= 0, DD_E_ARG, "dd_kf_gate: bad argument");
    DD_DEVICE(ctx);
    DD_REQUIRE(n == 0 || n_det == 0 || (xyah && out_d2), DD_E_ARG, "dd_kf_gate: NULL argument");
    return ddk::kf_gate(dd_pick_stream(ctx, stream), means, covs, slots, n, xyah, n_det, only_position, out_d2);
}

int dd_iou_cost(dd_ctx *ctx, const double *tlwh_t, const int *tsu, int n_t, const double *tlwh_d, int n_d,
                double *out, void *stream) {
    DD_REQUIRE(ctx && n_t >= 0 && n_d >= 0, DD_E_ARG, "dd_iou_cost: bad argument");
    DD_DEVICE(ctx);
    DD_REQUIRE(n_t == 0 || n_d == 0 || (tlwh_t && tlwh_d && out), DD_E_ARG, "dd_iou_cost: NULL argument");
    return ddk::iou_cost(dd_pick_stream(ctx, stream), tlwh_t, tsu, n_t, tlwh_d, n_d, out);
}

int dd_cosine_nn_cost(dd_ctx *ctx, const float *gallery, const int *offsets_host, int n_t, const float *feats,
                      int n_d, double *out, void *stream) {
    DD_REQUIRE(ctx && n_t >= 0 && n_d >= 0, DD_E_ARG, "dd_cosine_nn_cost: bad argument");
    DD_DEVICE(ctx);
    if (n_t == 0 || n_d == 0) return DD_OK;
    DD_REQUIRE(gallery && offsets_host && feats && out, DD_E_ARG, "dd_cosine_nn_cost: NULL argument");
    hipStream_t s = dd_pick_stream(ctx, stream);
    const int g = offsets_host[n_t];
    for (int t = 0; t < n_t; ++t)
        DD_REQUIRE(offsets_host[t + 1] > offsets_host[t], DD_E_ARG, "dd_cosine_nn_cost: target %d has no samples", t);
    int rc;
    const size_t nbytes = ((size_t)g + n_d) * 128 * sizeof(float);
    const size_t ibytes = (size_t)n_t * (sizeof(long long) + sizeof(int));
    if ((rc = ctx->scratch[0].reserve(nbytes)) != DD_OK) return rc;
    if ((rc = ctx->scratch[1].reserve(ibytes)) != DD_OK) return rc;
    if ((rc = ctx->pin[0].reserve(ibytes)) != DD_OK) return rc;
    long long *h_start = ctx->pin[0].as<long long>();
    int *h_count = reinterpret_cast<int *>(h_start + n_t);
    for (int t = 0; t < n_t; ++t) {
        h_start[t] = offsets_host[t];
        h_count[t] = offsets_host[t + 1] - offsets_host[t];
    }
    if ((rc = upload(ctx, s, ctx->scratch[1].p, h_start, ibytes)) != DD_OK) return rc;
    float *gal_n = ctx->scratch[0].as<float>(), *feat_n = gal_n + (size_t)g * 128;
    if ((rc = ddk::normalize_rows(s, gallery, gal_n, g)) != DD_OK) return rc;
    if ((rc = ddk::normalize_rows(s, feats, feat_n, n_d)) != DD_OK) return rc;
    const long long *d_start = ctx->scratch[1].as<long long>();
    const int *d_count = reinterpret_cast<const int *>(d_start + n_t);
    return ddk::cosine_nn_cost(s, gal_n, d_start, d_count, n_t, feat_n, n_d, out, n_d);
}

static int nms_common(dd_ctx *ctx, const double *boxes, const double *keys, int k, double thr, int mode,
                      int *out_idx, int *out_n, void *stream, const char *who) {
    DD_REQUIRE(ctx && k >= 0 && out_n, DD_E_ARG, "%s: bad argument", who);
    DD_DEVICE(ctx);
    DD_REQUIRE(k == 0 || (boxes && keys && out_idx), DD_E_ARG, "%s: NULL argument", who);
    hipStream_t s = dd_pick_stream(ctx, stream);
    int rc;
    const size_t need = k > 0 ? ddk::nms_scratch_bytes(k) : 0;
    if (need && (rc = ctx->scratch[2].reserve(need)) != DD_OK) return rc;
    return ddk::nms(s, boxes, keys, k, thr, mode, out_idx, out_n, ctx->scratch[2].p, ctx->scratch[2].cap);
}

int dd_nms(dd_ctx *ctx, const double *tlwh, const double *keys, int k, double max_overlap, int *out_idx,
           int *out_n, void *stream) {
    return nms_common(ctx, tlwh, keys, k, max_overlap, 0, out_idx, out_n, stream, "dd_nms");
}

int dd_nms_ssd(dd_ctx *ctx, const double *xyxy, const double *scores, int k, double iou_thr, int *out_idx,
               int *out_n, void *stream) {
    return nms_common(ctx, xyxy, scores, k, iou_thr, 1, out_idx, out_n, stream, "dd_nms_ssd");
}

}  // extern "C"
