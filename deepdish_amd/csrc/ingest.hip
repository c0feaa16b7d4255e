// Frame ingest ring: the step in front of the hot path (SURVEY.md section 8 f, n1).
//
// Replaces the CPU side of Pipeline.capture (deepdish.py:837-878): frames arrive in host memory
// (cv2.VideoCapture), are optionally flipped (cv2.flip(frame, 0), :864) and stretched to the pipeline's
// input size (cv2.resize(frame, input_size), :867, INTER_LINEAR) before they reach the detector.  Here a
// decoder writes raw frames of S streams straight into a pinned slot; submit() queues the host->device
// copy on a private copy stream followed by the flip + resize on the GPU (the crop_resize_k restatement of
// cv2.resize; skipped when sizes match and no flip is asked), and the consumer stream waits on an event,
// never on the host.  With >= 2 slots the upload of step t+1 overlaps the kernels of step t.
#include <vector>
#include "common.h"

struct dd_ingest {
    dd_ctx *ctx = nullptr;
    int slots = 0, S = 0, sh = 0, sw = 0, dh = 0, dw = 0, flip = 0;
    bool transform = false;
    hipStream_t copy = nullptr;
    std::vector<uint8_t *> h_raw, d_raw, d_out;
    std::vector<hipEvent_t> ready, done;
    std::vector<char> used, submitted;      // done[slot] / ready[slot] has been recorded at least once
    void *d_boxes = nullptr;                // S full-frame CropBox records
    size_t raw_bytes = 0, out_bytes = 0;
};

extern "C" {

int dd_ingest_create(dd_ctx *ctx, int slots, int n_streams, int src_h, int src_w, int dst_h, int dst_w, int flip,
                     dd_ingest **out) {
    DD_REQUIRE(ctx && out && slots > 0 && n_streams > 0 && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, DD_E_ARG,
               "dd_ingest_create: bad argument");
    DD_HIP(hipSetDevice(ctx->device));
    dd_ingest *g = new dd_ingest();
    g->ctx = ctx; g->slots = slots; g->S = n_streams; g->sh = src_h; g->sw = src_w; g->dh = dst_h; g->dw = dst_w;
    g->flip = flip != 0;
    g->transform = g->flip || src_h != dst_h || src_w != dst_w;
    g->raw_bytes = (size_t)n_streams * src_h * src_w * 3;
    g->out_bytes = (size_t)n_streams * dst_h * dst_w * 3;
    DD_HIP(hipStreamCreateWithFlags(&g->copy, hipStreamNonBlocking));
    for (int i = 0; i < slots; ++i) {
        uint8_t *h = nullptr, *d = nullptr, *o = nullptr;
        DD_HIP(hipHostMalloc(reinterpret_cast<void **>(&h), g->raw_bytes, hipHostMallocDefault));
        DD_HIP(hipMalloc(reinterpret_cast<void **>(&d), g->raw_bytes + 64));
        if (g->transform) DD_HIP(hipMalloc(reinterpret_cast<void **>(&o), g->out_bytes + 64));
        else o = d;
        hipEvent_t r, dn;
        DD_HIP(hipEventCreateWithFlags(&r, hipEventDisableTiming));
        DD_HIP(hipEventCreateWithFlags(&dn, hipEventDisableTiming));
        g->h_raw.push_back(h); g->d_raw.push_back(d); g->d_out.push_back(o);
        g->ready.push_back(r); g->done.push_back(dn); g->used.push_back(0); g->submitted.push_back(0);
    }
    if (g->transform) {
        std::vector<int> boxes((size_t)n_streams * 8, 0);
        for (int z = 0; z < n_streams; ++z) {
            int *b = boxes.data() + (size_t)z * 8;
            b[0] = 0; b[1] = 0; b[2] = src_w; b[3] = src_h; b[4] = z; b[5] = g->flip;
        }
        DD_HIP(hipMalloc(&g->d_boxes, boxes.size() * sizeof(int)));
        DD_HIP(hipMemcpy(g->d_boxes, boxes.data(), boxes.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    *out = g;
    return DD_OK;
}

int dd_ingest_destroy(dd_ingest *g) {
    if (!g) return DD_OK;
    (void)hipStreamSynchronize(g->copy);
    for (int i = 0; i < g->slots; ++i) {
        (void)hipHostFree(g->h_raw[i]);
        if (g->transform) (void)hipFree(g->d_out[i]);
        (void)hipFree(g->d_raw[i]);
        (void)hipEventDestroy(g->ready[i]);
        (void)hipEventDestroy(g->done[i]);
    }
    if (g->d_boxes) (void)hipFree(g->d_boxes);
    (void)hipStreamDestroy(g->copy);
    delete g;
    return DD_OK;
}

int dd_ingest_host_slot(dd_ingest *g, int slot, uint8_t **host_ptr, int64_t *n_bytes) {
    DD_REQUIRE(g && host_ptr && slot >= 0 && slot < g->slots, DD_E_ARG, "dd_ingest_host_slot: bad argument");
    *host_ptr = g->h_raw[slot];
    if (n_bytes) *n_bytes = (int64_t)g->raw_bytes;
    return DD_OK;
}

// The host may refill a pinned slot once its previous upload has left it.
int dd_ingest_wait_uploaded(dd_ingest *g, int slot) {
    DD_REQUIRE(g && slot >= 0 && slot < g->slots, DD_E_ARG, "dd_ingest_wait_uploaded: bad slot");
    DD_DEVICE(g->ctx);
    if (g->submitted[slot]) DD_HIP(hipEventSynchronize(g->ready[slot]));
    return DD_OK;
}

int dd_ingest_submit(dd_ingest *g, int slot) {
    DD_REQUIRE(g && slot >= 0 && slot < g->slots, DD_E_ARG, "dd_ingest_submit: bad slot");
    DD_DEVICE(g->ctx);
    if (g->used[slot]) DD_HIP(hipStreamWaitEvent(g->copy, g->done[slot], 0));       // the previous consumer of this slot
    DD_HIP(hipMemcpyAsync(g->d_raw[slot], g->h_raw[slot], g->raw_bytes, hipMemcpyHostToDevice, g->copy));
    if (g->transform) {
        int rc = ddk::crop_resize(g->copy, g->d_raw[slot], g->sh, g->sw, g->d_boxes, g->S, g->dh, g->dw, g->d_out[slot]);
        if (rc != DD_OK) return rc;
    }
    DD_HIP(hipEventRecord(g->ready[slot], g->copy));
    g->submitted[slot] = 1;
    return DD_OK;
}

int dd_ingest_acquire(dd_ingest *g, int slot, void *consumer_stream, const uint8_t **frames_dev) {
    DD_REQUIRE(g && frames_dev && slot >= 0 && slot < g->slots, DD_E_ARG, "dd_ingest_acquire: bad argument");
    DD_DEVICE(g->ctx);
    DD_HIP(hipStreamWaitEvent(dd_pick_stream(g->ctx, consumer_stream), g->ready[slot], 0));
    *frames_dev = g->d_out[slot];
    return DD_OK;
}

int dd_ingest_release(dd_ingest *g, int slot, void *consumer_stream) {
    DD_REQUIRE(g && slot >= 0 && slot < g->slots, DD_E_ARG, "dd_ingest_release: bad slot");
    DD_DEVICE(g->ctx);
    DD_HIP(hipEventRecord(g->done[slot], dd_pick_stream(g->ctx, consumer_stream)));
    g->used[slot] = 1;
    return DD_OK;
}

}  // extern "C"
