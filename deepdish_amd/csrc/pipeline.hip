// Multi-stream hot path: S independent video streams advanced one frame each per step, on one GPU.
//
// One step = the four calls the reference's Pipeline makes per frame (deepdish.py upstream):
// run_object_detector :880-885 -> box hygiene :940-960 -> non_max_suppression :995 -> encoder :1008 ->
// tracker.predict/update :1028-1029 -> count-line logic :1035-1114, for every stream, with the
// device work batched ACROSS streams (one detector forward for S frames, one crop launch and one MARS
// forward for all boxes of all streams, one batched NMS launch) and four host<->device round trips
// per STEP instead of ~10 per frame.  Streams share nothing (own tracker, ids, counters), exactly as
// independent DeepDish processes would.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <string>
#include "common.h"
#include "hostpool.h"
#include "host_phases.h"

struct dd_tracker;
struct dd_mog2;
namespace ddk {
int mog2_apply(dd_mog2 *m, hipStream_t s, const uint8_t *frames, double learning_rate, uint8_t *mask, uint8_t *masked);
int mask_box_count(hipStream_t s, const uint8_t *mask, int H, int W, const int *d_boxes, const int *d_box_stream, int K, int *d_counts);
int yolov5_decode(hipStream_t s, const float *raw, int n_rows, int n_cls, float thr, float img_w, float img_h, float *out_boxes,
                  float *out_scores, int *out_cls, int cap, int *out_n, int batch, void *scratch);
int yolov5_select(hipStream_t s, const float *boxes4, const float *conf, const int *cls, int n_rows, float thr, float img_w, float img_h,
                  float *out_boxes, float *out_scores, int *out_cls, int cap, int *out_n, int batch);
size_t ssd_post_decoded_scratch_bytes(int n_anchors, int batch);
int ssd_postprocess_decoded(hipStream_t s, const float *d_boxes, const float *d_score, const int *d_cls, const float *d_keys,
                            int n_anchors, int max_det, float score_thr, float iou_thr, float *boxes, float *classes, float *scores,
                            int *count, int batch, void *scratch, size_t scratch_bytes);
int yolov5_pack(hipStream_t s, const float *boxes, const float *scores, const int *cls, const int *n_rows, int cap, int batch, float *packed);
int ssd_finish(hipStream_t s, const float *boxes, const float *cls, const float *scores, int batch, int max_det, double conf,
               double iou_thr, double img_w, double img_h, double *out_boxes, int *out_cls, double *out_scores, int *out_n);
int tracker_group_create(dd_ctx *ctx, int n, double max_cos, double max_iou, int max_age, int n_init, int budget, int tcap,
                         int gcap, dd_tracker **out);
int trackers_predict(dd_tracker **ts, int S);
int trackers_update_begin(dd_tracker **ts, int S, const double *tlwh_host, const float *feats, int feats_on_device,
                          const int *det_off);
int trackers_update_match(dd_tracker **ts, int S);
int trackers_update_end(dd_tracker **ts, int S);
int tracker_read_host(dd_tracker *t, int which, int64_t *ints6_host, double *means_host);
}
extern "C" int dd_net_max_batch(dd_net *net, int *out_host);
extern "C" int dd_net_ssd_decode(dd_net *net, const float *anchors_host, int n_anchors, float score_thr, int enable);
extern "C" int dd_net_ssd_decoded(dd_net *net, float **boxes, float **scores, int **classes, float **keys);
extern "C" int dd_net_yolo_decode(dd_net *net, int enable);
extern "C" int dd_net_yolo_decoded(dd_net *net, float **boxes, float **conf, int **classes, int *rows);
extern "C" int dd_net_input_size(dd_net *net, int *h_host, int *w_host);

namespace {

constexpr int MAX_DET_DEFAULT = 10, MAX_DET_CAP = 64;          // max_detections of the stock SSD post-process op; what csrc/post.hip takes
constexpr float SSD_SCORE_THR = 1e-8f, SSD_IOU_THR = 0.6f;    // its score / NMS thresholds (dd_pipeline_ssd_options: the model file's own)
constexpr int YOLO_HOST_ROWS = 128;      // YOLOv5 rows per stream the first device-to-host copy of a step has room for (the rest, if any, follows)
enum { DET_SSD = 0, DET_YOLOV5 = 1, DET_TFLITE = 2 };
enum { CONFIRMED = 2, DELETED = 3 };

struct Votes {                         // track.py:78-81,147-151: label -> confidences, in first-seen order
    std::vector<int> cls;
    std::vector<int> cnt;
    std::vector<double> sum;
    void add(int c, double conf) {
        for (size_t i = 0; i < cls.size(); ++i) if (cls[i] == c) { cnt[i]++; sum[i] += conf; return; }
        cls.push_back(c); cnt.push_back(1); sum.push_back(conf);
    }
};

struct StreamState {
    dd_tracker *trk = nullptr;
    std::map<int64_t, std::vector<std::pair<double, double>>> db;    // deepdish.py:519 self.db
    std::map<int64_t, Votes> votes;
    std::vector<int64_t> counts;                                     // [n_wanted][4] pos, neg, int, del
    std::vector<int> det_cls;                                        // class of every detection of this step
    std::vector<double> det_conf;
    // per-step scratch of the host phases (kept across steps: no allocation in the steady state)
    std::vector<double> boxes0, scores0;                             // detector adaptor output: tlwh f64 rows, scores
    std::vector<int> cls0;
    std::vector<int64_t> ib;                                         // after box hygiene: int boxes, scores, classes
    std::vector<double> is;
    std::vector<int> ic, keep;
    std::vector<int64_t> ints;                                       // count line: track table rows
    std::vector<double> means;
};

using ddhost::cross2;
using ddhost::seg_intersect;

}  // namespace

struct dd_pipeline {
    dd_ctx *ctx = nullptr;
    int S = 0, H = 0, W = 0;
    dd_net *det = nullptr, *enc = nullptr;
    int det_kind = DET_SSD, det_in = 300, det_in_w = 300, n_anchors = 0, n_classes = 0, enc_batch = 0;
    int label_offset = 1;                      // class id c is line c + label_offset of the label file (SSD 1, YOLOv5 0)
    float *d_anchors = nullptr;
    double nms_overlap = 0.6, det_conf = 0.5;
    int max_det = MAX_DET_DEFAULT; float ssd_score_thr = SSD_SCORE_THR, ssd_iou_thr = SSD_IOU_THR;   // TFLite_Detection_PostProcess options (dd_pipeline_ssd_options)
    double line[4] = {0, 0, 0, 0};
    std::vector<std::string> labels;           // label file lines (index = class id + 1, ssd_mobilenet.py:142-147)
    std::vector<std::string> wanted;
    std::vector<StreamState> st;
    std::vector<dd_tracker *> trks;
    DevBuf d_resized, d_tmp, d_post, d_det, d_fin, d_pack, d_nms, d_crop, d_patches, d_feats;
    DevBuf d_tfl_boxes;                        // generic TFLite adaptor: one whole-frame CropBox per stream (bilinear stretch, BGR -> RGB)
    std::vector<std::string> tfl_labels;       // its label list: the label file's lines after the first, empty lines dropped (tflite.py:22, tflite_object_detector.py)
    bool det_late = true;                      // where the look-ahead detector run is queued (dd_pipeline_step2)
    bool ssd_dec = false;                      // SSD: the head layers decode in their epilogue (dd_net_ssd_decode)
    bool yolo_dec = false;                     // YOLOv5: the Detect layers reduce their rows in their epilogue (dd_net_yolo_decode)
    size_t yolo_host_rows = 0;                 // YOLOv5: packed rows the first copy of a step brings to the host
    std::vector<size_t> ybase;
    PinBuf h_fin, h_nms, h_crop;
    int crop_cap = 0;
    double t_det = 0, t_nms = 0, t_enc = 0, t_trk = 0;    // host wall seconds per stage, accumulated
    // GPU time per stage (dd_pipeline_stage_gpu_ms): HIP events on the streams the kernels run on -- the detector chain on its own stream
    // (Lanczos .. host copy; read when the step consumes it), and on the main stream the pairs 0 predict, 1 NMS, 2 crops + encoder,
    // 3 association, 4 Kalman update / gallery, 5 track management -- read lazily (at the next step or by the getter) so that no step
    // waits for anything it did not wait for before.  host = the step's wall time outside its waits for the GPU.
    hipEvent_t ev_det[2] = {nullptr, nullptr}, ev_main[12] = {};
    bool ev_pair[6] = {false, false, false, false, false, false}, ev_pending = false;
    double g_det = 0, g_nms = 0, g_enc = 0, g_trk = 0, g_host = 0, g_wall = 0;     // milliseconds, accumulated
    double step_wait = 0;                                                          // seconds of this step spent in stream / event waits
    long long steps = 0;
    // The detector has its own stream: like the reference, which keeps one detector call and one encoder call
    // in flight on different frames (deepdish.py:935,985,1008), the detector of frame t+1 can be queued while
    // frame t goes through NMS / encoder / tracker on the main stream.
    hipStream_t det_stream = nullptr;
    hipEvent_t det_done = nullptr, main_mark = nullptr;
    const uint8_t *det_pending = nullptr;      // frames the queued detector run belongs to
    // Background subtraction (deepdish.py:889,920-924,957; the reference's default, off in its benchmarks):
    // MOG2 over every frame on the main stream, then only boxes with enough moving pixels reach NMS.
    dd_mog2 *mog2 = nullptr;
    double motion_ratio = -1.0;                // --background-subtraction-ratio; < 0 = disabled
    bool bg_masking = false;                   // --enable-background-masking
    DevBuf d_mask, d_masked, d_mbox;
    PinBuf h_mbox;
    long long motion_rejected = 0;
    std::vector<int> off, doff;                // per-step prefix sums (candidates, kept detections) over the streams
    std::vector<double> tlwh;
};

namespace {
int flush_stage_events(dd_pipeline *p);

int wanted_index(const dd_pipeline *p, const std::string &name) {
    for (size_t i = 0; i < p->wanted.size(); ++i) if (p->wanted[i] == name) return (int)i;
    return -1;
}

std::string class_name(const dd_pipeline *p, int c) {
    if (c >= 0 && c + p->label_offset < (int)p->labels.size()) return p->labels[c + p->label_offset];
    return std::string();
}

// track.py:154-188 get_label (Dirichlet-multinomial expectation + the motorbike/bicycle rule)
std::string vote_label(const dd_pipeline *p, const Votes &v) {
    if (v.cls.empty()) return std::string();
    double csum = 0, asum = 0;
    std::vector<double> avg(v.cls.size());
    for (size_t i = 0; i < v.cls.size(); ++i) { avg[i] = v.sum[i] / v.cnt[i]; csum += v.cnt[i]; asum += avg[i]; }
    std::vector<std::pair<double, std::string>> e;
    for (size_t i = 0; i < v.cls.size(); ++i) e.emplace_back((avg[i] + v.cnt[i]) / (csum + asum), class_name(p, v.cls[i]));
    std::sort(e.begin(), e.end());
    std::reverse(e.begin(), e.end());
    if (e.size() > 1 && e[0].second == "motorbike" && e[1].second == "bicycle")
        return e[0].first > e[1].first * 4 ? "motorbike" : "bicycle";
    return e[0].second;
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

}  // namespace

extern "C" {

int dd_tracker_create(dd_ctx *, double, double, int, int, int, int, int, dd_tracker **);
int dd_tracker_destroy(dd_tracker *);
int dd_tracker_count(dd_tracker *, int, int *);
int dd_tracker_read(dd_tracker *, int, int64_t *, double *, double *);
int dd_net_forward(dd_net *, const uint8_t *, int, void *);
int dd_net_output(dd_net *, int, void **, int *, int *, int *, int *, int *);
int dd_net_read(dd_net *, int, int, void *, int, void *);
int dd_mog2_create(dd_ctx *, int, int, int, int, double, int, dd_mog2 **);
int dd_mog2_destroy(dd_mog2 *);

int dd_pipeline_create(dd_ctx *ctx, int n_streams, int frame_h, int frame_w, dd_net *detector,
                       const float *anchors_host, int n_anchors, int n_classes, dd_net *encoder,
                       const char *labels_nl, const char *wanted_nl, double max_cosine_distance, double nms_max_overlap,
                       double max_iou_distance, int max_age, int n_init, const double *line_host, int track_capacity,
                       int gallery_capacity, dd_pipeline **out) {
    DD_REQUIRE(ctx && encoder && out && n_streams > 0 && frame_h > 0 && frame_w > 0 && labels_nl && wanted_nl && line_host,
               DD_E_ARG, "dd_pipeline_create: bad argument");
    DD_REQUIRE(!detector || (n_anchors > 64 && n_classes > 1), DD_E_ARG, "dd_pipeline_create: detector needs its head shape");
    dd_pipeline *p = new dd_pipeline();
    p->ctx = ctx; p->S = n_streams; p->H = frame_h; p->W = frame_w;
    p->det = detector; p->enc = encoder; p->n_anchors = n_anchors; p->n_classes = n_classes;
    p->nms_overlap = nms_max_overlap;
    for (int i = 0; i < 4; ++i) p->line[i] = line_host[i];
    auto split = [](const char *s, std::vector<std::string> &out) {
        std::string cur;
        for (const char *c = s; *c; ++c) { if (*c == '\n') { out.push_back(cur); cur.clear(); } else cur.push_back(*c); }
        if (!cur.empty()) out.push_back(cur);
    };
    split(labels_nl, p->labels);
    split(wanted_nl, p->wanted);
    DD_HIP(hipSetDevice(ctx->device));
    int rc;
    if ((rc = dd_net_max_batch(encoder, &p->enc_batch)) != DD_OK) return rc;
    if (detector) {
        int db = 0;
        if ((rc = dd_net_max_batch(detector, &db)) != DD_OK) return rc;
        DD_REQUIRE(db >= n_streams, DD_E_CAPACITY, "dd_pipeline_create: detector max_batch %d < %d streams", db, n_streams);
        if ((rc = dd_net_input_size(detector, &p->det_in, &p->det_in_w)) != DD_OK) return rc;
        p->det_kind = anchors_host ? DET_SSD : DET_YOLOV5;
        if (p->det_kind == DET_YOLOV5) { p->label_offset = 0; p->det_conf = 0.25; }          // yolov5.py:38,134
        {   // the look-ahead detector run yields to the main stream's short kernels (see dd_ctx_create)
            int least = 0, greatest = 0;
            const char *e = getenv("DD_STREAM_PRIO");
            DD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            if (e && atoi(e) == 1 && greatest != least) DD_HIP(hipStreamCreateWithPriority(&p->det_stream, hipStreamNonBlocking, least));
            else DD_HIP(hipStreamCreateWithFlags(&p->det_stream, hipStreamNonBlocking));
        }
        DD_HIP(hipEventCreateWithFlags(&p->det_done, hipEventDisableTiming));
        DD_HIP(hipEventCreateWithFlags(&p->main_mark, hipEventDisableTiming));
        for (hipEvent_t &e : p->ev_det) DD_HIP(hipEventCreate(&e));
        const size_t S = n_streams;
        if ((rc = p->d_resized.reserve(S * p->det_in * p->det_in_w * 3)) != DD_OK) return rc;
        if ((rc = p->d_tmp.reserve(S * frame_h * p->det_in_w * 3 + 64)) != DD_OK) return rc;
        if (p->det_kind == DET_SSD) {
            DD_HIP(hipMalloc(&p->d_anchors, (size_t)n_anchors * 4 * sizeof(float)));
            DD_HIP(hipMemcpy(p->d_anchors, anchors_host, (size_t)n_anchors * 4 * sizeof(float), hipMemcpyHostToDevice));
            if ((rc = p->d_post.reserve(ddk::ssd_post_scratch_bytes(n_anchors, n_streams))) != DD_OK) return rc;
            // the op's first stage (best class, anchor decode, sigmoid, threshold) runs in the head layers' epilogues: the head
            // matrix is never written (DD_SSD_DEC=0 keeps the separate ssd_decode_k pass: same bits, for A/B runs)
            const char *e = getenv("DD_SSD_DEC");
            p->ssd_dec = !(e && atoi(e) == 0);
            if (p->ssd_dec && (rc = dd_net_ssd_decode(detector, anchors_host, n_anchors, SSD_SCORE_THR, 1)) != DD_OK) return rc;
            // sized for the largest max_detections the post-process kernels take: dd_pipeline_ssd_options may raise it before the first step
            if ((rc = p->d_det.reserve(S * MAX_DET_CAP * 6 * sizeof(float) + S * sizeof(int) + 256)) != DD_OK) return rc;
            if ((rc = p->d_fin.reserve(S * (MAX_DET_CAP * (4 * 8 + 4 + 8) + 4) + 256)) != DD_OK) return rc;
            if ((rc = p->h_fin.reserve(S * (MAX_DET_CAP * (4 * 8 + 4 + 8) + 4) + 256)) != DD_OK) return rc;
        } else {
            if ((rc = p->d_post.reserve(S * n_anchors * 8 + 256)) != DD_OK) return rc;              // per-row confidence + class
            // yolov5.py:126-128 (cls *= obj, argmax, confidence) runs in the Detect layers' epilogues when the program carries their
            // per-anchor weight copy: the [rows][85] matrix is never written (DD_YOLO_DEC=0: the separate yolo_conf_k pass, same bits)
            if (p->det_kind == DET_YOLOV5) {
                const char *e = getenv("DD_YOLO_DEC");
                p->yolo_dec = !(e && atoi(e) == 0) && dd_net_yolo_decode(detector, 1) == DD_OK;
            }
            // every row of the head may pass the threshold: yolov5.py:120-145 has no limit, so neither has this (device: per image
            // n_anchors rows of boxes f32x4, score, class + a count; then the same rows packed over the images)
            const size_t fin = S * ((size_t)n_anchors * 24 + 4) + 256;
            if ((rc = p->d_fin.reserve(fin)) != DD_OK) return rc;
            if ((rc = p->d_pack.reserve(S * (size_t)n_anchors * 24 + 256)) != DD_OK) return rc;
            p->yolo_host_rows = (size_t)S * YOLO_HOST_ROWS;
            if ((rc = p->h_fin.reserve(S * 4 + 64 + p->yolo_host_rows * 24)) != DD_OK) return rc;
        }
    }
    // late look-ahead (dd_pipeline_step2) when a step's kernels fill the GPU; with a handful of streams they do not, and the detector of
    // the next frame runs beside the encoder from the consume point on (one stream: 0.46 ms per frame early, 0.66 ms late)
    { const char *e = getenv("DD_DET_LATE"); p->det_late = e ? atoi(e) != 0 : n_streams >= 64; }
    p->st.resize(n_streams);
    p->trks.resize(n_streams);
    if ((rc = ddk::tracker_group_create(ctx, n_streams, max_cosine_distance, max_iou_distance, max_age, n_init, 0,
                                        track_capacity, gallery_capacity, p->trks.data())) != DD_OK) return rc;
    for (int z = 0; z < n_streams; ++z) {
        p->st[z].trk = p->trks[z];
        p->st[z].counts.assign(p->wanted.size() * 4, 0);
    }
    for (hipEvent_t &e : p->ev_main) DD_HIP(hipEventCreate(&e));
    *out = p;
    return DD_OK;
}

int dd_pipeline_destroy(dd_pipeline *p) {
    if (!p) return DD_OK;
    for (hipEvent_t e : p->ev_det) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_main) if (e) (void)hipEventDestroy(e);
    if (p->det_stream) { (void)hipStreamSynchronize(p->det_stream); (void)hipStreamDestroy(p->det_stream); }
    if (p->det_done) (void)hipEventDestroy(p->det_done);
    if (p->main_mark) (void)hipEventDestroy(p->main_mark);
    for (auto &s : p->st) dd_tracker_destroy(s.trk);
    (void)hipFree(p->d_anchors);
    dd_mog2_destroy(p->mog2);
    for (DevBuf *b : {&p->d_resized, &p->d_tmp, &p->d_post, &p->d_det, &p->d_fin, &p->d_pack, &p->d_tfl_boxes, &p->d_nms, &p->d_crop, &p->d_patches, &p->d_feats,
                      &p->d_mask, &p->d_masked, &p->d_mbox}) b->release();
    for (PinBuf *b : {&p->h_fin, &p->h_nms, &p->h_crop, &p->h_mbox}) b->release();
    delete p;
    return DD_OK;
}

// Which adaptor consumes an SSD-type detector (deepdish.py:482-502 picks the plugin by the model's file name): 0 =
// tools/ssd_mobilenet.py (the default of a pipeline created with anchors: Pillow Lanczos stretch, predict tail with per-class
// nms_boxes), 2 = the generic TFLite-Task adaptor (tools/tflite.py + tools/tflite_object_detector.py: cv2 bilinear stretch of
// the RGB frame, the post-process op's rows with score >= 0.5, int() corners, sorted by score).  Call before the first step.
// TFLite_Detection_PostProcess runs with the options the model file states (the interpreter at tools/ssd_mobilenet.py:100-109 upstream does):
// max_detections rows per frame, nms_score_threshold, nms_iou_threshold -- the stock export's 10 / 1e-8 / 0.6 until this is called.
int dd_pipeline_ssd_options(dd_pipeline *p, int max_detections, float nms_score_threshold, float nms_iou_threshold) {
    DD_REQUIRE(p && p->det && p->det_kind != DET_YOLOV5, DD_E_ARG, "dd_pipeline_ssd_options: needs a pipeline with an SSD-type detector");
    DD_REQUIRE(max_detections >= 1 && max_detections <= MAX_DET_CAP && nms_iou_threshold > 0.f && nms_iou_threshold <= 1.f, DD_E_ARG,
               "dd_pipeline_ssd_options: max_detections %d (1 .. %d), nms_iou_threshold %g (0 < t <= 1)", max_detections, MAX_DET_CAP, (double)nms_iou_threshold);
    DD_REQUIRE(p->steps == 0 && !p->det_pending, DD_E_STATE, "dd_pipeline_ssd_options: call before the first step");
    DD_DEVICE(p->ctx);
    if (p->ssd_dec && nms_score_threshold != p->ssd_score_thr) {   // the head layers' decode epilogue carries the score threshold
        std::vector<float> anchors((size_t)p->n_anchors * 4);
        DD_HIP(hipMemcpy(anchors.data(), p->d_anchors, anchors.size() * sizeof(float), hipMemcpyDeviceToHost));
        const int rc = dd_net_ssd_decode(p->det, anchors.data(), p->n_anchors, nms_score_threshold, 1);
        if (rc != DD_OK) return rc;
    }
    p->max_det = max_detections; p->ssd_score_thr = nms_score_threshold; p->ssd_iou_thr = nms_iou_threshold;
    return DD_OK;
}

int dd_pipeline_detector_adaptor(dd_pipeline *p, int adaptor) {
    DD_REQUIRE(p && p->det && p->det_kind != DET_YOLOV5 && (adaptor == DET_SSD || adaptor == DET_TFLITE), DD_E_ARG,
               "dd_pipeline_detector_adaptor: needs a pipeline with an SSD-type detector; adaptor 0 (ssd_mobilenet) or 2 (tflite)");
    DD_REQUIRE(p->steps == 0 && !p->det_pending, DD_E_STATE, "dd_pipeline_detector_adaptor: call before the first step");
    DD_DEVICE(p->ctx);
    p->det_kind = adaptor;
    if (adaptor == DET_TFLITE) {
        p->tfl_labels.clear();
        for (size_t i = 1; i < p->labels.size(); ++i) if (!p->labels[i].empty()) p->tfl_labels.push_back(p->labels[i]);
        std::vector<int> hb((size_t)p->S * 8, 0);
        for (int z = 0; z < p->S; ++z) { int *b = hb.data() + (size_t)z * 8; b[2] = p->W; b[3] = p->H; b[4] = z; b[6] = 1; }   // whole frame, swap_rb
        int rc;
        if ((rc = p->d_tfl_boxes.reserve(hb.size() * sizeof(int))) != DD_OK) return rc;
        DD_HIP(hipMemcpy(p->d_tfl_boxes.p, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    return DD_OK;
}

// deepdish.py:512,889: background subtraction on (ratio = --background-subtraction-ratio, default 0.25) or off
// (ratio < 0 = --disable-background-subtraction).  Turning it on starts a fresh model; masking =
// --enable-background-masking (the detector and the encoder then see cv2.bitwise_and(frame, frame, mask=fgMask)).
int dd_pipeline_background_subtraction(dd_pipeline *p, double ratio, int masking) {
    DD_REQUIRE(p && !(ratio > 1.0), DD_E_ARG, "dd_pipeline_background_subtraction: ratio must be <= 1 (negative = off)");
    DD_DEVICE(p->ctx);
    DD_HIP(hipStreamSynchronize(p->ctx->stream));
    if (p->det_stream) DD_HIP(hipStreamSynchronize(p->det_stream));
    p->det_pending = nullptr;                    // a detector run queued ahead is dropped: it may not match the new setting
    dd_mog2_destroy(p->mog2);
    p->mog2 = nullptr; p->motion_ratio = -1.0; p->bg_masking = false;
    if (ratio < 0) return DD_OK;
    int rc;
    if ((rc = dd_mog2_create(p->ctx, p->S, p->H, p->W, 500, 16.0, 1, &p->mog2)) != DD_OK) return rc;
    if ((rc = p->d_mask.reserve((size_t)p->S * p->H * p->W)) != DD_OK) return rc;
    if (masking && (rc = p->d_masked.reserve((size_t)p->S * p->H * p->W * 3)) != DD_OK) return rc;
    p->motion_ratio = ratio; p->bg_masking = masking != 0;
    return DD_OK;
}

// Foreground mask of the last step, u8 [S][H][W], copied to dst (host or device memory); dst may be NULL to read
// only the number of boxes the motion test has rejected so far.
int dd_pipeline_motion_mask(dd_pipeline *p, uint8_t *dst, int dst_on_device, long long *rejected_host) {
    DD_REQUIRE(p, DD_E_ARG, "dd_pipeline_motion_mask: NULL argument");
    DD_DEVICE(p->ctx);
    if (rejected_host) *rejected_host = p->motion_rejected;
    if (!dst) return DD_OK;
    DD_REQUIRE(p->mog2, DD_E_ARG, "dd_pipeline_motion_mask: background subtraction is off");
    DD_HIP(hipMemcpyAsync(dst, p->d_mask.p, (size_t)p->S * p->H * p->W, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                          p->ctx->stream));
    DD_HIP(hipStreamSynchronize(p->ctx->stream));
    return DD_OK;
}

int dd_pipeline_tracker(dd_pipeline *p, int stream, dd_tracker **out) {
    DD_REQUIRE(p && out && stream >= 0 && stream < p->S, DD_E_ARG, "dd_pipeline_tracker: bad argument");
    *out = p->st[stream].trk;
    return DD_OK;
}

int dd_pipeline_counts(dd_pipeline *p, int64_t *counts_host) {
    DD_REQUIRE(p && counts_host, DD_E_ARG, "dd_pipeline_counts: NULL argument");
    const size_t n = p->wanted.size() * 4;
    for (int s = 0; s < p->S; ++s) memcpy(counts_host + (size_t)s * n, p->st[s].counts.data(), n * sizeof(int64_t));
    return DD_OK;
}

int dd_pipeline_detector_stream(dd_pipeline *p, void **stream_out) {
    DD_REQUIRE(p && stream_out, DD_E_ARG, "dd_pipeline_detector_stream: NULL argument");
    *stream_out = p->det ? reinterpret_cast<void *>(p->det_stream) : nullptr;
    return DD_OK;
}

int dd_pipeline_stage_gpu_ms(dd_pipeline *p, double *out6_host, long long *steps_host) {
    DD_REQUIRE(p && out6_host, DD_E_ARG, "dd_pipeline_stage_gpu_ms: NULL argument");
    DD_DEVICE(p->ctx);
    const int rc = flush_stage_events(p);
    if (rc != DD_OK) return rc;
    out6_host[0] = p->g_det; out6_host[1] = p->g_nms; out6_host[2] = p->g_enc; out6_host[3] = p->g_trk; out6_host[4] = p->g_host; out6_host[5] = p->g_wall;
    if (steps_host) *steps_host = p->steps;
    return DD_OK;
}

int dd_pipeline_detections(dd_pipeline *p, int stream, double *boxes_host, double *scores_host, int *classes_host, int cap, int *n_host) {
    DD_REQUIRE(p && n_host && stream >= 0 && stream < p->S && cap >= 0, DD_E_ARG, "dd_pipeline_detections: bad argument");
    const StreamState &q = p->st[stream];
    const int n = (int)q.scores0.size();
    *n_host = n;
    DD_REQUIRE(n <= cap || (!boxes_host && !scores_host && !classes_host), DD_E_CAPACITY, "dd_pipeline_detections: %d rows, room for %d", n, cap);
    if (boxes_host) memcpy(boxes_host, q.boxes0.data(), (size_t)n * 4 * sizeof(double));
    if (scores_host) memcpy(scores_host, q.scores0.data(), (size_t)n * sizeof(double));
    if (classes_host) memcpy(classes_host, q.cls0.data(), (size_t)n * sizeof(int));
    return DD_OK;
}

int dd_pipeline_stage_seconds(dd_pipeline *p, double *out4_host, long long *steps_host) {
    DD_REQUIRE(p && out4_host, DD_E_ARG, "dd_pipeline_stage_seconds: NULL argument");
    out4_host[0] = p->t_det; out4_host[1] = p->t_nms; out4_host[2] = p->t_enc; out4_host[3] = p->t_trk;
    if (steps_host) *steps_host = p->steps;
    return DD_OK;
}

}  // extern "C"

namespace {

// resize -> forward -> post-process -> adaptor tail -> pinned host block, all streams at once, on the
// detector stream; det_done fires when the host block is complete.
// Stage events of the previous step -> the accumulated GPU milliseconds (they completed long ago unless the caller asks right after a step).
int flush_stage_events(dd_pipeline *p) {
    if (!p->ev_pending) return DD_OK;
    p->ev_pending = false;
    double *into[6] = {&p->g_trk, &p->g_nms, &p->g_enc, &p->g_trk, &p->g_trk, &p->g_trk};
    for (int k = 0; k < 6; ++k) {
        if (!p->ev_pair[k]) continue;
        p->ev_pair[k] = false;
        float ms = 0.f;
        DD_HIP(hipEventSynchronize(p->ev_main[2 * k + 1]));
        DD_HIP(hipEventElapsedTime(&ms, p->ev_main[2 * k], p->ev_main[2 * k + 1]));
        *into[k] += (double)ms;
    }
    return DD_OK;
}
// DD_STAGE_EVENTS=0: no stage events (dd_pipeline_stage_gpu_ms then reports zeros for the GPU stages); fourteen event records and seven
// reads per step are ~10 us of host time -- noise against a multi-millisecond batched step, a few % of a single stream's 0.45 ms frame
static const bool g_stage_events = !(getenv("DD_STAGE_EVENTS") && atoi(getenv("DD_STAGE_EVENTS")) == 0);
#define DD_STAGE_BEGIN(k) do { if (g_stage_events) DD_HIP(hipEventRecord(p->ev_main[2 * (k)], s)); } while (0)
#define DD_STAGE_END(k) do { if (g_stage_events) { DD_HIP(hipEventRecord(p->ev_main[2 * (k) + 1], s)); p->ev_pair[k] = true; } } while (0)
#define DD_TIMED_WAIT(expr) do { const double w0_ = now_s(); DD_HIP(expr); p->step_wait += now_s() - w0_; } while (0)

int enqueue_detector(dd_pipeline *p, const uint8_t *frames) {
    hipStream_t s = p->det_stream;
    const int S = p->S, MAX_DET = p->max_det;
    int rc;
    // order after whatever the caller has queued on the main stream so far (e.g. the ingest ring's wait for the
    // upload of these frames)
    DD_HIP(hipEventRecord(p->main_mark, p->ctx->stream));
    DD_HIP(hipStreamWaitEvent(s, p->main_mark, 0));
    if (g_stage_events) DD_HIP(hipEventRecord(p->ev_det[0], s));
    if (p->det_kind == DET_TFLITE) {                           // tflite_object_detector.py:207-211: cv2.resize (INTER_LINEAR) of the RGB frame
        if ((rc = ddk::crop_resize(s, frames, p->H, p->W, p->d_tfl_boxes.p, S, p->det_in, p->det_in_w, p->d_resized.as<uint8_t>())) != DD_OK) return rc;
    } else if ((rc = ddk::resize_lanczos(s, p->ctx->device, frames, p->H, p->W, 3, 1, p->d_resized.as<uint8_t>(), p->det_in,
                                         p->det_in_w, p->d_tmp.as<uint8_t>(), S)) != DD_OK) return rc;    // ssd_mobilenet.py:54-57, yolov5.py:99
    if ((rc = dd_net_forward(p->det, p->d_resized.as<uint8_t>(), S, s)) != DD_OK) return rc;       // :102-103 / yolov5.py:107-109
    void *raw = nullptr;
    if ((rc = dd_net_output(p->det, -1, &raw, nullptr, nullptr, nullptr, nullptr, nullptr)) != DD_OK) return rc;
    if (p->det_kind == DET_YOLOV5) {                                                               // yolov5.py:120-131
        const size_t cap = (size_t)p->n_anchors;
        float *yb = p->d_fin.as<float>(), *ys = yb + S * cap * 4;
        int *yc = reinterpret_cast<int *>(ys + S * cap), *yn = yc + S * cap;
        if (p->yolo_dec) {                                       // :126-128 ran in the Detect layers' epilogues
            float *b4 = nullptr, *cf = nullptr; int *cl = nullptr;
            if ((rc = dd_net_yolo_decoded(p->det, &b4, &cf, &cl, nullptr)) != DD_OK) return rc;
            if ((rc = ddk::yolov5_select(s, b4, cf, cl, p->n_anchors, (float)p->det_conf, (float)p->W, (float)p->H, yb, ys, yc,
                                         p->n_anchors, yn, S)) != DD_OK) return rc;
        } else
        if ((rc = ddk::yolov5_decode(s, static_cast<const float *>(raw), p->n_anchors, p->n_classes, (float)p->det_conf, (float)p->W,
                                     (float)p->H, yb, ys, yc, p->n_anchors, yn, S, p->d_post.p)) != DD_OK) return rc;
        // the passing rows of all streams packed behind one another; the host block is [S counts | pad to 64 B | rows x 24 B]: the
        // counts and the first yolo_host_rows rows now, whatever lies beyond them when the step consumes the block
        if ((rc = ddk::yolov5_pack(s, yb, ys, yc, yn, p->n_anchors, S, p->d_pack.as<float>())) != DD_OK) return rc;
        const size_t head = ((size_t)S * 4 + 63) / 64 * 64;
        DD_HIP(hipMemcpyAsync(p->h_fin.p, yn, (size_t)S * 4, hipMemcpyDeviceToHost, s));
        DD_HIP(hipMemcpyAsync(p->h_fin.as<char>() + head, p->d_pack.p, p->yolo_host_rows * 24, hipMemcpyDeviceToHost, s));
        if (g_stage_events) DD_HIP(hipEventRecord(p->ev_det[1], s));
        DD_HIP(hipEventRecord(p->det_done, s));
        p->det_pending = frames;
        return DD_OK;
    }
    float *db = p->d_det.as<float>(), *dc = db + (size_t)S * MAX_DET * 4, *ds = dc + (size_t)S * MAX_DET;
    int *dn = reinterpret_cast<int *>(ds + (size_t)S * MAX_DET);
    if (p->ssd_dec) {
        float *eb = nullptr, *es = nullptr, *ek = nullptr;
        int *ec = nullptr;
        if ((rc = dd_net_ssd_decoded(p->det, &eb, &es, &ec, &ek)) != DD_OK) return rc;
        if ((rc = ddk::ssd_postprocess_decoded(s, eb, es, ec, ek, p->n_anchors, MAX_DET, p->ssd_score_thr, p->ssd_iou_thr, db, dc, ds, dn, S,
                                               p->d_post.p, p->d_post.cap)) != DD_OK) return rc;
    } else if ((rc = ddk::ssd_postprocess(s, static_cast<const float *>(raw), p->d_anchors, p->n_anchors, p->n_classes, MAX_DET,
                                          p->ssd_score_thr, p->ssd_iou_thr, db, dc, ds, dn, S, p->d_post.p, p->d_post.cap)) != DD_OK) return rc;
    if (p->det_kind == DET_TFLITE) {                           // the generic adaptor's tail is a few integer truncations: on the host (step2)
        const size_t dbytes = (size_t)S * MAX_DET * 6 * sizeof(float) + (size_t)S * sizeof(int);
        DD_HIP(hipMemcpyAsync(p->h_fin.p, p->d_det.p, dbytes, hipMemcpyDeviceToHost, s));
        if (g_stage_events) DD_HIP(hipEventRecord(p->ev_det[1], s));
        DD_HIP(hipEventRecord(p->det_done, s));
        p->det_pending = frames;
        return DD_OK;
    }
    double *fb = p->d_fin.as<double>(), *fs = fb + (size_t)S * MAX_DET * 4;
    int *fc = reinterpret_cast<int *>(fs + (size_t)S * MAX_DET), *fn = fc + (size_t)S * MAX_DET;
    if ((rc = ddk::ssd_finish(s, db, dc, ds, S, MAX_DET, p->det_conf, 0.5, (double)p->W, (double)p->H, fb, fc, fs, fn)) != DD_OK)
        return rc;                                                                                    // :111-150
    const size_t fbytes = (size_t)S * (MAX_DET * (4 * 8 + 8 + 4) + 4);
    DD_HIP(hipMemcpyAsync(p->h_fin.p, p->d_fin.p, fbytes, hipMemcpyDeviceToHost, s));
    if (g_stage_events) DD_HIP(hipEventRecord(p->ev_det[1], s));
    DD_HIP(hipEventRecord(p->det_done, s));
    p->det_pending = frames;
    return DD_OK;
}

// Count-line logic of one stream after its tracker update (deepdish.py:1035-1114, 1303-1312); host only.
void count_line_one_stream(dd_pipeline *p, StreamState &st) {
    int nd = 0, nl = 0;
    dd_tracker_count(st.trk, 1, &nd);
    dd_tracker_count(st.trk, 0, &nl);
    std::vector<int64_t> &ints = st.ints;
    std::vector<double> &means = st.means;
    std::map<std::string, int> delcounts;
    if (nd) {
        ints.resize((size_t)nd * 6);
        ddk::tracker_read_host(st.trk, 1, ints.data(), nullptr);
        for (int i = 0; i < nd; ++i) {
            const int64_t id = ints[(size_t)i * 6];
            delcounts.clear();                                  // overwritten per deleted track (:1040-1044)
            auto it = st.db.find(id);
            if (it != st.db.end() && it->second.size() > 1) {
                bool hit = false;
                for (size_t q = 0; q + 1 < it->second.size() && !hit; ++q) {
                    const double a[2] = {it->second[q].first, it->second[q].second};
                    const double b[2] = {it->second[q + 1].first, it->second[q + 1].second};
                    hit = seg_intersect(p->line, p->line + 2, a, b);
                }
                if (hit) delcounts[vote_label(p, st.votes[id])] += 1;
                it->second.clear();
            }
            st.votes.erase(id);
        }
    }
    if (nl) {
        ints.resize((size_t)nl * 6);
        means.resize((size_t)nl * 8);
        ddk::tracker_read_host(st.trk, 0, ints.data(), means.data());
    }
    std::vector<std::pair<std::string, double>> events;
    for (int i = 0; i < nl; ++i) {
        const int64_t *r = ints.data() + (size_t)i * 6;
        const int64_t id = r[0];
        if (r[5] >= 0) st.votes[id].add(st.det_cls[r[5]], st.det_conf[r[5]]);
        if (r[1] != CONFIRMED || r[2] > 1) continue;
        const double *m = means.data() + (size_t)i * 8;
        const double w = m[2] * m[3];                            // track.py:84-111 to_tlbr
        const double x1 = m[0] - w / 2, y1 = m[1] - m[3] / 2;
        const double x2 = x1 + w, y2 = y1 + m[3];
        auto &pts = st.db[id];
        pts.emplace_back((x1 + x2) / 2.0, y2);
        if (pts.size() > 1) {
            const double p2[2] = {pts.back().first, pts.back().second};
            const double q2[2] = {pts[pts.size() - 2].first, pts[pts.size() - 2].second};
            const double cp = cross2(p->line[2] - p->line[0], p->line[3] - p->line[1], q2[0] - p2[0], q2[1] - p2[1]);
            if (seg_intersect(p->line, p->line + 2, p2, q2)) events.emplace_back(vote_label(p, st.votes[id]), cp);
        }
    }
    for (auto &e : events) {
        const int wi = wanted_index(p, e.first);
        if (wi < 0) continue;
        st.counts[wi * 4 + (e.second >= 0 ? 0 : 1)] += 1;
        st.counts[wi * 4 + 2] += 1;
    }
    for (auto &d : delcounts) {
        const int wi = wanted_index(p, d.first);
        if (wi >= 0) st.counts[wi * 4 + 3] += d.second;
    }
}

}  // namespace

extern "C" {

// frames: device u8 [S][H][W][3] BGR.  inj_*: optional injected detections that REPLACE the detector's
// output (the detector still runs): boxes tlwh as the detector adaptor would return them (f64),
// scores, class ids, stream s owns rows [inj_offsets[s], inj_offsets[s+1]).
// frames_next (optional): the frames of the following step; their detector run is queued on the detector
// stream as soon as this step has read its own detections, and overlaps this step's NMS / encoder / tracker.
int dd_pipeline_step2(dd_pipeline *p, const uint8_t *frames, const uint8_t *frames_next, const double *inj_boxes_host,
                      const double *inj_scores_host, const int *inj_cls_host, const int *inj_offsets_host);

int dd_pipeline_step(dd_pipeline *p, const uint8_t *frames, const double *inj_boxes_host, const double *inj_scores_host,
                     const int *inj_cls_host, const int *inj_offsets_host) {
    return dd_pipeline_step2(p, frames, nullptr, inj_boxes_host, inj_scores_host, inj_cls_host, inj_offsets_host);
}

int dd_pipeline_step2(dd_pipeline *p, const uint8_t *frames_in, const uint8_t *frames_next, const double *inj_boxes_host,
                      const double *inj_scores_host, const int *inj_cls_host, const int *inj_offsets_host) {
    DD_REQUIRE(p && frames_in, DD_E_ARG, "dd_pipeline_step: NULL argument");
    DD_REQUIRE(frames_next != frames_in, DD_E_ARG,
               "dd_pipeline_step2: frames_next must be a different buffer from frames (its detector run is queued while "
               "this step still reads frames, and the next step consumes it by address)");
    DD_DEVICE(p->ctx);
    const uint8_t *frames = frames_in;
    hipStream_t s = p->ctx->stream;
    const int S = p->S, MAX_DET = p->max_det;
    int rc;
    if ((rc = flush_stage_events(p)) != DD_OK) return rc;
    const double t0 = now_s();
    p->step_wait = 0;
    DD_STAGE_BEGIN(0);
    if ((rc = ddk::trackers_predict(p->trks.data(), S)) != DD_OK) return rc;            // deepdish.py:1028
    DD_STAGE_END(0);
    if (p->mog2) {                                                                      // :920-924
        if ((rc = ddk::mog2_apply(p->mog2, s, frames_in, -1.0, p->d_mask.as<uint8_t>(),
                                  p->bg_masking ? p->d_masked.as<uint8_t>() : nullptr)) != DD_OK) return rc;
        if (p->bg_masking) { frames = p->d_masked.as<uint8_t>(); frames_next = nullptr; p->det_pending = nullptr; }
    }

    // ---------------- detector: resize -> forward -> post-process -> adaptor tail, all streams at once
    // (host phases below: one parallel_for over the streams each -- they share nothing; hostpool.h)
    constexpr int GRAIN = 16;
    std::vector<StreamState> &st = p->st;
    if (p->det) {
        if (p->det_pending != frames && (rc = enqueue_detector(p, frames)) != DD_OK) return rc;       // not queued ahead: run it now
        DD_TIMED_WAIT(hipEventSynchronize(p->det_done));                                               // round trip 1
        p->det_pending = nullptr;
        if (g_stage_events) {   // the detector chain of these frames on its stream: resize, forward, post-process, adaptor tail, host copy
            float ms = 0.f;
            DD_HIP(hipEventElapsedTime(&ms, p->ev_det[0], p->ev_det[1]));
            p->g_det += (double)ms;
        }
        if (p->det_kind == DET_YOLOV5) {
            const size_t head = ((size_t)S * 4 + 63) / 64 * 64;
            const int *yn = p->h_fin.as<int>();
            std::vector<size_t> &ybase = p->ybase;
            ybase.assign(S + 1, 0);
            for (int z = 0; z < S; ++z) ybase[z + 1] = ybase[z] + (size_t)std::min(yn[z], p->n_anchors);
            if (ybase[S] > p->yolo_host_rows) {                    // a busy step: fetch the rest (the detector stream is idle here) and
                const size_t have = p->yolo_host_rows, total = ybase[S];      // make room for twice as many from now on
                // ... but never more than the packed block holds (S * n_anchors rows): the first copy of every later step reads that many
                const size_t grown = std::min(2 * total, (size_t)S * (size_t)p->n_anchors);
                PinBuf bigger;
                if ((rc = bigger.reserve(head + grown * 24)) != DD_OK) return rc;
                memcpy(bigger.p, p->h_fin.p, head + have * 24);
                hipError_t he = hipMemcpyAsync(bigger.as<char>() + head + have * 24, p->d_pack.as<char>() + have * 24, (total - have) * 24,
                                               hipMemcpyDeviceToHost, p->det_stream);
                if (he == hipSuccess) he = hipStreamSynchronize(p->det_stream);
                if (he != hipSuccess) { bigger.release(); DD_HIP(he); }
                p->h_fin.release();
                p->h_fin = bigger;
                p->yolo_host_rows = grown;
                yn = p->h_fin.as<int>();
            }
            const float *rows = reinterpret_cast<const float *>(p->h_fin.as<char>() + head);
            ddk::parallel_for(S, GRAIN, [&](int z0, int z1) {
                for (int z = z0; z < z1; ++z) {
                    StreamState &q = st[z];
                    q.boxes0.clear(); q.scores0.clear(); q.cls0.clear();
                    const int n = (int)(ybase[z + 1] - ybase[z]);
                    for (int i = 0; i < n; ++i) {                                                      // yolov5.py:137-145
                        const float *b = rows + (ybase[z] + i) * 6;
                        int c;
                        memcpy(&c, b + 5, 4);
                        const float sc = b[4];
                        if (wanted_index(p, class_name(p, c)) < 0 || !(sc >= (float)p->det_conf)) continue;
                        q.boxes0.insert(q.boxes0.end(), {(double)b[0], (double)b[1], (double)(b[2] - b[0]), (double)(b[3] - b[1])});   // f32 arithmetic, :140-142
                        q.scores0.push_back((double)sc);
                        q.cls0.push_back(c);
                    }
                }
            });
        } else if (p->det_kind == DET_TFLITE) {
            // tflite_object_detector.py:234-295 (_postprocess) + tools/tflite.py:26-41: score >= threshold, int() of the scaled
            // corners (f32 arithmetic, truncation), label = label_list[class], stable sort by descending score, wanted labels only
            const float *hb = p->h_fin.as<float>(), *hc = hb + (size_t)S * MAX_DET * 4, *hs = hc + (size_t)S * MAX_DET;
            const int *hn = reinterpret_cast<const int *>(hs + (size_t)S * MAX_DET);
            ddk::parallel_for(S, GRAIN, [&](int z0, int z1) {
                for (int z = z0; z < z1; ++z) {
                    StreamState &q = st[z];
                    q.boxes0.clear(); q.scores0.clear(); q.cls0.clear();
                    int order[MAX_DET_CAP], n = 0;
                    for (int i = 0; i < hn[z] && i < MAX_DET; ++i)
                        if ((double)hs[z * MAX_DET + i] >= p->det_conf) order[n++] = i;      // np.float32 >= python float: compared in float64
                    std::stable_sort(order, order + n, [&](int a, int b) { return hs[z * MAX_DET + a] > hs[z * MAX_DET + b]; });
                    for (int k = 0; k < n; ++k) {
                        const int i = order[k], c = (int)hc[z * MAX_DET + i];
                        if (c < 0 || c >= (int)p->tfl_labels.size() || wanted_index(p, p->tfl_labels[c]) < 0) continue;
                        // the class stored for the vote is resolved through class_name() = labels[c + label_offset]: the same line of the
                        // label file only while no blank line precedes it -- store the index class_name() maps back to this very label
                        int cs = -1;
                        for (int li = p->label_offset; li < (int)p->labels.size() && cs < 0; ++li) if (p->labels[li] == p->tfl_labels[c]) cs = li - p->label_offset;
                        if (cs < 0) continue;
                        const float *b = hb + ((size_t)z * MAX_DET + i) * 4;                          // ymin, xmin, ymax, xmax (normalised)
                        // int(np.float32 * python int): the reference's pinned NumPy promotes the product to float64 (exact), then truncates
                        const int top = (int)((double)b[0] * (double)p->H), left = (int)((double)b[1] * (double)p->W);
                        const int bottom = (int)((double)b[2] * (double)p->H), right = (int)((double)b[3] * (double)p->W);
                        q.boxes0.insert(q.boxes0.end(), {(double)left, (double)top, (double)(right - left), (double)(bottom - top)});
                        q.scores0.push_back((double)hs[z * MAX_DET + i]);
                        q.cls0.push_back(cs);
                    }
                }
            });
        } else {
            const double *hb = p->h_fin.as<double>(), *hs = hb + (size_t)S * MAX_DET * 4;
            const int *hc = reinterpret_cast<const int *>(hs + (size_t)S * MAX_DET), *hn = hc + (size_t)S * MAX_DET;
            ddk::parallel_for(S, GRAIN, [&](int z0, int z1) {
                for (int z = z0; z < z1; ++z) {
                    StreamState &q = st[z];
                    q.boxes0.clear(); q.scores0.clear(); q.cls0.clear();
                    for (int i = 0; i < hn[z]; ++i) {                                                  // :204-212
                        const double sc = hs[z * MAX_DET + i];
                        if (!(sc >= p->det_conf) || wanted_index(p, class_name(p, hc[z * MAX_DET + i])) < 0) continue;
                        const double *b = hb + ((size_t)z * MAX_DET + i) * 4;
                        q.boxes0.insert(q.boxes0.end(), {b[0], b[1], b[2] - b[0], b[3] - b[1]});
                        q.scores0.push_back(sc);
                        q.cls0.push_back(hc[z * MAX_DET + i]);
                    }
                }
            });
        }
        // the host block has been consumed: the detector buffers are free for the next frames
        if (frames_next && !p->det_late && (rc = enqueue_detector(p, frames_next)) != DD_OK) return rc;
    } else {
        for (auto &q : st) { q.boxes0.clear(); q.scores0.clear(); q.cls0.clear(); }
    }
    if (inj_offsets_host)
        DD_REQUIRE(inj_boxes_host && inj_scores_host && inj_cls_host, DD_E_ARG, "dd_pipeline_step: injected arrays missing");
    const double t1 = now_s();

    // ---------------- box hygiene (deepdish.py:940-960, background subtraction off) + batched NMS (:995)
    std::vector<int> &off = p->off;
    off.assign(S + 1, 0);
    ddk::parallel_for(S, GRAIN, [&](int z0, int z1) {
        for (int z = z0; z < z1; ++z) {
            StreamState &q = st[z];
            if (inj_offsets_host) {
                const int a = inj_offsets_host[z], b = inj_offsets_host[z + 1];
                q.boxes0.assign(inj_boxes_host + (size_t)a * 4, inj_boxes_host + (size_t)b * 4);
                q.scores0.assign(inj_scores_host + a, inj_scores_host + b);
                q.cls0.assign(inj_cls_host + a, inj_cls_host + b);
            }
            q.ib.clear(); q.is.clear(); q.ic.clear();
            ddhost::clean_boxes(q.boxes0, q.scores0, q.cls0, p->W, p->H, q.ib, q.is, q.ic);
        }
    });
    for (int z = 0; z < S; ++z) off[z + 1] = off[z] + (int)st[z].is.size();
    if (p->mog2 && off[S] > 0) {                                                        // :957 motion test on every candidate
        const int K0 = off[S];
        const size_t in_bytes = (size_t)K0 * 5 * sizeof(int), all_bytes = in_bytes + (size_t)K0 * sizeof(int);
        if ((rc = p->h_mbox.reserve(all_bytes)) != DD_OK) return rc;
        if ((rc = p->d_mbox.reserve(all_bytes)) != DD_OK) return rc;
        int *hm = p->h_mbox.as<int>(), *dm = p->d_mbox.as<int>();
        for (int z = 0; z < S; ++z)
            for (size_t i = 0; i < st[z].is.size(); ++i) {
                for (int q = 0; q < 4; ++q) hm[(size_t)(off[z] + i) * 4 + q] = (int)st[z].ib[i * 4 + q];
                hm[(size_t)K0 * 4 + off[z] + i] = z;
            }
        DD_HIP(hipMemcpyAsync(dm, hm, in_bytes, hipMemcpyHostToDevice, s));
        if ((rc = ddk::mask_box_count(s, p->d_mask.as<uint8_t>(), p->H, p->W, dm, dm + (size_t)K0 * 4, K0, dm + (size_t)K0 * 5)) != DD_OK) return rc;
        DD_HIP(hipMemcpyAsync(hm + (size_t)K0 * 5, dm + (size_t)K0 * 5, (size_t)K0 * sizeof(int), hipMemcpyDeviceToHost, s));
        DD_TIMED_WAIT(hipStreamSynchronize(s));                                                        // extra round trip
        const int *cnt = hm + (size_t)K0 * 5;
        std::vector<int> off0 = off;
        for (int z = 0; z < S; ++z) {
            StreamState &q = st[z];
            size_t n = 0;
            for (size_t i = 0; i < q.is.size(); ++i) {
                const int64_t w = q.ib[i * 4 + 2], h = q.ib[i * 4 + 3];
                if (!((double)cnt[off0[z] + i] >= p->motion_ratio * (double)w * (double)h)) { p->motion_rejected++; continue; }
                for (int c = 0; c < 4; ++c) q.ib[n * 4 + c] = q.ib[i * 4 + c];
                q.is[n] = q.is[i]; q.ic[n] = q.ic[i];
                ++n;
            }
            q.ib.resize(n * 4); q.is.resize(n); q.ic.resize(n);
            off[z + 1] = off[z] + (int)n;
        }
    }
    const int K = off[S];
    if (K > 0) {
        bool small = true;
        for (int z = 0; z < S; ++z) if (off[z + 1] - off[z] > 64) small = false;
        const size_t in_bytes = (size_t)K * 5 * sizeof(double) + (size_t)(S + 1) * sizeof(int);
        const size_t out_bytes = (size_t)(K + S) * sizeof(int);
        if ((rc = p->h_nms.reserve(in_bytes + out_bytes + 256)) != DD_OK) return rc;
        if ((rc = p->d_nms.reserve(in_bytes + out_bytes + 256 + (small ? 0 : ddk::nms_scratch_bytes(4096)))) != DD_OK) return rc;
        double *hb = p->h_nms.as<double>(), *hk = hb + (size_t)K * 4;
        int *ho = reinterpret_cast<int *>(hk + K);
        ddk::parallel_for(S, GRAIN, [&](int z0, int z1) {
            for (int z = z0; z < z1; ++z) {
                const StreamState &q = st[z];
                for (size_t i = 0; i < q.ib.size(); ++i) hb[(size_t)off[z] * 4 + i] = (double)q.ib[i];   // astype(np.float)
                for (size_t i = 0; i < q.is.size(); ++i) hk[off[z] + i] = q.is[i];
            }
        });
        memcpy(ho, off.data(), (size_t)(S + 1) * sizeof(int));
        char *d = p->d_nms.as<char>();
        DD_STAGE_BEGIN(1);
        DD_HIP(hipMemcpyAsync(d, hb, in_bytes, hipMemcpyHostToDevice, s));
        const double *dbx = reinterpret_cast<const double *>(d), *dk = dbx + (size_t)K * 4;
        const int *doff = reinterpret_cast<const int *>(dk + K);
        int *didx = reinterpret_cast<int *>(d + ((in_bytes + 63) / 64) * 64), *dnk = didx + K;
        if (small) {
            if ((rc = ddk::nms_batched_small(s, dbx, dk, doff, S, p->nms_overlap, 0, didx, dnk)) != DD_OK) return rc;
        } else {
            void *scr = d + ((in_bytes + 63) / 64) * 64 + ((out_bytes + 63) / 64) * 64;
            for (int z = 0; z < S; ++z) {
                const int k = off[z + 1] - off[z];
                if ((rc = ddk::nms_ex(s, dbx + (size_t)off[z] * 4, dk + off[z], k, p->nms_overlap, 0, 0, didx + off[z], dnk + z,
                                      scr, ddk::nms_scratch_bytes(4096))) != DD_OK) return rc;
            }
        }
        int *hidx = reinterpret_cast<int *>(p->h_nms.as<char>() + ((in_bytes + 63) / 64) * 64);
        DD_HIP(hipMemcpyAsync(hidx, didx, out_bytes, hipMemcpyDeviceToHost, s));
        DD_STAGE_END(1);
        DD_TIMED_WAIT(hipStreamSynchronize(s));                                                        // round trip 2
        const int *hnk = hidx + K;
        for (int z = 0; z < S; ++z) st[z].keep.assign(hidx + off[z], hidx + off[z] + hnk[z]);
    } else {
        for (auto &q : st) q.keep.clear();
    }
    const double t2 = now_s();

    // ---------------- crops + MARS for every kept box of every stream (deepdish.py:1008)
    std::vector<int> &doff = p->doff;
    doff.assign(S + 1, 0);
    for (int z = 0; z < S; ++z) doff[z + 1] = doff[z] + (int)st[z].keep.size();
    const int D = doff[S];
    std::vector<double> &tlwh = p->tlwh;
    tlwh.resize((size_t)D * 4);
    if (D > 0) {
        if ((rc = p->h_crop.reserve((size_t)D * 32)) != DD_OK) return rc;
        if ((rc = p->d_crop.reserve((size_t)D * 32)) != DD_OK) return rc;
        if ((rc = p->d_patches.reserve((size_t)D * 64 * 32 * 3)) != DD_OK) return rc;
        if ((rc = p->d_feats.reserve((size_t)D * 128 * sizeof(float))) != DD_OK) return rc;
        int *hc = p->h_crop.as<int>();
        ddk::parallel_for(S, GRAIN, [&](int z0, int z1) {
            for (int z = z0; z < z1; ++z) {
                StreamState &q = st[z];
                q.det_cls.clear(); q.det_conf.clear();
                for (size_t j = 0; j < q.keep.size(); ++j) {
                    const int i = q.keep[j];
                    const int64_t *b = q.ib.data() + (size_t)i * 4;
                    int *c = hc + (size_t)(doff[z] + j) * 8;
                    ddk::crop_box_host(b, 64, 32, p->H, p->W, c, c + 1, c + 2, c + 3);     // generate_detections.py:63-80
                    c[4] = z; c[5] = c[6] = c[7] = 0;
                    for (int c4 = 0; c4 < 4; ++c4) tlwh[(size_t)(doff[z] + j) * 4 + c4] = (double)b[c4];
                    q.det_cls.push_back(q.ic[i]);
                    q.det_conf.push_back(q.is[i]);
                }
            }
        });
        DD_STAGE_BEGIN(2);
        DD_HIP(hipMemcpyAsync(p->d_crop.p, hc, (size_t)D * 32, hipMemcpyHostToDevice, s));
        if ((rc = ddk::crop_resize(s, frames, p->H, p->W, p->d_crop.p, D, 64, 32, p->d_patches.as<uint8_t>())) != DD_OK) return rc;
        for (int a = 0; a < D; a += p->enc_batch) {
            const int n = std::min(p->enc_batch, D - a);
            if ((rc = dd_net_forward(p->enc, p->d_patches.as<uint8_t>() + (size_t)a * 64 * 32 * 3, n, s)) != DD_OK) return rc;
            if ((rc = dd_net_read(p->enc, -1, n, p->d_feats.as<float>() + (size_t)a * 128, 1, s)) != DD_OK) return rc;
        }
        DD_STAGE_END(2);
    } else {
        for (auto &q : st) { q.det_cls.clear(); q.det_conf.clear(); }
    }
    const double t3 = now_s();

    // ---------------- deep_sort update, phase-split so all streams share two round trips (:1029)
    DD_STAGE_BEGIN(3);
    if ((rc = ddk::trackers_update_begin(p->trks.data(), S, tlwh.data(), D ? p->d_feats.as<float>() : nullptr, 1,
                                         doff.data())) != DD_OK) return rc;
    DD_STAGE_END(3);
    // Look-ahead, late form (default): the detector run of the next frames is queued HERE, behind this step's NMS / crops / encoder /
    // association kernels (enqueue_detector orders the detector stream after what the main stream holds so far).  Queued at the
    // consume point instead (DD_DET_LATE=0) it shares the GPU with the encoder from the start, the chain the host waits for takes
    // twice as long, and then the GPU idles through the host's matching / count-line tail: one worker group ran 6.3 ms per
    // 384-frame step for 5.0 ms of kernels.  Here the chain runs alone, and the detector fills the tail.
    if (p->det && frames_next && p->det_late && (rc = enqueue_detector(p, frames_next)) != DD_OK) return rc;
    DD_TIMED_WAIT(hipStreamSynchronize(s));                                                            // round trip 3
    DD_STAGE_BEGIN(4);
    if ((rc = ddk::trackers_update_match(p->trks.data(), S)) != DD_OK) return rc;
    DD_STAGE_END(4);
    DD_TIMED_WAIT(hipStreamSynchronize(s));                                                            // round trip 4
    DD_STAGE_BEGIN(5);
    if ((rc = ddk::trackers_update_end(p->trks.data(), S)) != DD_OK) return rc;
    DD_STAGE_END(5);
    ddk::parallel_for(S, GRAIN, [&](int z0, int z1) {
        for (int z = z0; z < z1; ++z) count_line_one_stream(p, st[z]);
    });
    const double t4 = now_s();
    p->t_det += t1 - t0; p->t_nms += t2 - t1; p->t_enc += t3 - t2; p->t_trk += t4 - t3;
    p->g_wall += 1e3 * (t4 - t0); p->g_host += 1e3 * ((t4 - t0) - p->step_wait);
    p->ev_pending = true;
    p->steps += 1;
    return DD_OK;
}

}  // extern "C"
