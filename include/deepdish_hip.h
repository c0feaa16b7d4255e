/* deepdish_hip.h -- C ABI of libdeepdish_hip.so (MI355X / gfx950 only).
 *
 * The upstream project (AdaptiveCity/deepdish) is pure Python and has NO FFI: its
 * hot path is a set of duck-typed Python plugin objects.  This header is therefore
 * the boundary a maintainer would bind with ctypes; every entry
 * point cites the reference interface (file:line, relative to the upstream tree)
 * whose arithmetic it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (DD_E_*); the message of the
 *     last error on the calling thread is dd_last_error().  Nothing throws.
 *   - pointers are DEVICE pointers unless the parameter name ends in _host.
 *   - matrices are dense row-major; f64 = double, f32 = float, boxes are tlwh
 *     (top-left x, top-left y, width, height) unless stated.
 *   - `stream` is a hipStream_t passed as void* (NULL = the context's own stream).
 *     Calls only enqueue work; the caller synchronises, except for *_host outputs,
 *     which are complete on return.
 *   - the library never takes ownership of caller memory.
 */
#ifndef DEEPDISH_HIP_H
#define DEEPDISH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_OK            0
#define DD_E_ARG        -1   /* bad argument (NULL, negative size, capacity exceeded) */
#define DD_E_HIP        -2   /* a HIP runtime call failed */
#define DD_E_STATE      -3   /* call order / handle state */
#define DD_E_CAPACITY   -4   /* a fixed capacity (tracks, boxes, batch) would be exceeded */

#define DD_FEATURE_DIM 128   /* re-ID feature length, tools/freeze_model.py:139-147 */

typedef struct dd_ctx dd_ctx;
typedef struct dd_tracker dd_tracker;
typedef struct dd_net dd_net;
typedef struct dd_pipeline dd_pipeline;

const char *dd_last_error(void);
int dd_version(void);

/* One context per (thread, device): owns a stream and scratch memory.  The reference
 * runs the detector and the encoder on different pool threads (deepdish.py:935,1008);
 * give each its own context. */
int dd_ctx_create(int device, dd_ctx **out);
int dd_ctx_destroy(dd_ctx *ctx);
int dd_ctx_stream(dd_ctx *ctx, void **out_stream);
int dd_ctx_sync(dd_ctx *ctx);

/* ---------------------------------------------------------------- Kalman filter (f64)
 * State layout: means[slot][8] = (cx, cy, a, h, vx, vy, va, vh), covs[slot][8][8].
 * `slots` selects rows of the state arrays (NULL = rows 0..n-1). */

/* deep_sort/kalman_filter.py:55-86  KalmanFilter.initiate */
int dd_kf_initiate(dd_ctx *ctx, double *means, double *covs, const int *slots,
                   const double *xyah, int n, void *stream);
/* deep_sort/kalman_filter.py:88-123  KalmanFilter.predict (Track.predict, track.py:113-125) */
int dd_kf_predict(dd_ctx *ctx, double *means, double *covs, const int *slots, int n, void *stream);
/* deep_sort/kalman_filter.py:125-152  KalmanFilter.project -> proj_mean[n][4], proj_cov[n][4][4] */
int dd_kf_project(dd_ctx *ctx, const double *means, const double *covs, const int *slots, int n,
                  double *proj_mean, double *proj_cov, void *stream);
/* deep_sort/kalman_filter.py:154-186  KalmanFilter.update; pair i updates slots[i] with xyah[i] */
int dd_kf_update(dd_ctx *ctx, double *means, double *covs, const int *slots,
                 const double *xyah, int n, void *stream);
/* deep_sort/kalman_filter.py:188-229  KalmanFilter.gating_distance for every (track, detection):
 * out_d2[n][n_det] squared Mahalanobis distance; only_position != 0 uses (cx, cy) only. */
int dd_kf_gate(dd_ctx *ctx, const double *means, const double *covs, const int *slots, int n,
               const double *xyah, int n_det, int only_position, double *out_d2, void *stream);

/* ---------------------------------------------------------------- cost matrices */

/* deep_sort/iou_matching.py:7-39 iou + :42-81 iou_cost: out[n_t][n_d] = 1 - IoU (no +1 pixel).
 * rows with tsu[i] > 1 are filled with 1e5 (tsu may be NULL). */
int dd_iou_cost(dd_ctx *ctx, const double *tlwh_t, const int *tsu, int n_t,
                const double *tlwh_d, int n_d, double *out, void *stream);

/* deep_sort/nn_matching.py:31-54,78-96,156-177: per target t, min over its gallery rows of
 * (1 - a.b) with a, b L2-normalised in f32; result widened to f64 as nn_matching.py:174-176.
 * Target t owns gallery rows [offsets[t], offsets[t+1]).  out[n_t][n_d]. */
int dd_cosine_nn_cost(dd_ctx *ctx, const float *gallery, const int *offsets_host, int n_t,
                      const float *feats, int n_d, double *out, void *stream);

/* deep_sort/preprocessing.py:6-73 non_max_suppression (greedy, +1 pixel, inter/area_other > thr).
 * `keys` is the sort key (scores, or y2 when the reference is called with scores=None).
 * out_idx[k] receives the surviving indices in pick order, *out_n their number (both device). */
int dd_nms(dd_ctx *ctx, const double *tlwh, const double *keys, int k, double max_overlap,
           int *out_idx, int *out_n, void *stream);

/* tools/ssd_mobilenet.py:59-98 nms_boxes for ONE class: xyxy boxes, IoU with +1 on the
 * intersection only, keep while IoU <= thr. Same outputs as dd_nms. */
int dd_nms_ssd(dd_ctx *ctx, const double *xyxy, const double *scores, int k, double iou_thr,
               int *out_idx, int *out_n, void *stream);

/* scipy.optimize.linear_sum_assignment as called at deep_sort/linear_assignment.py:58
 * (scipy is a third-party dependency of the reference; rectangular shortest-augmenting-path,
 * Crouse 2016).  Host code.  row_ind/col_ind get min(nr,nc) pairs sorted by row. */
int dd_lsap_host(const double *cost_host, int nr, int nc, int *row_ind_host, int *col_ind_host);

/* Iteration order of Python's list(set(a) - set(b)) for small non-negative ints (CPython 3.10 set
 * layout): deep_sort/linear_assignment.py:140 feeds that order into the IoU stage (tracker.py:120-123),
 * where it decides new-track-id order under cost ties.  Host code; exported for tests. */
int dd_pyset_difference_order_host(const int *a_host, int na, const int *b_host, int nb, int *out_host,
                                   int *out_n_host);

/* ---------------------------------------------------------------- tracker (state in HBM)
 * deep_sort/tracker.py:40-138 Tracker + track.py:67-196 Track state machine +
 * linear_assignment.py:11-190 (threshold, LSAP, cascade, gating) + nn_matching.py:137-154
 * partial_fit.  Kalman state and the appearance gallery live in device memory; the integer
 * book-keeping (ids, states, hits, age, time_since_update) lives on the host.
 * nn_budget <= 0 is the reference's budget=None (deepdish.py:515): a track keeps EVERY sample it was ever matched
 * with -- galleries grow in 32-row chunks from a pool shared by the group and are bounded by device memory only
 * (DD_E_CAPACITY when an allocation fails, never a silent overwrite).  nn_budget = B keeps the last B samples
 * (nn_matching.py:150-153).  gallery_capacity is a sizing hint: the rows per track the chunk table starts with. */
int dd_tracker_create(dd_ctx *ctx, double max_cosine_distance, double max_iou_distance,
                      int max_age, int n_init, int nn_budget /* <=0: None */,
                      int track_capacity, int gallery_capacity, dd_tracker **out);
int dd_tracker_destroy(dd_tracker *trk);
/* tracker.py:51-57 */
int dd_tracker_predict(dd_tracker *trk);
/* tracker.py:59-93.  tlwh_host[n][4] f64; feats[n][128] f32 on device when feats_on_device,
 * else on the host. */
int dd_tracker_update(dd_tracker *trk, const double *tlwh_host, const float *feats,
                      int feats_on_device, int n);
/* which: 0 = tracker.tracks, 1 = tracker.deleted_tracks (tracker.py:80-81) */
int dd_tracker_count(dd_tracker *trk, int which, int *out_n_host);
/* Any output pointer may be NULL.  ints6[n][6] = (track_id, state, time_since_update, hits, age,
 * index of the detection that updated or founded the track in the last update, else -1);
 * means[n][8], covs[n][64]. */
int dd_tracker_read(dd_tracker *trk, int which, int64_t *ints6_host, double *means_host,
                    double *covs_host);
/* Per-track calls the host makes between two updates (deepdish/framerecords.py:133-165 via deepdish.py:1047):
 * dd_tracker_track_update = Track.update(kf, detection) (deep_sort/track.py:127-152) for one live track -- Kalman
 * update with the box, feature (128 f32, host or device) appended to the track's gallery, hits += 1,
 * time_since_update = 0, Tentative -> Confirmed at n_init hits; the mirrored mean is refreshed (synchronises).
 * dd_tracker_track_set = host assignment to track.state (1 Tentative, 2 Confirmed) and track.time_since_update
 * (< 0: unchanged).  dd_tracker_remove = the host reassigned tracker.tracks without these ids. */
int dd_tracker_track_update(dd_tracker *t, int64_t track_id, const double *tlwh_host, const float *feat, int feat_on_device);
int dd_tracker_track_predict(dd_tracker *t, int64_t track_id);   /* Track.predict(kf), deep_sort/track.py:113-125, one track */
int dd_tracker_track_set(dd_tracker *t, int64_t track_id, int state, int time_since_update);
int dd_tracker_remove(dd_tracker *t, const int64_t *track_ids_host, int n);
int dd_tracker_next_id(dd_tracker *trk, int64_t *out_host);
/* Rows of tracker.tracks reassigned by the host (deepdish.py:1047) are not supported: the
 * identity pass-through of framerecords.py:183 is the only behaviour reproduced. */

/* last association, for parity tests: app_host / iou_host [rows][cols] = the appearance cost matrix (nn_matching.py:156-177
 * after gate_cost_matrix, linear_assignment.py:181-189: gated entries 1e5; rows of unconfirmed tracks unspecified) and the
 * IoU cost matrix (iou_matching.py:42-81) of the last update; rows = tracks before that update, cols = detections. */
int dd_tracker_last_cost(dd_tracker *trk, double *app_host, double *iou_host, int cap, int *rows_host, int *cols_host);
/* matches[m][2] = (track row, detection), in update order */
int dd_tracker_last_matches(dd_tracker *trk, int *pairs_host, int cap, int *out_m_host);

/* ---------------------------------------------------------------- crops
 * tools/generate_detections.py:40-84 extract_image_patch for every box (integer box math on the
 * host side of the call, bilinear u8 resample as cv2.resize INTER_LINEAR on the device).
 * frame: u8 [H][W][3] BGR.  boxes_host: int64 tlwh [n][4] exactly as the reference receives them.
 * out: u8 [n][ph][pw][3].  valid_host[i] = 0 where the reference would return None. */
int dd_crop_resize(dd_ctx *ctx, const uint8_t *frame, int H, int W, const int64_t *boxes_host,
                   int n, int ph, int pw, uint8_t *out, int *valid_host, void *stream);
/* The same for float boxes (f64 [n][4] tlwh): generate_detections.py:64-74 evaluated in floating point with one
 * truncation at `astype(np.int)` -- what happens upstream when a box comes from a CVAT annotation
 * (deepdish/framerecords.py:109-118 feeds annotation boxes to the encoder). */
int dd_crop_resize_f64(dd_ctx *ctx, const uint8_t *frame, int H, int W, const double *boxes_host, int n, int ph,
                       int pw, uint8_t *out, int *valid_host, void *stream);

/* tools/generate_detections.py:86-116 DummyImageEncoder (mode 0: channel-mean of 16x8 patches, -128,
 * L2-normalised) and ConstantImageEncoder (mode 1: e0): the reference's model-free test encoders.
 * patches u8 [n][16][8][3] -> out f32 [n][128]. */
int dd_fake_encode(dd_ctx *ctx, const uint8_t *patches, int n, int mode, float *out, void *stream);

/* PIL Image.resize(LANCZOS) of an RGB(A) u8 image as tools/ssd_mobilenet.py:54-57 and
 * tools/yolov5.py:99 call it (stretch, no letterbox).  src u8 [H][W][src_c] (first 3 channels
 * used, optionally swapped BGR->RGB), dst u8 [h][w][3]. */
int dd_resize_lanczos(dd_ctx *ctx, const uint8_t *src, int H, int W, int src_c, int swap_rb,
                      uint8_t *dst, int h, int w, void *stream);
/* The same resize of `batch` frames of one geometry, densely packed on both sides (what the batched pipeline
 * does with a step's frames: one launch, both passes through LDS when the geometry allows). */
int dd_resize_lanczos_batch(dd_ctx *ctx, const uint8_t *src, int batch, int H, int W, int src_c, int swap_rb,
                            uint8_t *dst, int h, int w, void *stream);
/* cv2.resize INTER_LINEAR stretch as tools/tflite_object_detector.py:211 */
int dd_resize_bilinear(dd_ctx *ctx, const uint8_t *src, int H, int W, int c,
                       uint8_t *dst, int h, int w, void *stream);

/* ---------------------------------------------------------------- frame ingest
 * The CPU side of Pipeline.capture (deepdish.py:837-878): cv2.flip(frame, 0) (:864) + cv2.resize(frame,
 * input_size) (:867) on frames that a decoder put in host memory.  A ring of `slots` pinned host buffers,
 * each [n_streams][src_h][src_w][3] u8 BGR, with device twins; dd_ingest_submit queues host->device copy
 * + flip + INTER_LINEAR stretch on a private copy stream; dd_ingest_acquire makes the consumer's stream
 * wait for that slot (no host wait) and returns the device frames [n_streams][dst_h][dst_w][3];
 * dd_ingest_release marks the consumer's last use so the slot can be refilled. */
typedef struct dd_ingest dd_ingest;
int dd_ingest_create(dd_ctx *ctx, int slots, int n_streams, int src_h, int src_w, int dst_h, int dst_w, int flip,
                     dd_ingest **out);
int dd_ingest_destroy(dd_ingest *g);
int dd_ingest_host_slot(dd_ingest *g, int slot, uint8_t **host_ptr, int64_t *n_bytes);
int dd_ingest_wait_uploaded(dd_ingest *g, int slot);    /* host may overwrite the pinned slot after this returns */
int dd_ingest_submit(dd_ingest *g, int slot);
int dd_ingest_acquire(dd_ingest *g, int slot, void *consumer_stream, const uint8_t **frames_dev);
int dd_ingest_release(dd_ingest *g, int slot, void *consumer_stream);

/* ---------------------------------------------------------------- background subtraction
 * cv2.createBackgroundSubtractorMOG2(history, varThreshold, detectShadows) (deepdish.py:889) for n_streams
 * independent streams of [height][width][3] u8 frames; every other parameter keeps OpenCV's default (5 modes,
 * backgroundRatio 0.9, varThresholdGen 9, varInit 15, varMin 4, varMax 75, complexity reduction 0.05, shadow
 * value 127, shadow threshold 0.5).  The model (101 bytes per pixel) lives in HBM inside the handle.
 * dd_mog2_apply = backSub.apply(frame, learningRate) (deepdish.py:922) for all streams in one launch:
 * frames device u8 [n_streams][H][W][3] -> mask device u8 [n_streams][H][W] (0 background, 127 shadow, 255
 * foreground); learning_rate < 0 = OpenCV's automatic 1/min(2 nframes, history).  masked_frames (optional,
 * device, same shape as frames) receives cv2.bitwise_and(frame, frame, mask=fgMask) (deepdish.py:924).
 * Calls on one handle must be issued in frame order (the model update is sequential per stream). */
typedef struct dd_mog2 dd_mog2;
int dd_mog2_create(dd_ctx *ctx, int n_streams, int height, int width, int history, double var_threshold, int detect_shadows,
                   dd_mog2 **out);
int dd_mog2_destroy(dd_mog2 *m);
int dd_mog2_apply(dd_mog2 *m, const uint8_t *frames, double learning_rate, uint8_t *mask, uint8_t *masked_frames, void *stream);
/* Test aid: the model of one stream as host arrays -- weight, variance f32 [5][H*W], mean f32 [5][3][H*W]
 * (zero past a pixel's mode count), nmodes u8 [H*W].  Synchronises. */
int dd_mog2_state(dd_mog2 *m, int stream_index, float *weight_host, float *variance_host, float *mean_host, uint8_t *nmodes_host);
/* np.count_nonzero(fgMask[y:y+h, x:x+w]) (deepdish.py:957) for n_boxes boxes: mask device u8
 * [n_streams][height][width]; boxes_xywh_host int32 [n_boxes][4], already clipped to the frame as
 * deepdish.py:951-952 does (anything else is DD_E_ARG); box_stream_host int32 [n_boxes]; counts_host int32
 * [n_boxes].  Synchronises the stream. */
int dd_mask_box_count(dd_ctx *ctx, const uint8_t *mask, int n_streams, int height, int width, const int *boxes_xywh_host,
                      const int *box_stream_host, int n_boxes, int *counts_host, void *stream);

/* ---------------------------------------------------------------- networks
 * Replaces tflite_runtime.Interpreter(model_path).invoke() at tools/ssd_mobilenet.py:35-38,102-109,
 * tools/yolov5.py:71-79,107-109 and tools/generate_detections.py:153-154,169-171.  A model is an op
 * program + one weight blob compiled on the host (deepdish_amd/nets.py: MARS per
 * tools/freeze_model.py:88-157, SSD-MobileNet-v1, YOLOv5s per detectors/yolov5/yolov5s.yaml) -- the
 * analogue of the reference's .tflite file; the word layout is documented in csrc/nets.hip. */
int dd_net_create(dd_ctx *ctx, const int32_t *program_host, int n_words, const void *weights_host,
                  int64_t n_weight_bytes, int max_batch, dd_net **out);
/* The same engine with its activation buffers overlaid by lifetime (a buffer is live from its first writer to its last reader; the
 * output tensor to the end): what a pipeline that never reads intermediate tensors wants -- the reference's interpreter keeps one
 * arena per model too (tflite_runtime: `interpreter.allocate_tensors()`, tools/ssd_mobilenet.py:38).  f16 programs only; dd_net_read
 * of anything but the output tensor is DD_E_STATE. */
int dd_net_create_shared(dd_ctx *ctx, const int32_t *program_host, int n_words, const void *weights_host,
                         int64_t n_weight_bytes, int max_batch, dd_net **out);
/* device bytes of the engine's activation buffers (one number: the arena, or the sum of the per-tensor buffers) */
int dd_net_activation_bytes(dd_net *net, int64_t *out_host);
int dd_net_destroy(dd_net *net);
/* input u8 [n][in_h][in_w][3]; results stay in the net's own device tensors (dd_net_output). */
int dd_net_forward(dd_net *net, const uint8_t *input, int n, void *stream);
/* tensor < 0 selects the program's declared output.  dtype: 0 f16, 1 f32, 2 u8.  Row n of the
 * tensor starts at dev_ptr + n * h * w * cs elements. */
int dd_net_output(dd_net *net, int tensor, void **dev_ptr_host, int *h_host, int *w_host, int *c_host,
                  int *cs_host, int *dtype_host);

/* Latency mode for small batches (the reference runs ONE stream: deepdish.py:1324-1340): the kernels of a forward are
 * captured once per (input pointer, n) into a hipGraph and replayed with one launch.  Results are identical. */
int dd_net_use_graph(dd_net *net, int enable);
int dd_net_input_size(dd_net *net, int *h_host, int *w_host);   /* the model's input height / width (ssd_mobilenet.py:43, yolov5.py:79) */
int dd_net_max_batch(dd_net *net, int *out_host);
int dd_net_last_batch(dd_net *net, int *out_host);   /* images in the most recent forward */
/* Copy the first n images of a (whole, un-sliced) tensor to caller memory: n*h*w*cs elements. */
int dd_net_read(dd_net *net, int tensor, int n, void *dst, int dst_on_device, void *stream);

/* Measurement aid (not on the product path): bracket every op of the next forwards with HIP events
 * on the launch stream; read back the per-op milliseconds of the last forward. */
int dd_net_profile(dd_net *net, int enable);
int dd_net_profile_read(dd_net *net, float *ms_host, int cap, int *n_ops_host);
/* Which launch ran each op of the last forward: 0 the op's own kernel, 1 none (folded into the next op's launch),
 * 2 conv3x3_pool_rows_k, 3 conv3x3_pool_rows_k with the first layer folded in, 4 res_unit_rows_k, 5 ssd_front_k,
 * 6 conv3x3_c64_rows_k, 7 conv3x3_s2_rows_k, 8 conv_ws_k, 9 conv_ws_dw_k, 10 dwpw_rows_k, 11 SSD head with the decode in its
 * epilogue, 12 res_pair_rows_k, 13 YOLOv5 Detect head with the row reduction in its epilogue, 14 mars_ws128_k, 15 none (the op ran
 * inside the previous op's launch), 16 mars_pair64_k, 17 q_dwm_k, 18 conv3x3_c64_rows_k on 8-column strips -- so that a per-kernel time table attributes a fused launch to the kernel that ran. */
int dd_net_op_launches(dd_net *net, int32_t *codes_host, int cap, int *n_ops_host);

/* TFLite_Detection_PostProcess (inside the reference's .tflite graph, tools/ssd_mobilenet.py:103-109):
 * anchor decode, sigmoid, per-class NMS, top max_det.  raw f32 [n_anchors][4+n_classes] ->
 * boxes f32 [max_det][4] (ymin,xmin,ymax,xmax normalised), classes f32, scores f32, count. */
int dd_ssd_postprocess(dd_ctx *ctx, const float *raw, const float *anchors, int n_anchors,
                       int n_classes, int max_det, float score_thr, float iou_thr,
                       float *boxes, float *classes, float *scores, int *count, void *stream);

/* The op's first stage alone, `batch` images: raw f32 [batch][n_anchors][4+n_classes] -> per anchor boxes f32 [.][4]
 * (ymin,xmin,ymax,xmax), scores (sigmoid of the best class logit, background excluded), classes int32 (id - 1, lowest
 * on ties), keys f32 (score, or -1 below score_thr) -- and its second stage (class-agnostic NMS, top max_det) from
 * those arrays.  dd_ssd_postprocess = the two in one call for one image. */
int dd_ssd_decode(dd_ctx *ctx, const float *raw, const float *anchors, int n_anchors, int n_classes, float score_thr,
                  float *boxes, float *scores, int *classes, float *keys, int batch, void *stream);
int dd_ssd_postprocess_decoded(dd_ctx *ctx, const float *dec_boxes, const float *dec_scores, const int *dec_classes,
                               const float *dec_keys, int n_anchors, int max_det, float score_thr, float iou_thr,
                               float *boxes, float *classes, float *scores, int *count, int batch, void *stream);
/* SSD detector engines (tools/ssd_mobilenet.py:102-109): run that first stage inside the head layers' GEMM epilogues,
 * straight from the accumulators -- the [n][n_anchors][4+n_classes] head matrix is then never written (dd_net_read of
 * it is DD_E_STATE) and dd_net_ssd_decoded hands out the per-anchor arrays of the last forward ([n][n_anchors] each,
 * same bits as dd_ssd_decode on the head matrix).  anchors_host f32 [n_anchors][4] (yc, xc, h, w). */
int dd_net_ssd_decode(dd_net *net, const float *anchors_host, int n_anchors, float score_thr, int enable);
int dd_net_ssd_decoded(dd_net *net, float **boxes_dev, float **scores_dev, int **classes_dev, float **keys_dev);
/* the same arrays of the first n images of the last forward copied to host memory (any pointer may be NULL) */
int dd_net_ssd_decoded_read(dd_net *net, int n, float *boxes_host, float *scores_host, int *classes_host, float *keys_host);

/* YOLOv5 detector: tools/yolov5.py:126-128 (cls *= obj, np.argmax, confidence) inside the Detect layers' epilogues -- per row of the
 * head the decoded box (x, y, w, h: the matrix's first four columns), the confidence and the class; the f32 [rows][5 + C] matrix
 * (yolov5.py:109 `pred`) is then never written and dd_net_read of it is an error.  Needs a program compiled with the per-anchor
 * copy of the Detect weights (deepdish_amd.nets.compile_yolov5s emits it).  enable = 0 switches back. */
int dd_net_yolo_decode(dd_net *net, int enable);
/* Device pointers to what the last forward's heads wrote: boxes f32 [n][rows][4], conf f32 [n][rows], classes i32 [n][rows]
 * (a NaN product makes conf NaN and the class the first NaN's index, as np.argmax does); rows = rows per image. */
int dd_net_yolo_decoded(dd_net *net, float **boxes, float **conf, int **classes, int *rows);
/* The same, copied to the host for the first n images (any pointer may be NULL). */
int dd_net_yolo_decoded_read(dd_net *net, int n, float *boxes_host, float *conf_host, int *classes_host);

/* tools/ssd_mobilenet.py:111-150, SSDMobileNet.predict after its four get_tensor calls, for `batch` images at once:
 * NaN scrub (:111-116), score >= confidence (:119), reorder [1,0,3,2] and scale by (w,h,w,h) in f64 (:121-127),
 * per-class nms_boxes (:59-98, see dd_nms_ssd).  Inputs are the outputs of dd_ssd_postprocess (device):
 * boxes f32 [batch][max_det][4], classes f32, scores f32 [batch][max_det], max_det <= 16.  Outputs (device):
 * out_boxes f64 [batch][max_det][4] xyxy pixels, out_cls int32 (class id = label-file line - 1, :142-147),
 * out_scores f64, out_n int32 [batch].  Rows come class by class in ascending id (the reference walks a Python
 * set) and inside a class in nms_boxes' pick order. */
int dd_ssd_detections(dd_ctx *ctx, const float *boxes, const float *classes, const float *scores, int batch,
                      int max_det, double confidence, double iou_thr, double img_w, double img_h, double *out_boxes,
                      int *out_cls, double *out_scores, int *out_n, void *stream);

/* tools/yolov5.py:120-131: xywh->xyxy, cls*=obj, argmax, conf >= thr, scale to image.
 * raw f32 [n_rows][5+n_cls] -> out_boxes f32 [cap][4] xyxy pixels, out_scores, out_cls, *out_n
 * (rows in ascending row order, like np.where). */
int dd_yolov5_decode(dd_ctx *ctx, const float *raw, int n_rows, int n_cls, float thr,
                     float img_w, float img_h, float *out_boxes, float *out_scores,
                     int *out_cls, int cap, int *out_n, void *stream);

/* ---------------------------------------------------------------- multi-stream hot path
 * The per-frame call sequence of the reference's Pipeline (deepdish.py:880-885 run_object_detector,
 * :940-960 box hygiene, :995 NMS, :1008 encoder, :1028-1029 tracker, :1035-1114 count line) for
 * n_streams independent streams, one frame each per step, device work batched across streams.
 * detector may be NULL (detections are then always injected).  Which detector adaptor runs is chosen like the
 * reference chooses its plugin (deepdish.py:482-502): anchors_host != NULL = SSD-MobileNet (tools/ssd_mobilenet.py:
 * resize to the net's input, forward, TFLite_Detection_PostProcess, predict tail; labels_nl line i+1 names class
 * id i, :142-147; score threshold 0.5); anchors_host == NULL = YOLOv5 (tools/yolov5.py:97-146: resize, forward with
 * the Detect decode fused, dd_yolov5_decode, label filter, xyxy -> tlwh; n_anchors = rows of the head tensor,
 * n_classes = classes; labels_nl line i names class id i, :134; score threshold 0.25; no NMS of its own -- every
 * candidate goes on to deep_sort's NMS, however many pass the threshold).
 * labels_nl: the label file's lines joined by '\n'; wanted_nl: --wanted-labels.
 * line_host: count line x1,y1,x2,y2 (deepdish.py:739-744). */
int dd_pipeline_create(dd_ctx *ctx, int n_streams, int frame_h, int frame_w, dd_net *detector,
                       const float *anchors_host, int n_anchors, int n_classes, dd_net *encoder,
                       const char *labels_nl, const char *wanted_nl, double max_cosine_distance,
                       double nms_max_overlap, double max_iou_distance, int max_age, int n_init,
                       const double *line_host, int track_capacity, int gallery_capacity, dd_pipeline **out);
int dd_pipeline_destroy(dd_pipeline *p);
/* Adaptor that consumes an SSD-type detector's output, chosen like the reference chooses its plugin class from the model
 * file name (deepdish.py:482-502): 0 = tools/ssd_mobilenet.py (default), 2 = the generic TFLite-Task adaptor
 * (tools/tflite.py:9-41 over tools/tflite_object_detector.py:180-295: cv2.resize INTER_LINEAR of the RGB frame, rows of the
 * post-process op with score >= 0.5, int() of the scaled corners, sorted by score).  Before the first step. */
int dd_pipeline_detector_adaptor(dd_pipeline *p, int adaptor);
/* Options of the TFLite_Detection_PostProcess op inside an SSD-type model file, which the reference's interpreter applies as the file
 * states them (tools/ssd_mobilenet.py:100-109: invoke() + the four output tensors of max_detections rows): rows per frame (<= 64),
 * nms_score_threshold, nms_iou_threshold (fast class-agnostic NMS).  Defaults: the stock export's 10 / 1e-8 / 0.6.  Before the first step.
 * Above 16 rows the ORDER of equal-score rows of one class is the reference's only up to its NumPy's unstable sort (tools/ssd_mobilenet.py:73
 * `s.argsort()[::-1]`: stable for <= 16 elements, unspecified beyond) -- INTEGRATION.md, "Ties inside a class". */
int dd_pipeline_ssd_options(dd_pipeline *p, int max_detections, float nms_score_threshold, float nms_iou_threshold);
/* frames: device u8 [n_streams][H][W][3] BGR.  inj_*: optional detections that REPLACE the detector's
 * output (it still runs): tlwh f64 rows, scores, class ids; stream s owns rows
 * [inj_offsets[s], inj_offsets[s+1]).  Blocks until the step is complete. */
int dd_pipeline_step(dd_pipeline *p, const uint8_t *frames, const double *inj_boxes_host,
                     const double *inj_scores_host, const int *inj_cls_host, const int *inj_offsets_host);
/* Same step with a look-ahead: the detector run of `frames_next` (may be NULL) is queued on the pipeline's detector
 * stream once this step has read its own detections -- the reference keeps one detector call and one encoder call
 * in flight on different frames the same way (deepdish.py:935,985,1008).  Contract: the next call must pass
 * `frames_next` as its `frames`, and the memory behind it must stay UNMODIFIED until that call returns -- the queued
 * result is matched to the next step by address, so a capture buffer that is refilled in place (a 1-slot ring) must
 * not be handed in here (pass NULL instead).  frames_next == frames is rejected (DD_E_ARG). */
int dd_pipeline_step2(dd_pipeline *p, const uint8_t *frames, const uint8_t *frames_next, const double *inj_boxes_host,
                      const double *inj_scores_host, const int *inj_cls_host, const int *inj_offsets_host);
/* The stream the look-ahead detector run of dd_pipeline_step2 is queued on (NULL for a pipeline without a detector).  A caller whose
 * `frames_next` are still on their way to the device hands it to dd_ingest_acquire as the consumer of THAT slot: the upload of frame
 * t + 1 is then waited for by the detector run of frame t + 1 alone, not by step t's own kernels -- the reference's capture thread fills
 * the next frame while the stages work on the current one the same way (deepdish.py:837-878, FreshQueue :192-203). */
int dd_pipeline_detector_stream(dd_pipeline *p, void **stream_out);
/* counts_host: int64 [n_streams][n_wanted][4] = poscount, negcount, intcount, delcount */
/* Background subtraction for every stream of the pipeline (deepdish.py:512,889,920-924,957): ratio =
 * --background-subtraction-ratio (reference default 0.25), ratio < 0 = --disable-background-subtraction (the state
 * a new pipeline starts in, as the reference's benchmarks run); masking = --enable-background-masking.  Each
 * step then updates the MOG2 model with the step's frames and drops detector boxes with fewer than
 * ratio * w * h moving pixels before NMS.  (Re-)enabling starts a fresh model. */
int dd_pipeline_background_subtraction(dd_pipeline *p, double ratio, int masking);
/* Foreground mask of the last step, u8 [n_streams][H][W], copied to dst (host, or device when dst_on_device);
 * dst may be NULL to read only the number of boxes the motion test has rejected so far. */
int dd_pipeline_motion_mask(dd_pipeline *p, uint8_t *dst, int dst_on_device, long long *rejected_host);
int dd_pipeline_counts(dd_pipeline *p, int64_t *counts_host);
int dd_pipeline_tracker(dd_pipeline *p, int stream, dd_tracker **out);
/* accumulated host wall time per stage (objd, nms, feat, trak as in deepdish.py's TimingInfo labels) */
int dd_pipeline_stage_seconds(dd_pipeline *p, double *out4_host, long long *steps_host);
/* GPU time per stage, as the reference names its per-frame timers (deepdish.py:975-981 objd, :1018-1021 feat, :1031-1032 trak; nms is the
 * deep_sort NMS of :995): milliseconds between HIP events recorded on the streams the stage's kernels run on, summed over the steps so far --
 * out6 = {objd (resize, detector forward, post-process, adaptor tail, host copy; on the detector stream), nms, feat (crops + encoder), trak
 * (Kalman predict, association, update, track management kernels), host (the steps' wall time outside their waits for the GPU: adaptor
 * filter, box hygiene, LSAP, count line), wall}. */
int dd_pipeline_stage_gpu_ms(dd_pipeline *p, double *out6_host, long long *steps_host);
/* What the detector adaptor returned for one stream in the last step -- the reference's object_detector.detect_image(...) result
 * (deepdish.py:935,985; tools/ssd_mobilenet.py:198-213): tlwh rows (f64), scores, class ids (index into the label file minus the
 * adaptor's label offset); with injected detections, those.  cap rows of room; n_host always gets the row count. */
int dd_pipeline_detections(dd_pipeline *p, int stream, double *boxes_host, double *scores_host, int *classes_host, int cap, int *n_host);

/* ---------------------------------------------------------------- multi-GPU
 * Sum of the per-stream count vectors (pos, neg, int, del per label; deepdish.py:1141-1145).
 * The collective itself is issued by the host through torch.distributed (RCCL); this entry
 * only packs/accumulates the int64 vector on the device. */
int dd_counts_accumulate(dd_ctx *ctx, int64_t *acc, const int64_t *counts_host, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPDISH_HIP_H */
