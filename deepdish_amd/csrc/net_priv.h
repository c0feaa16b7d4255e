// Engine state shared by the two executor translation units: csrc/nets.hip (f16 programs) and csrc/netsq.hip (uint8 programs).
#pragma once
#include <map>
#include <utility>
#include <vector>
#include "common.h"

struct TensorDesc { int buf, h, w, c, cs, coff, dtype, pad; };

struct dd_net {
    dd_ctx *ctx = nullptr;
    int max_batch = 0;
    std::vector<int32_t> prog;
    std::vector<TensorDesc> tensors;
    std::vector<int64_t> buf_elems;          // per image
    std::vector<void *> bufs;
    std::vector<int> buf_dtype;
    int n_ops = 0, ops_off = 0;
    void *arena = nullptr;                   // dd_net_create_shared: one allocation, the buffers are offsets into it
    int64_t arena_bytes = 0;
    char *d_weights = nullptr;
    int64_t weight_bytes = 0;
    int in_h = 0, in_w = 0, out_tensor = -1;
    bool profile = false;
    int last_batch = 0;
    DevBuf slab;                             // split-K partial sums (sized for max_batch: see launch_conv)
    // dd_net_ssd_decode: the SSD head ops decode in their epilogue into these per-anchor arrays ([max_batch][n_anchors] each)
    bool ssd_dec = false, yolo_dec = false;                      // dd_net_yolo_decode: the Detect heads reduce their rows to (box, confidence, class); dec_anchors = rows per image
    int dec_anchors = 0; float dec_thr = 0.f;
    float *d_anchors = nullptr, *dec_boxes = nullptr, *dec_score = nullptr, *dec_keys = nullptr; int *dec_cls = nullptr;
    bool slab_moved = false;                 // the slab was reallocated during the last eager forward: captured graphs hold a dead pointer
    _Float16 *d_zero = nullptr;              // 256 bytes of zeros (padding taps of the direct-to-LDS fills)
    bool use_glds = true;
    bool use_rw = true;                      // DD_NO_RW=1: 3x3x32x32 layers fall back to the implicit-GEMM kernels (A/B measurements)
    int tile_mode = 0;                       // DD_TILE_MODE=1 forces the 64 x 64 tile everywhere (A/B measurements)
    std::vector<hipEvent_t> events;           // n_ops + 1 when profiling
    std::vector<int32_t> op_launch;           // per op of the last forward: DD_OPK_* (which launch ran it)
    // Latency mode (dd_net_use_graph): the launch train of one forward -- 20 to 75 short kernels at batch 1 -- captured
    // once per (input pointer, batch) and replayed as one hipGraph launch; the first call of a key runs eagerly (it may
    // still allocate split-K slabs and set function attributes), the second captures.
    struct GraphEntry { int calls = 0; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
    bool use_graph = false;
    std::map<std::pair<const void *, int>, GraphEntry> graphs;
};

// csrc/netsq.hip: ops of the uint8 programs (kinds >= 16).  `handled` = 0 when the kind is not one of them.
int netq_run_op(dd_net *net, int op_index, const int32_t *o, const uint8_t *input, int nimg, hipStream_t s, int *handled);
// csrc/netsq_front.hip: the first three ops of a uint8 SSD-MobileNet-v1 program (first layer, blocks 1 and 2) as one launch; *ran = 0: not this
// program / geometry / quantisation -- run the ops one by one
int netq_run_front(dd_net *net, const int32_t *o0, const int32_t *o1, const int32_t *o2, const uint8_t *input, int nimg, hipStream_t s, int *ran);
// csrc/netsq_mid.hip: two consecutive block ops (MobileNet blocks 3 and 4 of a uint8 SSD-MobileNet-v1 program) as one launch; *ran = 0: not those
int netq_run_mid(dd_net *net, const int32_t *o3, const int32_t *o4, int nimg, hipStream_t s, int *ran);
int netq_prepare(dd_net *net);           // after dd_net_create allocated the buffers: borders of the uint8 tensors
