// deep_sort multi-target tracker with its Kalman state and appearance gallery resident in HBM.
//
// Reference (upstream paths): deep_sort/tracker.py:40-138, deep_sort/track.py:67-196,
// deep_sort/linear_assignment.py:11-190, deep_sort/nn_matching.py:137-177,
// deep_sort/iou_matching.py:42-81.
//
// Trackers are driven in GROUPS (one tracker per video stream; a stand-alone dd_tracker is a group
// of one): per step the device does a handful of launches for ALL streams together and the host
// two round trips:
//   predict : 1 kernel over every live track of every stream
//   update  : normalise all detection features (1 launch);
//             K1 "assoc"  -- for every (track row, detection of the SAME stream): gated cosine NN cost
//                            (exact-f32 MFMA + f64 Mahalanobis gate) and the IoU cost, ALL cascade
//                            levels at once (the cost of a pair does not depend on the level);
//             host        -- per stream: matching cascade + LSAP on the small cost matrices;
//             K2 "apply"  -- Kalman update / initiate + gallery append for the decided pairs;
//             gather      -- means of all live / just-deleted tracks back to the host.
// Layout: means[slot][8] f64, covs[slot][64] f64; a stream owns the slot range [slot_base, slot_base + tcap) of
// a pool, slots recycled through a per-stream free list.  Appearance gallery: rows of 128 f32 (already
// L2-normalised) in 32-row chunks drawn from 64 MiB arenas shared by the group; a track owns a chunk LIST (device
// table tab[slot][stride]), extended whenever the track is matched -- nn_budget=None upstream keeps every sample
// (deepdish.py:515, nn_matching.py:137-154), so a gallery has no fixed capacity here either; with nn_budget = B
// the last B samples are kept (a ring over ceil(B / 32) chunks).  Chunks return to the pool with the track's slot.
#include <algorithm>
#include <atomic>
#include <numeric>
#include "common.h"
#include "hostpool.h"
#include "kalman_dev.h"
#include "cost_dev.h"

namespace {
using namespace kfdev;
using namespace costdev;

constexpr double GATE_4DOF = 9.4877;      // kalman_filter.py:14
constexpr double INFTY_COST = 1e5;        // linear_assignment.py:8
enum { TENTATIVE = 1, CONFIRMED = 2, DELETED = 3 };   // track.py:15-17

// detection.py:43-50
__device__ __forceinline__ void tlwh_to_xyah(const double *b, double z[4]) {
    z[0] = b[0] + b[2] / 2;
    z[1] = b[1] + b[3] / 2;
    z[2] = b[2] / b[3];
    z[3] = b[3];
}

// grid (R rows of all streams, ceil(max n_det / 64)); 4 waves, 16 detections per wave.  Row r belongs to
// one stream and sees only that stream's detections [det_off, det_off + n_det).
__global__ __launch_bounds__(256) void tracker_assoc_k(
    const double *__restrict__ means, const double *__restrict__ covs, float *const *__restrict__ arenas,
    const int *__restrict__ tab, int tab_stride, const int *__restrict__ row_slot, const int *__restrict__ row_state,
    const int *__restrict__ row_tsu, const int *__restrict__ row_det_off, const int *__restrict__ row_ndet,
    const int *__restrict__ row_cost_off, const int *__restrict__ row_iou_delta, const int *__restrict__ row_gcount,
    const double *__restrict__ det_tlwh, const float *__restrict__ feats_n, double *__restrict__ cost) {
    const int row = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_det = row_ndet[row];
    const int d0 = blockIdx.y * 64 + wave * 16;
    if (d0 >= n_det) return;                                  // wave-uniform
    const int slot = row_slot[row];
    const int det_off = row_det_off[row];
    const bool confirmed = row_state[row] == CONFIRMED;
    float best = 0.f;
    if (confirmed)                                            // wave-uniform branch around the MFMAs
        best = nn_max_dot(ChunkRows{arenas, tab + (size_t)slot * tab_stride}, row_gcount[row],
                          feats_n + (size_t)det_off * 128, d0, n_det, lane);
    const int d = d0 + (lane & 15);
    if ((lane >> 4) != 0 || d >= n_det) return;
    const double *m = means + (size_t)slot * 8;
    const double *b = det_tlwh + (size_t)(det_off + d) * 4;
    const size_t o = (size_t)row_cost_off[row] + d;
    if (confirmed) {
        double S[16], z[4];
        innovation_cov(covs + (size_t)slot * 64, m[3], S);
        const Chol4 c = chol4(S);
        tlwh_to_xyah(b, z);
        const double mm[4] = {m[0], m[1], m[2], m[3]};
        const double d2 = maha2(c, mm, z, 0);
        cost[o] = d2 > GATE_4DOF ? INFTY_COST : (double)(1.0f - best);   // linear_assignment.py:181-189
    }
    double ci = INFTY_COST;                                   // iou_matching.py:74-76
    if (row_tsu[row] <= 1) {
        double t[4];                                          // track.py:84-97 to_tlwh
        t[3] = m[3];
        t[2] = m[2] * m[3];
        t[0] = m[0] - t[2] / 2;
        t[1] = m[1] - t[3] / 2;
        ci = 1.0 - iou_tlwh(t, b);
    }
    cost[o + row_iou_delta[row]] = ci;
}

// One wave per decided pair: [0, n_upd) Kalman update + gallery append; [n_upd, n_upd+n_new) new track.
// pair_row = chunk * 32 + row inside the chunk: where the host placed this sample (track.py:140 features.append).
__global__ __launch_bounds__(256) void tracker_apply_k(
    double *__restrict__ means, double *__restrict__ covs, float *const *__restrict__ arenas,
    const int *__restrict__ pair_slot, const int *__restrict__ pair_det, const int *__restrict__ pair_row, int n_upd,
    int n_new, const double *__restrict__ det_tlwh, const float *__restrict__ feats_n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_upd + n_new) return;
    const int slot = pair_slot[w], det = pair_det[w], grow = pair_row[w];
    double z[4];
    tlwh_to_xyah(det_tlwh + (size_t)det * 4, z);
    double *P = covs + (size_t)slot * 64, *m = means + (size_t)slot * 8;
    if (w < n_upd) update_wave(P, m, z, lane);                // track.py:138-139
    else initiate_wave(P, m, z, lane);                        // tracker.py:135-138
    const float2 f = reinterpret_cast<const float2 *>(feats_n + (size_t)det * 128)[lane];
    reinterpret_cast<float2 *>(gal_row(arenas, grow >> GAL_CH_SHIFT, grow & (GAL_CH - 1)))[lane] = f;
}

// tab[idx[i]] = val[i]: the chunk-table entries of this step's new chunks
__global__ void tab_scatter_k(int *__restrict__ tab, const int *__restrict__ upd, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[upd[2 * i]] = upd[2 * i + 1];
}

struct TrackRec {
    int64_t id;
    int state, tsu, hits, age, slot, last_det;           // slot is pool-global
};

}  // namespace

// Device state shared by the trackers of one group + the group's staging buffers.
constexpr int GAL_MAX_ARENAS = 4096;                      // x 64 MiB = 256 GiB: bounded by device memory, not by this
struct TrackerPool {
    dd_ctx *ctx = nullptr;
    int slots = 0, refs = 0;
    double *d_means = nullptr, *d_covs = nullptr;
    // gallery: arenas of GAL_ARENA_CHUNKS chunks, per-slot chunk lists (host) mirrored in d_tab[slot][tab_stride]
    std::vector<float *> arenas;
    float **d_arenas = nullptr;
    std::vector<int> free_chunks;
    std::vector<std::vector<int>> slot_chunks;
    std::vector<int> slot_total;                          // samples appended to the slot's track so far
    int *d_tab = nullptr, tab_stride = 0;
    std::vector<int> tab_upd;                             // (table index, chunk) pairs not yet on the device
    int tab_need = 0;                                     // longest chunk list placed so far; the stride follows it in gallery_flush ONLY
                                                          // (tab_stride always describes the d_tab the device holds, also on an error path)
    DevBuf d_tabupd;
    PinBuf h_tabupd;
    DevBuf d_pred, d_in, d_feats_raw, d_feats_n, d_cost, d_pairs, d_gather;
    PinBuf h_pred, h_in, h_cost, h_pairs, h_gather;
    bool pred_inflight = false;
    // phase state of the current group update
    std::vector<dd_tracker *> cur;
    std::vector<int> det_off, cost_base;
    int D = 0, R = 0, ng = 0;
    bool have_cost = false;
};

struct dd_tracker {
    TrackerPool *pool = nullptr;
    int slot_base = 0;
    double max_cos = 0.2, max_iou = 0.7;
    int max_age = 30, n_init = 3, budget = 0, tcap = 0;
    std::vector<TrackRec> tracks, deleted;
    std::vector<int> free_slots, pending_free;
    std::vector<double> live_means, dead_means;           // host mirrors, [n][8]
    std::vector<int> last_pairs;                          // (track row before update, detection)
    int64_t next_id = 1;
    int ph_n = 0, ph_T = 0, ph_cost_base = -1;            // ph_cost_base: where this tracker's cost matrices sit in pool->h_cost
    std::vector<int> m_upd, m_new;                        // decisions of the current update: (slot, detection) pairs, matched / new tracks
};

namespace {

// linear_assignment.py:11-75 on a host sub-matrix; `full` is [n_rows_total][n_det] row-major.
void min_cost_matching(const double *full, int n_det, double max_distance, const std::vector<int> &rows,
                       const std::vector<int> &dets, std::vector<std::pair<int, int>> &matches,
                       std::vector<int> &un_rows, std::vector<int> &un_dets) {
    un_rows.clear();
    un_dets.clear();
    if (rows.empty() || dets.empty()) {
        un_rows = rows;
        un_dets = dets;
        return;
    }
    const int nr = (int)rows.size(), nc = (int)dets.size();
    std::vector<double> c((size_t)nr * nc);
    for (int r = 0; r < nr; ++r)
        for (int q = 0; q < nc; ++q) {
            const double v = full[(size_t)rows[r] * n_det + dets[q]];
            c[(size_t)r * nc + q] = v > max_distance ? max_distance + 1e-5 : v;   // :57
        }
    std::vector<int> ri(std::min(nr, nc)), ci(std::min(nr, nc));
    const int np = ddk::lsap(c.data(), nr, nc, ri.data(), ci.data());
    std::vector<char> row_used(nr, 0), col_used(nc, 0);
    for (int p = 0; p < np; ++p) { row_used[ri[p]] = 1; col_used[ci[p]] = 1; }
    for (int q = 0; q < nc; ++q) if (!col_used[q]) un_dets.push_back(dets[q]);    // :62-64
    for (int r = 0; r < nr; ++r) if (!row_used[r]) un_rows.push_back(rows[r]);    // :65-67
    for (int p = 0; p < np; ++p) {                                                // :68-74
        if (c[(size_t)ri[p] * nc + ci[p]] > max_distance) {
            un_rows.push_back(rows[ri[p]]);
            un_dets.push_back(dets[ci[p]]);
        } else {
            matches.emplace_back(rows[ri[p]], dets[ci[p]]);
        }
    }
}

int pool_create(dd_ctx *ctx, int slots, int gallery_rows_hint, TrackerPool **out) {
    TrackerPool *p = new TrackerPool();
    p->ctx = ctx; p->slots = slots;
    DD_HIP(hipSetDevice(ctx->device));
    DD_HIP(hipMalloc(&p->d_means, (size_t)slots * 8 * sizeof(double)));
    DD_HIP(hipMalloc(&p->d_covs, (size_t)slots * 64 * sizeof(double)));
    DD_HIP(hipMalloc(&p->d_arenas, (size_t)GAL_MAX_ARENAS * sizeof(float *)));
    DD_HIP(hipMemset(p->d_arenas, 0, (size_t)GAL_MAX_ARENAS * sizeof(float *)));
    p->tab_stride = std::max(1, dd_ceil_div(gallery_rows_hint, GAL_CH));
    DD_HIP(hipMalloc(&p->d_tab, (size_t)slots * p->tab_stride * sizeof(int)));
    DD_HIP(hipMemset(p->d_tab, 0, (size_t)slots * p->tab_stride * sizeof(int)));
    p->slot_chunks.resize(slots);
    p->slot_total.assign(slots, 0);
    *out = p;
    return DD_OK;
}

// One more 64 MiB arena of gallery chunks.  Running out of device memory is the only capacity limit of a gallery.
int pool_add_arena(TrackerPool *p) {
    DD_REQUIRE((int)p->arenas.size() < GAL_MAX_ARENAS, DD_E_CAPACITY, "tracker gallery: %d arenas of 64 MiB exhausted", GAL_MAX_ARENAS);
    float *a = nullptr;
    const hipError_t e = hipMalloc(&a, (size_t)GAL_ARENA_CHUNKS * GAL_CH * 128 * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        dd_set_error("tracker gallery: device memory exhausted after %zu arenas of 64 MiB (%s); nn_budget bounds a gallery",
                     p->arenas.size(), hipGetErrorString(e));
        return DD_E_CAPACITY;
    }
    const int k = (int)p->arenas.size();
    p->arenas.push_back(a);
    DD_HIP(hipMemcpy(p->d_arenas + k, &a, sizeof(float *), hipMemcpyHostToDevice));
    for (int c = GAL_ARENA_CHUNKS - 1; c >= 0; --c) p->free_chunks.push_back(k * GAL_ARENA_CHUNKS + c);
    return DD_OK;
}

// Where the next sample of `slot`'s track goes (track.py:140 / nn_matching.py:150-153): row `total` of its chunk
// list, or row total % budget with nn_budget; a new chunk is taken from the pool when the list is too short.
int gallery_place(TrackerPool *p, int slot, int budget, int *row_out) {
    const int total = p->slot_total[slot];
    const int pos = budget > 0 ? total % budget : total;
    const int ci = pos >> GAL_CH_SHIFT;
    std::vector<int> &chs = p->slot_chunks[slot];
    if (ci >= (int)chs.size()) {
        p->tab_need = std::max(p->tab_need, ci + 1);          // rare: beyond the stride -> gallery_flush doubles and rewrites the table
        if (p->free_chunks.empty()) {
            const int rc = pool_add_arena(p);
            if (rc != DD_OK) return rc;
        }
        const int c = p->free_chunks.back();
        p->free_chunks.pop_back();
        chs.push_back(c);
        p->tab_upd.push_back(slot);                           // resolved against the final stride in gallery_flush
        p->tab_upd.push_back(ci);
        p->tab_upd.push_back(c);
    }
    *row_out = chs[ci] * GAL_CH + (pos & (GAL_CH - 1));
    p->slot_total[slot] = total + 1;
    return DD_OK;
}

inline int gallery_count(const TrackerPool *p, int slot, int budget) {
    const int total = p->slot_total[slot];
    return budget > 0 ? std::min(total, budget) : total;
}

void gallery_free_slot(TrackerPool *p, int slot) {
    for (int c : p->slot_chunks[slot]) p->free_chunks.push_back(c);
    p->slot_chunks[slot].clear();
    p->slot_total[slot] = 0;
}

// Bring the device chunk table up to date (same stream, after the launches that used the old contents).
int gallery_flush(TrackerPool *p, hipStream_t s) {
    if (p->tab_need > p->tab_stride) {
        int stride = p->tab_stride;
        while (p->tab_need > stride) stride *= 2;
        DD_HIP(hipStreamSynchronize(s));                       // nothing in flight may still read the old table
        const size_t n = (size_t)p->slots * stride;
        int *fresh = nullptr;
        DD_HIP(hipMalloc(&fresh, n * sizeof(int)));
        std::vector<int> host(n, 0);
        for (int sl = 0; sl < p->slots; ++sl)
            for (size_t i = 0; i < p->slot_chunks[sl].size(); ++i) host[(size_t)sl * stride + i] = p->slot_chunks[sl][i];
        const hipError_t e = hipMemcpy(fresh, host.data(), n * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(fresh); DD_HIP(e); }
        (void)hipFree(p->d_tab);
        p->d_tab = fresh; p->tab_stride = stride;              // committed together, only now
        p->tab_upd.clear();
        return DD_OK;
    }
    const int n = (int)p->tab_upd.size() / 3;
    if (n == 0) return DD_OK;
    int rc;
    if ((rc = p->h_tabupd.reserve((size_t)n * 2 * sizeof(int))) != DD_OK) return rc;
    if ((rc = p->d_tabupd.reserve((size_t)n * 2 * sizeof(int))) != DD_OK) return rc;
    int *h = p->h_tabupd.as<int>();
    for (int i = 0; i < n; ++i) {
        h[2 * i] = p->tab_upd[3 * i] * p->tab_stride + p->tab_upd[3 * i + 1];
        h[2 * i + 1] = p->tab_upd[3 * i + 2];
    }
    p->tab_upd.clear();
    DD_HIP(hipMemcpyAsync(p->d_tabupd.p, h, (size_t)n * 2 * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(tab_scatter_k, dim3(dd_ceil_div(n, 256)), dim3(256), 0, s, p->d_tab, p->d_tabupd.as<int>(), n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

void pool_release(TrackerPool *p) {
    if (!p || --p->refs > 0) return;
    (void)hipFree(p->d_means); (void)hipFree(p->d_covs);
    for (float *a : p->arenas) (void)hipFree(a);
    (void)hipFree(p->d_arenas); (void)hipFree(p->d_tab);
    for (DevBuf *b : {&p->d_pred, &p->d_in, &p->d_feats_raw, &p->d_feats_n, &p->d_cost, &p->d_pairs, &p->d_gather, &p->d_tabupd}) b->release();
    for (PinBuf *b : {&p->h_pred, &p->h_in, &p->h_cost, &p->h_pairs, &p->h_gather, &p->h_tabupd}) b->release();
    delete p;
}

dd_tracker *tracker_new(TrackerPool *pool, int slot_base, int tcap, double max_cos, double max_iou, int max_age, int n_init,
                        int budget) {
    dd_tracker *t = new dd_tracker();
    t->pool = pool; pool->refs += 1;
    t->slot_base = slot_base; t->tcap = tcap;
    t->max_cos = max_cos; t->max_iou = max_iou; t->max_age = max_age; t->n_init = n_init;
    t->budget = budget > 0 ? budget : 0;
    t->free_slots.resize(tcap);
    for (int i = 0; i < tcap; ++i) t->free_slots[i] = slot_base + tcap - 1 - i;      // pop_back hands out base+0,1,2,...
    return t;
}

}  // namespace

namespace ddk {

// n trackers sharing one pool (slots [i*tcap, (i+1)*tcap) each)
int tracker_group_create(dd_ctx *ctx, int n, double max_cos, double max_iou, int max_age, int n_init, int budget, int tcap,
                         int gcap, dd_tracker **out) {
    TrackerPool *pool = nullptr;
    int rc = pool_create(ctx, n * tcap, gcap, &pool);             // gcap: initial rows per track the chunk table is sized for
    if (rc != DD_OK) return rc;
    for (int i = 0; i < n; ++i) out[i] = tracker_new(pool, i * tcap, tcap, max_cos, max_iou, max_age, n_init, budget);
    return DD_OK;
}

// tracker.py:51-57 + track.py:113-125 for every tracker of the group: one copy, one launch, no sync.
int trackers_predict(dd_tracker **ts, int S) {
    if (S <= 0) return DD_OK;
    TrackerPool *p = ts[0]->pool;
    hipStream_t s = p->ctx->stream;
    int n = 0;
    for (int z = 0; z < S; ++z) {
        dd_tracker *t = ts[z];
        for (int sl : t->pending_free) { gallery_free_slot(p, sl); t->free_slots.push_back(sl); }
        t->pending_free.clear();
        n += (int)t->tracks.size();
    }
    if (n == 0) return DD_OK;
    int rc;
    if (p->pred_inflight) DD_HIP(hipStreamSynchronize(s));      // two predicts in a row: h_pred is still being read
    p->pred_inflight = true;
    if ((rc = p->h_pred.reserve((size_t)n * sizeof(int))) != DD_OK) return rc;
    if ((rc = p->d_pred.reserve((size_t)n * sizeof(int))) != DD_OK) return rc;
    int *h = p->h_pred.as<int>();
    int k = 0;
    for (int z = 0; z < S; ++z)
        for (auto &tr : ts[z]->tracks) {
            h[k++] = tr.slot;
            tr.age += 1;
            tr.tsu += 1;
        }
    DD_HIP(hipMemcpyAsync(p->d_pred.p, h, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    return ddk::kf_predict(s, p->d_means, p->d_covs, p->d_pred.as<int>(), n);
}

// tracker.py:59-93, phase 1 for the whole group.  tlwh_host: [D][4] f64 for all streams, stream z owns
// detections [det_off[z], det_off[z+1]); feats likewise [D][128] f32 (device or host).  Enqueue only.
int trackers_update_begin(dd_tracker **ts, int S, const double *tlwh_host, const float *feats, int feats_on_device,
                          const int *det_off) {
    TrackerPool *p = ts[0]->pool;
    hipStream_t s = p->ctx->stream;
    int rc;
    // an update that failed between gallery_place and gallery_flush (capacity) leaves table entries pending: the association
    // launch below must see them
    if ((p->tab_need > p->tab_stride || !p->tab_upd.empty()) && (rc = gallery_flush(p, s)) != DD_OK) return rc;
    p->cur.assign(ts, ts + S);
    p->det_off.assign(det_off, det_off + S + 1);
    p->cost_base.assign(S, 0);
    const int D = det_off[S];
    int R = 0, maxn = 0;
    size_t cost_total = 0;
    for (int z = 0; z < S; ++z) {
        dd_tracker *t = ts[z];
        t->last_pairs.clear();
        for (auto &tr : t->tracks) tr.last_det = -1;
        t->ph_n = det_off[z + 1] - det_off[z];
        t->ph_T = (int)t->tracks.size();
        p->cost_base[z] = (int)cost_total;
        cost_total += (size_t)2 * t->ph_T * t->ph_n;
        R += t->ph_T;
        maxn = std::max(maxn, t->ph_n);
    }
    p->D = D; p->R = R; p->have_cost = false;
    if (D == 0) return DD_OK;
    // ---- stage inputs: [det tlwh f64 D*4][8 int arrays of R rows]
    const size_t off_rows = (size_t)D * 4 * sizeof(double);
    const size_t in_bytes = off_rows + (size_t)8 * R * sizeof(int);
    if ((rc = p->h_in.reserve(in_bytes + 64)) != DD_OK) return rc;
    if ((rc = p->d_in.reserve(in_bytes + 64)) != DD_OK) return rc;
    char *h = p->h_in.as<char>();
    memcpy(h, tlwh_host, (size_t)D * 4 * sizeof(double));
    int *hr = reinterpret_cast<int *>(h + off_rows);
    int *h_slot = hr, *h_state = hr + R, *h_tsu = hr + 2 * R, *h_doff = hr + 3 * R, *h_nd = hr + 4 * R, *h_coff = hr + 5 * R,
        *h_idel = hr + 6 * R, *h_gcnt = hr + 7 * R;
    int r = 0;
    for (int z = 0; z < S; ++z) {
        dd_tracker *t = ts[z];
        for (int i = 0; i < t->ph_T; ++i, ++r) {
            h_slot[r] = t->tracks[i].slot;
            h_state[r] = t->tracks[i].state;
            h_tsu[r] = t->tracks[i].tsu;
            h_doff[r] = det_off[z];
            h_nd[r] = t->ph_n;
            h_coff[r] = p->cost_base[z] + i * t->ph_n;
            h_idel[r] = t->ph_T * t->ph_n;
            h_gcnt[r] = gallery_count(p, t->tracks[i].slot, t->budget);
        }
    }
    char *d = p->d_in.as<char>();
    DD_HIP(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    const float *raw = feats;
    if (!feats_on_device) {
        if ((rc = p->d_feats_raw.reserve((size_t)D * 128 * sizeof(float))) != DD_OK) return rc;
        DD_HIP(hipMemcpyAsync(p->d_feats_raw.p, feats, (size_t)D * 128 * sizeof(float), hipMemcpyHostToDevice, s));
        raw = p->d_feats_raw.as<float>();
    }
    if ((rc = p->d_feats_n.reserve((size_t)D * 128 * sizeof(float))) != DD_OK) return rc;
    if ((rc = ddk::normalize_rows(s, raw, p->d_feats_n.as<float>(), D)) != DD_OK) return rc;
    if (R > 0 && cost_total > 0) {
        if ((rc = p->d_cost.reserve(cost_total * sizeof(double))) != DD_OK) return rc;
        if ((rc = p->h_cost.reserve(cost_total * sizeof(double))) != DD_OK) return rc;
        const int *dr = reinterpret_cast<const int *>(d + off_rows);
        hipLaunchKernelGGL(tracker_assoc_k, dim3(R, dd_ceil_div(maxn, 64)), dim3(256), 0, s, p->d_means, p->d_covs,
                           p->d_arenas, p->d_tab, p->tab_stride, dr, dr + R, dr + 2 * R, dr + 3 * R, dr + 4 * R, dr + 5 * R,
                           dr + 6 * R, dr + 7 * R, reinterpret_cast<const double *>(d), p->d_feats_n.as<float>(),
                           p->d_cost.as<double>());
        DD_LAUNCH_CHECK();
        DD_HIP(hipMemcpyAsync(p->h_cost.p, p->d_cost.p, cost_total * sizeof(double), hipMemcpyDeviceToHost, s));
        p->have_cost = true;
    }
    return DD_OK;
}

// phase 2 (the cost matrices have landed): per stream matching cascade + LSAP + track management on
// the host, then ONE launch for all Kalman updates / new tracks / gallery appends of the group and
// ONE gather of the means.  The per-stream part touches only that stream's tracker, so the streams are spread over
// the host pool (hostpool.h); gallery chunks are then handed out serially in stream order, as a single thread would.
namespace {
// tracker.py:59-93 for one stream: decisions + integer book-keeping.  Fills t->m_upd / t->m_new with (slot, detection) pairs.
int match_one_stream(TrackerPool *p, dd_tracker *t, int z) {
    const int n = t->ph_n, T = t->ph_T, doff = p->det_off[z];
    std::vector<std::pair<int, int>> matches;
    std::vector<int> un_rows_final, un_dets, lvl_rows, tmp_rows, tmp_dets, confirmed, unconfirmed;
    t->m_upd.clear(); t->m_new.clear();
    t->ph_cost_base = (n > 0 && T > 0 && p->have_cost) ? p->cost_base[z] : -1;   // parity aid (dd_tracker_last_cost)
    if (n > 0 && T > 0) {
        const double *app = p->h_cost.as<double>() + p->cost_base[z], *iou = app + (size_t)T * n;
        // ---- tracker.py:95-133 _match
        for (int i = 0; i < T; ++i) (t->tracks[i].state == CONFIRMED ? confirmed : unconfirmed).push_back(i);
        un_dets.resize(n);
        std::iota(un_dets.begin(), un_dets.end(), 0);
        for (int level = 0; level < t->max_age; ++level) {            // linear_assignment.py:78-141
            if (un_dets.empty()) break;
            lvl_rows.clear();
            for (int k : confirmed) if (t->tracks[k].tsu == 1 + level) lvl_rows.push_back(k);
            if (lvl_rows.empty()) continue;
            min_cost_matching(app, n, t->max_cos, lvl_rows, un_dets, matches, tmp_rows, tmp_dets);
            un_dets = tmp_dets;
        }
        // unmatched_tracks_a = list(set(track_indices) - set(k for k, _ in matches)) (linear_assignment.py:140):
        // the reference's order is CPython's set iteration order, reproduced by csrc/pyset.cpp, because it
        // becomes the row order of the IoU assignment below (tracker.py:120-123).
        std::vector<int> matched_rows, un_a_all;
        for (auto &m : matches) matched_rows.push_back(m.first);
        ddk::pyset_difference_order(confirmed, matched_rows, un_a_all);
        std::vector<int> iou_rows = unconfirmed, un_rows_a;
        for (int k : un_a_all) {
            if (t->tracks[k].tsu == 1) iou_rows.push_back(k); else un_rows_a.push_back(k);
        }
        std::vector<int> un_rows_b;
        min_cost_matching(iou, n, t->max_iou, iou_rows, un_dets, matches, un_rows_b, tmp_dets);
        un_dets = tmp_dets;
        un_rows_final = un_rows_a;
        un_rows_final.insert(un_rows_final.end(), un_rows_b.begin(), un_rows_b.end());
    } else {
        un_dets.resize(n);
        std::iota(un_dets.begin(), un_dets.end(), 0);
        for (int i = 0; i < T; ++i) un_rows_final.push_back(i);
    }
    // ---- tracker.py:70-79 apply to the integer book-keeping
    for (auto &m : matches) {                                  // track.py:127-152
        TrackRec &tr = t->tracks[m.first];
        tr.hits += 1;
        tr.tsu = 0;
        tr.last_det = m.second;
        if (tr.state == TENTATIVE && tr.hits >= t->n_init) tr.state = CONFIRMED;
        t->m_upd.push_back(tr.slot); t->m_upd.push_back(doff + m.second);
        t->last_pairs.push_back(m.first);
        t->last_pairs.push_back(m.second);
    }
    for (int r : un_rows_final) {                              // track.py:190-196
        TrackRec &tr = t->tracks[r];
        if (tr.state == TENTATIVE) tr.state = DELETED;
        else if (tr.tsu > t->max_age) tr.state = DELETED;
    }
    DD_REQUIRE((int)t->free_slots.size() >= (int)un_dets.size(), DD_E_CAPACITY,
               "dd_tracker_update: track capacity %d exhausted", t->tcap);
    for (int dd : un_dets) {                                   // tracker.py:135-138
        TrackRec tr;
        tr.id = t->next_id++;
        tr.state = TENTATIVE; tr.tsu = 0; tr.hits = 1; tr.age = 1;
        tr.slot = t->free_slots.back();
        t->free_slots.pop_back();
        tr.last_det = dd;
        t->tracks.push_back(tr);
        t->m_new.push_back(tr.slot); t->m_new.push_back(doff + dd);
    }
    // ---- tracker.py:80-81 split live / deleted
    std::vector<TrackRec> live;
    t->deleted.clear();
    for (auto &tr : t->tracks) (tr.state == DELETED ? t->deleted : live).push_back(tr);
    t->tracks.swap(live);
    for (auto &tr : t->deleted) t->pending_free.push_back(tr.slot);
    return DD_OK;
}
}  // namespace

int trackers_update_match(dd_tracker **ts, int S) {
    TrackerPool *p = ts[0]->pool;
    hipStream_t s = p->ctx->stream;
    p->pred_inflight = false;                                   // the caller synchronised before this phase
    int rc;
    const int D = p->D;
    size_t max_pairs = (size_t)p->R + D;
    if ((rc = p->h_pairs.reserve(max_pairs * 3 * sizeof(int) + 64)) != DD_OK) return rc;
    if ((rc = p->d_pairs.reserve(max_pairs * 3 * sizeof(int) + 64)) != DD_OK) return rc;
    std::vector<int> upd_slot, upd_det, upd_row, new_slot, new_det, new_row;
    std::atomic<int> first_err{DD_OK};
    ddk::parallel_for(S, 8, [&](int z0, int z1) {
        for (int z = z0; z < z1; ++z) {
            const int r = match_one_stream(p, ts[z], z);
            if (r != DD_OK) { int ok = DD_OK; first_err.compare_exchange_strong(ok, r); }
        }
    });
    if (first_err.load() != DD_OK) {                               // (dd_last_error is per thread: restate it on the caller's)
        // Every stream has already advanced its host state (tracks, free slots, ids) while no device update followed: the group's host
        // mirrors and its device state now disagree -- the handle must be destroyed and recreated with a larger track_capacity.
        int zbad = -1;
        for (int z = 0; z < S && zbad < 0; ++z) if (ts[z]->free_slots.empty()) zbad = z;
        dd_set_error("dd_tracker_update: track capacity %d exhausted (stream %d of %d in the group); the group's state is no longer consistent: "
                     "recreate it with a larger track_capacity", zbad >= 0 ? ts[zbad]->tcap : ts[0]->tcap, zbad, S);
        return first_err.load();
    }
    for (int z = 0; z < S; ++z) {                                  // gallery placement: serial, stream order (track.py:140 features.append)
        dd_tracker *t = ts[z];
        for (size_t i = 0; i + 1 < t->m_upd.size(); i += 2) {
            int grow = 0;
            if ((rc = gallery_place(p, t->m_upd[i], t->budget, &grow)) != DD_OK) return rc;
            upd_slot.push_back(t->m_upd[i]); upd_det.push_back(t->m_upd[i + 1]); upd_row.push_back(grow);
        }
        for (size_t i = 0; i + 1 < t->m_new.size(); i += 2) {
            int grow = 0;
            if ((rc = gallery_place(p, t->m_new[i], t->budget, &grow)) != DD_OK) return rc;
            new_slot.push_back(t->m_new[i]); new_det.push_back(t->m_new[i + 1]); new_row.push_back(grow);
        }
    }
    const int n_upd = (int)upd_slot.size(), n_new = (int)new_slot.size(), np = n_upd + n_new;
    if (np > 0) {
        int *hp = p->h_pairs.as<int>();
        for (int i = 0; i < n_upd; ++i) { hp[i] = upd_slot[i]; hp[np + i] = upd_det[i]; hp[2 * np + i] = upd_row[i]; }
        for (int i = 0; i < n_new; ++i) { hp[n_upd + i] = new_slot[i]; hp[np + n_upd + i] = new_det[i]; hp[2 * np + n_upd + i] = new_row[i]; }
        int *dp = p->d_pairs.as<int>();
        DD_HIP(hipMemcpyAsync(dp, hp, (size_t)3 * np * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(tracker_apply_k, dim3(dd_ceil_div(np, 4)), dim3(256), 0, s, p->d_means, p->d_covs, p->d_arenas,
                           dp, dp + np, dp + 2 * np, n_upd, n_new, p->d_in.as<double>(), p->d_feats_n.as<float>());
        DD_LAUNCH_CHECK();
    }
    if ((rc = gallery_flush(p, s)) != DD_OK) return rc;           // new chunks are in the table before the next association
    // ---- mirror the means of live + just-deleted tracks of every stream
    int ng = 0;
    for (int z = 0; z < S; ++z) ng += (int)(ts[z]->tracks.size() + ts[z]->deleted.size());
    p->ng = ng;
    if (ng > 0) {
        if ((rc = p->h_gather.reserve((size_t)ng * (sizeof(int) + 8 * sizeof(double)))) != DD_OK) return rc;
        if ((rc = p->d_gather.reserve((size_t)ng * (sizeof(int) + 8 * sizeof(double)))) != DD_OK) return rc;
        double *hg = p->h_gather.as<double>();
        int *hs = reinterpret_cast<int *>(hg + (size_t)ng * 8);
        int k = 0;
        for (int z = 0; z < S; ++z) {
            for (auto &tr : ts[z]->tracks) hs[k++] = tr.slot;
            for (auto &tr : ts[z]->deleted) hs[k++] = tr.slot;
        }
        double *dg = p->d_gather.as<double>();
        int *ds = reinterpret_cast<int *>(dg + (size_t)ng * 8);
        DD_HIP(hipMemcpyAsync(ds, hs, (size_t)ng * sizeof(int), hipMemcpyHostToDevice, s));
        if ((rc = ddk::gather_state(s, p->d_means, p->d_covs, ds, ng, dg, nullptr)) != DD_OK) return rc;
        DD_HIP(hipMemcpyAsync(hg, dg, (size_t)ng * 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    return DD_OK;
}

// phase 3 (the means have landed): mirror them for dd_tracker_read.
int trackers_update_end(dd_tracker **ts, int S) {
    TrackerPool *p = ts[0]->pool;
    const double *hg = p->h_gather.as<double>();
    size_t k = 0;
    for (int z = 0; z < S; ++z) {
        dd_tracker *t = ts[z];
        const size_t nl = t->tracks.size(), nd = t->deleted.size();
        t->live_means.assign(hg + k * 8, hg + (k + nl) * 8);
        k += nl;
        t->dead_means.assign(hg + k * 8, hg + (k + nd) * 8);
        k += nd;
    }
    return DD_OK;
}

// The host mirrors of a tracker (what dd_tracker_read returns without the covariances): no device call, so the pipeline's
// pool threads may use it.  which: 0 live tracks, 1 the tracks the last update deleted.
int tracker_read_host(dd_tracker *t, int which, int64_t *ints6_host, double *means_host) {
    const std::vector<TrackRec> &v = which == 0 ? t->tracks : t->deleted;
    const std::vector<double> &mm = which == 0 ? t->live_means : t->dead_means;
    const int n = (int)v.size();
    if (ints6_host)
        for (int i = 0; i < n; ++i) {
            int64_t *o = ints6_host + (size_t)i * 6;
            o[0] = v[i].id; o[1] = v[i].state; o[2] = v[i].tsu; o[3] = v[i].hits; o[4] = v[i].age; o[5] = v[i].last_det;
        }
    if (means_host && n) {
        DD_REQUIRE(mm.size() == (size_t)n * 8, DD_E_STATE, "dd_tracker_read: means are only mirrored after update()");
        memcpy(means_host, mm.data(), (size_t)n * 8 * sizeof(double));
    }
    return DD_OK;
}

}  // namespace ddk

extern "C" {

int dd_tracker_create(dd_ctx *ctx, double max_cosine_distance, double max_iou_distance, int max_age, int n_init,
                      int nn_budget, int track_capacity, int gallery_capacity, dd_tracker **out) {
    DD_REQUIRE(ctx && out, DD_E_ARG, "dd_tracker_create: NULL argument");
    DD_REQUIRE(track_capacity > 0 && gallery_capacity > 0, DD_E_ARG, "dd_tracker_create: capacities must be > 0");
    return ddk::tracker_group_create(ctx, 1, max_cosine_distance, max_iou_distance, max_age, n_init, nn_budget,
                                     track_capacity, gallery_capacity, out);
}

int dd_tracker_destroy(dd_tracker *t) {
    if (!t) return DD_OK;
    pool_release(t->pool);
    delete t;
    return DD_OK;
}

int dd_tracker_predict(dd_tracker *t) {
    DD_REQUIRE(t, DD_E_ARG, "dd_tracker_predict: NULL tracker");
    DD_DEVICE(t->pool->ctx);
    return ddk::trackers_predict(&t, 1);
}

int dd_tracker_update(dd_tracker *t, const double *tlwh_host, const float *feats, int feats_on_device, int n) {
    DD_REQUIRE(t && n >= 0, DD_E_ARG, "dd_tracker_update: bad argument");
    DD_DEVICE(t->pool->ctx);
    DD_REQUIRE(n == 0 || (tlwh_host && feats), DD_E_ARG, "dd_tracker_update: NULL detections");
    int rc;
    const int off[2] = {0, n};
    hipStream_t s = t->pool->ctx->stream;
    if ((rc = ddk::trackers_update_begin(&t, 1, tlwh_host, feats, feats_on_device, off)) != DD_OK) return rc;
    DD_HIP(hipStreamSynchronize(s));
    if ((rc = ddk::trackers_update_match(&t, 1)) != DD_OK) return rc;
    DD_HIP(hipStreamSynchronize(s));
    return ddk::trackers_update_end(&t, 1);
}

int dd_tracker_count(dd_tracker *t, int which, int *out_n_host) {
    DD_REQUIRE(t && out_n_host, DD_E_ARG, "dd_tracker_count: NULL argument");
    *out_n_host = (int)(which == 0 ? t->tracks.size() : t->deleted.size());
    return DD_OK;
}

int dd_tracker_read(dd_tracker *t, int which, int64_t *ints6_host, double *means_host, double *covs_host) {
    DD_REQUIRE(t, DD_E_ARG, "dd_tracker_read: NULL tracker");
    int rc0 = ddk::tracker_read_host(t, which, ints6_host, means_host);
    if (rc0 != DD_OK) return rc0;
    const std::vector<TrackRec> &v = which == 0 ? t->tracks : t->deleted;
    const int n = (int)v.size();
    if (covs_host && n) {
        DD_DEVICE(t->pool->ctx);
        TrackerPool *p = t->pool;
        hipStream_t s = p->ctx->stream;
        int rc;
        const size_t bytes = (size_t)n * (sizeof(int) + 72 * sizeof(double));
        if ((rc = p->h_gather.reserve(bytes)) != DD_OK) return rc;
        if ((rc = p->d_gather.reserve(bytes)) != DD_OK) return rc;
        double *hg = p->h_gather.as<double>();
        int *hs = reinterpret_cast<int *>(hg + (size_t)n * 72);
        for (int i = 0; i < n; ++i) hs[i] = v[i].slot;
        double *dg = p->d_gather.as<double>();
        int *ds = reinterpret_cast<int *>(dg + (size_t)n * 72);
        DD_HIP(hipMemcpyAsync(ds, hs, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
        if ((rc = ddk::gather_state(s, p->d_means, p->d_covs, ds, n, dg, dg + (size_t)n * 8)) != DD_OK) return rc;
        DD_HIP(hipMemcpyAsync(hg, dg, (size_t)n * 72 * sizeof(double), hipMemcpyDeviceToHost, s));
        DD_HIP(hipStreamSynchronize(s));
        memcpy(covs_host, hg + (size_t)n * 8, (size_t)n * 64 * sizeof(double));
    }
    return DD_OK;
}

// ---- per-track calls the host makes outside Tracker.update (deepdish/framerecords.py:133-165 upstream)
namespace {
int find_live(const dd_tracker *t, int64_t id) {
    for (size_t i = 0; i < t->tracks.size(); ++i) if (t->tracks[i].id == id) return (int)i;
    return -1;
}
}  // namespace

// Track.update(kf, detection) (track.py:127-152) for ONE live track: Kalman update with the detection's box, feature
// appended to the track's gallery, hits += 1, time_since_update = 0, Tentative -> Confirmed once hits >= n_init.
// The mirrored mean of the track is refreshed.  feat: 128 f32 (host or device), normalised here like every feature.
int dd_tracker_track_update(dd_tracker *t, int64_t track_id, const double *tlwh_host, const float *feat, int feat_on_device) {
    DD_REQUIRE(t && tlwh_host && feat, DD_E_ARG, "dd_tracker_track_update: NULL argument");
    DD_DEVICE(t->pool->ctx);
    const int i = find_live(t, track_id);
    DD_REQUIRE(i >= 0, DD_E_ARG, "dd_tracker_track_update: no live track with id %lld", (long long)track_id);
    DD_REQUIRE(t->live_means.size() == t->tracks.size() * 8, DD_E_STATE, "dd_tracker_track_update: call update() first");
    TrackerPool *p = t->pool;
    hipStream_t s = p->ctx->stream;
    TrackRec &tr = t->tracks[i];
    int rc;
    // staging: [tlwh f64 x4][pairs int x3][mean f64 x8 out] | raw feature | normalised feature
    if ((rc = p->h_in.reserve(256)) != DD_OK) return rc;
    if ((rc = p->d_in.reserve(256)) != DD_OK) return rc;
    if ((rc = p->d_feats_raw.reserve(128 * sizeof(float))) != DD_OK) return rc;
    if ((rc = p->d_feats_n.reserve(128 * sizeof(float))) != DD_OK) return rc;
    if ((rc = p->d_gather.reserve(sizeof(int) + 8 * sizeof(double))) != DD_OK) return rc;
    if ((rc = p->h_gather.reserve(sizeof(int) + 8 * sizeof(double))) != DD_OK) return rc;
    char *h = p->h_in.as<char>(), *d = p->d_in.as<char>();
    memcpy(h, tlwh_host, 4 * sizeof(double));
    int *hp = reinterpret_cast<int *>(h + 32);
    int grow = 0;
    if ((rc = gallery_place(p, tr.slot, t->budget, &grow)) != DD_OK) return rc;
    hp[0] = tr.slot; hp[1] = 0; hp[2] = grow;
    DD_HIP(hipMemcpyAsync(d, h, 32 + 3 * sizeof(int), hipMemcpyHostToDevice, s));
    DD_HIP(hipMemcpyAsync(p->d_feats_raw.p, feat, 128 * sizeof(float), feat_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if ((rc = ddk::normalize_rows(s, p->d_feats_raw.as<float>(), p->d_feats_n.as<float>(), 1)) != DD_OK) return rc;
    const int *dp = reinterpret_cast<const int *>(d + 32);
    hipLaunchKernelGGL(tracker_apply_k, dim3(1), dim3(256), 0, s, p->d_means, p->d_covs, p->d_arenas, dp, dp + 1, dp + 2, 1, 0,
                       reinterpret_cast<const double *>(d), p->d_feats_n.as<float>());
    DD_LAUNCH_CHECK();
    if ((rc = gallery_flush(p, s)) != DD_OK) return rc;
    double *dg = p->d_gather.as<double>();
    if ((rc = ddk::gather_state(s, p->d_means, p->d_covs, dp, 1, dg, nullptr)) != DD_OK) return rc;
    DD_HIP(hipMemcpyAsync(p->h_gather.p, dg, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    DD_HIP(hipStreamSynchronize(s));
    memcpy(t->live_means.data() + (size_t)i * 8, p->h_gather.p, 8 * sizeof(double));
    tr.hits += 1;
    tr.tsu = 0;
    if (tr.state == TENTATIVE && tr.hits >= t->n_init) tr.state = CONFIRMED;
    return DD_OK;
}

// Track.predict(kf) (track.py:113-125) for ONE live track: Kalman predict, age += 1, time_since_update += 1.
int dd_tracker_track_predict(dd_tracker *t, int64_t track_id) {
    DD_REQUIRE(t, DD_E_ARG, "dd_tracker_track_predict: NULL tracker");
    DD_DEVICE(t->pool->ctx);
    const int i = find_live(t, track_id);
    DD_REQUIRE(i >= 0, DD_E_ARG, "dd_tracker_track_predict: no live track with id %lld", (long long)track_id);
    DD_REQUIRE(t->live_means.size() == t->tracks.size() * 8, DD_E_STATE, "dd_tracker_track_predict: call update() first");
    TrackerPool *p = t->pool;
    hipStream_t s = p->ctx->stream;
    int rc;
    if (p->pred_inflight) DD_HIP(hipStreamSynchronize(s));
    if ((rc = p->h_pred.reserve(sizeof(int))) != DD_OK) return rc;
    if ((rc = p->d_pred.reserve(sizeof(int))) != DD_OK) return rc;
    if ((rc = p->d_gather.reserve(sizeof(int) + 8 * sizeof(double))) != DD_OK) return rc;
    if ((rc = p->h_gather.reserve(sizeof(int) + 8 * sizeof(double))) != DD_OK) return rc;
    *p->h_pred.as<int>() = t->tracks[i].slot;
    DD_HIP(hipMemcpyAsync(p->d_pred.p, p->h_pred.p, sizeof(int), hipMemcpyHostToDevice, s));
    if ((rc = ddk::kf_predict(s, p->d_means, p->d_covs, p->d_pred.as<int>(), 1)) != DD_OK) return rc;
    if ((rc = ddk::gather_state(s, p->d_means, p->d_covs, p->d_pred.as<int>(), 1, p->d_gather.as<double>(), nullptr)) != DD_OK) return rc;
    DD_HIP(hipMemcpyAsync(p->h_gather.p, p->d_gather.p, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    DD_HIP(hipStreamSynchronize(s));
    p->pred_inflight = false;
    memcpy(t->live_means.data() + (size_t)i * 8, p->h_gather.p, 8 * sizeof(double));
    t->tracks[i].age += 1;
    t->tracks[i].tsu += 1;
    return DD_OK;
}

// Host assignment to track.state / track.time_since_update (framerecords.py:160-161).  state: 1 Tentative, 2 Confirmed
// (a track is deleted with dd_tracker_remove); time_since_update < 0 leaves it unchanged.
int dd_tracker_track_set(dd_tracker *t, int64_t track_id, int state, int time_since_update) {
    DD_REQUIRE(t && (state == TENTATIVE || state == CONFIRMED), DD_E_ARG, "dd_tracker_track_set: state must be 1 or 2");
    const int i = find_live(t, track_id);
    DD_REQUIRE(i >= 0, DD_E_ARG, "dd_tracker_track_set: no live track with id %lld", (long long)track_id);
    t->tracks[i].state = state;
    if (time_since_update >= 0) t->tracks[i].tsu = time_since_update;
    return DD_OK;
}

// The host dropped tracks from tracker.tracks (deepdish.py:1047 assigns framerecords.process_tracking's list): they
// leave the live set at once, their state slots are recycled at the next predict(); unknown ids are an error.
int dd_tracker_remove(dd_tracker *t, const int64_t *track_ids_host, int n) {
    DD_REQUIRE(t && n >= 0 && (n == 0 || track_ids_host), DD_E_ARG, "dd_tracker_remove: bad argument");
    const bool have_means = t->live_means.size() == t->tracks.size() * 8;
    for (int k = 0; k < n; ++k) {
        const int i = find_live(t, track_ids_host[k]);
        DD_REQUIRE(i >= 0, DD_E_ARG, "dd_tracker_remove: no live track with id %lld", (long long)track_ids_host[k]);
        t->pending_free.push_back(t->tracks[i].slot);
        t->tracks.erase(t->tracks.begin() + i);
        if (have_means) t->live_means.erase(t->live_means.begin() + (size_t)i * 8, t->live_means.begin() + (size_t)(i + 1) * 8);
    }
    return DD_OK;
}

int dd_tracker_next_id(dd_tracker *t, int64_t *out_host) {
    DD_REQUIRE(t && out_host, DD_E_ARG, "dd_tracker_next_id: NULL argument");
    *out_host = t->next_id;
    return DD_OK;
}

// Parity aid: the cost matrices the last update() associated with -- appearance [T][n] (gated entries are 1e5, rows of
// unconfirmed tracks unspecified) and IoU [T][n]; T = tracks before that update, n = its detections.  Valid until the
// next update of any tracker of the group.
int dd_tracker_last_cost(dd_tracker *t, double *app_host, double *iou_host, int cap, int *rows_host, int *cols_host) {
    DD_REQUIRE(t && rows_host && cols_host, DD_E_ARG, "dd_tracker_last_cost: NULL argument");
    const bool have = t->ph_cost_base >= 0;
    *rows_host = have ? t->ph_T : 0;
    *cols_host = have ? t->ph_n : 0;
    if (!have || (!app_host && !iou_host)) return DD_OK;
    const size_t tn = (size_t)t->ph_T * t->ph_n;
    DD_REQUIRE((size_t)cap >= tn, DD_E_ARG, "dd_tracker_last_cost: cap %d < %zu", cap, tn);
    const double *c0 = t->pool->h_cost.as<double>() + t->ph_cost_base;
    if (app_host) memcpy(app_host, c0, tn * sizeof(double));
    if (iou_host) memcpy(iou_host, c0 + tn, tn * sizeof(double));
    return DD_OK;
}

int dd_tracker_last_matches(dd_tracker *t, int *pairs_host, int cap, int *out_m_host) {
    DD_REQUIRE(t && out_m_host, DD_E_ARG, "dd_tracker_last_matches: NULL argument");
    const int m = (int)t->last_pairs.size() / 2;
    *out_m_host = m;
    if (pairs_host) {
        DD_REQUIRE(cap >= m, DD_E_ARG, "dd_tracker_last_matches: cap %d < %d", cap, m);
        memcpy(pairs_host, t->last_pairs.data(), (size_t)m * 2 * sizeof(int));
    }
    return DD_OK;
}

}  // extern "C"
