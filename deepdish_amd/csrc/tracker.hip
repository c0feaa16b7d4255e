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
// Layout: means[slot][8] f64, covs[slot][64] f64, gallery[slot][gcap][128] f32 (rows already
// L2-normalised, ring buffer); a stream owns the slot range [slot_base, slot_base + tcap) of a pool,
// slots recycled through a per-stream free list.
#include <algorithm>
#include <numeric>
#include "common.h"
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
    const double *__restrict__ means, const double *__restrict__ covs, const float *__restrict__ gallery,
    int gcap, const int *__restrict__ row_slot, const int *__restrict__ row_state, const int *__restrict__ row_tsu,
    const int *__restrict__ row_det_off, const int *__restrict__ row_ndet, const int *__restrict__ row_cost_off,
    const int *__restrict__ row_iou_delta, const int *__restrict__ gal_count, const double *__restrict__ det_tlwh,
    const float *__restrict__ feats_n, double *__restrict__ cost) {
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
        best = nn_max_dot(gallery + (size_t)slot * gcap * 128, gal_count[slot], feats_n + (size_t)det_off * 128, d0,
                          n_det, lane);
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
__global__ __launch_bounds__(256) void tracker_apply_k(
    double *__restrict__ means, double *__restrict__ covs, float *__restrict__ gallery, int gcap,
    int *__restrict__ gal_count, int *__restrict__ gal_total, const int *__restrict__ pair_slot,
    const int *__restrict__ pair_det, const int *__restrict__ pair_cap, int n_upd, int n_new,
    const double *__restrict__ det_tlwh, const float *__restrict__ feats_n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_upd + n_new) return;
    const int slot = pair_slot[w], det = pair_det[w], cap_eff = pair_cap[w];
    double z[4];
    tlwh_to_xyah(det_tlwh + (size_t)det * 4, z);
    double *P = covs + (size_t)slot * 64, *m = means + (size_t)slot * 8;
    int total;
    if (w < n_upd) {
        update_wave(P, m, z, lane);                           // track.py:138-139
        total = gal_total[slot];
    } else {
        initiate_wave(P, m, z, lane);                         // tracker.py:135-138
        total = 0;
    }
    const int pos = total % cap_eff;                          // track.py:140 features.append (ring when full)
    const float2 f = reinterpret_cast<const float2 *>(feats_n + (size_t)det * 128)[lane];
    reinterpret_cast<float2 *>(gallery + ((size_t)slot * gcap + pos) * 128)[lane] = f;
    if (lane == 0) {
        gal_total[slot] = total + 1;
        gal_count[slot] = min(total + 1, cap_eff);
    }
}

struct TrackRec {
    int64_t id;
    int state, tsu, hits, age, slot, last_det;           // slot is pool-global
};

}  // namespace

// Device state shared by the trackers of one group + the group's staging buffers.
struct TrackerPool {
    dd_ctx *ctx = nullptr;
    int slots = 0, gcap = 0, refs = 0;
    double *d_means = nullptr, *d_covs = nullptr;
    float *d_gallery = nullptr;
    int *d_gal_count = nullptr, *d_gal_total = nullptr;
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
    int ph_n = 0, ph_T = 0;
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

int pool_create(dd_ctx *ctx, int slots, int gcap, TrackerPool **out) {
    TrackerPool *p = new TrackerPool();
    p->ctx = ctx; p->slots = slots; p->gcap = gcap;
    DD_HIP(hipSetDevice(ctx->device));
    DD_HIP(hipMalloc(&p->d_means, (size_t)slots * 8 * sizeof(double)));
    DD_HIP(hipMalloc(&p->d_covs, (size_t)slots * 64 * sizeof(double)));
    DD_HIP(hipMalloc(&p->d_gallery, (size_t)slots * gcap * 128 * sizeof(float)));
    DD_HIP(hipMalloc(&p->d_gal_count, (size_t)slots * sizeof(int)));
    DD_HIP(hipMalloc(&p->d_gal_total, (size_t)slots * sizeof(int)));
    DD_HIP(hipMemsetAsync(p->d_gal_count, 0, (size_t)slots * sizeof(int), ctx->stream));
    DD_HIP(hipMemsetAsync(p->d_gal_total, 0, (size_t)slots * sizeof(int), ctx->stream));
    DD_HIP(hipStreamSynchronize(ctx->stream));
    *out = p;
    return DD_OK;
}

void pool_release(TrackerPool *p) {
    if (!p || --p->refs > 0) return;
    (void)hipFree(p->d_means); (void)hipFree(p->d_covs); (void)hipFree(p->d_gallery);
    (void)hipFree(p->d_gal_count); (void)hipFree(p->d_gal_total);
    for (DevBuf *b : {&p->d_pred, &p->d_in, &p->d_feats_raw, &p->d_feats_n, &p->d_cost, &p->d_pairs, &p->d_gather}) b->release();
    for (PinBuf *b : {&p->h_pred, &p->h_in, &p->h_cost, &p->h_pairs, &p->h_gather}) b->release();
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
    int rc = pool_create(ctx, n * tcap, gcap, &pool);
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
        for (int sl : t->pending_free) t->free_slots.push_back(sl);
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
    // ---- stage inputs: [det tlwh f64 D*4][7 int arrays of R rows]
    const size_t off_rows = (size_t)D * 4 * sizeof(double);
    const size_t in_bytes = off_rows + (size_t)7 * R * sizeof(int);
    if ((rc = p->h_in.reserve(in_bytes + 64)) != DD_OK) return rc;
    if ((rc = p->d_in.reserve(in_bytes + 64)) != DD_OK) return rc;
    char *h = p->h_in.as<char>();
    memcpy(h, tlwh_host, (size_t)D * 4 * sizeof(double));
    int *hr = reinterpret_cast<int *>(h + off_rows);
    int *h_slot = hr, *h_state = hr + R, *h_tsu = hr + 2 * R, *h_doff = hr + 3 * R, *h_nd = hr + 4 * R, *h_coff = hr + 5 * R,
        *h_idel = hr + 6 * R;
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
                           p->d_gallery, p->gcap, dr, dr + R, dr + 2 * R, dr + 3 * R, dr + 4 * R, dr + 5 * R, dr + 6 * R,
                           p->d_gal_count, reinterpret_cast<const double *>(d), p->d_feats_n.as<float>(),
                           p->d_cost.as<double>());
        DD_LAUNCH_CHECK();
        DD_HIP(hipMemcpyAsync(p->h_cost.p, p->d_cost.p, cost_total * sizeof(double), hipMemcpyDeviceToHost, s));
        p->have_cost = true;
    }
    return DD_OK;
}

// phase 2 (the cost matrices have landed): per stream matching cascade + LSAP + track management on
// the host, then ONE launch for all Kalman updates / new tracks / gallery appends of the group and
// ONE gather of the means.
int trackers_update_match(dd_tracker **ts, int S) {
    TrackerPool *p = ts[0]->pool;
    hipStream_t s = p->ctx->stream;
    p->pred_inflight = false;                                   // the caller synchronised before this phase
    int rc;
    const int D = p->D;
    size_t max_pairs = (size_t)p->R + D;
    if ((rc = p->h_pairs.reserve(max_pairs * 3 * sizeof(int) + 64)) != DD_OK) return rc;
    if ((rc = p->d_pairs.reserve(max_pairs * 3 * sizeof(int) + 64)) != DD_OK) return rc;
    std::vector<int> upd_slot, upd_det, upd_cap, new_slot, new_det, new_cap;
    std::vector<std::pair<int, int>> matches;
    std::vector<int> un_rows_final, un_dets, lvl_rows, tmp_rows, tmp_dets, confirmed, unconfirmed;
    for (int z = 0; z < S; ++z) {
        dd_tracker *t = ts[z];
        const int n = t->ph_n, T = t->ph_T, doff = p->det_off[z];
        const int cap_eff = t->budget > 0 ? std::min(t->budget, p->gcap) : p->gcap;
        matches.clear(); un_rows_final.clear(); un_dets.clear();
        if (n > 0 && T > 0) {
            const double *app = p->h_cost.as<double>() + p->cost_base[z], *iou = app + (size_t)T * n;
            // ---- tracker.py:95-133 _match
            confirmed.clear(); unconfirmed.clear();
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
            upd_slot.push_back(tr.slot); upd_det.push_back(doff + m.second); upd_cap.push_back(cap_eff);
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
            new_slot.push_back(tr.slot); new_det.push_back(doff + dd); new_cap.push_back(cap_eff);
        }
        // ---- tracker.py:80-81 split live / deleted
        std::vector<TrackRec> live;
        t->deleted.clear();
        for (auto &tr : t->tracks) (tr.state == DELETED ? t->deleted : live).push_back(tr);
        t->tracks.swap(live);
        for (auto &tr : t->deleted) t->pending_free.push_back(tr.slot);
    }
    const int n_upd = (int)upd_slot.size(), n_new = (int)new_slot.size(), np = n_upd + n_new;
    if (np > 0) {
        int *hp = p->h_pairs.as<int>();
        for (int i = 0; i < n_upd; ++i) { hp[i] = upd_slot[i]; hp[np + i] = upd_det[i]; hp[2 * np + i] = upd_cap[i]; }
        for (int i = 0; i < n_new; ++i) { hp[n_upd + i] = new_slot[i]; hp[np + n_upd + i] = new_det[i]; hp[2 * np + n_upd + i] = new_cap[i]; }
        int *dp = p->d_pairs.as<int>();
        DD_HIP(hipMemcpyAsync(dp, hp, (size_t)3 * np * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(tracker_apply_k, dim3(dd_ceil_div(np, 4)), dim3(256), 0, s, p->d_means, p->d_covs, p->d_gallery,
                           p->gcap, p->d_gal_count, p->d_gal_total, dp, dp + np, dp + 2 * np, n_upd, n_new,
                           p->d_in.as<double>(), p->d_feats_n.as<float>());
        DD_LAUNCH_CHECK();
    }
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
    return ddk::trackers_predict(&t, 1);
}

int dd_tracker_update(dd_tracker *t, const double *tlwh_host, const float *feats, int feats_on_device, int n) {
    DD_REQUIRE(t && n >= 0, DD_E_ARG, "dd_tracker_update: bad argument");
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
    if (covs_host && n) {
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
    const int cap_eff = t->budget > 0 ? std::min(t->budget, p->gcap) : p->gcap;
    hp[0] = tr.slot; hp[1] = 0; hp[2] = cap_eff;
    DD_HIP(hipMemcpyAsync(d, h, 32 + 3 * sizeof(int), hipMemcpyHostToDevice, s));
    DD_HIP(hipMemcpyAsync(p->d_feats_raw.p, feat, 128 * sizeof(float), feat_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if ((rc = ddk::normalize_rows(s, p->d_feats_raw.as<float>(), p->d_feats_n.as<float>(), 1)) != DD_OK) return rc;
    const int *dp = reinterpret_cast<const int *>(d + 32);
    hipLaunchKernelGGL(tracker_apply_k, dim3(1), dim3(256), 0, s, p->d_means, p->d_covs, p->d_gallery, p->gcap, p->d_gal_count,
                       p->d_gal_total, dp, dp + 1, dp + 2, 1, 0, reinterpret_cast<const double *>(d), p->d_feats_n.as<float>());
    DD_LAUNCH_CHECK();
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
