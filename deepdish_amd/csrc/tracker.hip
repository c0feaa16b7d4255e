// deep_sort multi-target tracker with its Kalman state and appearance gallery resident in HBM.
//
// Reference (upstream paths): deep_sort/tracker.py:40-138, deep_sort/track.py:67-196,
// deep_sort/linear_assignment.py:11-190, deep_sort/nn_matching.py:137-177,
// deep_sort/iou_matching.py:42-81.
//
// Per frame the device does two launches and the host two small round trips:
//   predict : 1 kernel over all live tracks
//   update  : K1 "assoc"  -- for every (track row, detection): gated cosine NN cost (exact-f32
//                            MFMA + f64 Mahalanobis gate) and the IoU cost, ALL cascade levels at
//                            once (the cost of a pair does not depend on the level);
//             host        -- matching cascade + LSAP on the tiny cost matrices (integer outputs);
//             K2 "apply"  -- Kalman update / initiate + gallery append for the decided pairs.
// Layout: means[slot][8] f64, covs[slot][64] f64, gallery[slot][gcap][128] f32 (rows already
// L2-normalised, ring buffer), one slot per track, slots recycled through a free list.
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

// grid (n_rows, ceil(n_det/64)); 4 waves, 16 detections per wave.
__global__ __launch_bounds__(256) void tracker_assoc_k(
    const double *__restrict__ means, const double *__restrict__ covs, const float *__restrict__ gallery,
    int gcap, const int *__restrict__ row_slot, const int *__restrict__ row_state, const int *__restrict__ row_tsu,
    const int *__restrict__ gal_count, const double *__restrict__ det_tlwh, const float *__restrict__ feats_n,
    int n_det, double *__restrict__ cost_app, double *__restrict__ cost_iou) {
    const int row = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d0 = blockIdx.y * 64 + wave * 16;
    if (d0 >= n_det) return;                                  // wave-uniform
    const int slot = row_slot[row];
    const bool confirmed = row_state[row] == CONFIRMED;
    float best = 0.f;
    if (confirmed)                                            // wave-uniform branch around the MFMAs
        best = nn_max_dot(gallery + (size_t)slot * gcap * 128, gal_count[slot], feats_n, d0, n_det, lane);
    const int d = d0 + (lane & 15);
    if ((lane >> 4) != 0 || d >= n_det) return;
    const double *m = means + (size_t)slot * 8;
    const double *b = det_tlwh + (size_t)d * 4;
    const size_t o = (size_t)row * n_det + d;
    if (confirmed) {
        double S[16], z[4];
        innovation_cov(covs + (size_t)slot * 64, m[3], S);
        const Chol4 c = chol4(S);
        tlwh_to_xyah(b, z);
        const double mm[4] = {m[0], m[1], m[2], m[3]};
        const double d2 = maha2(c, mm, z, 0);
        cost_app[o] = d2 > GATE_4DOF ? INFTY_COST : (double)(1.0f - best);   // linear_assignment.py:181-189
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
    cost_iou[o] = ci;
}

// One wave per decided pair: [0, n_upd) Kalman update + gallery append; [n_upd, n_upd+n_new) new track.
__global__ __launch_bounds__(256) void tracker_apply_k(
    double *__restrict__ means, double *__restrict__ covs, float *__restrict__ gallery, int gcap, int cap_eff,
    int *__restrict__ gal_count, int *__restrict__ gal_total, const int *__restrict__ pair_slot,
    const int *__restrict__ pair_det, int n_upd, int n_new, const double *__restrict__ det_tlwh,
    const float *__restrict__ feats_n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n_upd + n_new) return;
    const int slot = pair_slot[w], det = pair_det[w];
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
    int state, tsu, hits, age, slot, last_det;
};

}  // namespace

struct dd_tracker {
    dd_ctx *ctx = nullptr;
    double max_cos = 0.2, max_iou = 0.7;
    int max_age = 30, n_init = 3, budget = 0, tcap = 0, gcap = 0;
    double *d_means = nullptr, *d_covs = nullptr;
    float *d_gallery = nullptr;
    int *d_gal_count = nullptr, *d_gal_total = nullptr;
    DevBuf d_stage, d_pred, d_feats_raw, d_feats_n, d_cost, d_gather;
    PinBuf h_stage, h_pred, h_cost, h_gather;
    // state carried between the phases of one update (begin -> match -> end)
    int ph_n = 0, ph_T = 0, ph_ng = 0, ph_nl = 0;
    size_t ph_off_pairs = 0;
    bool ph_have_cost = false, pred_inflight = false;
    std::vector<TrackRec> tracks, deleted;
    std::vector<int> free_slots, pending_free;
    std::vector<double> live_means, dead_means;           // host mirrors, [n][8]
    std::vector<int> last_pairs;                          // (track row before update, detection)
    int64_t next_id = 1;
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

}  // namespace

extern "C" {

int dd_tracker_create(dd_ctx *ctx, double max_cosine_distance, double max_iou_distance, int max_age, int n_init,
                      int nn_budget, int track_capacity, int gallery_capacity, dd_tracker **out) {
    DD_REQUIRE(ctx && out, DD_E_ARG, "dd_tracker_create: NULL argument");
    DD_REQUIRE(track_capacity > 0 && gallery_capacity > 0, DD_E_ARG, "dd_tracker_create: capacities must be > 0");
    dd_tracker *t = new dd_tracker();
    t->ctx = ctx;
    t->max_cos = max_cosine_distance;
    t->max_iou = max_iou_distance;
    t->max_age = max_age;
    t->n_init = n_init;
    t->budget = nn_budget > 0 ? nn_budget : 0;
    t->tcap = track_capacity;
    t->gcap = gallery_capacity;
    DD_HIP(hipSetDevice(ctx->device));
    DD_HIP(hipMalloc(&t->d_means, (size_t)t->tcap * 8 * sizeof(double)));
    DD_HIP(hipMalloc(&t->d_covs, (size_t)t->tcap * 64 * sizeof(double)));
    DD_HIP(hipMalloc(&t->d_gallery, (size_t)t->tcap * t->gcap * 128 * sizeof(float)));
    DD_HIP(hipMalloc(&t->d_gal_count, (size_t)t->tcap * sizeof(int)));
    DD_HIP(hipMalloc(&t->d_gal_total, (size_t)t->tcap * sizeof(int)));
    DD_HIP(hipMemsetAsync(t->d_gal_count, 0, (size_t)t->tcap * sizeof(int), ctx->stream));
    DD_HIP(hipMemsetAsync(t->d_gal_total, 0, (size_t)t->tcap * sizeof(int), ctx->stream));
    DD_HIP(hipStreamSynchronize(ctx->stream));
    t->free_slots.resize(t->tcap);
    for (int i = 0; i < t->tcap; ++i) t->free_slots[i] = t->tcap - 1 - i;     // pop_back hands out 0,1,2,...
    *out = t;
    return DD_OK;
}

int dd_tracker_destroy(dd_tracker *t) {
    if (!t) return DD_OK;
    (void)hipFree(t->d_means);
    (void)hipFree(t->d_covs);
    (void)hipFree(t->d_gallery);
    (void)hipFree(t->d_gal_count);
    (void)hipFree(t->d_gal_total);
    t->d_stage.release(); t->d_pred.release(); t->h_pred.release(); t->d_feats_raw.release(); t->d_feats_n.release(); t->d_cost.release(); t->d_gather.release();
    t->h_stage.release(); t->h_cost.release(); t->h_gather.release();
    delete t;
    return DD_OK;
}

}  // extern "C"

namespace ddk {

// tracker.py:51-57 + track.py:113-125.  Enqueue only (own pinned staging block, so the copy may
// still be in flight when update_begin stages its inputs).
int tracker_predict_async(dd_tracker *t) {
    hipStream_t s = t->ctx->stream;
    for (int sl : t->pending_free) t->free_slots.push_back(sl);
    t->pending_free.clear();
    const int n = (int)t->tracks.size();
    if (n == 0) return DD_OK;
    int rc;
    if (t->pred_inflight) DD_HIP(hipStreamSynchronize(s));      // two predicts in a row: h_pred is still being read
    t->pred_inflight = true;
    if ((rc = t->h_pred.reserve((size_t)n * sizeof(int))) != DD_OK) return rc;
    if ((rc = t->d_pred.reserve((size_t)n * sizeof(int))) != DD_OK) return rc;
    int *h = t->h_pred.as<int>();
    for (int i = 0; i < n; ++i) {
        h[i] = t->tracks[i].slot;
        t->tracks[i].age += 1;
        t->tracks[i].tsu += 1;
    }
    DD_HIP(hipMemcpyAsync(t->d_pred.p, h, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
    return ddk::kf_predict(s, t->d_means, t->d_covs, t->d_pred.as<int>(), n);
}

// tracker.py:59-93, phase 1: stage the detections, enqueue the association kernel and the copy of the
// cost matrices back to pinned host memory.  No synchronisation.
int tracker_update_begin(dd_tracker *t, const double *tlwh_host, const float *feats, int feats_on_device, int n) {
    hipStream_t s = t->ctx->stream;
    const int T = (int)t->tracks.size();
    int rc;
    t->last_pairs.clear();
    for (auto &tr : t->tracks) tr.last_det = -1;
    t->ph_n = n; t->ph_T = T; t->ph_have_cost = false;

    // ---- stage inputs: [det tlwh f64 n*4][row_slot T][row_state T][row_tsu T][pairs 2*(T+n)]
    const size_t off_tlwh = 0;
    const size_t off_rows = (size_t)n * 4 * sizeof(double);
    const size_t off_pairs = off_rows + (size_t)3 * T * sizeof(int);
    const size_t stage_bytes = off_pairs + (size_t)2 * (T + n) * sizeof(int) + 64;
    if ((rc = t->h_stage.reserve(stage_bytes)) != DD_OK) return rc;
    if ((rc = t->d_stage.reserve(stage_bytes)) != DD_OK) return rc;
    char *h = t->h_stage.as<char>();
    char *d = t->d_stage.as<char>();
    if (n) memcpy(h + off_tlwh, tlwh_host, (size_t)n * 4 * sizeof(double));
    int *h_slot = reinterpret_cast<int *>(h + off_rows), *h_state = h_slot + T, *h_tsu = h_state + T;
    for (int i = 0; i < T; ++i) {
        h_slot[i] = t->tracks[i].slot;
        h_state[i] = t->tracks[i].state;
        h_tsu[i] = t->tracks[i].tsu;
    }
    t->ph_off_pairs = off_pairs;
    const double *d_tlwh = reinterpret_cast<const double *>(d + off_tlwh);
    const int *d_slot = reinterpret_cast<const int *>(d + off_rows), *d_state = d_slot + T, *d_tsu = d_state + T;
    const float *d_feats_n = nullptr;

    if (n > 0) {
        DD_HIP(hipMemcpyAsync(d, h, off_pairs, hipMemcpyHostToDevice, s));
        const float *raw = feats;
        if (!feats_on_device) {
            if ((rc = t->d_feats_raw.reserve((size_t)n * 128 * sizeof(float))) != DD_OK) return rc;
            DD_HIP(hipMemcpyAsync(t->d_feats_raw.p, feats, (size_t)n * 128 * sizeof(float), hipMemcpyHostToDevice, s));
            raw = t->d_feats_raw.as<float>();
        }
        if ((rc = t->d_feats_n.reserve((size_t)n * 128 * sizeof(float))) != DD_OK) return rc;
        if ((rc = ddk::normalize_rows(s, raw, t->d_feats_n.as<float>(), n)) != DD_OK) return rc;
        d_feats_n = t->d_feats_n.as<float>();
    }

    if (n > 0 && T > 0) {
        const size_t cbytes = (size_t)2 * T * n * sizeof(double);
        if ((rc = t->d_cost.reserve(cbytes)) != DD_OK) return rc;
        if ((rc = t->h_cost.reserve(cbytes)) != DD_OK) return rc;
        double *d_app = t->d_cost.as<double>(), *d_iou = d_app + (size_t)T * n;
        hipLaunchKernelGGL(tracker_assoc_k, dim3(T, dd_ceil_div(n, 64)), dim3(256), 0, s, t->d_means, t->d_covs,
                           t->d_gallery, t->gcap, d_slot, d_state, d_tsu, t->d_gal_count, d_tlwh, d_feats_n, n,
                           d_app, d_iou);
        DD_LAUNCH_CHECK();
        DD_HIP(hipMemcpyAsync(t->h_cost.p, t->d_cost.p, cbytes, hipMemcpyDeviceToHost, s));
        t->ph_have_cost = true;
    }
    return DD_OK;
}

// phase 2 (the cost matrices have landed): matching cascade + LSAP + track management on the host,
// then enqueue the Kalman updates / new tracks / gallery appends and the copy of the means.
int tracker_update_match(dd_tracker *t) {
    hipStream_t s = t->ctx->stream;
    t->pred_inflight = false;                                   // the caller synchronised before this phase
    const int n = t->ph_n, T = t->ph_T;
    int rc;
    char *h = t->h_stage.as<char>();
    char *d = t->d_stage.as<char>();
    const size_t off_pairs = t->ph_off_pairs;
    const double *d_tlwh = reinterpret_cast<const double *>(d);
    const float *d_feats_n = n > 0 ? t->d_feats_n.as<float>() : nullptr;
    std::vector<std::pair<int, int>> matches;
    std::vector<int> un_rows_final, un_dets;
    if (t->ph_have_cost) {
        const double *app = t->h_cost.as<double>(), *iou = app + (size_t)T * n;

        // ---- tracker.py:95-133 _match
        std::vector<int> confirmed, unconfirmed;
        for (int i = 0; i < T; ++i) (t->tracks[i].state == CONFIRMED ? confirmed : unconfirmed).push_back(i);
        un_dets.resize(n);
        std::iota(un_dets.begin(), un_dets.end(), 0);
        // linear_assignment.py:78-141 matching_cascade
        std::vector<int> lvl_rows, tmp_rows, tmp_dets;
        for (int level = 0; level < t->max_age; ++level) {
            if (un_dets.empty()) break;
            lvl_rows.clear();
            for (int k : confirmed) if (t->tracks[k].tsu == 1 + level) lvl_rows.push_back(k);
            if (lvl_rows.empty()) continue;
            min_cost_matching(app, n, t->max_cos, lvl_rows, un_dets, matches, tmp_rows, tmp_dets);
            un_dets = tmp_dets;
        }
        std::vector<char> matched_row(T, 0);
        for (auto &m : matches) matched_row[m.first] = 1;
        // set(track_indices) - matched: ascending row order here (the reference's order is CPython's
        // set iteration order; it only permutes LSAP rows, see DESIGN.md "known order dependence")
        std::vector<int> iou_rows = unconfirmed, un_rows_a;
        for (int k : confirmed) {
            if (matched_row[k]) continue;
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
    int *h_pair_slot = reinterpret_cast<int *>(h + off_pairs), *h_pair_det = h_pair_slot + (T + n);
    int np = 0;
    for (auto &m : matches) {                                  // track.py:127-152
        TrackRec &tr = t->tracks[m.first];
        tr.hits += 1;
        tr.tsu = 0;
        tr.last_det = m.second;
        if (tr.state == TENTATIVE && tr.hits >= t->n_init) tr.state = CONFIRMED;
        h_pair_slot[np] = tr.slot;
        h_pair_det[np] = m.second;
        ++np;
        t->last_pairs.push_back(m.first);
        t->last_pairs.push_back(m.second);
    }
    const int n_upd = np;
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
        tr.state = TENTATIVE;
        tr.tsu = 0;
        tr.hits = 1;
        tr.age = 1;
        tr.slot = t->free_slots.back();
        t->free_slots.pop_back();
        tr.last_det = dd;
        t->tracks.push_back(tr);
        h_pair_slot[np] = tr.slot;
        h_pair_det[np] = dd;
        ++np;
    }
    const int n_new = np - n_upd;
    if (np > 0) {
        // ship both index lists as one block: [slot 0..np)[det 0..np)]
        memmove(h_pair_slot + np, h_pair_det, (size_t)np * sizeof(int));
        int *d_pair = reinterpret_cast<int *>(d + off_pairs);
        DD_HIP(hipMemcpyAsync(d_pair, h_pair_slot, (size_t)2 * np * sizeof(int), hipMemcpyHostToDevice, s));
        const int cap_eff = t->budget > 0 ? std::min(t->budget, t->gcap) : t->gcap;
        hipLaunchKernelGGL(tracker_apply_k, dim3(dd_ceil_div(np, 4)), dim3(256), 0, s, t->d_means, t->d_covs,
                           t->d_gallery, t->gcap, cap_eff, t->d_gal_count, t->d_gal_total, d_pair, d_pair + np,
                           n_upd, n_new, d_tlwh, d_feats_n);
        DD_LAUNCH_CHECK();
    }

    // ---- tracker.py:80-81 split live / deleted, then mirror the means of both on the host
    std::vector<TrackRec> live;
    t->deleted.clear();
    for (auto &tr : t->tracks) (tr.state == DELETED ? t->deleted : live).push_back(tr);
    t->tracks.swap(live);
    for (auto &tr : t->deleted) t->pending_free.push_back(tr.slot);
    const int nl = (int)t->tracks.size(), nd = (int)t->deleted.size(), ng = nl + nd;
    t->live_means.assign((size_t)nl * 8, 0.0);
    t->dead_means.assign((size_t)nd * 8, 0.0);
    if (ng > 0) {
        if ((rc = t->h_gather.reserve((size_t)ng * (sizeof(int) + 8 * sizeof(double)))) != DD_OK) return rc;
        if ((rc = t->d_gather.reserve((size_t)ng * (sizeof(int) + 8 * sizeof(double)))) != DD_OK) return rc;
        double *hg = t->h_gather.as<double>();
        int *hs = reinterpret_cast<int *>(hg + (size_t)ng * 8);
        for (int i = 0; i < nl; ++i) hs[i] = t->tracks[i].slot;
        for (int i = 0; i < nd; ++i) hs[nl + i] = t->deleted[i].slot;
        double *dg = t->d_gather.as<double>();
        int *ds = reinterpret_cast<int *>(dg + (size_t)ng * 8);
        DD_HIP(hipMemcpyAsync(ds, hs, (size_t)ng * sizeof(int), hipMemcpyHostToDevice, s));
        if ((rc = ddk::gather_state(s, t->d_means, t->d_covs, ds, ng, dg, nullptr)) != DD_OK) return rc;
        DD_HIP(hipMemcpyAsync(hg, dg, (size_t)ng * 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    t->ph_ng = ng; t->ph_nl = nl;
    return DD_OK;
}

// phase 3 (the means have landed): mirror them for dd_tracker_read.
int tracker_update_end(dd_tracker *t) {
    const int ng = t->ph_ng, nl = t->ph_nl, nd = ng - nl;
    if (ng > 0) {
        const double *hg = t->h_gather.as<double>();
        memcpy(t->live_means.data(), hg, (size_t)nl * 8 * sizeof(double));
        memcpy(t->dead_means.data(), hg + (size_t)nl * 8, (size_t)nd * 8 * sizeof(double));
    }
    return DD_OK;
}

}  // namespace ddk

extern "C" {

int dd_tracker_predict(dd_tracker *t) {
    DD_REQUIRE(t, DD_E_ARG, "dd_tracker_predict: NULL tracker");
    return ddk::tracker_predict_async(t);
}

int dd_tracker_update(dd_tracker *t, const double *tlwh_host, const float *feats, int feats_on_device, int n) {
    DD_REQUIRE(t && n >= 0, DD_E_ARG, "dd_tracker_update: bad argument");
    DD_REQUIRE(n == 0 || (tlwh_host && feats), DD_E_ARG, "dd_tracker_update: NULL detections");
    int rc;
    if ((rc = ddk::tracker_update_begin(t, tlwh_host, feats, feats_on_device, n)) != DD_OK) return rc;
    DD_HIP(hipStreamSynchronize(t->ctx->stream));
    if ((rc = ddk::tracker_update_match(t)) != DD_OK) return rc;
    DD_HIP(hipStreamSynchronize(t->ctx->stream));
    return ddk::tracker_update_end(t);
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
        hipStream_t s = t->ctx->stream;
        int rc;
        const size_t bytes = (size_t)n * (sizeof(int) + 72 * sizeof(double));
        if ((rc = t->h_gather.reserve(bytes)) != DD_OK) return rc;
        if ((rc = t->d_gather.reserve(bytes)) != DD_OK) return rc;
        double *hg = t->h_gather.as<double>();
        int *hs = reinterpret_cast<int *>(hg + (size_t)n * 72);
        for (int i = 0; i < n; ++i) hs[i] = v[i].slot;
        double *dg = t->d_gather.as<double>();
        int *ds = reinterpret_cast<int *>(dg + (size_t)n * 72);
        DD_HIP(hipMemcpyAsync(ds, hs, (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
        if ((rc = ddk::gather_state(s, t->d_means, t->d_covs, ds, n, dg, dg + (size_t)n * 8)) != DD_OK) return rc;
        DD_HIP(hipMemcpyAsync(hg, dg, (size_t)n * 72 * sizeof(double), hipMemcpyDeviceToHost, s));
        DD_HIP(hipStreamSynchronize(s));
        memcpy(covs_host, hg + (size_t)n * 8, (size_t)n * 64 * sizeof(double));
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
