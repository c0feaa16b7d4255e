// Batched 8-state constant-velocity Kalman filter in f64 -- one 64-lane wave per track.
//
// Follows deep_sort/kalman_filter.py (upstream paths): initiate :55-86, predict :88-123,
// project :125-152, update :154-186, gating_distance :188-229.  The 8x8 covariance maps onto a
// wave exactly (lane = 8*i + j), so predict/update are a coalesced 512-byte load, a handful of
// wave shuffles and a coalesced store: HBM-bound, 1152 B read + 1152 B written per track.
#include "common.h"
#include "kalman_dev.h"

namespace {
using namespace kfdev;
constexpr int WAVES_PER_BLOCK = 4;

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void kf_initiate_k(
    double *__restrict__ means, double *__restrict__ covs, const int *__restrict__ slots,
    const double *__restrict__ xyah, int n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const int s = slots ? slots[w] : w;
    initiate_wave(covs + (size_t)s * 64, means + (size_t)s * 8, xyah + (size_t)w * 4, lane);
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void kf_predict_k(
    double *__restrict__ means, double *__restrict__ covs, const int *__restrict__ slots, int n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const int s = slots ? slots[w] : w;
    predict_wave(covs + (size_t)s * 64, means + (size_t)s * 8, lane);
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void kf_project_k(
    const double *__restrict__ means, const double *__restrict__ covs, const int *__restrict__ slots,
    int n, double *__restrict__ pm, double *__restrict__ pc) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const int s = slots ? slots[w] : w;
    const double *P = covs + (size_t)s * 64;
    const double *m = means + (size_t)s * 8;
    if (lane < 16) {
        const int a = lane >> 2, b = lane & 3;
        double v = P[a * 8 + b];
        if (a == b) {
            const double sd = a == 2 ? 1e-1 : W_POS * m[3];
            v += sd * sd;
        }
        pc[(size_t)w * 16 + lane] = v;
    }
    if (lane < 4) pm[(size_t)w * 4 + lane] = m[lane];
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void kf_update_k(
    double *__restrict__ means, double *__restrict__ covs, const int *__restrict__ slots,
    const double *__restrict__ xyah, int n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const int s = slots ? slots[w] : w;
    update_wave(covs + (size_t)s * 64, means + (size_t)s * 8, xyah + (size_t)w * 4, lane);
}

// One block per track row; threads stride over detections.
__global__ __launch_bounds__(128) void kf_gate_k(
    const double *__restrict__ means, const double *__restrict__ covs, const int *__restrict__ slots,
    const double *__restrict__ xyah, int n_det, int only_position, double *__restrict__ out) {
    const int row = blockIdx.x;
    const int s = slots ? slots[row] : row;
    const double *P = covs + (size_t)s * 64;
    const double *m = means + (size_t)s * 8;
    double S[16];
    innovation_cov(P, m[3], S);
    const Chol4 c = chol4(S);
    const double mm[4] = {m[0], m[1], m[2], m[3]};
    for (int d = threadIdx.x; d < n_det; d += blockDim.x)
        out[(size_t)row * n_det + d] = maha2(c, mm, xyah + (size_t)d * 4, only_position);
}

__global__ void gather_state_k(const double *__restrict__ means, const double *__restrict__ covs,
                               const int *__restrict__ slots, int n, double *__restrict__ om,
                               double *__restrict__ oc) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const int s = slots[w];
    if (oc) oc[(size_t)w * 64 + lane] = covs[(size_t)s * 64 + lane];
    if (lane < 8) om[(size_t)w * 8 + lane] = means[(size_t)s * 8 + lane];
}

inline int wave_grid(int n) { return dd_ceil_div(n, WAVES_PER_BLOCK); }

}  // namespace

namespace ddk {

int kf_initiate(hipStream_t s, double *means, double *covs, const int *slots, const double *xyah, int n) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(kf_initiate_k, dim3(wave_grid(n)), dim3(64 * WAVES_PER_BLOCK), 0, s, means, covs, slots, xyah, n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int kf_predict(hipStream_t s, double *means, double *covs, const int *slots, int n) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(kf_predict_k, dim3(wave_grid(n)), dim3(64 * WAVES_PER_BLOCK), 0, s, means, covs, slots, n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int kf_project(hipStream_t s, const double *means, const double *covs, const int *slots, int n,
               double *pm, double *pc) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(kf_project_k, dim3(wave_grid(n)), dim3(64 * WAVES_PER_BLOCK), 0, s, means, covs, slots, n, pm, pc);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int kf_update(hipStream_t s, double *means, double *covs, const int *slots, const double *xyah, int n) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(kf_update_k, dim3(wave_grid(n)), dim3(64 * WAVES_PER_BLOCK), 0, s, means, covs, slots, xyah, n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int kf_gate(hipStream_t s, const double *means, const double *covs, const int *slots, int n,
            const double *xyah, int n_det, int only_position, double *out_d2) {
    if (n <= 0 || n_det <= 0) return DD_OK;
    hipLaunchKernelGGL(kf_gate_k, dim3(n), dim3(128), 0, s, means, covs, slots, xyah, n_det, only_position, out_d2);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int gather_state(hipStream_t s, const double *means, const double *covs, const int *slots, int n,
                 double *out_means, double *out_covs) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(gather_state_k, dim3(wave_grid(n)), dim3(64 * WAVES_PER_BLOCK), 0, s, means, covs, slots, n,
                       out_means, out_covs);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace ddk
