// Association cost matrices: IoU (f64) and nearest-neighbour cosine appearance cost (f32 -> f64).
//
// Reference (upstream paths): deep_sort/iou_matching.py:7-81, deep_sort/nn_matching.py:31-54,
// 78-96, 156-177.  The cosine cost is a [G_t x 128] . [128 x D] contraction per target followed by
// a column min; it runs on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, a k-ordered fmaf chain)
// so the values stay f32-faithful to the reference's float32 GEMM.
#include "common.h"
#include "cost_dev.h"

namespace {
using namespace costdev;

__global__ __launch_bounds__(256) void iou_cost_k(const double *__restrict__ tlwh_t, const int *__restrict__ tsu,
                                                  int n_t, const double *__restrict__ tlwh_d, int n_d,
                                                  double *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_t * n_d) return;
    const int t = idx / n_d, d = idx - t * n_d;
    double v;
    if (tsu && tsu[t] > 1) v = 1e5;                          // iou_matching.py:74-76
    else v = 1.0 - iou_tlwh(tlwh_t + (size_t)t * 4, tlwh_d + (size_t)d * 4);
    out[idx] = v;
}

// One wave per 128-float row: a / ||a||   (nn_matching.py:52-53, all f32)
__global__ __launch_bounds__(256) void normalize_rows_k(const float *__restrict__ in, float *__restrict__ out, int n) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    const float2 v = reinterpret_cast<const float2 *>(in + (size_t)w * 128)[lane];
    float ss = v.x * v.x + v.y * v.y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = sqrtf(ss);
    float2 r;
    r.x = v.x / nrm;
    r.y = v.y / nrm;
    reinterpret_cast<float2 *>(out + (size_t)w * 128)[lane] = r;
}

// grid (n_t, ceil(n_d / 64)), 4 waves, each wave owns 16 detections and sweeps the target's gallery.
__global__ __launch_bounds__(256) void cosine_nn_k(const float *__restrict__ gal, const long long *__restrict__ row_start,
                                                   const int *__restrict__ row_count, const float *__restrict__ feats,
                                                   int n_d, double *__restrict__ out, int ld_out) {
    const int t = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d0 = blockIdx.y * 64 + wave * 16;
    if (d0 >= n_d) return;                                    // wave-uniform, no barriers below
    const float best = nn_max_dot(FlatRows{gal + (size_t)row_start[t] * 128}, row_count[t], feats, d0, n_d, lane);
    const int c = lane & 15;
    if ((lane >> 4) == 0 && d0 + c < n_d) out[(size_t)t * ld_out + d0 + c] = (double)(1.0f - best);
}

}  // namespace

namespace ddk {

int iou_cost(hipStream_t s, const double *tlwh_t, const int *tsu, int n_t, const double *tlwh_d, int n_d,
             double *out) {
    if (n_t <= 0 || n_d <= 0) return DD_OK;
    hipLaunchKernelGGL(iou_cost_k, dim3(dd_ceil_div(n_t * n_d, 256)), dim3(256), 0, s, tlwh_t, tsu, n_t, tlwh_d, n_d, out);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int normalize_rows(hipStream_t s, const float *in, float *out, int n) {
    if (n <= 0) return DD_OK;
    hipLaunchKernelGGL(normalize_rows_k, dim3(dd_ceil_div(n, 4)), dim3(256), 0, s, in, out, n);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int cosine_nn_cost(hipStream_t s, const float *gallery_n, const long long *row_start, const int *row_count,
                   int n_t, const float *feats_n, int n_d, double *out, int ld_out) {
    if (n_t <= 0 || n_d <= 0) return DD_OK;
    hipLaunchKernelGGL(cosine_nn_k, dim3(n_t, dd_ceil_div(n_d, 64)), dim3(256), 0, s, gallery_n, row_start, row_count,
                       feats_n, n_d, out, ld_out);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace ddk
