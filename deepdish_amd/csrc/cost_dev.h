// Device-side bodies shared by cost.hip and tracker.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace costdev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// deep_sort/iou_matching.py:7-39 -- plain IoU of two tlwh boxes, no +1 pixel; same operation
// order as the reference so the f64 result is bit-identical (file is built with -ffp-contract=off).
__device__ __forceinline__ double iou_tlwh(const double *a, const double *b) {
    const double ax2 = a[0] + a[2], ay2 = a[1] + a[3];
    const double bx2 = b[0] + b[2], by2 = b[1] + b[3];
    const double tlx = fmax(a[0], b[0]), tly = fmax(a[1], b[1]);
    const double brx = fmin(ax2, bx2), bry = fmin(ay2, by2);
    const double w = fmax(0.0, brx - tlx), h = fmax(0.0, bry - tly);
    const double inter = w * h;
    const double area_a = a[2] * a[3];
    const double area_b = b[2] * b[3];
    return inter / (area_a + area_b - inter);
}

// Where a track's gallery rows live.  FlatRows: one contiguous [count][128] block (dd_cosine_nn_cost).
// ChunkRows: the tracker's growable gallery -- rows in 32-row chunks (16 KiB) drawn from arenas of 4096 chunks
// (64 MiB); `tab` is the track's chunk list.  nn_budget=None upstream keeps every sample of a track
// (deep_sort/nn_matching.py:137-154), so a gallery has no fixed size: chunks are added as a track is matched.
constexpr int GAL_CH_SHIFT = 5, GAL_CH = 1 << GAL_CH_SHIFT;
constexpr int GAL_ARENA_SHIFT = 12, GAL_ARENA_CHUNKS = 1 << GAL_ARENA_SHIFT;

__device__ __forceinline__ float *gal_row(float *const *arenas, int chunk, int off) {
    return arenas[chunk >> GAL_ARENA_SHIFT] + ((size_t)(chunk & (GAL_ARENA_CHUNKS - 1)) * GAL_CH + off) * 128;
}

struct FlatRows {
    const float *base;
    __device__ __forceinline__ const float *row(int r) const { return base + (size_t)r * 128; }
};

struct ChunkRows {
    float *const *arenas;
    const int *tab;
    __device__ __forceinline__ const float *row(int r) const { return gal_row(arenas, tab[r >> GAL_CH_SHIFT], r & (GAL_CH - 1)); }
};

// max over `count` gallery rows of <row[g], feats[d]> for the 16 detections d0..d0+15 of this wave.
// Both operands are already L2-normalised.  Lane (c = lane & 15, q = lane >> 4) returns the max
// for detection d0 + c in every q (reduced across q).  v_mfma_f32_16x16x4_f32: lane supplies
// A[row = lane & 15][k = lane >> 4] and B[k = lane >> 4][col = lane & 15]; D row = 4*(lane>>4)+reg,
// col = lane & 15.  Each lane fetches 4 consecutive k as one 16-byte load and feeds them to four
// MFMAs; A and B use the same k permutation, so every k is summed exactly once.
template <class Rows>
__device__ __forceinline__ float nn_max_dot(const Rows rows, int count,
                                            const float *__restrict__ feats, int d0, int n_d, int lane) {
    const int c = lane & 15, q = lane >> 4;
    const int dd = min(d0 + c, n_d - 1);
    f32x4 b[8];
    const float *fp = feats + (size_t)dd * 128 + 4 * q;
#pragma unroll
    for (int s = 0; s < 8; ++s) b[s] = *reinterpret_cast<const f32x4 *>(fp + 16 * s);
    float best = -__builtin_inff();
    for (int g = 0; g < count; g += 16) {
        const int r = min(g + c, count - 1);                 // clamp: a repeated row cannot change the max
        const float *ap = rows.row(r) + 4 * q;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ap + 16 * s);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[s][j], acc, 0, 0, 0);
        }
        best = fmaxf(best, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
    }
    best = fmaxf(best, __shfl_xor(best, 16, 64));
    best = fmaxf(best, __shfl_xor(best, 32, 64));
    return best;
}

}  // namespace costdev
