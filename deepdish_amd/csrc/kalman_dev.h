// Device-side bodies of the f64 Kalman filter, shared by kalman.hip and tracker.hip.
// One 64-lane wave owns one track: lane = 8*i + j holds covariance element P[i][j].
// Reference: deep_sort/kalman_filter.py (initiate :55-86, predict :88-123, project :125-152,
// update :154-186, gating_distance :188-229).
#pragma once
#include <hip/hip_runtime.h>

namespace kfdev {

constexpr double W_POS = 1.0 / 20;    // kalman_filter.py:52
constexpr double W_VEL = 1.0 / 160;   // kalman_filter.py:53

__device__ __forceinline__ double shfl64(double v, int src) { return __shfl(v, src, 64); }

// Lower Cholesky factor of a symmetric 4x4 (s is row-major, lower triangle read).
struct Chol4 {
    double l00, l10, l11, l20, l21, l22, l30, l31, l32, l33;
};

__device__ __forceinline__ Chol4 chol4(const double s[16]) {
    Chol4 c;
    c.l00 = sqrt(s[0]);
    c.l10 = s[4] / c.l00;
    c.l20 = s[8] / c.l00;
    c.l30 = s[12] / c.l00;
    c.l11 = sqrt(s[5] - c.l10 * c.l10);
    c.l21 = (s[9] - c.l20 * c.l10) / c.l11;
    c.l31 = (s[13] - c.l30 * c.l10) / c.l11;
    c.l22 = sqrt(s[10] - c.l20 * c.l20 - c.l21 * c.l21);
    c.l32 = (s[14] - c.l30 * c.l20 - c.l31 * c.l21) / c.l22;
    c.l33 = sqrt(s[15] - c.l30 * c.l30 - c.l31 * c.l31 - c.l32 * c.l32);
    return c;
}

// S = P[:4,:4] + diag((h/20)^2, (h/20)^2, 1e-1^2, (h/20)^2)   kalman_filter.py:140-152
__device__ __forceinline__ void innovation_cov(const double *P, double h, double s[16]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) s[a * 4 + b] = P[a * 8 + b];
    const double sp = W_POS * h;
    s[0] += sp * sp;
    s[5] += sp * sp;
    s[10] += 1e-1 * 1e-1;
    s[15] += sp * sp;
}


// kalman_filter.py:55-86 -- z = xyah of the founding detection
__device__ __forceinline__ void initiate_wave(double *P, double *m, const double *z, int lane) {
    const double h = z[3];
    const int i = lane >> 3, j = lane & 7;
    double v = 0.0;
    if (i == j) {
        double sd;
        if (i == 2) sd = 1e-2;
        else if (i == 6) sd = 1e-5;
        else if (i < 4) sd = 2 * W_POS * h;
        else sd = 10 * W_VEL * h;
        v = sd * sd;
    }
    P[lane] = v;
    if (lane < 8) m[lane] = lane < 4 ? z[lane] : 0.0;
}

// kalman_filter.py:88-123
__device__ __forceinline__ void predict_wave(double *P, double *m, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const double p = P[lane];
    const double mv = m[j];
    const double h = shfl64(mv, 3);                       // h BEFORE the step, kalman_filter.py:107-117
    // F (P F^T): B = P F^T adds column j+4 into column j (j < 4); F B adds row i+4 into row i.
    const double pr = shfl64(p, (lane + 4) & 63);
    const double b = j < 4 ? p + pr : p;
    const double bd = shfl64(b, (lane + 32) & 63);
    double r = i < 4 ? b + bd : b;
    if (i == j) {
        double sd;
        if (i == 2) sd = 1e-2;
        else if (i == 6) sd = 1e-5;
        else if (i < 4) sd = W_POS * h;
        else sd = W_VEL * h;
        r += sd * sd;
    }
    const double mo = shfl64(mv, (lane + 4) & 63);        // m[(j+4)&7]
    P[lane] = r;
    if (lane < 8) m[lane] = lane < 4 ? mv + mo : mv;
}

// kalman_filter.py:154-186 -- z = xyah of the matched detection
__device__ __forceinline__ void update_wave(double *P, double *m, const double *z, int lane) {
    const int i = lane >> 3, j = lane & 7;
    const double p = P[lane];
    const double mj = m[j];
    const double h = shfl64(mj, 3);
    double S[16];
    innovation_cov(P, h, S);
    const Chol4 c = chol4(S);
    // Gain row for state component j: solve S x = P[:4, j]   (cho_solve, kalman_filter.py:176-178)
    const double r0 = P[0 * 8 + j], r1 = P[1 * 8 + j], r2 = P[2 * 8 + j], r3 = P[3 * 8 + j];
    const double y0 = r0 / c.l00;
    const double y1 = (r1 - c.l10 * y0) / c.l11;
    const double y2 = (r2 - c.l20 * y0 - c.l21 * y1) / c.l22;
    const double y3 = (r3 - c.l30 * y0 - c.l31 * y1 - c.l32 * y2) / c.l33;
    double kj[4];
    kj[3] = y3 / c.l33;
    kj[2] = (y2 - c.l32 * kj[3]) / c.l22;
    kj[1] = (y1 - c.l21 * kj[2] - c.l31 * kj[3]) / c.l11;
    kj[0] = (y0 - c.l10 * kj[1] - c.l20 * kj[2] - c.l30 * kj[3]) / c.l00;
    // (S K^T)[k][j] then P' = P - K (S K^T)      kalman_filter.py:182-184
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double mkj = S[k * 4 + 0] * kj[0] + S[k * 4 + 1] * kj[1] + S[k * 4 + 2] * kj[2] + S[k * 4 + 3] * kj[3];
        const double kik = shfl64(kj[k], i);              // K[i][k] lives in lane i (row 0, column i)
        acc += kik * mkj;
    }
    // x' = x + (z - Hx) K^T                       kalman_filter.py:179-181
    double dx = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double innov = z[k] - shfl64(mj, k);
        dx += innov * kj[k];
    }
    P[lane] = p - acc;
    if (lane < 8) m[lane] = mj + dx;
}

// kalman_filter.py:188-229 for one (track, detection); c = chol of the track's innovation cov
__device__ __forceinline__ double maha2(const Chol4 &c, const double *m, const double *z, int only_position) {
    const double y0 = (z[0] - m[0]) / c.l00;
    const double y1 = ((z[1] - m[1]) - c.l10 * y0) / c.l11;
    if (only_position) return y0 * y0 + y1 * y1;
    const double y2 = ((z[2] - m[2]) - c.l20 * y0 - c.l21 * y1) / c.l22;
    const double y3 = ((z[3] - m[3]) - c.l30 * y0 - c.l31 * y1 - c.l32 * y2) / c.l33;
    return y0 * y0 + y1 * y1 + y2 * y2 + y3 * y3;
}

}  // namespace kfdev
