// Per-stream host arithmetic of the batched pipeline that needs no device: tools/intersection.py:4-24 and the box hygiene of
// deepdish.py:940-960.  A header of its own so that csrc/pipeline.hip and the sanitizer harness (tests/sanitize/host_harness.cpp,
// built with -fsanitize=thread / address,undefined by tests/test_sanitizers.py) compile the same statements.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace ddhost {

inline double cross2(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }

// tools/intersection.py:4-24
inline bool seg_intersect(const double p[2], const double pr[2], const double q[2], const double qs[2]) {
    const double eps = 2.220446049250313e-16;
    const double rx = pr[0] - p[0], ry = pr[1] - p[1], sx = qs[0] - q[0], sy = qs[1] - q[1];
    const double rxs = cross2(rx, ry, sx, sy);
    const double mx = q[0] - p[0], my = q[1] - p[1];
    const double qpxr = cross2(mx, my, rx, ry);
    if (fabs(rxs) < eps) {
        if (fabs(qpxr) >= eps) return false;
        const double rr = rx * rx + ry * ry;
        const double ex = rx / rr, ey = ry / rr;
        double t0 = mx * ex + my * ey;
        double t1 = t0 + sx * ex + sy * ey;
        if (t0 > t1) std::swap(t0, t1);
        return !(t1 < 0 || t0 > 1);
    }
    const double t = cross2(mx, my, sx, sy) / rxs, u = qpxr / rxs;
    return 0.0 <= t && t <= 1.0 && 0.0 <= u && u <= 1.0;
}

// deepdish.py:947-953 for one stream: nothing survives a NaN anywhere; x, y clipped to the frame and truncated, w, h clipped to what is
// left of it; boxes over 90 % of the frame dropped.  boxes tlwh f64 rows; appends to the three output vectors.
inline void clean_boxes(const std::vector<double> &boxes, const std::vector<double> &scores, const std::vector<int> &cls, int W, int H,
                        std::vector<int64_t> &ib, std::vector<double> &is, std::vector<int> &ic) {
    const int k0 = (int)scores.size();
    for (double v : boxes) if (v != v) return;
    for (int i = 0; i < k0; ++i) {
        const double *b = boxes.data() + (size_t)i * 4;
        auto clipi = [](double v, double lo, double hi) { return (int64_t)(v < lo ? lo : (v > hi ? hi : v)); };
        const int64_t x = clipi(b[0], 0, W), y = clipi(b[1], 0, H);
        const int64_t w = clipi(b[2], 0, (double)(W - x)), h = clipi(b[3], 0, (double)(H - y));
        if ((double)(w * h) > 0.9 * W * H) continue;
        ib.insert(ib.end(), {x, y, w, h});
        is.push_back(scores[i]);
        ic.push_back(cls[i]);
    }
}

}  // namespace ddhost
