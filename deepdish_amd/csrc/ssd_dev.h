// The per-anchor arithmetic of TFLite_Detection_PostProcess's first stage (the custom op inside the reference's SSD
// .tflite graph, invoked at tools/ssd_mobilenet.py:103): anchor decode with scales (10, 10, 5, 5), sigmoid of the best class
// logit, score threshold.  One definition for the two places that run it -- ssd_decode_k (post.hip: from the raw head
// matrix) and the SSD head GEMM's epilogue (nets.hip: straight from the accumulators, the head matrix is never written) --
// so that both give the same bits whatever -ffp-contract their translation unit is built with.
#pragma once
#include <hip/hip_runtime.h>

namespace ssddev {

// r = the four box encodings (ty, tx, th, tw) of the anchor, an = its (yc, xc, h, w); best = the largest class logit
// (background excluded).  Writes ymin, xmin, ymax, xmax and returns the score.
__device__ __forceinline__ float decode_anchor(const float r[4], const float an[4], float best, float box[4]) {
#pragma clang fp contract(off)
    const float ay = an[0], ax = an[1], ah = an[2], aw = an[3];
    const float yc = r[0] / 10.f * ah + ay;
    const float xc = r[1] / 10.f * aw + ax;
    const float hh = 0.5f * expf(r[2] / 5.f) * ah;
    const float hw = 0.5f * expf(r[3] / 5.f) * aw;
    box[0] = yc - hh;
    box[1] = xc - hw;
    box[2] = yc + hh;
    box[3] = xc + hw;
    return 1.f / (1.f + expf(-best));
}

}  // namespace ssddev
