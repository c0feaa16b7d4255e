// Background subtraction + motion test: the step between the detector and NMS in the reference's default
// configuration (SURVEY.md section 8 f, n2).
//
// Replaces cv2.createBackgroundSubtractorMOG2() / backSub.apply(frame) (deepdish.py:889,922) and the per-box
// np.count_nonzero(fgMask[y:y+h, x:x+w]) (deepdish.py:957) for S independent streams at once.  The arithmetic
// is OpenCV's (third party, absent from the reference tree and from this image): the adaptive Gaussian-mixture
// model of Zivkovic (ICPR 2004; Zivkovic & van der Heijden, PRL 2006) as published in OpenCV 4.x
// modules/video/src/bgfg_gaussmix2.cpp (MOG2Invoker, detectShadowGMM) -- up to 5 modes per pixel kept sorted
// by weight, f32 throughout, every operation in the published order with FMA contraction off so that the
// numpy restatement in oracle/mog2_np.py is reproduced bit for bit.  Parity with OpenCV itself is unpinned.
//
// HBM layout (per stream z, P = H*W pixels, mode-major planes so a wave touches only the modes its pixels use
// and every access is a full coalesced line):
//   rec [z][5][P] float4 {weight, variance, mean_b, mean_g}     m2 [z][5][P] float mean_r     nm [z][P] u8 modes used
// Algorithmic bytes per pixel with n live modes: 3 (frame) + 1 + 20 n read, 20 n' + 1 + 1 (mask) written
// (n = n' = 1 on a settled background: 46 B).  One lane per pixel, all mode loops unrolled over registers.
#include <algorithm>
#include <cfloat>
#include <vector>
#include "common.h"

namespace {

constexpr int NMIX = 5;

struct Mog2P {
    const uint8_t *frames;
    float4 *rec;
    float *m2;
    uint8_t *nm, *mask, *masked;
    int hw;
    float alphaT, alpha1, prune, Tb, TB, Tg, var_init, var_min, var_max, tau;
    int detect_shadows, shadow_val;
};

struct Mode { float w, var, mu0, mu1, mu2; };

__device__ __forceinline__ void mode_swap(Mode &a, Mode &b) { const Mode t = a; a = b; b = t; }

__global__ __launch_bounds__(256) void mog2_apply_k(const Mog2P P) {
#pragma clang fp contract(off)
    const int px = blockIdx.x * 256 + threadIdx.x;
    if (px >= P.hw) return;
    const size_t z = blockIdx.y;
    const size_t gp = z * (size_t)P.hw + px;
    const uint8_t *src = P.frames + gp * 3;
    const uint8_t b0 = src[0], b1 = src[1], b2 = src[2];
    const float d0 = (float)b0, d1 = (float)b1, d2 = (float)b2;
    int nmodes = P.nm[gp];
    float4 *rec = P.rec + z * NMIX * (size_t)P.hw + px;
    float *m2 = P.m2 + z * NMIX * (size_t)P.hw + px;
    Mode g[NMIX];
#pragma unroll
    for (int m = 0; m < NMIX; ++m) {
        g[m] = Mode{0.f, 0.f, 0.f, 0.f, 0.f};
        if (m < nmodes) {
            const float4 r = rec[(size_t)m * P.hw];
            g[m] = Mode{r.x, r.y, r.z, r.w, m2[(size_t)m * P.hw]};
        }
    }

    // ---- bgfg_gaussmix2.cpp MOG2Invoker::operator(): walk the modes in weight order
    bool background = false, fits = false;
    float total = 0.f;
#pragma unroll
    for (int m = 0; m < NMIX; ++m) {
        if (m < nmodes) {                                   // nmodes shrinks inside the loop, as upstream
            float weight = P.alpha1 * g[m].w + P.prune;
            int pos = m;
            if (!fits) {
                const float var = g[m].var;
                const float e0 = g[m].mu0 - d0, e1 = g[m].mu1 - d1, e2 = g[m].mu2 - d2;
                const float dist2 = e0 * e0 + e1 * e1 + e2 * e2;
                if (total < P.TB && dist2 < P.Tb * var) background = true;
                if (dist2 < P.Tg * var) {
                    fits = true;
                    weight += P.alphaT;
                    const float k = P.alphaT / weight;
                    g[m].mu0 -= k * e0; g[m].mu1 -= k * e1; g[m].mu2 -= k * e2;
                    float vn = var + k * (dist2 - var);
                    vn = vn < P.var_min ? P.var_min : vn;
                    g[m].var = vn > P.var_max ? P.var_max : vn;
                    bool stop = false;
#pragma unroll
                    for (int i = m; i > 0; --i)
                        if (!stop) {
                            if (weight < g[i - 1].w) stop = true;
                            else { mode_swap(g[i], g[i - 1]); pos = i - 1; }
                        }
                }
            }
            if (weight < -P.prune) { weight = 0.f; nmodes--; }
#pragma unroll
            for (int j = 0; j <= m; ++j) if (j == pos) g[j].w = weight;
            total += weight;
        }
    }
    float inv = 0.f;
    if (fabsf(total) > FLT_EPSILON) inv = 1.f / total;
#pragma unroll
    for (int m = 0; m < NMIX; ++m) if (m < nmodes) g[m].w *= inv;

    if (!fits && P.alphaT > 0.f) {                          // replace the weakest mode or add one
        int mode;
        if (nmodes == NMIX) mode = NMIX - 1; else { mode = nmodes; nmodes++; }
#pragma unroll
        for (int j = 0; j < NMIX; ++j) {
            if (j == mode) g[j] = Mode{nmodes == 1 ? 1.f : P.alphaT, P.var_init, d0, d1, d2};
            else if (nmodes != 1 && j < nmodes - 1) g[j].w *= P.alpha1;
        }
        bool stop = false;
#pragma unroll
        for (int i = NMIX - 1; i > 0; --i)
            if (i <= nmodes - 1 && !stop) {
                if (P.alphaT < g[i - 1].w) stop = true;
                else mode_swap(g[i], g[i - 1]);
            }
    }

    P.nm[gp] = (uint8_t)nmodes;
#pragma unroll
    for (int m = 0; m < NMIX; ++m)
        if (m < nmodes) {
            rec[(size_t)m * P.hw] = float4{g[m].w, g[m].var, g[m].mu0, g[m].mu1};
            m2[(size_t)m * P.hw] = g[m].mu2;
        }

    // ---- mask value; detectShadowGMM on the updated model
    int out = 255;
    if (background) out = 0;
    else if (P.detect_shadows) {
        float tw = 0.f;
        bool decided = false, shadow = false;
#pragma unroll
        for (int m = 0; m < NMIX; ++m)
            if (m < nmodes && !decided) {
                float num = 0.f, den = 0.f;
                num += d0 * g[m].mu0; den += g[m].mu0 * g[m].mu0;
                num += d1 * g[m].mu1; den += g[m].mu1 * g[m].mu1;
                num += d2 * g[m].mu2; den += g[m].mu2 * g[m].mu2;
                if (den == 0.f) { decided = true; continue; }
                if (num <= den && num >= P.tau * den) {
                    const float a = num / den;
                    float dist2a = 0.f;
                    const float q0 = a * g[m].mu0 - d0, q1 = a * g[m].mu1 - d1, q2 = a * g[m].mu2 - d2;
                    dist2a += q0 * q0; dist2a += q1 * q1; dist2a += q2 * q2;
                    if (dist2a < P.Tb * g[m].var * a * a) { decided = true; shadow = true; continue; }
                }
                tw += g[m].w;
                if (tw > P.TB) decided = true;
            }
        if (shadow) out = P.shadow_val;
    }
    P.mask[gp] = (uint8_t)out;
    if (P.masked) {                                         // cv2.bitwise_and(frame, frame, mask=fgMask), deepdish.py:924
        uint8_t *dst = P.masked + gp * 3;
        dst[0] = out ? b0 : 0; dst[1] = out ? b1 : 0; dst[2] = out ? b2 : 0;
    }
}

// np.count_nonzero(fgMask[y:y+h, x:x+w]) (deepdish.py:957), one block per box.
__global__ __launch_bounds__(256) void mask_box_count_k(const uint8_t *__restrict__ mask, int H, int W, const int *__restrict__ boxes,
                                                        const int *__restrict__ box_stream, int *__restrict__ counts) {
    __shared__ int part[4];
    const int k = blockIdx.x;
    const int x = boxes[k * 4], y = boxes[k * 4 + 1], w = boxes[k * 4 + 2], h = boxes[k * 4 + 3];
    const uint8_t *m = mask + (size_t)box_stream[k] * H * W;
    int c = 0;
    if (w > 0)
        for (int i = threadIdx.x; i < w * h; i += 256) {
            const int r = i / w, q = i - r * w;
            c += m[(size_t)(y + r) * W + x + q] != 0;
        }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[k] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace

struct dd_mog2 {
    dd_ctx *ctx = nullptr;
    int S = 0, H = 0, W = 0;
    int history = 500, detect_shadows = 1;
    float var_threshold = 16.f;
    long long nframes = 0;
    float4 *rec = nullptr;
    float *m2 = nullptr;
    uint8_t *nm = nullptr;
};

namespace ddk {

int mog2_apply(dd_mog2 *m, hipStream_t s, const uint8_t *frames, double learning_rate, uint8_t *mask, uint8_t *masked) {
    // BackgroundSubtractorMOG2Impl::apply: ++nframes; learningRate < 0 (or first frame) -> 1 / min(2 nframes, history)
    ++m->nframes;
    const double lr = (learning_rate >= 0 && m->nframes > 1) ? learning_rate
                                                              : 1.0 / (double)std::min<long long>(2 * m->nframes, m->history);
    DD_REQUIRE(lr >= 0 && lr <= 1, DD_E_ARG, "dd_mog2_apply: learning rate %g outside [0, 1]", lr);
    Mog2P P;
    P.frames = frames; P.rec = m->rec; P.m2 = m->m2; P.nm = m->nm; P.mask = mask; P.masked = masked;
    P.hw = m->H * m->W;
    P.alphaT = (float)lr; P.alpha1 = 1.f - P.alphaT;
    const float fCT = 0.05f;
    P.prune = -P.alphaT * fCT;
    P.Tb = m->var_threshold; P.TB = 0.9f; P.Tg = 3.0f * 3.0f;
    P.var_init = 15.0f; P.var_min = 4.0f; P.var_max = 5 * 15.0f; P.tau = 0.5f;
    P.detect_shadows = m->detect_shadows; P.shadow_val = 127;
    hipLaunchKernelGGL(mog2_apply_k, dim3((unsigned)dd_ceil_div(P.hw, 256), (unsigned)m->S), dim3(256), 0, s, P);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int mask_box_count(hipStream_t s, const uint8_t *mask, int H, int W, const int *d_boxes, const int *d_box_stream, int K, int *d_counts) {
    if (K <= 0) return DD_OK;
    hipLaunchKernelGGL(mask_box_count_k, dim3((unsigned)K), dim3(256), 0, s, mask, H, W, d_boxes, d_box_stream, d_counts);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace ddk

extern "C" {

int dd_mog2_create(dd_ctx *ctx, int n_streams, int height, int width, int history, double var_threshold, int detect_shadows,
                   dd_mog2 **out) {
    DD_REQUIRE(ctx && out && n_streams > 0 && height > 0 && width > 0 && history > 0 && var_threshold > 0, DD_E_ARG,
               "dd_mog2_create: bad argument");
    DD_REQUIRE((long long)height * width < (1ll << 30), DD_E_ARG, "dd_mog2_create: frame too large");
    DD_HIP(hipSetDevice(ctx->device));
    dd_mog2 *m = new dd_mog2();
    m->ctx = ctx; m->S = n_streams; m->H = height; m->W = width;
    m->history = history; m->var_threshold = (float)var_threshold; m->detect_shadows = detect_shadows != 0;
    const size_t px = (size_t)n_streams * height * width;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->rec), px * NMIX * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->m2), px * NMIX * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->nm), px);
    if (e == hipSuccess) e = hipMemsetAsync(m->nm, 0, px, ctx->stream);        // no modes yet: the first frame creates them
    if (e != hipSuccess) {
        dd_set_error("dd_mog2_create: %s (model state is %zu bytes)", hipGetErrorString(e), px * (NMIX * 20 + 1));
        (void)hipFree(m->rec); (void)hipFree(m->m2); (void)hipFree(m->nm);
        delete m;
        return DD_E_HIP;
    }
    DD_HIP(hipStreamSynchronize(ctx->stream));
    *out = m;
    return DD_OK;
}

int dd_mog2_destroy(dd_mog2 *m) {
    if (!m) return DD_OK;
    (void)hipFree(m->rec); (void)hipFree(m->m2); (void)hipFree(m->nm);
    delete m;
    return DD_OK;
}

int dd_mog2_apply(dd_mog2 *m, const uint8_t *frames, double learning_rate, uint8_t *mask, uint8_t *masked_frames, void *stream) {
    DD_REQUIRE(m && frames && mask, DD_E_ARG, "dd_mog2_apply: NULL argument");
    DD_DEVICE(m->ctx);
    return ddk::mog2_apply(m, dd_pick_stream(m->ctx, stream), frames, learning_rate, mask, masked_frames);
}

int dd_mog2_state(dd_mog2 *m, int stream_index, float *weight_host, float *variance_host, float *mean_host, uint8_t *nmodes_host) {
    DD_REQUIRE(m && stream_index >= 0 && stream_index < m->S && weight_host && variance_host && mean_host && nmodes_host, DD_E_ARG,
               "dd_mog2_state: bad argument");
    DD_DEVICE(m->ctx);
    const size_t hw = (size_t)m->H * m->W;
    std::vector<float4> rec(hw * NMIX);
    std::vector<float> m2(hw * NMIX);
    DD_HIP(hipStreamSynchronize(m->ctx->stream));
    DD_HIP(hipMemcpy(rec.data(), m->rec + (size_t)stream_index * NMIX * hw, rec.size() * sizeof(float4), hipMemcpyDeviceToHost));
    DD_HIP(hipMemcpy(m2.data(), m->m2 + (size_t)stream_index * NMIX * hw, m2.size() * sizeof(float), hipMemcpyDeviceToHost));
    DD_HIP(hipMemcpy(nmodes_host, m->nm + (size_t)stream_index * hw, hw, hipMemcpyDeviceToHost));
    for (int k = 0; k < NMIX; ++k)
        for (size_t i = 0; i < hw; ++i) {
            const bool live = k < nmodes_host[i];                 // planes past a pixel's mode count hold stale values
            const float4 r = rec[k * hw + i];
            weight_host[k * hw + i] = live ? r.x : 0.f;
            variance_host[k * hw + i] = live ? r.y : 0.f;
            mean_host[(k * 3 + 0) * hw + i] = live ? r.z : 0.f;
            mean_host[(k * 3 + 1) * hw + i] = live ? r.w : 0.f;
            mean_host[(k * 3 + 2) * hw + i] = live ? m2[k * hw + i] : 0.f;
        }
    return DD_OK;
}

int dd_mask_box_count(dd_ctx *ctx, const uint8_t *mask, int n_streams, int height, int width, const int *boxes_xywh_host,
                      const int *box_stream_host, int n_boxes, int *counts_host, void *stream) {
    DD_REQUIRE(ctx && mask && n_boxes >= 0 && (n_boxes == 0 || (boxes_xywh_host && box_stream_host && counts_host)), DD_E_ARG,
               "dd_mask_box_count: bad argument");
    DD_DEVICE(ctx);
    if (n_boxes == 0) return DD_OK;
    for (int k = 0; k < n_boxes; ++k) {                     // shapes are checked here, never by a faulting kernel
        const int *b = boxes_xywh_host + (size_t)k * 4;
        DD_REQUIRE(box_stream_host[k] >= 0 && box_stream_host[k] < n_streams && b[0] >= 0 && b[1] >= 0 && b[2] >= 0 && b[3] >= 0 &&
                       b[0] + b[2] <= width && b[1] + b[3] <= height,
                   DD_E_ARG, "dd_mask_box_count: box %d (%d,%d,%d,%d) of stream %d leaves the %dx%d mask", k, b[0], b[1], b[2], b[3],
                   box_stream_host[k], width, height);
    }
    hipStream_t s = dd_pick_stream(ctx, stream);
    int rc;
    const size_t in_bytes = (size_t)n_boxes * 5 * sizeof(int), out_bytes = (size_t)n_boxes * sizeof(int);
    if ((rc = ctx->scratch[0].reserve(in_bytes + out_bytes)) != DD_OK) return rc;
    if ((rc = ctx->pin[0].reserve(in_bytes + out_bytes)) != DD_OK) return rc;
    int *h = ctx->pin[0].as<int>(), *d = ctx->scratch[0].as<int>();
    memcpy(h, boxes_xywh_host, (size_t)n_boxes * 4 * sizeof(int));
    memcpy(h + (size_t)n_boxes * 4, box_stream_host, (size_t)n_boxes * sizeof(int));
    DD_HIP(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, s));
    if ((rc = ddk::mask_box_count(s, mask, height, width, d, d + (size_t)n_boxes * 4, n_boxes, d + (size_t)n_boxes * 5)) != DD_OK) return rc;
    DD_HIP(hipMemcpyAsync(h + (size_t)n_boxes * 5, d + (size_t)n_boxes * 5, out_bytes, hipMemcpyDeviceToHost, s));
    DD_HIP(hipStreamSynchronize(s));
    memcpy(counts_host, h + (size_t)n_boxes * 5, out_bytes);
    return DD_OK;
}

}  // extern "C"
