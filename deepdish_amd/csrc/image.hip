// Image front-end kernels (u8, integer arithmetic, HBM/L2-bound):
//   * crop + bilinear resize of detection boxes   tools/generate_detections.py:40-84 (cv2.resize)
//   * Lanczos stretch resize                      tools/ssd_mobilenet.py:54-57, tools/yolov5.py:99 (PIL)
//   * bilinear stretch resize                     tools/tflite_object_detector.py:207-211 (cv2.resize)
// Frames are BGR u8 [H][W][3]; adjacent lanes walk adjacent output pixels of one row, so loads of a
// source row are contiguous and the 9..15 taps of neighbouring outputs hit in L1/L2.
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include <algorithm>
#include <cstdint>
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;       // Pillow Resample.c

// ------------------------------------------------------------------ cv2 INTER_LINEAR on u8
// OpenCV resize.cpp: coordinates in float, coefficients rounded to 11-bit shorts.
__device__ __forceinline__ void lin_coeff(int d, int dst, int src, int &s, int &a0, int &a1) {
    const double scale = 1.0 / ((double)dst / (double)src);
    float f = (float)((d + 0.5) * scale - 0.5);
    int si = (int)floorf(f);
    f -= (float)si;
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= src - 1) { f = 0.f; si = src - 1; }
    s = si;
    a0 = (int)rintf((1.f - f) * 2048.f);
    a1 = (int)rintf(f * 2048.f);
}

struct CropBox { int sx, sy, cw, ch, frame, flip, swap_rb, r2; };   // cw <= 0 marks a box the reference rejects; flip: rows read bottom-up (cv2.flip(frame, 0));
                                                                    // swap_rb: channels written in reverse order (a BGR frame resampled into the RGB a detector reads)

// grid (ceil(oh*ow/256), n); one thread per output pixel (3 channels).
__global__ __launch_bounds__(256) void crop_resize_k(const uint8_t *__restrict__ frames, int H, int W,
                                                     const CropBox *__restrict__ boxes, int oh, int ow,
                                                     uint8_t *__restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= oh * ow) return;
    const CropBox b = boxes[blockIdx.y];
    uint8_t *o = out + ((size_t)blockIdx.y * oh * ow + p) * 3;
    if (b.cw <= 0) { o[0] = o[1] = o[2] = 0; return; }
    const int dy = p / ow, dx = p - dy * ow;
    // flip: row y of the (virtually) flipped frame is row H-1-y of the stored one: start at the flipped first
    // row and walk with a negative row stride
    const size_t row0 = b.flip ? (size_t)(H - 1 - b.sy) : (size_t)b.sy;
    const uint8_t *base = frames + ((size_t)b.frame * H * W + row0 * W + b.sx) * 3;
    const ptrdiff_t rs = b.flip ? -(ptrdiff_t)W * 3 : (ptrdiff_t)W * 3;
    if (b.cw == 2 * ow && b.ch == 2 * oh) {       // exact 2x decimation: INTER_AREA shortcut
        const uint8_t *r0 = base + (ptrdiff_t)(2 * dy) * rs + (size_t)(2 * dx) * 3, *r1 = r0 + rs;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[b.swap_rb ? 2 - c : c] = (uint8_t)((r0[c] + r0[c + 3] + r1[c] + r1[c + 3] + 2) >> 2);
        return;
    }
    int sx, xa0, xa1, sy, ya0, ya1;
    lin_coeff(dx, ow, b.cw, sx, xa0, xa1);
    lin_coeff(dy, oh, b.ch, sy, ya0, ya1);
    const int sx1 = min(sx + 1, b.cw - 1), sy1 = min(sy + 1, b.ch - 1);
    const uint8_t *r0 = base + (ptrdiff_t)sy * rs, *r1 = base + (ptrdiff_t)sy1 * rs;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = r0[sx * 3 + c] * xa0 + r0[sx1 * 3 + c] * xa1;      // scale 2^11
        const int h1 = r1[sx * 3 + c] * xa0 + r1[sx1 * 3 + c] * xa1;
        const int v = (((ya0 * (h0 >> 4)) >> 16) + ((ya1 * (h1 >> 4)) >> 16) + 2) >> 2;
        o[b.swap_rb ? 2 - c : c] = (uint8_t)min(max(v, 0), 255);
    }
}

// The same arithmetic with one thread per FOUR output pixels of a row (ow % 4 == 0): the two source pixels a bilinear tap pair
// needs are six adjacent bytes -- one unaligned 8-byte load per source row instead of six byte loads -- and the twelve output bytes
// leave as three dwords instead of twelve bytes.  crop_resize_k issued 15 memory instructions per output pixel, each a wave of 64
// scattered single bytes: 81 us per 7 680 crops, bound by the address path.  (A load that would run past its frame -- the last
// two pixels of the last row -- falls back to bytes.)
typedef unsigned u2u __attribute__((ext_vector_type(2), aligned(1)));
__device__ __forceinline__ void crop_pair(const uint8_t *p, const uint8_t *end, uint32_t &lo, uint32_t &hi) {      // six bytes at p: lo = p[0..3], hi = p[4..5]
    if (p + 8 <= end) {
        const u2u v = *reinterpret_cast<const u2u *>(p);
        lo = v[0]; hi = v[1];
    } else {                                                      // the last pixels of a frame: byte by byte, nothing past `end`
        uint32_t b[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = p + i < end ? (uint32_t)p[i] : 0u;
        lo = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        hi = b[4] | (b[5] << 8);
    }
}
__device__ __forceinline__ int crop_byte(uint32_t lo, uint32_t hi, int i) { return (int)((i < 4 ? lo >> (8 * i) : hi >> (8 * (i - 4))) & 255u); }

__global__ __launch_bounds__(256) void crop_resize4_k(const uint8_t *__restrict__ frames, int H, int W, const CropBox *__restrict__ boxes,
                                                      int oh, int ow, uint8_t *__restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;        // quad of output pixels
    const int qpr = ow >> 2;
    if (q >= oh * qpr) return;
    const CropBox b = boxes[blockIdx.y];
    const int dy = q / qpr, dx0 = (q - dy * qpr) * 4;
    uint32_t *o = reinterpret_cast<uint32_t *>(out + ((size_t)blockIdx.y * oh * ow + (size_t)dy * ow + dx0) * 3);
    if (b.cw <= 0) { o[0] = o[1] = o[2] = 0; return; }
    const size_t row0 = b.flip ? (size_t)(H - 1 - b.sy) : (size_t)b.sy;
    const uint8_t *fbase = frames + (size_t)b.frame * H * W * 3;
    const uint8_t *base = fbase + (row0 * W + b.sx) * 3;
    const ptrdiff_t rs = b.flip ? -(ptrdiff_t)W * 3 : (ptrdiff_t)W * 3;
    const uint8_t *fend = fbase + (size_t)H * W * 3;
    uint8_t px[4][3];
    if (b.cw == 2 * ow && b.ch == 2 * oh) {                     // exact 2x decimation: INTER_AREA shortcut
        const uint8_t *r0 = base + (ptrdiff_t)(2 * dy) * rs, *r1 = r0 + rs;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint8_t *p0 = r0 + (size_t)(2 * (dx0 + i)) * 3, *p1 = r1 + (size_t)(2 * (dx0 + i)) * 3;
            uint32_t l0, h0, l1, h1;
            crop_pair(p0, fend, l0, h0);
            crop_pair(p1, fend, l1, h1);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                px[i][b.swap_rb ? 2 - c : c] = (uint8_t)((crop_byte(l0, h0, c) + crop_byte(l0, h0, c + 3) + crop_byte(l1, h1, c) + crop_byte(l1, h1, c + 3) + 2) >> 2);
        }
    } else {
        int sy, ya0, ya1;
        lin_coeff(dy, oh, b.ch, sy, ya0, ya1);
        const int sy1 = min(sy + 1, b.ch - 1);
        const uint8_t *r0 = base + (ptrdiff_t)sy * rs, *r1 = base + (ptrdiff_t)sy1 * rs;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int sx, xa0, xa1;
            lin_coeff(dx0 + i, ow, b.cw, sx, xa0, xa1);
            const int second = min(sx + 1, b.cw - 1) == sx ? 0 : 3;        // at the crop's right edge both taps are pixel sx
            const uint8_t *p0 = r0 + (size_t)sx * 3, *p1 = r1 + (size_t)sx * 3;
            uint32_t l0, h0, l1, h1;
            crop_pair(p0, fend, l0, h0);
            crop_pair(p1, fend, l1, h1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int a0 = crop_byte(l0, h0, c) * xa0 + (second ? crop_byte(l0, h0, c + 3) : crop_byte(l0, h0, c)) * xa1;      // scale 2^11
                const int a1 = crop_byte(l1, h1, c) * xa0 + (second ? crop_byte(l1, h1, c + 3) : crop_byte(l1, h1, c)) * xa1;
                const int v = (((ya0 * (a0 >> 4)) >> 16) + ((ya1 * (a1 >> 4)) >> 16) + 2) >> 2;
                px[i][b.swap_rb ? 2 - c : c] = (uint8_t)min(max(v, 0), 255);
            }
        }
    }
    const uint8_t *f = &px[0][0];
#pragma unroll
    for (int w = 0; w < 3; ++w)
        o[w] = (uint32_t)f[4 * w] | ((uint32_t)f[4 * w + 1] << 8) | ((uint32_t)f[4 * w + 2] << 16) | ((uint32_t)f[4 * w + 3] << 24);
}

// ------------------------------------------------------------------ Pillow Lanczos, 2 passes
// Horizontal: src [H][W][src_c] -> tmp [H][w][3]; one thread per (y, xx).
__global__ __launch_bounds__(256) void lanczos_h_k(const uint8_t *__restrict__ src, int H, int W, int src_c,
                                                   int swap_rb, const int *__restrict__ bounds,
                                                   const int *__restrict__ kk, int ksize, int w,
                                                   uint8_t *__restrict__ tmp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * w) return;
    src += (size_t)blockIdx.y * H * W * src_c;
    tmp += (size_t)blockIdx.y * H * w * 3;
    const int y = idx / w, xx = idx - y * w;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int *k = kk + (size_t)xx * ksize;
    const uint8_t *row = src + ((size_t)y * W + xmin) * src_c;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < n; ++x) {
        const int kv = k[x];
        a0 += row[x * src_c + 0] * kv;
        a1 += row[x * src_c + 1] * kv;
        a2 += row[x * src_c + 2] * kv;
    }
    if (swap_rb) { const int t = a0; a0 = a2; a2 = t; }
    uint8_t *o = tmp + (size_t)idx * 3;
    o[0] = (uint8_t)min(max(a0 >> PRECISION_BITS, 0), 255);
    o[1] = (uint8_t)min(max(a1 >> PRECISION_BITS, 0), 255);
    o[2] = (uint8_t)min(max(a2 >> PRECISION_BITS, 0), 255);
}

// Vertical: tmp [H][w*3] -> dst [h][w*3]; one thread per output byte, coalesced along the row.
__global__ __launch_bounds__(256) void lanczos_v_k(const uint8_t *__restrict__ tmp, int H, int rowbytes,
                                                   const int *__restrict__ bounds, const int *__restrict__ kk,
                                                   int ksize, int h, uint8_t *__restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= h * rowbytes) return;
    tmp += (size_t)blockIdx.y * H * rowbytes;
    dst += (size_t)blockIdx.y * h * rowbytes;
    const int yy = idx / rowbytes, xb = idx - yy * rowbytes;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int *k = kk + (size_t)yy * ksize;
    int acc = 1 << (PRECISION_BITS - 1);
    const uint8_t *col = tmp + (size_t)ymin * rowbytes + xb;
    for (int y = 0; y < n; ++y) acc += col[(size_t)y * rowbytes] * k[y];
    dst[idx] = (uint8_t)min(max(acc >> PRECISION_BITS, 0), 255);
}

// Fast variants used when the row geometry allows 16-byte / 4-byte accesses (640x480 -> 300x300 does):
// horizontal: one block per source row, the row staged in LDS with 16-byte loads, taps read from LDS;
// vertical: four consecutive output bytes per lane, one 32-bit load per tap.
constexpr int HROWS = 8;            // source rows per block in the fast horizontal pass
constexpr int HTAPS = 24;           // register-resident taps per output (falls back when ksize is larger)

__global__ __launch_bounds__(320) void lanczos_h_row_k(const uint8_t *__restrict__ src, int H, int W, int src_c,
                                                       int swap_rb, const int *__restrict__ bounds,
                                                       const int *__restrict__ kk, int ksize, int w,
                                                       uint8_t *__restrict__ tmp) {
    extern __shared__ __attribute__((aligned(16))) uint8_t srow[];            // HROWS rows of W*src_c bytes
    const int y0 = blockIdx.x * HROWS;
    const int nr = min(HROWS, H - y0);
    const size_t img = blockIdx.y;
    const int rowbytes = W * src_c;                           // multiple of 16 (checked by the launcher)
    const uint4 *g = reinterpret_cast<const uint4 *>(src + (img * H + y0) * (size_t)rowbytes);
    for (int i = threadIdx.x; i < nr * (rowbytes / 16); i += blockDim.x) reinterpret_cast<uint4 *>(srow)[i] = g[i];
    __syncthreads();
    for (int xx = threadIdx.x; xx < w; xx += blockDim.x) {
        const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
        int kv[HTAPS];
#pragma unroll
        for (int x = 0; x < HTAPS; ++x) kv[x] = x < n ? kk[(size_t)xx * ksize + x] : 0;   // taps once, reused for all rows
        for (int r = 0; r < nr; ++r) {
            const uint8_t *row = srow + (size_t)r * rowbytes + xmin * src_c;
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
#pragma unroll
            for (int x = 0; x < HTAPS; ++x) {
                if (x < n) {
                    a0 += row[x * src_c + 0] * kv[x];
                    a1 += row[x * src_c + 1] * kv[x];
                    a2 += row[x * src_c + 2] * kv[x];
                }
            }
            if (swap_rb) { const int t = a0; a0 = a2; a2 = t; }
            uint8_t *o = tmp + ((img * H + y0 + r) * (size_t)w + xx) * 3;
            o[0] = (uint8_t)min(max(a0 >> PRECISION_BITS, 0), 255);
            o[1] = (uint8_t)min(max(a1 >> PRECISION_BITS, 0), 255);
            o[2] = (uint8_t)min(max(a2 >> PRECISION_BITS, 0), 255);
        }
    }
}

__global__ __launch_bounds__(256) void lanczos_v4_k(const uint8_t *__restrict__ tmp, int H, int rowbytes,
                                                    const int *__restrict__ bounds, const int *__restrict__ kk,
                                                    int ksize, int h, uint8_t *__restrict__ dst) {
    const int q = rowbytes >> 2;                              // rowbytes is a multiple of 4
    const int idx = dd_xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (idx >= h * q) return;
    tmp += (size_t)blockIdx.y * H * rowbytes;
    dst += (size_t)blockIdx.y * h * rowbytes;
    const int yy = idx / q, xb = (idx - yy * q) * 4;
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int *k = kk + (size_t)yy * ksize;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0, a3 = a0;
    const uint8_t *col = tmp + (size_t)ymin * rowbytes + xb;
    for (int y = 0; y < n; ++y) {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(col + (size_t)y * rowbytes);
        const int kv = k[y];
        a0 += (int)(v & 255u) * kv;
        a1 += (int)((v >> 8) & 255u) * kv;
        a2 += (int)((v >> 16) & 255u) * kv;
        a3 += (int)(v >> 24) * kv;
    }
    // hipcc (ROCm 7.2) lowers clamp(a >> 22) | clamp(b >> 22) << 8 to v_ashr_pk_u8_i32, whose upper 16 result
    // bits are not zero on gfx950; the following v_or3 then corrupts bytes 2-3.  Keep the four clamped
    // bytes opaque to the pattern matcher.
    uint32_t b0 = (uint32_t)min(max(a0 >> PRECISION_BITS, 0), 255), b1 = (uint32_t)min(max(a1 >> PRECISION_BITS, 0), 255);
    uint32_t b2 = (uint32_t)min(max(a2 >> PRECISION_BITS, 0), 255), b3 = (uint32_t)min(max(a3 >> PRECISION_BITS, 0), 255);
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    const uint32_t o = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    *reinterpret_cast<uint32_t *>(dst + (size_t)yy * rowbytes + xb) = o;
}

// ------------------------------------------------------------------ Pillow Lanczos on the i8 matrix cores
// Either pass of the separable resample is a banded matrix product over bytes,
//     outT[col][row] = clip8((sum_k coef[col][k] * src[row][start(col group) + k] + 2^21) >> 22),
// (horizontal: row = image row, col = (output x, channel), k runs over the interleaved bytes of the
// source row; vertical: row = (x, channel) of the transposed intermediate, col = output y, k = source y).
// Both write their result transposed, so two passes end in the natural [h][w][3] layout and neither
// needs a byte shuffle.  The 22-bit Pillow coefficients are split into three signed base-256 digits
// (balanced, so each fits i8) and the pixels are biased by -128 (x ^ 0x80); three
// v_mfma_i32_16x16x64_i8 per 64 source bytes give the digit sums exactly in i32 and
// d0 + (d1 << 8) + (d2 << 16) + 128 * sum(coef) is the integer Pillow computes -- bit-exact.
// One wave owns 16 output columns (its coefficient fragments stay in registers) and walks 16-row tiles;
// operands come straight from global memory in MFMA fragment shape: no LDS, no barriers.
typedef int i4v __attribute__((ext_vector_type(4)));

template <int KS>
__global__ __launch_bounds__(256) void band_resample_k(const uint8_t *__restrict__ src, size_t src_img_stride, int R,
                                                       int pitch_s, const int *__restrict__ start,
                                                       const i4v *__restrict__ coef, const int *__restrict__ bias,
                                                       int n_groups, int C, uint8_t *__restrict__ outT,
                                                       size_t out_img_stride, int pitch_o, int tiles_per_chunk, int nx, int ny) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 1-D grid, logical block (bx, by, bz) with bx fastest: the column groups of one image's row chunk read overlapping
    // windows of the same source rows, so they sit next to each other in the logical order and dd_xcd_remap puts them
    // on one XCD (one L2).  Dealt round-robin over the XCDs every image was fetched by all eight of them: PMC had the
    // horizontal pass of 192 640x480 frames at 505 MB read for 177 MB of frames.
    const unsigned v = dd_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = (int)(v % (unsigned)nx), by = (int)((v / (unsigned)nx) % (unsigned)ny), bz = (int)(v / (unsigned)(nx * ny));
    const int g = bx * 4 + wave;
    if (g >= n_groups) return;                                  // whole waves leave; the kernel has no barrier
    const int fr = lane & 15, fq = lane >> 4;
    i4v cf[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) cf[ks][p] = coef[((size_t)(g * KS + ks) * 3 + p) * 64 + lane];
    const int col = g * 16 + fr;
    const int b = col < C ? bias[col] : 0;
    const uint8_t *base = src + bz * src_img_stride + start[g] + fq * 16;
    uint8_t *obase = outT + bz * out_img_stride + (size_t)col * pitch_o + fq * 4;
    const int r_first = by * tiles_per_chunk * 16;
    for (int t = 0; t < tiles_per_chunk; ++t) {
        const int r0 = r_first + t * 16;
        if (r0 >= R) break;
        const uint8_t *p = base + (size_t)min(r0 + fr, R - 1) * pitch_s;
        i4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            i4v x = *reinterpret_cast<const i4v *>(p + ks * 64);
            x ^= (int)0x80808080;                                // u8 -> i8: p - 128
            a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, cf[ks][0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, cf[ks][1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, cf[ks][2], a2, 0, 0, 0);
        }
        uint32_t by[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = (int)((unsigned)a0[i] + ((unsigned)a1[i] << 8) + ((unsigned)a2[i] << 16) + (unsigned)b);
            by[i] = (uint32_t)min(max(v >> PRECISION_BITS, 0), 255);
        }
        asm volatile("" : "+v"(by[0]), "+v"(by[1]), "+v"(by[2]), "+v"(by[3]));     // see lanczos_v4_k: keep v_ashr_pk_u8_i32 away
        if (col < C && r0 + fq * 4 < R)
            *reinterpret_cast<uint32_t *>(obase + r0) = by[0] | (by[1] << 8) | (by[2] << 16) | (by[3] << 24);
    }
}

// Same product, 64 source rows per iteration in four 16-row MFMA tiles whose rows are interleaved (tile j takes
// rows r0 + 16*(i/4) + 4*j + i%4): the lane that owns accumulator rows 4*fq.. of every tile then holds the sixteen
// consecutive rows r0 + 16*fq .. +15 of its output column and writes them with one 16-byte store.  (With one tile
// per iteration a lane wrote 4 bytes at a time and the fabric saw twice the algorithmic write bytes -- PMC.)
// Needs R % 4 == 0 and 4-byte aligned output rows (both passes of the 640x480 -> 300x300 stretch: R = 480 and 900).
template <int KS>
__global__ __launch_bounds__(256) void band_resample_wide_k(const uint8_t *__restrict__ src, size_t src_img_stride, int R,
                                                            int pitch_s, const int *__restrict__ start,
                                                            const i4v *__restrict__ coef, const int *__restrict__ bias,
                                                            int n_groups, int C, uint8_t *__restrict__ outT,
                                                            size_t out_img_stride, int pitch_o, int tiles_per_chunk, int nx, int ny) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned v = dd_xcd_remap(blockIdx.x, gridDim.x);      // see band_resample_k
    const int bx = (int)(v % (unsigned)nx), by = (int)((v / (unsigned)nx) % (unsigned)ny), bz = (int)(v / (unsigned)(nx * ny));
    const int g = bx * 4 + wave;
    if (g >= n_groups) return;                                  // whole waves leave; the kernel has no barrier
    const int fr = lane & 15, fq = lane >> 4;
    i4v cf[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) cf[ks][p] = coef[((size_t)(g * KS + ks) * 3 + p) * 64 + lane];
    const int col = g * 16 + fr;
    const int b = col < C ? bias[col] : 0;
    // Which 16-byte chunks of the window does any of the group's 16 columns use?  This lane supplies chunk fq of every k step to
    // the MFMAs, and the coefficient lanes of chunk fq are its own 16-lane row: a chunk whose coefficients are all zero (the tail
    // of a window that is rounded up to whole 64-byte steps) is not loaded -- the kernel is bound by the number of 16-byte row
    // pieces it pulls through the texture path (PMC: 56 % of its wave cycles parked on s_waitcnt, 6 % issuing).
    bool need[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const i4v o = cf[ks][0] | cf[ks][1] | cf[ks][2];
        const unsigned long long m = __ballot((o[0] | o[1] | o[2] | o[3]) != 0);
        need[ks] = ((m >> (fq * 16)) & 0xFFFFull) != 0;
    }
    const uint8_t *base = src + bz * src_img_stride + start[g] + fq * 16;
    uint8_t *obase = outT + bz * out_img_stride + (size_t)col * pitch_o + fq * 16;
    const int r_first = by * tiles_per_chunk * 64;              // tiles_per_chunk counts 64-row iterations here
    const int row_in_tile = ((fr >> 2) << 4) + (fr & 3);        // + 4*j
    for (int t = 0; t < tiles_per_chunk; ++t) {
        const int r0 = r_first + t * 64;
        if (r0 >= R) break;
        uint32_t word[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint8_t *p = base + (size_t)min(r0 + row_in_tile + 4 * j, R - 1) * pitch_s;
            i4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                i4v x = {0, 0, 0, 0};
                if (need[ks]) x = *reinterpret_cast<const i4v *>(p + ks * 64);
                x ^= (int)0x80808080;                            // u8 -> i8: p - 128
                a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, cf[ks][0], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, cf[ks][1], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, cf[ks][2], a2, 0, 0, 0);
            }
            uint32_t by[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = (int)((unsigned)a0[i] + ((unsigned)a1[i] << 8) + ((unsigned)a2[i] << 16) + (unsigned)b);
                by[i] = (uint32_t)min(max(v >> PRECISION_BITS, 0), 255);
            }
            asm volatile("" : "+v"(by[0]), "+v"(by[1]), "+v"(by[2]), "+v"(by[3]));     // see lanczos_v4_k: keep v_ashr_pk_u8_i32 away
            word[j] = by[0] | (by[1] << 8) | (by[2] << 16) | (by[3] << 24);              // rows r0 + 16*fq + 4*j .. +3
        }
        typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));      // pitch_o is only a multiple of 4 in the vertical pass
        if (col < C) {
            if (r0 + fq * 16 + 16 <= R) {
                *reinterpret_cast<u4a *>(obase + r0) = u4a{word[0], word[1], word[2], word[3]};
            } else {                                             // R % 64 != 0: the last rows in words (R % 4 == 0)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (r0 + fq * 16 + 4 * j + 4 <= R) *reinterpret_cast<uint32_t *>(obase + r0 + 4 * j) = word[j];
            }
        }
    }
}

// Four values of one column -> (v >> 22) clipped to a byte each, packed into a word.  gfx950 has the instruction for it,
// v_ashr_pk_u8_i32: D[7:0] = sat_u8(S0 >> S2), D[15:8] = sat_u8(S1 >> S2) into the half of D that op_sel[3] names, the other half
// kept (which is why hipcc's own use of it, followed by an OR that assumes zeros there, corrupts bytes: lanczos_v4_k).
// The operands must come from ordinary VALU results: hipcc places no wait states between an MFMA and inline assembly that reads
// its destination (tried: 98 % of the bytes wrong).
__device__ __forceinline__ uint32_t band_pack4(int v0, int v1, int v2, int v3) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v0), "v"(v1), "v"(PRECISION_BITS));
    asm("v_ashr_pk_u8_i32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(v2), "v"(v3), "v"(PRECISION_BITS));
    return r;
#else
    return 0;
#endif
}

// The three digit products of four tiles: digit 2's accumulator shifted left by 8 is what digit 1 accumulates onto; digit 0
// accumulates onto the bias in a chain of its own; (hi << 8) + lo = d0 + (d1 << 8) + (d2 << 16) + bias modulo 2^32 -- two VALU
// operations per value instead of four, and with the MFMA results in VGPRs (__launch_bounds__(256, 2)) and band_pack4 a tile costs
// ~20 VALU issues instead of 110 (the first form of the fused kernel spent 72 % of its issue slots on that arithmetic).
template <int KS, int NJ>
__device__ __forceinline__ void band_digits(const i4v (&x)[NJ][KS], const i4v (&cf)[KS][3], int bias, uint32_t (&word)[NJ]) {
    i4v xx[NJ][KS], hi[NJ], lo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xx[j][ks] = x[j][ks] ^ (int)0x80808080;      // u8 -> i8: p - 128 (the bias carries 128 * sum(coef))
#pragma unroll
    for (int j = 0; j < NJ; ++j) { hi[j] = i4v{0, 0, 0, 0}; lo[j] = i4v{bias, bias, bias, bias}; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            hi[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xx[j][ks], cf[ks][2], hi[j], 0, 0, 0);
            lo[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xx[j][ks], cf[ks][0], lo[j], 0, 0, 0);
        }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) hi[j][i] = (int)((unsigned)hi[j][i] << 8);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NJ; ++j) hi[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xx[j][ks], cf[ks][1], hi[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (int)(((unsigned)hi[j][i] << 8) + (unsigned)lo[j][i]);
        word[j] = band_pack4(v[0], v[1], v[2], v[3]);
    }
}

// Both passes in ONE launch, through LDS.  band_resample_wide_k feeds its MFMAs straight from global memory in fragment
// shape: lane = (row fr, 16-byte chunk fq), so every 16-lane pass of a load touches 16 different rows -- 64 cache-line
// look-ups per wave load, and the passes ran at 92 CU cycles per wave load (1.3 M of them for 384 frames: the whole 224 us of
// the horizontal pass), with the transposed intermediate written to and read back from HBM on top (PMC round 3: 406 + 203
// and 169 + 179 MB for 354 MB of frames and 104 MB of output).  Here a block owns 64 columns of the transposed intermediate
// (four 16-column groups, one per wave) of ONE frame:
//   * the union of the four groups' source windows (<= 256 bytes of every row) is loaded coalesced -- 16 lanes x 16 B per
//     row, four rows per wave load -- 64 rows at a time, through registers into a double-buffered LDS tile S[64][272];
//   * the horizontal product reads its fragments from S (chunk c of row r at c ^ 4 (r / 16 % 4), as T below), same three digit MFMAs, same
//     packing; its bytes go to T[64 columns][<= 512 rows] in LDS, 16-byte chunk q of column c stored at q ^ 4 (c / 16 % 4)
//     so that both the writing lanes (16 columns) and the reading lanes (columns 16 a + 4 j + b) fall on 16 bank groups;
//   * the vertical product (output row groups dealt over the waves, coefficient fragments from L2) reads T and stores
//     16 bytes per lane into the [h][w][3] result.
// HBM sees the frame once and the output once.  Same integers as the two-pass form: the same bytes.
constexpr int LF_ROWS = 64, LF_SP = 272, LF_TP = 528, LF_NVG = 5;      // LF_NVG: output row groups per wave (h <= 320)
constexpr int lf_lds_bytes() { return 2 * LF_ROWS * LF_SP + 64 * LF_TP; }

template <int KH, int KV>
__global__ __launch_bounds__(256, 2) void lanczos_fused_k(const uint8_t *__restrict__ src, size_t src_img_stride, int H, int pitch_s,
                                                       const int *__restrict__ start_h, const int *__restrict__ slab_u0,
                                                       const i4v *__restrict__ coef_h, const int *__restrict__ bias_h,
                                                       int n_groups_h, int C, const int *__restrict__ start_v,
                                                       const i4v *__restrict__ coef_v, const int *__restrict__ bias_v,
                                                       int n_groups_v, int h, uint8_t *__restrict__ dst, size_t dst_img_stride,
                                                       int n_slabs) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lf_smem[];
    uint8_t *S = lf_smem, *T = lf_smem + 2 * LF_ROWS * LF_SP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const unsigned v = dd_xcd_remap(blockIdx.x, gridDim.x);      // the slabs of a frame sit next to each other: one XCD, one L2
    const int slab = (int)(v % (unsigned)n_slabs), n = (int)(v / (unsigned)n_slabs);
    const int g = slab * 4 + wave;
    const bool g_ok = g < n_groups_h;                            // wave-uniform; such a wave still loads and meets the barriers
    const int u0 = slab_u0[slab];
    i4v cf[KH][3];
#pragma unroll
    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p) cf[ks][p] = g_ok ? coef_h[((size_t)(g * KH + ks) * 3 + p) * 64 + lane] : i4v{0, 0, 0, 0};
    const int so = g_ok ? start_h[g] - u0 : 0;
    const int bh = (g_ok && g * 16 + fr < C) ? bias_h[g * 16 + fr] : 0;
    // vertical pass: fragments, bias and window of ALL the wave's row groups (every fourth), fetched now: fetched one group ahead
    // each round of that pass waited out an L2 round trip (47 us of 183 at 384 frames)
    i4v cva[LF_NVG][KV][3];
    int bva[LF_NVG], sva[LF_NVG];
#pragma unroll
    for (int i = 0; i < LF_NVG; ++i) {
        const int gg = min(wave + 4 * i, n_groups_v - 1);
#pragma unroll
        for (int ks = 0; ks < KV; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) cva[i][ks][p] = coef_v[((size_t)(gg * KV + ks) * 3 + p) * 64 + lane];
        bva[i] = gg * 16 + fr < h ? bias_v[gg * 16 + fr] : 0;
        sva[i] = start_v[gg];
    }
    const int lc = tid & 15, lr = tid >> 4;                      // load item: 16-byte chunk lc of rows lr + 16 i
    const uint8_t *img = src + (size_t)n * src_img_stride + min(u0 + lc * 16, pitch_s - 16);      // chunks past the row end belong to no window
    const int NC = (H + LF_ROWS - 1) / LF_ROWS;
    constexpr int PF = 4;                                         // chunks in flight (5 and 6 spill at 256 VGPRs and lose: 189 / 199 vs 171 us): a chunk is loaded PF iterations before its product
    i4v rg[PF][4];                                               // (one chunk ahead left the block waiting on HBM every iteration: 2 blocks x 16 KB per CU in flight)
    auto load_chunk = [&](int ch, i4v (&r)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const i4v *>(img + (size_t)min(ch * LF_ROWS + lr + 16 * i, H - 1) * pitch_s);
    };
    auto write_chunk = [&](int buf, const i4v (&r)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<i4v *>(S + buf * (LF_ROWS * LF_SP) + (lr + 16 * i) * LF_SP + ((lc ^ (4 * i)) << 4)) = r[i];
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_chunk(i, rg[i]);            // chunks past the last one re-read row H - 1 (never used)
    write_chunk(0, rg[0]);
    load_chunk(PF, rg[0]);
    __syncthreads();
    const int row_in_tile = ((fr >> 2) << 4) + (fr & 3);        // + 4 j: see band_resample_wide_k
    typedef uint32_t u4l __attribute__((ext_vector_type(4)));
    auto h_chunk = [&](int ch) {
        if (!g_ok) return;
        const uint8_t *sb = S + (ch & 1) * (LF_ROWS * LF_SP);
        const int sc = (so >> 4) + fq, ca = fr >> 2;              // chunk c of row r sits at c ^ 4 (r / 16 % 4): rows 16 a + b + 4 j on 16 bank groups
        // all fragment reads of the round, then all MFMAs, then the packing: written tile by tile hipcc kept that order and every
        // tile waited out its own LDS and MFMA latencies (2.7 k cycles per round, 35 k per block)
        i4v x[4][KH];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < KH; ++ks) x[j][ks] = *reinterpret_cast<const i4v *>(sb + (row_in_tile + 4 * j) * LF_SP + (((sc + ks * 4) ^ (4 * ca)) << 4));
        uint32_t word[4];
        band_digits<KH, 4>(x, cf, bh, word);
        // rows ch * 64 + fq * 16 .. + 15 of column wave * 16 + fr
        *reinterpret_cast<u4l *>(T + (wave * 16 + fr) * LF_TP + (((ch * 4 + fq) ^ (4 * wave)) << 4)) = u4l{word[0], word[1], word[2], word[3]};
    };
    // No branch around the loads or the LDS writes: with `if (ch + 1 < NC)` there hipcc's wait-count pass had paths with different
    // numbers of loads in flight and settled for vmcnt(0) in front of every write_chunk -- the prefetch depth was 1 whatever PF said.
    // Rounds past the last chunk load row H - 1 again and fill an S buffer nobody reads.
    for (int c0 = 0; c0 < NC; c0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {                            // chunk ch = c0 + u: set u held it, set (u + 1) % PF holds chunk ch + 1
            const int ch = c0 + u;
            if (ch >= NC) break;                                  // block-uniform; leaving is no branch AROUND a load
            h_chunk(ch);
            write_chunk((ch + 1) & 1, rg[(u + 1) % PF]);
            load_chunk(ch + 1 + PF, rg[(u + 1) % PF]);
            __syncthreads();
        }
    }
    // ---- vertical product: output rows gv * 16 + fr, bytes slab * 64 + fq * 16 .. + 15 of them
    uint8_t *out = dst + (size_t)n * dst_img_stride;
    const int ca = fr >> 2;                                       // column / 16 of this lane's fragment rows
#pragma unroll
    for (int vi = 0; vi < LF_NVG; ++vi) {
        const int gv = wave + 4 * vi;
        if (gv >= n_groups_v) break;                              // wave-uniform
        const i4v (&cv)[KV][3] = cva[vi];
        const int y = gv * 16 + fr;
        const int bv = bva[vi];
        const int q0 = (sva[vi] >> 4) + fq;
        i4v x[4][KV];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ks = 0; ks < KV; ++ks)
                x[j][ks] = *reinterpret_cast<const i4v *>(T + (row_in_tile + 4 * j) * LF_TP + (((q0 + ks * 4) ^ (4 * ca)) << 4));
        uint32_t word[4];
        band_digits<KV, 4>(x, cv, bv, word);
        if (y < h) {
            const int cb = slab * 64 + fq * 16;
            uint8_t *o = out + (size_t)y * C + cb;
            typedef uint32_t u4a __attribute__((ext_vector_type(4), aligned(4)));
            if (cb + 16 <= C) {
                *reinterpret_cast<u4a *>(o) = u4a{word[0], word[1], word[2], word[3]};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cb + 4 * j + 4 <= C) *reinterpret_cast<uint32_t *>(o + 4 * j) = word[j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void copy_rgb_k(const uint8_t *__restrict__ src, int n_px, int src_c, int swap_rb,
                                                  uint8_t *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px) return;
    const uint8_t *s = src + ((size_t)blockIdx.y * n_px + i) * src_c;
    uint8_t *d = dst + ((size_t)blockIdx.y * n_px + i) * 3;
    d[0] = s[swap_rb ? 2 : 0]; d[1] = s[1]; d[2] = s[swap_rb ? 0 : 2];
}

// The same for packed 3-byte pixels with the red / blue swap, four pixels (three dwords) per thread: the byte kernel above issues six
// single-byte memory instructions per pixel and moved a 640x640 batch at 2.1 TB/s (config 3: 298 us per 256 frames, 2.5 % of a step).
__global__ __launch_bounds__(256) void copy_swap_rb4_k(const uint32_t *__restrict__ src, int n_quads, uint32_t *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_quads) return;
    const size_t o = ((size_t)blockIdx.y * n_quads + i) * 3;
    const uint32_t w0 = src[o], w1 = src[o + 1], w2 = src[o + 2];      // p0c0 p0c1 p0c2 p1c0 | p1c1 p1c2 p2c0 p2c1 | p2c2 p3c0 p3c1 p3c2
    const uint32_t t = __builtin_amdgcn_perm(w1, w0, 0x070c0304u);     // p1c1 p1c0 . p2c1
    dst[o] = __builtin_amdgcn_perm(w1, w0, 0x05000102u);               // p0c2 p0c1 p0c0 p1c2
    dst[o + 1] = __builtin_amdgcn_perm(w2, t, 0x03040100u);            // p1c1 p1c0 p2c2 p2c1
    dst[o + 2] = __builtin_amdgcn_perm(w2, w1, 0x05060702u);           // p2c0 p3c2 p3c1 p3c0
}

// tools/generate_detections.py:86-116 -- the reference's model-free test encoders.
// mode 0 (DummyImageEncoder): patches u8 [n][16][8][3] -> mean over the 3 channels -> 128 values - 128
// -> L2-normalised (e0 when the norm is 0).  mode 1 (ConstantImageEncoder): e0.  One wave per patch.
__global__ __launch_bounds__(256) void fake_encode_k(const uint8_t *__restrict__ patches, int n, int mode,
                                                     float *__restrict__ out) {
    const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= n) return;
    float v[2] = {0.f, 0.f};
    float ss = 0.f;
    if (mode == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint8_t *p = patches + ((size_t)w * 128 + lane + 64 * q) * 3;
            // np.average over the channel axis of a float32 array: sum in f32, divide by 3
            v[q] = ((float)p[0] + (float)p[1] + (float)p[2]) / 3.0f - 128.0f;
            ss += v[q] * v[q];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float l = sqrtf(ss);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        float r;
        if (mode == 1 || l == 0.f) r = (mode == 1) ? (j == 0 ? 1.f : 0.f) : (j == 0 ? 1.f : v[q]);
        else r = v[q] / l;
        out[(size_t)w * 128 + j] = r;
    }
}

// Host: Pillow precompute_coeffs + normalize_coeffs_8bpc (double math on the host so that sin()
// is the same libm the reference's Pillow uses; the device only sees integers).
struct LanczosTable {
    int ksize = 0;
    std::vector<int> bounds, kk;
};

double sinc_(double x) { if (x == 0.0) return 1.0; x *= M_PI; return sin(x) / x; }
double lanczos_(double x) { return (-3.0 <= x && x < 3.0) ? sinc_(x) * sinc_(x / 3) : 0.0; }

LanczosTable make_table(int in_size, int out_size) {
    LanczosTable t;
    double scale = (double)in_size / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 3.0 * filterscale;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.bounds.assign((size_t)out_size * 2, 0);
    t.kk.assign((size_t)out_size * t.ksize, 0);
    const double ss = 1.0 / filterscale;
    std::vector<double> w(t.ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = lanczos_((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            t.kk[(size_t)xx * t.ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS))
                                                   : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        t.bounds[2 * xx] = xmin;
        t.bounds[2 * xx + 1] = xmax;
    }
    return t;
}

struct DevTable { int ksize; int *bounds; int *kk; };
std::mutex g_tab_mu;
std::map<std::tuple<int, int, int>, DevTable> g_tabs;     // (device, in, out) -> device-resident table

int get_table(int device, int in_size, int out_size, DevTable *out) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto key = std::make_tuple(device, in_size, out_size);
    auto it = g_tabs.find(key);
    if (it == g_tabs.end()) {
        LanczosTable t = make_table(in_size, out_size);
        DevTable d;
        d.ksize = t.ksize;
        DD_HIP(hipMalloc(&d.bounds, t.bounds.size() * sizeof(int)));
        DD_HIP(hipMalloc(&d.kk, t.kk.size() * sizeof(int)));
        DD_HIP(hipMemcpy(d.bounds, t.bounds.data(), t.bounds.size() * sizeof(int), hipMemcpyHostToDevice));
        DD_HIP(hipMemcpy(d.kk, t.kk.data(), t.kk.size() * sizeof(int), hipMemcpyHostToDevice));
        it = g_tabs.emplace(key, d).first;
    }
    *out = it->second;
    return DD_OK;
}

// Banded-product tables for band_resample_k.  `taps(col)` yields (source byte index, coefficient)
// pairs of output column `col`; pitch = bytes per source row (the window of a column group must fit).
struct BandTable { int n_groups = 0, ksteps = 0, n_cols = 0; int *start = nullptr; void *coef = nullptr; int *bias = nullptr; bool ok = false;
                   int *slab_u0 = nullptr; int n_slabs = 0, slab_width = 0; };     // lanczos_fused_k: first window byte of every 4 groups, widest union
std::map<std::tuple<int, int, int, int, int, int, int>, BandTable> g_bands;     // (device, in, out, mode, src_c, swap, pitch)

template <class TapFn>
bool build_band(int n_cols, int pitch, TapFn taps, std::vector<int> &start, std::vector<int8_t> &coef, std::vector<int> &bias, int &ksteps) {
    const int n_groups = (n_cols + 15) / 16;
    std::vector<std::vector<std::pair<int, int>>> col_taps(n_cols);
    for (int c = 0; c < n_cols; ++c) taps(c, col_taps[c]);
    start.assign(n_groups, 0);
    ksteps = 1;
    std::vector<int> kmax(n_groups, 0);
    for (int g = 0; g < n_groups; ++g) {
        int lo = INT32_MAX, hi = -1;
        for (int c = g * 16; c < std::min(n_cols, g * 16 + 16); ++c)
            for (auto &t : col_taps[c]) { lo = std::min(lo, t.first); hi = std::max(hi, t.first); }
        if (hi < 0) { lo = hi = 0; }
        start[g] = lo & ~15;
        kmax[g] = hi;
        ksteps = std::max(ksteps, (hi + 1 - start[g] + 63) / 64);
    }
    if (ksteps > 4 || ksteps * 64 > pitch) return false;
    for (int g = 0; g < n_groups; ++g) {
        if (start[g] + ksteps * 64 > pitch) start[g] = (pitch - ksteps * 64) & ~15;      // keep the window inside the row
        if (start[g] < 0 || kmax[g] >= start[g] + ksteps * 64) return false;
    }
    coef.assign((size_t)n_groups * ksteps * 3 * 64 * 16, 0);
    bias.assign((size_t)n_groups * 16, 0);
    for (int c = 0; c < n_cols; ++c) {
        const int g = c / 16, j = c % 16;
        long long sum = 0;
        for (auto &t : col_taps[c]) {
            const int k = t.second;
            sum += k;
            const int d0 = ((k + 128) & 255) - 128, r1 = (k - d0) >> 8;
            const int d1 = ((r1 + 128) & 255) - 128, d2 = (r1 - d1) >> 8;
            if (d2 < -128 || d2 > 127) return false;
            const int off = t.first - start[g], ks = off / 64, kg = (off % 64) / 16, i = off % 16;
            const int dig[3] = {d0, d1, d2};
            for (int p = 0; p < 3; ++p) {
                int8_t &slot = coef[((((size_t)g * ksteps + ks) * 3 + p) * 64 + (kg * 16 + j)) * 16 + i];
                slot = (int8_t)(slot + dig[p]);                    // one tap per source byte and column: plain store
            }
        }
        bias[c] = (int)((1LL << (PRECISION_BITS - 1)) + 128 * sum);
    }
    return true;
}

// mode 0: horizontal (in = W, out = w, columns (x, c), source bytes x*src_c + channel); mode 1: vertical (in = H, out = h).
int get_band(int device, int in_size, int out_size, int mode, int src_c, int swap_rb, int pitch, BandTable *out) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto key = std::make_tuple(device, in_size, out_size, mode, src_c, swap_rb, pitch);
    auto it = g_bands.find(key);
    if (it == g_bands.end()) {
        const LanczosTable t = make_table(in_size, out_size);
        BandTable b;
        std::vector<int> start, bias;
        std::vector<int8_t> coef;
        int ksteps = 0;
        const int n_cols = mode == 0 ? out_size * 3 : out_size;
        auto taps = [&](int c, std::vector<std::pair<int, int>> &v) {
            const int o = mode == 0 ? c / 3 : c, ch = mode == 0 ? c % 3 : 0;
            const int lo = t.bounds[2 * o], n = t.bounds[2 * o + 1];
            for (int x = 0; x < n; ++x) {
                const int k = t.kk[(size_t)o * t.ksize + x];
                if (k == 0) continue;
                v.emplace_back(mode == 0 ? (lo + x) * src_c + (swap_rb ? 2 - ch : ch) : lo + x, k);
            }
        };
        b.ok = build_band(n_cols, pitch, taps, start, coef, bias, ksteps);
        if (b.ok) {
            b.n_groups = (int)start.size(); b.ksteps = ksteps; b.n_cols = n_cols;
            DD_HIP(hipMalloc(&b.start, start.size() * sizeof(int)));
            DD_HIP(hipMalloc(&b.coef, coef.size()));
            DD_HIP(hipMalloc(&b.bias, bias.size() * sizeof(int)));
            DD_HIP(hipMemcpy(b.start, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
            DD_HIP(hipMemcpy(b.coef, coef.data(), coef.size(), hipMemcpyHostToDevice));
            DD_HIP(hipMemcpy(b.bias, bias.data(), bias.size() * sizeof(int), hipMemcpyHostToDevice));
            b.n_slabs = (b.n_groups + 3) / 4;
            std::vector<int> u0(b.n_slabs);
            for (int sl = 0; sl < b.n_slabs; ++sl) {
                int lo = INT32_MAX, hi = 0;
                for (int g = sl * 4; g < std::min(b.n_groups, sl * 4 + 4); ++g) { lo = std::min(lo, start[g]); hi = std::max(hi, start[g] + ksteps * 64); }
                u0[sl] = lo;
                b.slab_width = std::max(b.slab_width, hi - lo);
            }
            DD_HIP(hipMalloc(&b.slab_u0, u0.size() * sizeof(int)));
            DD_HIP(hipMemcpy(b.slab_u0, u0.data(), u0.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        it = g_bands.emplace(key, b).first;
    }
    *out = it->second;
    return DD_OK;
}

int launch_band(hipStream_t s, const BandTable &b, const uint8_t *src, size_t src_img_stride, int R, int pitch_s, uint8_t *outT,
                size_t out_img_stride, int pitch_o, int batch) {
    const i4v *cf = static_cast<const i4v *>(b.coef);
    static const bool no_wide = getenv("DD_LANCZOS_NO_WIDE") != nullptr;
    if (!no_wide && R % 4 == 0 && pitch_o % 4 == 0 && (reinterpret_cast<uintptr_t>(outT) & 3) == 0 && out_img_stride % 4 == 0) {
        const int tiles = dd_ceil_div(R, 64);                     // 64-row iterations, 16-byte stores
        const int chunks = std::max(1, std::min(tiles, dd_ceil_div(8192, std::max(1, b.n_groups * batch))));
        const int tpc = dd_ceil_div(tiles, chunks);
        const int nx = dd_ceil_div(b.n_groups, 4), ny = dd_ceil_div(tiles, tpc);
        const dim3 grid((unsigned)(nx * ny * batch));
#define DD_BANDW(KS_) hipLaunchKernelGGL(band_resample_wide_k<KS_>, grid, dim3(256), 0, s, src, src_img_stride, R, pitch_s, b.start, cf, \
                                         b.bias, b.n_groups, b.n_cols, outT, out_img_stride, pitch_o, tpc, nx, ny)
        switch (b.ksteps) {
            case 1: DD_BANDW(1); break;
            case 2: DD_BANDW(2); break;
            case 3: DD_BANDW(3); break;
            default: DD_BANDW(4); break;
        }
#undef DD_BANDW
        DD_LAUNCH_CHECK();
        return DD_OK;
    }
    const int tiles = dd_ceil_div(R, 16);
    const int chunks = std::max(1, std::min(tiles, dd_ceil_div(8192, std::max(1, b.n_groups * batch))));
    const int tpc = dd_ceil_div(tiles, chunks);
    const int nx = dd_ceil_div(b.n_groups, 4), ny = dd_ceil_div(tiles, tpc);
    const dim3 grid((unsigned)(nx * ny * batch));
#define DD_BAND(KS_) hipLaunchKernelGGL(band_resample_k<KS_>, grid, dim3(256), 0, s, src, src_img_stride, R, pitch_s, b.start, cf, b.bias, \
                                        b.n_groups, b.n_cols, outT, out_img_stride, pitch_o, tpc, nx, ny)
    switch (b.ksteps) {
        case 1: DD_BAND(1); break;
        case 2: DD_BAND(2); break;
        case 3: DD_BAND(3); break;
        default: DD_BAND(4); break;
    }
#undef DD_BAND
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int launch_lanczos_fused(hipStream_t s, int device, const BandTable &bh, const BandTable &bv, const uint8_t *src, size_t src_img_stride,
                         int H, int pitch_s, uint8_t *dst, size_t dst_img_stride, int h, int batch) {
    static DevOnce once;
    const int rc = once.run(device, [&]() -> int {
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lanczos_fused_k<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lf_lds_bytes()));
        DD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lanczos_fused_k<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lf_lds_bytes()));
        return DD_OK;
    });
    if (rc != DD_OK) return rc;
    const dim3 grid((unsigned)(bh.n_slabs * batch));
#define DD_LF(KH_, KV_) hipLaunchKernelGGL((lanczos_fused_k<KH_, KV_>), grid, dim3(256), lf_lds_bytes(), s, src, src_img_stride, H, pitch_s, bh.start, \
                                           bh.slab_u0, static_cast<const i4v *>(bh.coef), bh.bias, bh.n_groups, bh.n_cols, bv.start,                 \
                                           static_cast<const i4v *>(bv.coef), bv.bias, bv.n_groups, h, dst, dst_img_stride, bh.n_slabs)
    if (bh.ksteps == 1) DD_LF(1, 1);
    else DD_LF(2, 1);
#undef DD_LF
    DD_LAUNCH_CHECK();
    return DD_OK;
}

}  // namespace

namespace ddk {

// tools/generate_detections.py:63-80 on the host: int64 tlwh -> clipped crop rectangle.
// Returns 1 when the reference would extract a patch, 0 when it returns None.
int crop_box_host(const int64_t *b, int ph, int pw, int H, int W, int *sx, int *sy, int *cw, int *ch) {
    int64_t x = b[0], y = b[1], w = b[2], h = b[3];
    const double aspect = (double)pw / ph;
    const double new_width = aspect * (double)h;
    x = (int64_t)((double)x - (new_width - (double)w) / 2);     // in-place int64 store truncates
    w = (int64_t)new_width;
    int64_t x2 = x + w, y2 = y + h;
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (x2 > W - 1) x2 = W - 1;
    if (y2 > H - 1) y2 = H - 1;
    if (x >= x2 || y >= y2) { *sx = *sy = 0; *cw = *ch = 0; return 0; }
    *sx = (int)x; *sy = (int)y; *cw = (int)(x2 - x); *ch = (int)(y2 - y);
    return 1;
}

// The same statements on a float box (generate_detections.py:64-74 when bbox is a float array, e.g. a CVAT annotation
// fed through framerecords.process_boxes): all arithmetic in f64, one truncation by astype(int) at the end.
int crop_box_host_f64(const double *b, int ph, int pw, int H, int W, int *sx, int *sy, int *cw, int *ch) {
    double x = b[0], y = b[1], w = b[2], h = b[3];
    const double new_width = (double)pw / ph * h;
    x -= (new_width - w) / 2;
    w = new_width;
    const double xe = x + w, ye = y + h;
    if (!(fabs(x) < 9e15 && fabs(y) < 9e15 && fabs(xe) < 9e15 && fabs(ye) < 9e15)) { *sx = *sy = 0; *cw = *ch = 0; return 0; }
    int64_t x1 = (int64_t)x, y1 = (int64_t)y, x2 = (int64_t)xe, y2 = (int64_t)ye;
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    if (x2 > W - 1) x2 = W - 1;
    if (y2 > H - 1) y2 = H - 1;
    if (x1 >= x2 || y1 >= y2) { *sx = *sy = 0; *cw = *ch = 0; return 0; }
    *sx = (int)x1; *sy = (int)y1; *cw = (int)(x2 - x1); *ch = (int)(y2 - y1);
    return 1;
}

int crop_resize(hipStream_t s, const uint8_t *frames, int H, int W, const void *d_boxes, int n, int oh, int ow,
                uint8_t *out) {
    if (n <= 0) return DD_OK;
    static const bool quad_off = getenv("DD_CROP_QUADS") && atoi(getenv("DD_CROP_QUADS")) == 0;
    if (!quad_off && ow % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0)
        hipLaunchKernelGGL(crop_resize4_k, dim3(dd_ceil_div(oh * (ow / 4), 256), n), dim3(256), 0, s, frames, H, W,
                           static_cast<const CropBox *>(d_boxes), oh, ow, out);
    else
    hipLaunchKernelGGL(crop_resize_k, dim3(dd_ceil_div(oh * ow, 256), n), dim3(256), 0, s, frames, H, W,
                       static_cast<const CropBox *>(d_boxes), oh, ow, out);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

// `batch` images of identical geometry, densely packed; tmp must hold batch*H*w*3 bytes.
int resize_lanczos(hipStream_t s, int device, const uint8_t *src, int H, int W, int src_c, int swap_rb,
                   uint8_t *dst, int h, int w, uint8_t *tmp, int batch) {
    // ---- matrix-core path: both passes as banded i8 products through a transposed intermediate [w*3][H]
    static const int dbg_band = getenv("DD_LANCZOS_DEBUG") ? atoi(getenv("DD_LANCZOS_DEBUG")) : 0;
    if (!(dbg_band & 4) && w != W && h != H && (src_c == 3 || src_c == 4) && (W * src_c) % 16 == 0 && H % 16 == 0 &&
        (w * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(tmp) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(dst) & 3) == 0 && ((size_t)H * W * src_c) % 16 == 0) {
        BandTable bh, bv;
        int rc = get_band(device, W, w, 0, src_c, swap_rb, W * src_c, &bh);
        if (rc != DD_OK) return rc;
        rc = get_band(device, H, h, 1, 1, 0, H, &bv);
        if (rc != DD_OK) return rc;
        static const bool fused_off = getenv("DD_LANCZOS_FUSED") && atoi(getenv("DD_LANCZOS_FUSED")) == 0;
        if (bh.ok && bv.ok && !fused_off && bh.ksteps <= 2 && bv.ksteps == 1 && bv.n_groups <= 4 * LF_NVG && H <= 512 && bh.slab_width <= 256 && W * src_c >= 256)
            return launch_lanczos_fused(s, device, bh, bv, src, (size_t)H * W * src_c, H, W * src_c, dst, (size_t)h * w * 3, h, batch);
        if (bh.ok && bv.ok) {
            rc = launch_band(s, bh, src, (size_t)H * W * src_c, H, W * src_c, tmp, (size_t)w * 3 * H, H, batch);
            if (rc != DD_OK) return rc;
            return launch_band(s, bv, tmp, (size_t)w * 3 * H, w * 3, H, dst, (size_t)h * w * 3, w * 3, batch);
        }
    }
    const uint8_t *mid = src;
    int mid_c = src_c;
    if (w != W) {
        DevTable th;
        int rc = get_table(device, W, w, &th);
        if (rc != DD_OK) return rc;
        uint8_t *o = (h != H) ? tmp : dst;
        static const int dbg = getenv("DD_LANCZOS_DEBUG") ? atoi(getenv("DD_LANCZOS_DEBUG")) : 0;
        if (!(dbg & 1) && (W * src_c) % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && W * src_c <= 6144 &&
            th.ksize <= HTAPS)
            hipLaunchKernelGGL(lanczos_h_row_k, dim3(dd_ceil_div(H, HROWS), batch), dim3(320), (size_t)HROWS * W * src_c, s, src,
                               H, W, src_c, swap_rb, th.bounds, th.kk, th.ksize, w, o);
        else
            hipLaunchKernelGGL(lanczos_h_k, dim3(dd_ceil_div(H * w, 256), batch), dim3(256), 0, s, src, H, W, src_c, swap_rb,
                               th.bounds, th.kk, th.ksize, w, o);
        DD_LAUNCH_CHECK();
        mid = o;
        mid_c = 3;
    } else if (src_c != 3 || swap_rb) {
        uint8_t *o = (h != H) ? tmp : dst;
        if (src_c == 3 && swap_rb && (H * W) % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(o)) & 3) == 0)
            hipLaunchKernelGGL(copy_swap_rb4_k, dim3(dd_ceil_div(H * W / 4, 256), batch), dim3(256), 0, s, reinterpret_cast<const uint32_t *>(src),
                               H * W / 4, reinterpret_cast<uint32_t *>(o));
        else
            hipLaunchKernelGGL(copy_rgb_k, dim3(dd_ceil_div(H * W, 256), batch), dim3(256), 0, s, src, H * W, src_c, swap_rb, o);
        DD_LAUNCH_CHECK();
        mid = o;
        mid_c = 3;
    }
    if (h != H) {
        DevTable tv;
        int rc = get_table(device, H, h, &tv);
        if (rc != DD_OK) return rc;
        static const int dbg2 = getenv("DD_LANCZOS_DEBUG") ? atoi(getenv("DD_LANCZOS_DEBUG")) : 0;
        if (!(dbg2 & 2) && (w * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(mid) & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0)
            hipLaunchKernelGGL(lanczos_v4_k, dim3(dd_ceil_div(h * (w * 3 / 4), 256), batch), dim3(256), 0, s, mid, H, w * 3,
                               tv.bounds, tv.kk, tv.ksize, h, dst);
        else
            hipLaunchKernelGGL(lanczos_v_k, dim3(dd_ceil_div(h * w * 3, 256), batch), dim3(256), 0, s, mid, H, w * 3, tv.bounds,
                               tv.kk, tv.ksize, h, dst);
        DD_LAUNCH_CHECK();
    } else if (mid != dst) {
        DD_HIP(hipMemcpyAsync(dst, mid, (size_t)batch * H * w * mid_c, hipMemcpyDeviceToDevice, s));
    }
    return DD_OK;
}

}  // namespace ddk

extern "C" {

int dd_crop_resize(dd_ctx *ctx, const uint8_t *frame, int H, int W, const int64_t *boxes_host, int n, int ph,
                   int pw, uint8_t *out, int *valid_host, void *stream) {
    DD_REQUIRE(ctx && n >= 0 && H > 0 && W > 0 && ph > 0 && pw > 0, DD_E_ARG, "dd_crop_resize: bad argument");
    DD_DEVICE(ctx);
    if (n == 0) return DD_OK;
    DD_REQUIRE(frame && boxes_host && out, DD_E_ARG, "dd_crop_resize: NULL argument");
    hipStream_t s = dd_pick_stream(ctx, stream);
    int rc;
    if ((rc = ctx->pin[1].reserve((size_t)n * 32)) != DD_OK) return rc;
    if ((rc = ctx->scratch[3].reserve((size_t)n * 32)) != DD_OK) return rc;
    int *hb = ctx->pin[1].as<int>();
    for (int i = 0; i < n; ++i) {
        const int ok = ddk::crop_box_host(boxes_host + (size_t)i * 4, ph, pw, H, W, hb + 8 * i, hb + 8 * i + 1,
                                          hb + 8 * i + 2, hb + 8 * i + 3);
        hb[8 * i + 4] = 0; hb[8 * i + 5] = 0; hb[8 * i + 6] = 0; hb[8 * i + 7] = 0;
        if (valid_host) valid_host[i] = ok;
    }
    DD_HIP(hipMemcpyAsync(ctx->scratch[3].p, hb, (size_t)n * 32, hipMemcpyHostToDevice, s));
    DD_HIP(hipStreamSynchronize(s));                        // the pinned block is reused by the next call
    return ddk::crop_resize(s, frame, H, W, ctx->scratch[3].p, n, ph, pw, out);
}

int dd_crop_resize_f64(dd_ctx *ctx, const uint8_t *frame, int H, int W, const double *boxes_host, int n, int ph,
                       int pw, uint8_t *out, int *valid_host, void *stream) {
    DD_REQUIRE(ctx && n >= 0 && H > 0 && W > 0 && ph > 0 && pw > 0, DD_E_ARG, "dd_crop_resize_f64: bad argument");
    DD_DEVICE(ctx);
    if (n == 0) return DD_OK;
    DD_REQUIRE(frame && boxes_host && out, DD_E_ARG, "dd_crop_resize_f64: NULL argument");
    hipStream_t s = dd_pick_stream(ctx, stream);
    int rc;
    if ((rc = ctx->pin[1].reserve((size_t)n * 32)) != DD_OK) return rc;
    if ((rc = ctx->scratch[3].reserve((size_t)n * 32)) != DD_OK) return rc;
    int *hb = ctx->pin[1].as<int>();
    for (int i = 0; i < n; ++i) {
        const int ok = ddk::crop_box_host_f64(boxes_host + (size_t)i * 4, ph, pw, H, W, hb + 8 * i, hb + 8 * i + 1,
                                              hb + 8 * i + 2, hb + 8 * i + 3);
        hb[8 * i + 4] = 0; hb[8 * i + 5] = 0; hb[8 * i + 6] = 0; hb[8 * i + 7] = 0;
        if (valid_host) valid_host[i] = ok;
    }
    DD_HIP(hipMemcpyAsync(ctx->scratch[3].p, hb, (size_t)n * 32, hipMemcpyHostToDevice, s));
    DD_HIP(hipStreamSynchronize(s));                        // the pinned block is reused by the next call
    return ddk::crop_resize(s, frame, H, W, ctx->scratch[3].p, n, ph, pw, out);
}

int dd_fake_encode(dd_ctx *ctx, const uint8_t *patches, int n, int mode, float *out, void *stream) {
    DD_REQUIRE(ctx && n >= 0 && (mode == 0 || mode == 1), DD_E_ARG, "dd_fake_encode: bad argument");
    DD_DEVICE(ctx);
    if (n == 0) return DD_OK;
    DD_REQUIRE(out && (mode == 1 || patches), DD_E_ARG, "dd_fake_encode: NULL argument");
    hipLaunchKernelGGL(fake_encode_k, dim3(dd_ceil_div(n, 4)), dim3(256), 0, dd_pick_stream(ctx, stream), patches, n, mode, out);
    DD_LAUNCH_CHECK();
    return DD_OK;
}

int dd_resize_lanczos(dd_ctx *ctx, const uint8_t *src, int H, int W, int src_c, int swap_rb, uint8_t *dst, int h,
                      int w, void *stream) {
    DD_REQUIRE(ctx && src && dst && H > 0 && W > 0 && h > 0 && w > 0, DD_E_ARG, "dd_resize_lanczos: bad argument");
    DD_DEVICE(ctx);
    DD_REQUIRE(src_c == 3 || src_c == 4, DD_E_ARG, "dd_resize_lanczos: src_c must be 3 or 4");
    int rc;
    if ((rc = ctx->scratch[3].reserve((size_t)H * w * 3 + 64)) != DD_OK) return rc;
    return ddk::resize_lanczos(dd_pick_stream(ctx, stream), ctx->device, src, H, W, src_c, swap_rb, dst, h, w,
                               ctx->scratch[3].as<uint8_t>(), 1);
}

int dd_resize_lanczos_batch(dd_ctx *ctx, const uint8_t *src, int batch, int H, int W, int src_c, int swap_rb, uint8_t *dst,
                            int h, int w, void *stream) {
    DD_REQUIRE(ctx && src && dst && batch > 0 && H > 0 && W > 0 && h > 0 && w > 0, DD_E_ARG, "dd_resize_lanczos_batch: bad argument");
    DD_DEVICE(ctx);
    DD_REQUIRE(src_c == 3 || src_c == 4, DD_E_ARG, "dd_resize_lanczos_batch: src_c must be 3 or 4");
    int rc;
    if ((rc = ctx->scratch[3].reserve((size_t)batch * H * w * 3 + 64)) != DD_OK) return rc;
    return ddk::resize_lanczos(dd_pick_stream(ctx, stream), ctx->device, src, H, W, src_c, swap_rb, dst, h, w,
                               ctx->scratch[3].as<uint8_t>(), batch);
}

int dd_resize_bilinear(dd_ctx *ctx, const uint8_t *src, int H, int W, int c, uint8_t *dst, int h, int w,
                       void *stream) {
    DD_REQUIRE(ctx && src && dst && H > 0 && W > 0 && h > 0 && w > 0, DD_E_ARG, "dd_resize_bilinear: bad argument");
    DD_DEVICE(ctx);
    DD_REQUIRE(c == 3, DD_E_ARG, "dd_resize_bilinear: 3-channel images only");
    hipStream_t s = dd_pick_stream(ctx, stream);
    int rc;
    if ((rc = ctx->pin[1].reserve(32)) != DD_OK) return rc;
    if ((rc = ctx->scratch[3].reserve(32)) != DD_OK) return rc;
    int *hb = ctx->pin[1].as<int>();
    hb[0] = 0; hb[1] = 0; hb[2] = W; hb[3] = H; hb[4] = 0; hb[5] = 0; hb[6] = 0; hb[7] = 0;
    DD_HIP(hipMemcpyAsync(ctx->scratch[3].p, hb, 32, hipMemcpyHostToDevice, s));
    DD_HIP(hipStreamSynchronize(s));
    return ddk::crop_resize(s, src, H, W, ctx->scratch[3].p, 1, h, w, dst);
}

}  // extern "C"
