// Is a depthwise 3x3 tap on the matrix pipe (B = diagonal weights, ONE nonzero product per output and MFMA) bit-identical to
// v_fma_mix_f32?  D[m][n] = C[m][n] + x[m][n] * w[n]: 16 pixels x 16 channels per MFMA, 9 MFMAs per 3x3 stencil.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float fmix(unsigned x, unsigned w, float acc, int hi) {
    if (hi) asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(w));
    else asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc) : "v"(x), "v"(w));
    return acc;
}
// x: [T taps][16 px][16 ch] f16, w: [T][16 ch] f16, bias [16] f32 -> out_valu / out_mfma [16 px][16 ch] f32, per block of 64 lanes; many blocks
__global__ void k(const _Float16 *x, const _Float16 *w, const float *bias, float *ov, float *om, int T, int reps) {
    const int lane = threadIdx.x, fr = lane & 15, fq = lane >> 4;
    const size_t blk = blockIdx.x;
    x += blk * (size_t)T * 256; w += blk * (size_t)T * 16; ov += blk * 256; om += blk * 256;
    // ---- VALU: lane = (pixel fr, channels fq*4 .. +3)
    float a[4];
    for (int i = 0; i < 4; ++i) a[i] = bias[fq * 4 + i];
    for (int r = 0; r < reps; ++r)
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 4; ++i) {
            const int c = fq * 4 + i;
            unsigned xv = 0, wv = 0;
            _Float16 xh = x[(t * 16 + fr) * 16 + c], wh = w[t * 16 + c];
            memcpy(&xv, &xh, 2); memcpy(&wv, &wh, 2);
            a[i] = fmix(xv, wv, a[i], 0);
        }
    for (int i = 0; i < 4; ++i) ov[fr * 16 + fq * 4 + i] = a[i];
    // ---- MFMA 16x16x32: A[m = px][k], B[k][n = ch]; k slots 0..15 = channels, 16..31 = zero.  Lane (fr, fq) holds A[fr][fq*8 .. +7]
    // and B[fq*8 .. +7][fr]; D rows 4 fq + i of column fr.  We want D[px][ch]: take A = x (m = px), B diagonal (n = ch).
    f4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = bias[fr];               // D[m = 4 fq + i][n = fr]: channel fr
    for (int r = 0; r < reps; ++r)
    for (int t = 0; t < T; ++t) {
        h8 av, bv;
        for (int j = 0; j < 8; ++j) {
            const int kk = fq * 8 + j;                          // k slot
            av[j] = kk < 16 ? x[(t * 16 + fr) * 16 + kk] : (_Float16)0.f;
            bv[j] = (kk < 16 && kk == fr) ? w[t * 16 + fr] : (_Float16)0.f;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) om[(fq * 4 + i) * 16 + fr] = acc[i];
}
int main(int argc, char **argv) {
    const int B = 4096, T = 9;
    std::vector<_Float16> hx((size_t)B * T * 256), hw((size_t)B * T * 16);
    std::vector<float> hb(16);
    srand(7);
    auto rnd = [&]() { return (float)rand() / RAND_MAX; };
    for (size_t i = 0; i < hx.size(); ++i) {
        const int m = rand() % 10;
        float v = m == 0 ? 0.f : m == 1 ? rnd() * 6e-5f : m == 2 ? rnd() * 1e-3f : rnd() * 6.f;      // zeros, f16 denormals, small, ReLU6 range
        hx[i] = (_Float16)v;
    }
    for (size_t i = 0; i < hw.size(); ++i) { const int m = rand() % 8; hw[i] = (_Float16)(m == 0 ? (rnd() - 0.5f) * 1e-4f : (rnd() - 0.5f) * 4.f); }
    for (int i = 0; i < 16; ++i) hb[i] = (rnd() - 0.5f) * 2.f;
    _Float16 *dx, *dw; float *db, *dov, *dom;
    hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&db, 64); hipMalloc(&dov, (size_t)B * 1024); hipMalloc(&dom, (size_t)B * 1024);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 64, hipMemcpyHostToDevice);
    k<<<B, 64>>>(dx, dw, db, dov, dom, T, 1);
    std::vector<float> ov((size_t)B * 256), om((size_t)B * 256);
    hipMemcpy(ov.data(), dov, ov.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(om.data(), dom, om.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0, signz = 0; double maxd = 0;
    for (size_t i = 0; i < ov.size(); ++i)
        if (memcmp(&ov[i], &om[i], 4)) { if (ov[i] == om[i]) ++signz; else { ++diff; double d = fabs((double)ov[i] - om[i]); if (d > maxd) maxd = d; } }
    printf("values %zu, bit-different %zu (max |d| %.3g), differ only in the sign of zero %zu; sample %g %g\n", ov.size(), diff, maxd, signz, ov[5], om[5]);
    return 0;
}
