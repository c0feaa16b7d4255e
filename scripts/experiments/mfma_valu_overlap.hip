// Can a SIMD run vector-ALU work under its matrix pipe?  Two waves per SIMD (as the row kernels of csrc/nets.hip run), each looping
// over a "round" of 80 v_mfma_f32_16x16x32_f16 (eight independent accumulators) and ~440 vector instructions (400 v_fma_f32 + 40
// v_exp_f32, the stem kernel's mix), in four arrangements: matrix only, vector only, one after the other, finely interleaved.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define MF(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#define VF(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define VE(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define VP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[i]) : "v"(cc));
#define VX(i) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define V5(i) VF(i) VF((i + 1) & 7) VF((i + 2) & 7) VF((i + 3) & 7) VF((i + 4) & 7)
// MODE 0: MFMA only; 1: VALU only; 2: 80 MFMA then 440 VALU; 3: per MFMA 5 fma (+ an exp with every second MFMA); 4: VALU only, no exps; 5: as 3 without exps
template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(unsigned long long *out, int iters, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
    f4 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f4{seed, seed, seed, seed}; v[i] = seed * i; }
    const float c0 = 0.999f, c1 = seed * 1e-3f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 w[8], cc = {0.999f, 1.001f};
    for (int i = 0; i < 8; ++i) w[i] = f2{seed + i, seed - i};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int r = 0; r < 10; ++r) { MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) }
        }
        if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
            for (int r = 0; r < 40; ++r) { V5(0) V5(5) if (MODE != 4) { VE(r & 7) } else { VF(r & 7) } }
        }
        if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < 55; ++r) { VP(0) VP(1) VP(2) VP(3) VP(4) VP(5) VP(6) VP(7) }
        }
        if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 55; ++r) { VX(0) VX(1) VX(2) VX(3) VX(4) VX(5) VX(6) VX(7) }
        }
        if (MODE == 3 || MODE == 5) {
#pragma unroll
            for (int r = 0; r < 40; ++r) { MF((2 * r) & 7) V5(0) MF((2 * r + 1) & 7) V5(5) if (MODE == 3) { VE(r & 7) } else { VF(r & 7) } }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += w[i][0] + w[i][1];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
    if (s == 0.12345f) out[0] = 1;
}
template <int MODE, int WPS> void run(const char *name) {
    unsigned long long *d; hipMalloc(&d, 8 * 8192);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * WPS;                                  // 256 CUs x WPS blocks of four waves = WPS waves per SIMD
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, d, 10, 1.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // a SIMD executed WPS * iters rounds
    printf("%-34s %d waves/SIMD: wall %.3f ms = %.0f ns per round per SIMD (%.0f cycles at 2.4 GHz); memtime ticks per round of wave 0: %.0f\n", name, WPS, ms,
           ms * 1e6 / (WPS * (double)iters), ms * 1e6 / (WPS * (double)iters) * 2.4, (double)h[0] / iters);
    hipFree(d);
}
int main() {
    run<0, 1>("80 MFMA"); run<0, 2>("80 MFMA");
    run<4, 1>("440 v_fma"); run<4, 2>("440 v_fma");
    run<1, 1>("400 v_fma + 40 v_exp"); run<1, 2>("400 v_fma + 40 v_exp");
    run<2, 1>("MFMA block, then VALU block"); run<2, 2>("MFMA block, then VALU block");
    run<3, 1>("interleaved (with exps)"); run<3, 2>("interleaved (with exps)");
    run<5, 1>("interleaved (fma only)"); run<5, 2>("interleaved (fma only)");
    run<6, 1>("440 v_pk_mul_f32"); run<6, 2>("440 v_pk_mul_f32");
    run<7, 1>("440 v_maximum3_f32"); run<7, 2>("440 v_maximum3_f32");
    return 0;
}
