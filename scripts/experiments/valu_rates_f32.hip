// Issue cost of the f32 / conversion instructions the row-streaming kernels of csrc/nets.hip are made of, two waves per SIMD
// (their occupancy): cycles of a SIMD per wave-instruction, eight independent registers round robin.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/valu_rates_f32.hip -o /tmp/vr && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define O_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define O_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c0));
#define O_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
#define O_MAX(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
#define O_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define O_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define O_MAXI3(i) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define O_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#define O_EXPC(i) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(v[i]));
#define O_CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
#define O_CVTF(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i]));
#define O_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
#define O_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define O_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]) : "v"(c1));
#define O_MAXDPP(i) asm volatile("v_max_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]) : "v"(c1));
#define O_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c1) : "vcc");
#define O_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[i]) : "v"(cc));
#define O_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i]) : "v"(cc));
#define O_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i]) : "v"(cc));
#define O_PKMAXH(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
#define O_PKMULH(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
#define O_FMAMIX(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %2" : "+v"(v[i]) : "v"(c0), "v"(c1));
#define O_MFMA(i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256, 2) void k(unsigned long long *out, int iters, float seed) {
    float v[8];
    f2 w[8], cc = {0.999f, 1.001f};
    h8 a, b;
    f4 acc[8];
    for (int i = 0; i < 8; ++i) { v[i] = seed * i + threadIdx.x; w[i] = f2{seed + i, seed - i}; a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); acc[i] = f4{seed, seed, seed, seed}; }
    const float c0 = 0.999f, c1 = seed * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            if (OP == 0) { R8(O_FMA) } if (OP == 1) { R8(O_MUL) } if (OP == 2) { R8(O_ADD) } if (OP == 3) { R8(O_MAX) } if (OP == 4) { R8(O_MAX3) }
            if (OP == 5) { R8(O_MED3) } if (OP == 6) { R8(O_MAXI3) } if (OP == 7) { R8(O_EXP) } if (OP == 8) { R8(O_EXPC) } if (OP == 9) { R8(O_CVTPK) }
            if (OP == 10) { R8(O_CVTF) } if (OP == 11) { R8(O_AND) } if (OP == 12) { R8(O_PERM) } if (OP == 13) { R8(O_DPP) } if (OP == 14) { R8(O_MAXDPP) }
            if (OP == 15) { R8(O_CND) } if (OP == 16) { R8(O_PKMUL) } if (OP == 17) { R8(O_PKADD) } if (OP == 18) { R8(O_PKFMA) } if (OP == 19) { R8(O_PKMAXH) }
            if (OP == 20) { R8(O_PKMULH) } if (OP == 21) { R8(O_FMAMIX) } if (OP == 22) { R8(O_MFMA) }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + w[i][0] + w[i][1] + acc[i][0] + acc[i][3];
    if (s == 0.12345f) out[0] = 1;
}
template <int OP> void run(const char *name) {
    unsigned long long *d; (void)hipMalloc(&d, 4096);
    const int iters = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(512), dim3(256), 0, 0, d, 4, 1.5f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(512), dim3(256), 0, 0, d, iters, 1.5f);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 2.0 * iters * 256.0;                  // wave-instructions a SIMD executed (two waves)
    printf("%-22s %.2f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    (void)hipFree(d);
}
int main() {
    run<0>("v_fma_f32"); run<1>("v_mul_f32"); run<2>("v_add_f32"); run<3>("v_max_f32"); run<4>("v_max3_f32"); run<5>("v_med3_f32"); run<6>("v_maximum3_f32");
    run<7>("v_exp_f32"); run<8>("v_exp_f32 clamp"); run<9>("v_cvt_pk_f16_f32"); run<10>("v_cvt_f32_f16"); run<11>("v_and_b32"); run<12>("v_perm_b32");
    run<13>("v_mov_b32_dpp"); run<14>("v_max_f32_dpp"); run<15>("v_cndmask_b32"); run<16>("v_pk_mul_f32"); run<17>("v_pk_add_f32"); run<18>("v_pk_fma_f32");
    run<19>("v_pk_max_f16"); run<20>("v_pk_mul_f16"); run<21>("v_fma_mixlo_f16"); run<22>("v_mfma_f32_16x16x32_f16");
    return 0;
}
