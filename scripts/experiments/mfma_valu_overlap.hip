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
// Specialised waves (the arrangement MI355X_MICROARCH.md's wave-scheduling section describes): per SIMD ONE wave that issues only the rounds' 80 MFMAs
// beside NV waves that issue only their 440 vector instructions (the NV waves share the rounds: each runs iters / NV of them).  One workgroup of
// 4 (1 + NV) waves per CU; wave w sits on SIMD w % 4 (checked through HW_ID), so waves 0..3 are the matrix waves.  VMODE 0: v_fma_f32 only,
// 1: 400 v_fma_f32 + 40 v_exp_f32 per round, 2: the integer mix of csrc/netsq.hip's requantisation (per 12: 4 v_mad_i64_i32, 2 v_cvt_pk_i16_i32,
// 2 v_pk_ashrrev_i16, 2 v_sat_pk_u8_i16, v_perm_b32, v_xor_b32), 3: nothing (the matrix waves alone in this launch shape).
// MMODE 0: v_mfma_f32_16x16x32_f16, 1: v_mfma_i32_16x16x64_i8, 2: no matrix waves' work (the vector waves alone in this launch shape).
typedef int i4 __attribute__((ext_vector_type(4)));
// PRIO 1: the vector waves raise their priority (s_setprio 2): the arbiter then takes a ready vector instruction before the matrix wave's next MFMA.
template <int NV, int VMODE, int MMODE, int PRIO = 0>
__global__ __launch_bounds__(256 * (1 + NV) > 1024 ? 1024 : 256 * (1 + NV)) void ks(unsigned long long *out, int iters, float seed, unsigned *simd_of) {
    const int wave = threadIdx.x >> 6;
    if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(2);
    if (PRIO == 2 && wave < 4) __builtin_amdgcn_s_setprio(2);
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) simd_of[wave] = (hw >> 4) & 3;      // HW_ID bits 5:4 = SIMD
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0;
    if (wave < 4) {
        if (MMODE == 0) {
            h8 a, b;
            for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f + threadIdx.x); }
            f4 acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = f4{seed, seed, seed, seed};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 10; ++r) { MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) }
            }
            for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        } else if (MMODE == 1) {
            i4 a = {(int)seed, 3, 5, 7}, b = {(int)threadIdx.x, 1, 2, 3}, acc[8];
            for (int i = 0; i < 8; ++i) acc[i] = i4{i, i, i, i};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 10; ++r) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                }
            }
            for (int i = 0; i < 8; ++i) s += (float)(acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]);
        }
    } else if (VMODE != 3) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = seed * i;
        const float c0 = 0.999f, c1 = seed * 1e-3f;
        int x[4] = {(int)threadIdx.x, 3, 5, 7};
        long long q[4] = {1, 2, 3, 4};
        const int M = 0x5a5a5a5a + (int)seed, sh = 0x00030003, sel = 0x05040100;
        const int my = iters / NV;
        for (int it = 0; it < my; ++it) {
            if (VMODE == 0) {
#pragma unroll
                for (int r = 0; r < 40; ++r) { V5(0) V5(5) VF(r & 7) }
            } else if (VMODE == 1) {
#pragma unroll
                for (int r = 0; r < 40; ++r) { V5(0) V5(5) VE(r & 7) }
            } else {
#pragma unroll
                for (int r = 0; r < 37; ++r) {                       // 37 x 12 = 444 instructions
                    unsigned p01, p23, q01, q23, o;
                    asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %6, %5, %1\n v_mad_i64_i32 %2, vcc, %7, %5, %2\n v_mad_i64_i32 %3, vcc, %8, %5, %3"
                                 : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : "v"(x[0]), "s"(M), "v"(x[1]), "v"(x[2]), "v"(x[3]) : "vcc");
                    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p01) : "v"((int)(q[0] >> 32)), "v"((int)(q[1] >> 32)));
                    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(p23) : "v"((int)(q[2] >> 32)), "v"((int)(q[3] >> 32)));
                    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(p01) : "s"(sh));
                    asm volatile("v_pk_ashrrev_i16 %0, %1, %0" : "+v"(p23) : "s"(sh));
                    asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(q01) : "v"(p01));
                    asm volatile("v_sat_pk_u8_i16 %0, %1" : "=v"(q23) : "v"(p23));
                    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(o) : "v"(q23), "v"(q01), "s"(sel));
                    asm volatile("v_xor_b32 %0, 0x80808080, %1" : "=v"(x[r & 3]) : "v"(o));
                }
            }
        }
        for (int i = 0; i < 8; ++i) s += v[i];
        s += (float)(x[0] + x[1] + x[2] + x[3]) + (float)(q[0] + q[1] + q[2] + q[3]);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (s == 0.12345f) out[0] = 1;
}
template <int NV, int VMODE, int MMODE, int PRIO = 0> void run_s(const char *name) {
    unsigned long long *d; hipMalloc(&d, 8 * 8192);
    unsigned *so; hipMalloc(&so, 64); hipMemset(so, 0xff, 64);
    const int iters = 1800;                                        // (divisible by 1, 2, 3)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    constexpr int NT = 256 * (1 + NV);
    hipLaunchKernelGGL((ks<NV, VMODE, MMODE, PRIO>), dim3(256), dim3(NT), 0, 0, d, 12, 1.5f, so);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((ks<NV, VMODE, MMODE, PRIO>), dim3(256), dim3(NT), 0, 0, d, iters, 1.5f, so);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned simd[16]; hipMemcpy(simd, so, sizeof(simd), hipMemcpyDeviceToHost);
    char place[64]; int n = 0;
    for (int w = 0; w < 4 * (1 + NV); ++w) n += snprintf(place + n, sizeof(place) - n, "%u", simd[w]);
    printf("%-44s 1 matrix + %d vector waves per SIMD: wall %.3f ms = %.0f ns per round per SIMD (%.0f cycles at 2.4 GHz); memtime ticks of matrix wave 0: %.0f per round, of vector wave 4: %.0f per round of its own; SIMD of waves: %s\n",
           name, NV, ms, ms * 1e6 / (double)iters, ms * 1e6 / (double)iters * 2.4, (double)h[0] / iters, (double)h[4] / (iters / NV), place);
    hipFree(d); hipFree(so);
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
    printf("-- specialised waves: per SIMD one matrix-only wave beside NV vector-only waves; a round = 80 MFMA + 440 (444) vector instructions\n");
    run_s<1, 3, 0>("f16 MFMA waves alone");
    run_s<1, 0, 2>("v_fma waves alone"); run_s<2, 0, 2>("v_fma waves alone"); run_s<3, 0, 2>("v_fma waves alone");
    run_s<1, 0, 0>("f16 MFMA wave || v_fma waves"); run_s<2, 0, 0>("f16 MFMA wave || v_fma waves"); run_s<3, 0, 0>("f16 MFMA wave || v_fma waves");
    run_s<1, 1, 2>("v_fma + v_exp waves alone"); run_s<2, 1, 2>("v_fma + v_exp waves alone"); run_s<3, 1, 2>("v_fma + v_exp waves alone");
    run_s<1, 1, 0>("f16 MFMA wave || v_fma + v_exp waves"); run_s<2, 1, 0>("f16 MFMA wave || v_fma + v_exp waves"); run_s<3, 1, 0>("f16 MFMA wave || v_fma + v_exp waves");
    run_s<1, 3, 1>("i8 MFMA waves alone");
    run_s<1, 2, 2>("requantisation-mix waves alone"); run_s<2, 2, 2>("requantisation-mix waves alone"); run_s<3, 2, 2>("requantisation-mix waves alone");
    run_s<1, 2, 1>("i8 MFMA wave || requantisation-mix waves"); run_s<2, 2, 1>("i8 MFMA wave || requantisation-mix waves"); run_s<3, 2, 1>("i8 MFMA wave || requantisation-mix waves");
    printf("-- the same with the vector waves at s_setprio 2 (then: the matrix waves at s_setprio 2)\n");
    run_s<1, 0, 0, 1>("f16 MFMA wave || v_fma waves, prio"); run_s<2, 0, 0, 1>("f16 MFMA wave || v_fma waves, prio"); run_s<3, 0, 0, 1>("f16 MFMA wave || v_fma waves, prio");
    run_s<1, 2, 1, 1>("i8 MFMA wave || requantisation mix, prio"); run_s<2, 2, 1, 1>("i8 MFMA wave || requantisation mix, prio"); run_s<3, 2, 1, 1>("i8 MFMA wave || requantisation mix, prio");
    run_s<1, 0, 0, 2>("f16 MFMA wave (prio) || v_fma waves"); run_s<1, 2, 1, 2>("i8 MFMA wave (prio) || requantisation mix");
    return 0;
}
