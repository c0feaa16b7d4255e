// Issue cost of the integer instructions of csrc/netsq.hip: cycles per wave-instruction with 1, 2 or 4 waves on a SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void k(unsigned long long *out, int iters, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1, c = 77;
    long long q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 1) { REP64(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 2) { REP64(asm volatile("v_dot2c_i32_i16 %0, %4, %5\n v_dot2c_i32_i16 %1, %4, %5\n v_dot2c_i32_i16 %2, %4, %5\n v_dot2c_i32_i16 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 3) { REP64(asm volatile("v_mad_i64_i32 %0, vcc, %4, %5, %0\n v_mad_i64_i32 %1, vcc, %4, %5, %1\n v_mad_i64_i32 %2, vcc, %4, %5, %2\n v_mad_i64_i32 %3, vcc, %4, %5, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(b), "v"(c) : "vcc");) }
        if (OP == 4) { REP64(asm volatile("v_mul_hi_i32 %0, %0, %4\n v_mul_hi_i32 %1, %1, %4\n v_mul_hi_i32 %2, %2, %4\n v_mul_hi_i32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 5) { REP64(asm volatile("v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 6) { REP64(asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 7) { REP64(asm volatile("v_mad_i32_i24 %0, %4, %5, %0\n v_mad_i32_i24 %1, %4, %5, %1\n v_mad_i32_i24 %2, %4, %5, %2\n v_mad_i32_i24 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 8) { REP64(asm volatile("v_pk_mad_i16 %0, %4, %5, %0\n v_pk_mad_i16 %1, %4, %5, %1\n v_pk_mad_i16 %2, %4, %5, %2\n v_pk_mad_i16 %3, %4, %5, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 9) { REP64(asm volatile("v_lshl_or_b32 %0, %0, 8, %4\n v_lshl_or_b32 %1, %1, 8, %4\n v_lshl_or_b32 %2, %2, 8, %4\n v_lshl_or_b32 %3, %3, 8, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 10) { REP64(asm volatile("v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte1 %2, %4\n v_cvt_f32_ubyte1 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 11) { REP64(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + (int)q0 + (int)q1 + (int)q2 + (int)q3 == 0x12345) out[0] = 1;
}
template <int OP> void run(const char *name) {
    unsigned long long *d; hipMalloc(&d, 8 * 8192);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves : {4, 8, 16, 32}) {                   // per CU: 1, 2, 4, 8 waves per SIMD (32 = blocks of 8 waves, four per CU)
        const int blocks = waves == 32 ? 1024 : 256, bw = waves == 32 ? 8 : waves;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(bw * 64), 0, 0, d, 10, 3);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(bw * 64), 0, 0, d, iters, 3);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        const double per = (double)h[0] / (iters * 64.0 * 4.0);
        const double instr_per_simd = (double)blocks * bw / 1024.0 * iters * 256.0;      // wave-instructions each SIMD executed
        printf("%-18s %2d waves/CU: %.2f memtime ticks per instruction per wave; wall %.3f ms = %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n",
               name, waves, per, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    hipFree(d);
}
int main() {
    run<0>("v_add_u32"); run<1>("v_perm_b32"); run<2>("v_dot2c_i32_i16"); run<3>("v_mad_i64_i32"); run<4>("v_mul_hi_i32");
    run<11>("v_pk_fma_f32");
    return 0;
}
