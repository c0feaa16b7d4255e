// What an i8 MFMA costs a SIMD by shape and by the number of waves issuing them: v_mfma_i32_16x16x64_i8 (16 K MACs, nominally 16 cycles of a SIMD
// at the 5 POP/s dense peak) against v_mfma_i32_32x32x32_i8 (32 K MACs, nominally 32), 1 / 2 / 4 waves per SIMD, NACC independent accumulators per
// wave in a round-robin chain (every kernel of csrc/netsq*.hip issues its MFMAs like that: 4-5 accumulators, each revisited every 4th-5th issue).
// Cycles = the launch's wall time (HIP events) x 2.4 GHz; profiles/r06_mfma_i8_shapes.txt.
//   hipcc --offload-arch=gfx950 -O3 scripts/experiments/mfma_i8_shapes.hip -o /tmp/mis && /tmp/mis
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(unsigned long long *out, int iters, int seed) {
    i4v a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, (int)threadIdx.x, seed - 7, seed + 11};
    i4v acc4[8];
    i16v acc16[4];
    for (int i = 0; i < 8; ++i) acc4[i] = i4v{seed, i, seed, i};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = seed + i + j;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 0) {
#pragma unroll
            for (int r = 0; r < 64; ++r) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc4[r % NACC]) : "v"(a), "v"(b));
        } else {
#pragma unroll
            for (int r = 0; r < 64; ++r) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc16[r % NACC]) : "v"(a), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 8; ++i) s += acc4[i][0] + acc4[i][3];
    for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][15];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
    if (s == 0x12345) out[0] = 1;
}


template <int SHAPE, int NACC, int WPS>
void run(const char *name) {
    unsigned long long *d;
    const int blocks = 256, waves = 4 * WPS, iters = 2000;
    (void)hipMalloc(&d, blocks * 16 * 8);
    (void)hipMemset(d, 0, blocks * 16 * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NACC, WPS>), dim3(blocks), dim3(64 * waves), 0, 0, d, 10, 3);      // warm
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NACC, WPS>), dim3(blocks), dim3(64 * waves), 0, 0, d, iters, 3);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), d, blocks * 16 * 8, hipMemcpyDeviceToHost);
    double ticks = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) ticks += (double)h[b * 16 + w];
    ticks /= blocks * waves;
    const double n = 64.0 * iters;                                    // MFMAs per wave
    const double macs = (SHAPE == 0 ? 16384.0 : 32768.0) * n * waves * blocks;
    // wall time of the launch -> cycles of a SIMD at 2.4 GHz per MFMA of ONE wave stream, and per MFMA issued on the SIMD (WPS streams share it)
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s NACC %d  waves/SIMD %d: %7.1f cycles per MFMA and wave (2.4 GHz wall), %6.1f per MFMA on the SIMD, %6.2f POP/s, s_memtime ticks per MFMA and wave %.2f\n",
           name, NACC, WPS, cyc / n, cyc / n / WPS, 2.0 * macs / (ms * 1e-3) * 1e-15, ticks / n);
    (void)hipFree(d);
}

int main() {
    run<0, 4, 1>("16x16x64"); run<0, 4, 2>("16x16x64"); run<0, 4, 4>("16x16x64");
    run<0, 8, 1>("16x16x64"); run<0, 8, 2>("16x16x64");
    run<0, 2, 1>("16x16x64"); run<0, 2, 2>("16x16x64");
    run<1, 4, 1>("32x32x32"); run<1, 4, 2>("32x32x32"); run<1, 4, 4>("32x32x32");
    run<1, 2, 1>("32x32x32"); run<1, 2, 2>("32x32x32");
    run<1, 1, 1>("32x32x32"); run<1, 1, 2>("32x32x32");
    return 0;
}
