// Bare fp32 MFMA issue-rate probe (gfx950): v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, NACC independent
// accumulators per wave, W waves per SIMD.  Prints TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_peak tools/micro/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned& s) {      // xorshift -> float in [-2, 2) with a full random mantissa
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return __uint_as_float(0x40000000u | (s & 0x007fffffu)) * ((s >> 31) ? 1.f : -1.f) - ((s >> 30) & 1 ? 0.f : 0.f);
}

// random operands, a different pair for every MFMA of the unrolled body (like GEMM fragments), NACC accumulators
template <int NACC>
__global__ __launch_bounds__(256) void probe_rand(float* out, unsigned long long* clk, int iters) {
    unsigned seed = 0x9e3779b9u * (blockIdx.x * 256 + threadIdx.x + 1);
    float av[16], bv[16];
    for (int i = 0; i < 16; ++i) { av[i] = rnd(seed) * 0.01f; bv[i] = rnd(seed) * 0.01f; }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + i) & 15], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run_rand(int blocks_per_cu, int iters) {
    const int grid = 256 * blocks_per_cu;
    float* out; unsigned long long* clk;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipMalloc(&clk, grid * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe_rand<NACC>), dim3(grid), dim3(256), 0, 0, out, clk, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe_rand<NACC>), dim3(grid), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), clk, grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double ghz = 0; for (int i = 0; i < grid; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; ghz /= grid;
    const double flops = (double)grid * 4 * iters * 16.0 * NACC * 4096.0;
    printf("RANDOM operands  nacc %d  waves/SIMD %d : %7.1f TFLOP/s  clock %.2f GHz  (%.3f ms)\n", NACC, blocks_per_cu,
           flops / ms / 1e9, ghz, ms);
    hipFree(out); hipFree(clk);
}

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* clk, int iters, float a0, float b0) {
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    f32x16 acc32[NACC];
    f32x4 acc16[NACC];
    for (int i = 0; i < NACC; ++i) {
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
        for (int r = 0; r < 4; ++r) acc16[i][r] = 0.f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 32) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[i], 0, 0, 0);
                else acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc16[i], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) {
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
        for (int r = 0; r < 4; ++r) s += acc16[i][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int NACC>
void run(int blocks_per_cu, int iters) {
    const int grid = 256 * blocks_per_cu;
    float* out; unsigned long long* clk;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipMalloc(&clk, grid * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.25f, 0.75f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<SHAPE, NACC>), dim3(grid), dim3(256), 0, 0, out, clk, iters, 1.25f, 0.75f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<unsigned long long> h(grid * 2);
    hipMemcpy(h.data(), clk, grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double ghz = 0; for (int i = 0; i < grid; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; ghz /= grid;
    const double flop_per_mfma = SHAPE == 32 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
    const double flops = (double)grid * 4 /*waves*/ * iters * 8.0 * NACC * flop_per_mfma;
    const double cyc_per_mfma = (double)h[0] / ((double)iters * 8 * NACC) / blocks_per_cu;   // per SIMD issue slot
    printf("shape %2d  nacc %d  waves/SIMD %d : %7.1f TFLOP/s  clock %.2f GHz  %.1f cyc/MFMA/SIMD  (%.3f ms)\n", SHAPE, NACC,
           blocks_per_cu, flops / ms / 1e9, ghz, cyc_per_mfma, ms);
    hipFree(out); hipFree(clk);
}

int main() {
    const int iters = 4000;
    run<32, 1>(1, iters); run<32, 2>(1, iters); run<32, 4>(1, iters);
    run<32, 1>(2, iters); run<32, 2>(2, iters); run<32, 1>(4, iters);
    run<16, 1>(1, iters); run<16, 2>(1, iters); run<16, 4>(1, iters);
    run<16, 4>(2, iters); run<16, 2>(4, iters); run<16, 4>(4, iters);
    run_rand<1>(1, 2000); run_rand<1>(2, 2000); run_rand<1>(4, 2000); run_rand<2>(2, 2000); run_rand<4>(1, 2000);
    return 0;
}
