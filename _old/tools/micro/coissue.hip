// What does a wave get to issue beside another wave's back-to-back fp32 MFMA stream on the same SIMD?
// 512-thread workgroups, one per CU: waves 0-3 issue v_mfma_f32_32x32x2_f32 (or 16x16x4) back to back; waves 4-7 run a
// loop of VALU / LDS / transcendental instructions.  Each wave times itself with s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int MODE, int PRIO, int MF, int NOPS = 0>   // NOPS: s_nop 15 (16 idle cycles) x NOPS after every MFMA; MODE 0: v_fma chains, 1: v_exp, 2: ds_read_b128, 3: nothing (MFMA alone);  MF 0: 32x32x2, 1: 16x16x4
__global__ __launch_bounds__(512) void coissue(unsigned long long* out, float* sink, int mfma_iters, int other_iters) {
    __shared__ float lds[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.f;
    __syncthreads();
    float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - threadIdx.x * 1e-3f;
    if (wave < 4) {
        f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        f4v b0 = {0}, b1 = {0}, b2 = {0}, b3 = {0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < mfma_iters; ++i) {
#define PAD() do { if (NOPS >= 1) asm volatile("s_nop 15"); if (NOPS >= 2) asm volatile("s_nop 15"); if (NOPS >= 3) asm volatile("s_nop 15"); if (NOPS >= 4) asm volatile("s_nop 7"); __builtin_amdgcn_sched_barrier(0); } while (0)
            if (MF == 0) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); PAD();
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); PAD();
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0); PAD();
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0); PAD();
            } else {
                b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b1, 0, 0, 0);
                b2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, b2, 0, 0, 0);
                b3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, b3, 0, 0, 0);
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
        sink[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + b0[0] + b1[1] + b2[2] + b3[3];
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        float c0 = x, c1 = y, c2 = x + y, c3 = x - y, c4 = x * y, c5 = 1.f, c6 = 2.f, c7 = 3.f;
        f4v l = {0, 0, 0, 0};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < other_iters; ++i) {
            if (MODE == 0) {
                c0 = __builtin_fmaf(c0, x, y); c1 = __builtin_fmaf(c1, x, y); c2 = __builtin_fmaf(c2, x, y); c3 = __builtin_fmaf(c3, x, y);
                c4 = __builtin_fmaf(c4, x, y); c5 = __builtin_fmaf(c5, x, y); c6 = __builtin_fmaf(c6, x, y); c7 = __builtin_fmaf(c7, x, y);
            } else if (MODE == 1) {
                c0 = __builtin_amdgcn_exp2f(c0); c1 = __builtin_amdgcn_exp2f(c1); c2 = __builtin_amdgcn_exp2f(c2); c3 = __builtin_amdgcn_exp2f(c3);
                c4 = __builtin_amdgcn_exp2f(c4); c5 = __builtin_amdgcn_exp2f(c5); c6 = __builtin_amdgcn_exp2f(c6); c7 = __builtin_amdgcn_exp2f(c7);
            } else if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) l += *reinterpret_cast<volatile f4v*>(&lds[((threadIdx.x & 63) * 4 + 256 * k) & 4095]);
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
        sink[blockIdx.x * 512 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + l[0] + l[1] + l[2] + l[3];
    }
}

template <int MODE, int PRIO, int MF, int NOPS = 0>
void run(const char* tag, unsigned long long* d, float* sink, int mi, int oi) {
    coissue<MODE, PRIO, MF, NOPS><<<256, 512>>>(d, sink, mi, oi);
    coissue<MODE, PRIO, MF, NOPS><<<256, 512>>>(d, sink, mi, oi);
    CK(hipDeviceSynchronize());
    unsigned long long h[256 * 8];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double m = 0, o = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : o) += (double)h[b * 8 + w] / (256 * 4);
    printf("%-44s MFMA wave: %7.1f cycles per MFMA | other wave: %7.1f cycles per instruction (8 per iteration)\n", tag, mi ? m / (4.0 * mi) : 0.0,
           MODE == 3 ? 0.0 : o / (8.0 * oi));
}

int main() {
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, 256 * 8 * 8)); CK(hipMalloc(&sink, 256 * 512 * 4));
    const int mi = 4000;
    run<3, 0, 0>("32x32x2 alone", d, sink, mi, 0);
    run<0, 0, 0>("v_fma alone (no MFMAs)", d, sink, 0, 3000);
    run<2, 0, 0>("ds_read_b128 alone (no MFMAs)", d, sink, 0, 1500);
    run<0, 0, 0>("32x32x2 + v_fma (prio 0)", d, sink, mi, 3000);
    run<0, 1, 0>("32x32x2 + v_fma (prio 3)", d, sink, mi, 3000);
    run<1, 0, 0>("32x32x2 + v_exp (prio 0)", d, sink, mi, 2000);
    run<1, 1, 0>("32x32x2 + v_exp (prio 3)", d, sink, mi, 2000);
    run<2, 0, 0>("32x32x2 + ds_read_b128 (prio 0)", d, sink, mi, 1500);
    run<2, 1, 0>("32x32x2 + ds_read_b128 (prio 3)", d, sink, mi, 1500);
    run<0, 0, 0, 1>("32x32x2 + 16 nop cycles + v_fma", d, sink, mi, 3000);
    run<0, 0, 0, 2>("32x32x2 + 32 nop cycles + v_fma", d, sink, mi, 6000);
    run<0, 0, 0, 3>("32x32x2 + 48 nop cycles + v_fma", d, sink, mi, 12000);
    run<0, 0, 0, 4>("32x32x2 + 56 nop cycles + v_fma", d, sink, mi, 12000);
    run<0, 1, 0, 3>("32x32x2 + 48 nop cycles + v_fma (prio 3)", d, sink, mi, 12000);
    run<1, 0, 0, 3>("32x32x2 + 48 nop cycles + v_exp", d, sink, mi, 8000);
    run<2, 0, 0, 3>("32x32x2 + 48 nop cycles + ds_read_b128", d, sink, mi, 6000);
    run<3, 0, 0, 3>("32x32x2 + 48 nop cycles alone", d, sink, mi, 0);
    run<3, 0, 0, 4>("32x32x2 + 56 nop cycles alone", d, sink, mi, 0);
    run<3, 0, 1>("16x16x4 alone", d, sink, 2 * mi, 0);
    run<0, 0, 1>("16x16x4 + v_fma (prio 0)", d, sink, 2 * mi, 3000);
    run<0, 1, 1>("16x16x4 + v_fma (prio 3)", d, sink, 2 * mi, 3000);
    run<1, 1, 1>("16x16x4 + v_exp (prio 3)", d, sink, 2 * mi, 2000);
    run<2, 1, 1>("16x16x4 + ds_read_b128 (prio 3)", d, sink, 2 * mi, 1500);
    return 0;
}
