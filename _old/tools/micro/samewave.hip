// Same-wave interleave: cost of K filler instructions (v_fma / v_exp / ds_read_b128) after every fp32 MFMA, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int KIND, int K, int WAVES>   // KIND 0 v_fma, 1 v_exp, 2 ds_read_b128, 3 v_pk_fma
__global__ __launch_bounds__(64 * WAVES) void kern(unsigned long long* out, float* sink, int iters) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1.f;
    __syncthreads();
    float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - threadIdx.x * 1e-3f;
    f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float c[16];
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v pk[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { c[k] = x + k; pk[k] = f2v{x + k, y - k}; }
    f4v l = {0, 0, 0, 0};
    const float* lp = &lds[(threadIdx.x & 63) * 4];
#define FILL()                                                                                              \
    do {                                                                                                    \
        _Pragma("unroll") for (int k = 0; k < K; ++k) {                                                     \
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(c[k]) : "v"(x), "v"(y));          \
            else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(c[k]));                              \
            else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[k]) : "v"(pk[(k + 1) & 15])); \
            else { f4v t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"((unsigned)(size_t)lp), "i"(k * 1024)); l += t; } \
        }                                                                                                   \
    } while (0)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); FILL();
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0); FILL();
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0); FILL();
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0); FILL();
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    float s = a0[0] + a1[1] + a2[2] + a3[3] + l[0] + l[1] + l[2] + l[3];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += c[k] + pk[k][0] + pk[k][1];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND, int K, int WAVES>
void run(unsigned long long* d, float* sink) {
    const int iters = 2000;
    kern<KIND, K, WAVES><<<256, 64 * WAVES>>>(d, sink, iters);
    kern<KIND, K, WAVES><<<256, 64 * WAVES>>>(d, sink, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[256 * 8];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    double m = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < WAVES; ++w) m += (double)h[b * 8 + w] / (256 * WAVES);
    const char* names[] = {"v_fma_f32", "v_exp_f32", "ds_read_b128 (lgkmcnt never waited)", "v_pk_fma_f32"};
    printf("%d waves/CU, %2d x %-36s per MFMA: %7.1f cycles per MFMA (wave view)\n", WAVES, K, names[KIND], m / (4.0 * iters));
}

int main() {
    unsigned long long* d; float* sink;
    CK(hipMalloc(&d, 256 * 8 * 8)); CK(hipMalloc(&sink, 256 * 512 * 4));
    run<0, 0, 4>(d, sink); run<0, 1, 4>(d, sink); run<0, 2, 4>(d, sink); run<0, 4, 4>(d, sink); run<0, 8, 4>(d, sink); run<0, 12, 4>(d, sink); run<0, 16, 4>(d, sink);
    run<3, 2, 4>(d, sink); run<3, 4, 4>(d, sink); run<3, 8, 4>(d, sink);
    run<1, 1, 4>(d, sink); run<1, 2, 4>(d, sink); run<1, 4, 4>(d, sink);
    run<2, 1, 4>(d, sink); run<2, 2, 4>(d, sink); run<2, 4, 4>(d, sink);
    run<0, 4, 8>(d, sink); run<0, 8, 8>(d, sink); run<0, 16, 8>(d, sink);
    return 0;
}
