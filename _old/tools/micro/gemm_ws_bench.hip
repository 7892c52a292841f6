// Standalone timing + correctness harness for csrc/gemm_ws.h (build: see tools/micro/Makefile; run on the GPU box).
//   gemm_ws_bench [M N K [act [residual]]]...   without arguments: the large shapes of the B=8,T=5 forward.
// Checks every output against an fp64-accumulated reference kernel and prints TFLOP/s of back-to-back launches.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd/csrc/gemm_ws.h"
#include "../../multilateral-temporal-view-pyramid-transformer-for-video-inpainting-detection_amd/csrc/gemm_ws64.h"

namespace mumpy {
void set_error(const char* fmt, ...) { fprintf(stderr, "error: %s\n", fmt); }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void ref_kernel(const float* X, const float* W, const float* bias, const float* res, float* Y, int M, int N, int K, int act) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    double s = 0;
    for (int k = 0; k < K; ++k) s += (double)X[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
    if (bias) s += bias[n];
    if (act == 1) s = 0.5 * s * (1.0 + erf(s * 0.70710678118654752440));
    if (res) s += res[(size_t)m * N + n];
    Y[(size_t)m * N + n] = (float)s;
}

__global__ void ref_conv_kernel(const float* X, const float* Wt, const float* bias, const float* res, float* Y, int B, int H, int Wd,
                                int C, int N, int kh, int kw, int act) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= B * H * Wd) return;
    const int img = m / (H * Wd), y = (m / Wd) % H, x = m % Wd;
    double s = 0;
    for (int r = 0; r < kh; ++r)
        for (int q = 0; q < kw; ++q) {
            const int yy = y + r - kh / 2, xx = x + q - kw / 2;
            if (yy < 0 || yy >= H || xx < 0 || xx >= Wd) continue;
            const float* xp = X + ((size_t)(img * H + yy) * Wd + xx) * C;
            const float* wp = Wt + ((size_t)(n * kh + r) * kw + q) * C;
            for (int c = 0; c < C; ++c) s += (double)xp[c] * (double)wp[c];
        }
    if (bias) s += bias[n];
    if (act == 1) s = 0.5 * s * (1.0 + erf(s * 0.70710678118654752440));
    if (res) s += res[(size_t)m * N + n];
    Y[(size_t)m * N + n] = (float)s;
}

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;      // uniform [-scale, scale)
    }
}

// yardstick: what the matrix pipe of THIS device sustains on a bare v_mfma_f32_32x32x2_f32 loop (boxes differ by >10 %)
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

struct Shape { int M, N, K, act, res; const char* tag; int B = 0, H = 0, W = 0, C = 0, kh = 0, kw = 0; };

int main(int argc, char** argv) {
    std::vector<Shape> shapes;
    if (argc >= 9 && !strcmp(argv[1], "conv")) {      // conv B H W Cin Cout kh kw [act res] ... (groups of 9)
        for (int a = 2; a + 6 < argc; a += 9) {
            Shape sh{0, 0, 0, a + 7 < argc ? atoi(argv[a + 7]) : 0, a + 8 < argc ? atoi(argv[a + 8]) : 0, "conv"};
            sh.B = atoi(argv[a]); sh.H = atoi(argv[a + 1]); sh.W = atoi(argv[a + 2]); sh.C = atoi(argv[a + 3]); sh.N = atoi(argv[a + 4]);
            sh.kh = atoi(argv[a + 5]); sh.kw = atoi(argv[a + 6]);
            sh.M = sh.B * sh.H * sh.W; sh.K = sh.kh * sh.kw * sh.C;
            shapes.push_back(sh);
        }
    } else if (argc >= 4) {
        for (int a = 1; a + 2 < argc; a += 5)
            shapes.push_back({atoi(argv[a]), atoi(argv[a + 1]), atoi(argv[a + 2]), a + 3 < argc ? atoi(argv[a + 3]) : 0,
                              a + 4 < argc ? atoi(argv[a + 4]) : 0, "cli"});
    } else {
        shapes = {{7840, 2048, 512, 1, 0, "v3s2 fc1"}, {7840, 512, 2048, 0, 1, "v3s2 fc2"}, {7840, 1536, 512, 0, 0, "v3s2 qkv"},
                  {7840, 512, 512, 0, 1, "v3s2 proj"}, {1960, 3072, 768, 1, 0, "g fc1"}, {1960, 768, 3072, 0, 1, "g fc2"},
                  {1960, 2304, 768, 0, 0, "g qkv"}, {125440, 512, 128, 1, 0, "v3s0 fc1"}, {125440, 128, 512, 0, 1, "v3s0 fc2"},
                  {31360, 1024, 256, 1, 0, "v3s1 fc1"}, {31360, 256, 1024, 0, 1, "v3s1 fc2"}, {1568, 1536, 384, 1, 0, "v1s2 fc1"},
                  {1000, 200, 96, 1, 1, "ragged"}};
    }
    int dev_cu = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    dev_cu = prop.multiProcessorCount;
    const int grid_override = getenv("WS_GRID") ? atoi(getenv("WS_GRID")) : 0;
    const int force_split = getenv("WS_SPLIT") ? atoi(getenv("WS_SPLIT")) : -1;
    const bool tile64 = getenv("WS_TILE") && atoi(getenv("WS_TILE")) == 64;      // the 64x64-tile kernel of gemm_ws64.h
    const int reps = getenv("WS_REPS") ? atoi(getenv("WS_REPS")) : 20;
    printf("%s, %d CUs, LDS %d B per workgroup, dbg %d\n", prop.name, dev_cu, mumpy::gemm_ws::LDS_BYTES, mumpy::gemm_ws::DBG);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    {
        float* o; CK(hipMalloc(&o, 256 * 256 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 20000;
        mfma_peak_kernel<<<256, 256, 0, s>>>(o, iters);
        CK(hipEventRecord(e0, s));
        mfma_peak_kernel<<<256, 256, 0, s>>>(o, iters);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("bare MFMA loop on this device: %.1f TF\n", 256.0 * 4 * iters * 4 * 4096.0 / (ms * 1e-3) / 1e12);
        CK(hipFree(o));
    }
    void* wsp = nullptr;
    const int64_t wsb = mumpy::gemm_ws::workspace_bytes(dev_cu);
    CK(hipMalloc(&wsp, wsb));
    CK(hipMemset(wsp, 0xff, wsb));       // (the launcher must not rely on the workspace's contents)
    unsigned long long* stamps = nullptr;
#ifdef MUMPY_WS_STAMP
    CK(hipMalloc(&stamps, 512 * 8 * 8));
#endif
    for (const Shape& sh : shapes) {
        const bool is_conv = sh.C != 0;
        const mumpy::gemm_ws::Conv cvd{sh.H, sh.W, sh.C, sh.kh, sh.kw};
        const mumpy::gemm_ws::Conv* cv = is_conv ? &cvd : nullptr;
        const size_t nx = is_conv ? (size_t)sh.M * sh.C : (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, ny = (size_t)sh.M * sh.N;
        float *X, *W, *B, *R, *Y, *Yr;
        CK(hipMalloc(&X, nx * 4)); CK(hipMalloc(&W, nw * 4)); CK(hipMalloc(&B, sh.N * 4));
        CK(hipMalloc(&R, ny * 4)); CK(hipMalloc(&Y, ny * 4)); CK(hipMalloc(&Yr, ny * 4));
        fill_kernel<<<1024, 256, 0, s>>>(X, nx, 1u, 1.0f);
        fill_kernel<<<1024, 256, 0, s>>>(W, nw, 2u, 1.0f / sqrtf((float)sh.K));
        fill_kernel<<<64, 256, 0, s>>>(B, sh.N, 3u, 0.5f);
        fill_kernel<<<1024, 256, 0, s>>>(R, ny, 4u, 1.0f);
        CK(hipMemsetAsync(Y, 0xff, ny * 4, s));
        if (is_conv ? !mumpy::gemm_ws::conv_eligible(sh.M, sh.N, cvd) : !mumpy::gemm_ws::eligible(sh.M, sh.N, sh.K)) { printf("%-10s %d %d %d not eligible\n", sh.tag, sh.M, sh.N, sh.K); continue; }
        const int cu = grid_override ? grid_override : dev_cu;
        auto go = [&]() {
            return tile64 ? mumpy::gemm_ws64::launch(X, W, B, sh.res ? R : nullptr, Y, sh.M, sh.N, sh.K, sh.act, cu, s, getenv("WS64_ONE") ? 1 : 2)
                          : mumpy::gemm_ws::launch(X, W, B, sh.res ? R : nullptr, Y, sh.M, sh.N, sh.K, sh.act, cu, s, wsp, wsb, force_split, stamps, false, cv);
        };
        if (go()) return 1;
        CK(hipGetLastError());
        if (is_conv)
            ref_conv_kernel<<<dim3((sh.N + 255) / 256, sh.M), 256, 0, s>>>(X, W, B, sh.res ? R : nullptr, Yr, sh.B, sh.H, sh.W, sh.C, sh.N, sh.kh, sh.kw, sh.act);
        else
        ref_kernel<<<dim3((sh.N + 255) / 256, sh.M), 256, 0, s>>>(X, W, B, sh.res ? R : nullptr, Yr, sh.M, sh.N, sh.K, sh.act);
        CK(hipStreamSynchronize(s));
        std::vector<float> hy(ny), hr(ny);
        CK(hipMemcpy(hy.data(), Y, ny * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), Yr, ny * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0; size_t bad = 0;
        for (size_t i = 0; i < ny; ++i) {
            const double d = fabs((double)hy[i] - (double)hr[i]);
            if (!(d <= 1e-4 * (1.0 + fabs(hr[i])))) ++bad;
            if (d > maxerr || d != d) maxerr = d;
            if (fabs(hr[i]) > maxref) maxref = fabs(hr[i]);
        }
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) go();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) go();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, tf = 2.0 * sh.M * sh.N * sh.K / us / 1e6;
        printf("%-10s M=%6d N=%5d K=%5d act=%d res=%d  %8.1f us %6.1f TF   max|err| %.2e (max|ref| %.2f) bad %zu%s\n", sh.tag, sh.M, sh.N,
               sh.K, sh.act, sh.res, us, tf, maxerr, maxref, bad, bad ? "  <-- MISMATCH" : "");
#ifdef MUMPY_WS_STAMP
        {
            std::vector<unsigned long long> h(256 * 8);
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            double a[8] = {0};
            const int nb = 256;
            for (int b = 0; b < nb; ++b) for (int k = 0; k < 8; ++k) a[k] += (double)h[b * 8 + k] / nb;
            printf("   per chunk (cycles, mean over blocks): matrix loop %.0f of which barrier wait %.0f | loader: work %.0f barrier %.0f | epilogue: work %.0f barrier %.0f   (prologue %.0f, chunks %.0f)\n",
                   a[1] / a[3], a[2] / a[3], a[4] / a[3], a[5] / a[3], a[6] / a[3], a[7] / a[3], a[0], a[3]);
        }
#endif
        fflush(stdout);
        CK(hipFree(X)); CK(hipFree(W)); CK(hipFree(B)); CK(hipFree(R)); CK(hipFree(Y)); CK(hipFree(Yr));
    }
    return 0;
}
