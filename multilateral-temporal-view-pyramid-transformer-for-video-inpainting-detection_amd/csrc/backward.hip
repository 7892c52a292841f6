// backward.hip — backward kernels of the Swin block's non-attention parts (SURVEY 8f-2, rows 6-7 of 8a in training):
// LayerNorm backward, exact-erf GELU forward/backward, 2-D transpose (operand layout for the weight-gradient GEMMs) and
// deterministic column sums (bias gradients).  All HBM-bound streaming kernels; every reduction runs in a fixed order
// (no atomics), so gradients are bitwise reproducible.
//
// Linear backward itself reuses mumpy_linear_fwd (y = x W^T):  dX = dY W  = linear(dY, W^T),  dW = dY^T X = linear(dY^T, X^T),
// with the transposes produced by mumpy_transpose_fwd (mumpy_hip/autograd.py).
#include "common.h"
using namespace mumpy;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm backward (nn.LayerNorm over the last dim, eps inside the sqrt; swin:266,305).  One wave per row:
//   xhat = (x - mean) * rstd,  g = dy * gamma,  dx = rstd * (g - mean(g) - xhat * mean(g * xhat))
// dgamma / dbeta: each block accumulates its rows in registers (lane owns columns lane, lane+64, ...) and writes one
// partial row per wave; ln_bwd_reduce_kernel sums the partials in order.
constexpr int LN_MAXC4 = 8;             // float4 columns per lane: C <= 64 * 4 * 8 = 2048 (PatchMerging's LayerNorm(4C) at C = 512)

__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, float* __restrict__ dx,
                                                     float* __restrict__ partial, int64_t rows, int C, float eps,
                                                     int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int n4 = C >> 2;                                  // float4 columns
    f32x4 dg[LN_MAXC4], db[LN_MAXC4];
#pragma unroll
    for (int i = 0; i < LN_MAXC4; ++i) { dg[i] = f32x4{0, 0, 0, 0}; db[i] = f32x4{0, 0, 0, 0}; }
    const int64_t r0 = gw * rows_per_wave;
    for (int64_t r = r0; r < r0 + rows_per_wave && r < rows; ++r) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + r * C);
        const f32x4* dyr = reinterpret_cast<const f32x4*>(dy + r * C);
        f32x4 xv[LN_MAXC4], gv[LN_MAXC4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < n4) { xv[i] = xr[c4]; s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w); }
        }
        const float mean = wave_sum(s, 64) / (float)C;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < n4) { xv[i] -= mean; v += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w); }
        }
        const float rstd = rsqrtf(wave_sum(v, 64) / (float)C + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < n4) {
                const f32x4 d = dyr[c4];
                const f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c4];
                xv[i] *= rstd;                                  // xhat
                gv[i] = d * gm;
                db[i] += d;
                dg[i] += d * xv[i];
                sg += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
                sgx += (gv[i].x * xv[i].x + gv[i].y * xv[i].y) + (gv[i].z * xv[i].z + gv[i].w * xv[i].w);
            }
        }
        const float mg = wave_sum(sg, 64) / (float)C, mgx = wave_sum(sgx, 64) / (float)C;
        f32x4* dxr = reinterpret_cast<f32x4*>(dx + r * C);
#pragma unroll
        for (int i = 0; i < LN_MAXC4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < n4) dxr[c4] = (gv[i] - mg - xv[i] * mgx) * rstd;
        }
    }
    f32x4* pg = reinterpret_cast<f32x4*>(partial + gw * 2 * C);
    f32x4* pb = reinterpret_cast<f32x4*>(partial + gw * 2 * C + C);
#pragma unroll
    for (int i = 0; i < LN_MAXC4; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < n4) { pg[c4] = dg[i]; pb[c4] = db[i]; }
    }
}

// out[c] = sum_p partial[p * stride + c]: 64 columns per block, 16 lane groups each summing every 16th partial, combined
// through LDS in group order -- a fixed summation tree (bitwise reproducible) without a thousands-long serial chain
__global__ __launch_bounds__(1024) void partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                              int64_t nparts, int64_t width, int64_t stride) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < width)
        for (int64_t p = grp; p < nparts; p += 16) s += partial[p * stride + i];
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < width) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        out[i] = t;
    }
}

// column sums of a (R, C) matrix: stage 1 writes one partial row per block of rows
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t R,
                                                              int C, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int64_t r = r0; r < r0 + rows_per_block && r < R; ++r) s += x[r * C + c];
    partial[(int64_t)blockIdx.y * C + c] = s;
}

__device__ __forceinline__ float gelu_grad(float x) {          // d/dx [0.5 x (1 + erf(x / sqrt 2))]
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

__global__ __launch_bounds__(256) void gelu_fwd_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = x[i];
        y[i] = f32x4{gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w)};
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ dy,
                                                       f32x4* __restrict__ dx, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = x[i], d = dy[i];
        dx[i] = f32x4{d.x * gelu_grad(v.x), d.y * gelu_grad(v.y), d.z * gelu_grad(v.z), d.w * gelu_grad(v.w)};
    }
}

// out (C, R) = in (R, C)^T, 64x64 tiles through LDS (stride 65: conflict-free both ways)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t R, int64_t C) {
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + ty + 4 * i, c = c0 + tx;
        if (r < R && c < C) tile[ty + 4 * i][tx] = in[r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < C && r < R) out[c * R + r] = tile[tx][ty + 4 * i];
    }
}

}  // namespace

static int64_t ln_bwd_waves(int64_t rows) {            // waves (= partial rows): enough to fill the chip, >= 8 rows each
    int64_t w = (rows + 7) / 8;
    if (w > 2048) w = 2048;
    if (w < 4) w = 4;
    return (w + 3) / 4 * 4;
}

extern "C" int64_t mumpy_layernorm_bwd_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return ln_bwd_waves(rows) * 2 * C * (int64_t)sizeof(float);
}

extern "C" int mumpy_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta,
                                   void* workspace, int64_t workspace_bytes, int64_t rows, int C, float eps, void* stream) {
    if (rows == 0) return 0;
    MUMPY_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && workspace, MUMPY_ENULL, "layernorm_bwd: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(gamma) && aligned16(dy) && aligned16(dx) && aligned16(workspace), MUMPY_EALIGN,
                  "layernorm_bwd: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXC4, MUMPY_EINVAL, "layernorm_bwd: unsupported C=%d", C);
    MUMPY_REQUIRE(workspace_bytes >= mumpy_layernorm_bwd_workspace_bytes(rows, C), MUMPY_EINVAL, "layernorm_bwd: workspace too small");
    const int64_t waves = ln_bwd_waves(rows);
    const int rpw = (int)((rows + waves - 1) / waves);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)(waves / 4)), dim3(256), 0, as_stream(stream), x, gamma, dy, dx, partial,
                       rows, C, eps, rpw);
    MUMPY_CHECK_LAUNCH("layernorm_bwd");
    // partial rows are [dgamma(C) | dbeta(C)] per wave; dgamma and dbeta are separate caller buffers: two strided reduces
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, as_stream(stream), partial, dgamma,
                       waves, (int64_t)C, (int64_t)2 * C);
    MUMPY_CHECK_LAUNCH("layernorm_bwd(reduce dgamma)");
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, as_stream(stream), partial + C, dbeta,
                       waves, (int64_t)C, (int64_t)2 * C);
    MUMPY_CHECK_LAUNCH("layernorm_bwd(reduce dbeta)");
    return 0;
}

extern "C" int mumpy_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (n == 0) return 0;
    MUMPY_REQUIRE(x && y, MUMPY_ENULL, "gelu: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(y) && n % 4 == 0, MUMPY_EALIGN, "gelu: need 16-byte aligned buffers and n %% 4 == 0");
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), reinterpret_cast<const f32x4*>(x),
                       reinterpret_cast<f32x4*>(y), n / 4);
    MUMPY_CHECK_LAUNCH("gelu_fwd");
    return 0;
}

extern "C" int mumpy_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    if (n == 0) return 0;
    MUMPY_REQUIRE(x && dy && dx, MUMPY_ENULL, "gelu_bwd: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && n % 4 == 0, MUMPY_EALIGN,
                  "gelu_bwd: need 16-byte aligned buffers and n %% 4 == 0");
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), reinterpret_cast<const f32x4*>(x),
                       reinterpret_cast<const f32x4*>(dy), reinterpret_cast<f32x4*>(dx), n / 4);
    MUMPY_CHECK_LAUNCH("gelu_bwd");
    return 0;
}

extern "C" int mumpy_transpose_fwd(const float* in, float* out, int64_t R, int64_t C, void* stream) {
    if (R == 0 || C == 0) return 0;
    MUMPY_REQUIRE(in && out, MUMPY_ENULL, "transpose: null pointer");
    MUMPY_REQUIRE(R > 0 && C > 0 && (R + 63) / 64 < 65536, MUMPY_EINVAL, "transpose: bad shape");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64)), dim3(256), 0, as_stream(stream),
                       in, out, R, C);
    MUMPY_CHECK_LAUNCH("transpose");
    return 0;
}

static int64_t col_sum_blocks(int64_t R) {
    int64_t b = (R + 255) / 256;
    if (b > 512) b = 512;
    return b < 1 ? 1 : b;
}

extern "C" int64_t mumpy_col_sum_workspace_bytes(int64_t R, int C) {
    if (R <= 0 || C <= 0) return 0;
    return col_sum_blocks(R) * C * (int64_t)sizeof(float);
}

extern "C" int mumpy_col_sum_fwd(const float* x, float* out, void* workspace, int64_t workspace_bytes, int64_t R, int C,
                                 void* stream) {
    MUMPY_REQUIRE(x && out && workspace, MUMPY_ENULL, "col_sum: null pointer");
    MUMPY_REQUIRE(R > 0 && C > 0, MUMPY_EINVAL, "col_sum: bad shape");
    MUMPY_REQUIRE(workspace_bytes >= mumpy_col_sum_workspace_bytes(R, C), MUMPY_EINVAL, "col_sum: workspace too small");
    const int64_t nb = col_sum_blocks(R);
    const int rpb = (int)((R + nb - 1) / nb);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)nb), dim3(256), 0, as_stream(stream), x,
                       partial, R, C, rpb);
    MUMPY_CHECK_LAUNCH("col_sum(partial)");
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, as_stream(stream), partial, out, nb,
                       (int64_t)C, (int64_t)C);
    MUMPY_CHECK_LAUNCH("col_sum(reduce)");
    return 0;
}
