// deform_bwd.hip — training kernels of the deformable cross-view attention ("backward HIP kernels for row 10").
//
// In training the offset network of SwinDAttention (deform:334-349) runs unfused so that the tape can keep its
// intermediates: depthwise 5x5 conv (this file) -> LayerNorm (mumpy_layernorm_fwd/bwd) -> GELU (mumpy_gelu_fwd/bwd) ->
// 1x1 conv to 2 offsets (mumpy_linear_fwd) -> tanh * 2/7 + reference points.  This file holds what has no counterpart
// elsewhere: the windowed depthwise convolution (forward + backward) and the backward of the bilinear window sampling.
// Layout everywhere: token-major windows (N, 49, C'), pixel p = 7*y + x, channels contiguous.  Deterministic reductions.
#include "common.h"
using namespace mumpy;

namespace {

// u[n][p][c] = b[c] + sum_{dy,dx in [-2,2]} x[n][p + (dy,dx)][c] * w[c][(dy+2)*5 + (dx+2)]   (zero padding inside the 7x7 window)
__global__ __launch_bounds__(256) void dwconv5_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ u, int C) {
    extern __shared__ float sm[];                     // the window: 49 * C floats
    const int n = blockIdx.x;
    const float* xn = x + (int64_t)n * WT * C;
    for (int i = threadIdx.x; i < WT * C; i += 256) sm[i] = xn[i];
    __syncthreads();
    for (int i = threadIdx.x; i < WT * C; i += 256) {
        const int c = i % C, p = i / C, y = p / WS, xx = p - y * WS;
        float acc = b[c];
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int yy = y + dy, xq = xx + dx;
                if ((unsigned)yy < (unsigned)WS && (unsigned)xq < (unsigned)WS)
                    acc = fmaf(sm[(yy * WS + xq) * C + c], w[c * 25 + (dy + 2) * 5 + (dx + 2)], acc);
            }
        u[(int64_t)n * WT * C + i] = acc;
    }
}

// dx[n][p][c] = sum_taps du[n][p - tap][c] w[c][tap];  part[n][26][C] = {dw[tap][c] (25 rows), db[c]} of this window
__global__ __launch_bounds__(256) void dwconv5_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ du, float* __restrict__ dx,
                                                          float* __restrict__ part, int C) {
    extern __shared__ float sm[];                     // x window then du window: 2 * 49 * C floats
    float* sx = sm;
    float* sd = sm + WT * C;
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < WT * C; i += 256) {
        sx[i] = x[(int64_t)n * WT * C + i];
        sd[i] = du[(int64_t)n * WT * C + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < WT * C; i += 256) {
        const int c = i % C, p = i / C, y = p / WS, xx = p - y * WS;
        float acc = 0.f;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx_ = -2; dx_ <= 2; ++dx_) {
                const int yy = y - dy, xq = xx - dx_;             // output pixel that read this input through tap (dy,dx)
                if ((unsigned)yy < (unsigned)WS && (unsigned)xq < (unsigned)WS)
                    acc = fmaf(sd[(yy * WS + xq) * C + c], w[c * 25 + (dy + 2) * 5 + (dx_ + 2)], acc);
            }
        dx[(int64_t)n * WT * C + i] = acc;
    }
    // weight / bias gradient partials of this window: one thread per (tap or bias, channel), pixels in order
    for (int i = threadIdx.x; i < 26 * C; i += 256) {
        const int c = i % C, tap = i / C;
        float acc = 0.f;
        if (tap == 25) {
            for (int p = 0; p < WT; ++p) acc += sd[p * C + c];
        } else {
            const int dy = tap / 5 - 2, dx_ = tap % 5 - 2;
            for (int p = 0; p < WT; ++p) {
                const int y = p / WS, xx = p - y * WS, yy = y + dy, xq = xx + dx_;
                if ((unsigned)yy < (unsigned)WS && (unsigned)xq < (unsigned)WS) acc = fmaf(sd[p * C + c], sx[(yy * WS + xq) * C + c], acc);
            }
        }
        part[((int64_t)n * 26 + tap) * C + c] = acc;
    }
}

// fixed-order sum over windows: out[i] = sum_n part[n * width + i]   (16 lane groups, as partial_reduce in backward.hip)
__global__ __launch_bounds__(1024) void window_partial_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                     int64_t nparts, int64_t width) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < width)
        for (int64_t p = grp; p < nparts; p += 16) s += part[p * width + i];
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < width) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        out[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bilinear window sampling (grid_sample, align_corners=True, zeros padding; deform:353-356) backward.
// forward (deform.hip): sampled[b2][p][g*Cg + c] = sum_{4 corners} wgt * x2[b2][corner][g*Cg + c], sample point
// (fy, fx) = ((pos_y + 1) * 3, (pos_x + 1) * 3), pos = pos[b2 % nq][g][p].
//   dx2[b2][pix][ch]      = sum_p wgt(p, pix) dsampled[b2][p][ch]            (gather over the 49 points of the channel's group)
//   dpos_part[b2][g][p]   = 3 * sum_{c in g} dsampled[b2][p][c] * d(sample)/d(fy, fx)     (summed over the r windows that share
//                           a q window by the caller, in order)
struct Corner { int y0, x0; float ly, lx; };
__device__ __forceinline__ Corner corner_of(float py, float px) {
    Corner k;
    const float fy = (py + 1.0f) * 3.0f, fx = (px + 1.0f) * 3.0f;
    const float y0f = floorf(fy), x0f = floorf(fx);
    k.y0 = (int)y0f; k.x0 = (int)x0f; k.ly = fy - y0f; k.lx = fx - x0f;
    return k;
}

// grid (B2, 3): one block per (kv window, channel group) -- a group's x2 / dsampled slices (2 * 49 * Cg floats) fit LDS
__global__ __launch_bounds__(256) void deform_sample_bwd_kernel(const float* __restrict__ x2, const float* __restrict__ pos,
                                                                const float* __restrict__ ds, float* __restrict__ dx2,
                                                                float* __restrict__ dpos_part, int C, int nq) {
    extern __shared__ float sm[];                     // x2 slice (49*Cg) | dsampled slice (49*Cg) | pos of the group (49*2)
    const int b2 = blockIdx.x, g = blockIdx.y, Cg = C / 3;
    float* sx = sm;
    float* sd = sm + WT * Cg;
    float* sp = sm + 2 * WT * Cg;
    for (int i = threadIdx.x; i < WT * Cg; i += 256) {
        const int c = i % Cg, p = i / Cg;
        sx[i] = x2[((int64_t)b2 * WT + p) * C + g * Cg + c];
        sd[i] = ds[((int64_t)b2 * WT + p) * C + g * Cg + c];
    }
    const float* pq = pos + ((int64_t)(b2 % nq) * 3 + g) * WT * 2;
    for (int i = threadIdx.x; i < WT * 2; i += 256) sp[i] = pq[i];
    __syncthreads();
    // dx2: thread per (pixel, channel); the 49 points in order
    for (int i = threadIdx.x; i < WT * Cg; i += 256) {
        const int c = i % Cg, pix = i / Cg, y = pix / WS, x = pix - y * WS;
        float acc = 0.f;
        for (int p = 0; p < WT; ++p) {
            const Corner k = corner_of(sp[p * 2], sp[p * 2 + 1]);
            const float wy = (k.y0 == y ? 1.0f - k.ly : 0.f) + (k.y0 + 1 == y ? k.ly : 0.f);
            const float wx = (k.x0 == x ? 1.0f - k.lx : 0.f) + (k.x0 + 1 == x ? k.lx : 0.f);
            const float wgt = wy * wx;
            if (wgt != 0.f) acc = fmaf(wgt, sd[p * Cg + c], acc);
        }
        dx2[((int64_t)b2 * WT + pix) * C + g * Cg + c] = acc;
    }
    // dpos: one thread per point; channels of the group in order
    for (int p = threadIdx.x; p < WT; p += 256) {
        const Corner k = corner_of(sp[p * 2], sp[p * 2 + 1]);
        auto px = [&](int yy, int xx, int c) -> float {
            return ((unsigned)yy < (unsigned)WS && (unsigned)xx < (unsigned)WS) ? sx[(yy * WS + xx) * Cg + c] : 0.f;
        };
        float gy = 0.f, gx = 0.f;
        for (int c = 0; c < Cg; ++c) {
            const float v00 = px(k.y0, k.x0, c), v01 = px(k.y0, k.x0 + 1, c), v10 = px(k.y0 + 1, k.x0, c), v11 = px(k.y0 + 1, k.x0 + 1, c);
            const float d = sd[p * Cg + c];
            gy = fmaf(d, (v10 - v00) * (1.0f - k.lx) + (v11 - v01) * k.lx, gy);
            gx = fmaf(d, (v01 - v00) * (1.0f - k.ly) + (v11 - v10) * k.ly, gx);
        }
        float* o = dpos_part + (((int64_t)b2 * 3 + g) * WT + p) * 2;
        o[0] = 3.0f * gy;                                    // d f / d pos = 3 (half the 6-pixel span)
        o[1] = 3.0f * gx;
    }
}

}  // namespace

extern "C" int mumpy_dwconv5_window_fwd(const float* x, const float* w, const float* b, float* u, int64_t N, int C, void* stream) {
    if (N == 0) return 0;
    MUMPY_REQUIRE(x && w && b && u, MUMPY_ENULL, "dwconv5: null pointer");
    MUMPY_REQUIRE(N > 0 && N < (1ll << 31) && C > 0 && C <= 384, MUMPY_EINVAL, "dwconv5: bad shape (C=%d)", C);
    const size_t lds = WT * C * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv5_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        MUMPY_REQUIRE(e == hipSuccess, (int)e, "dwconv5: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(dwconv5_fwd_kernel, dim3((unsigned)N), dim3(256), lds, as_stream(stream), x, w, b, u, C);
    MUMPY_CHECK_LAUNCH("dwconv5_fwd");
    return 0;
}

extern "C" int64_t mumpy_dwconv5_window_bwd_workspace_bytes(int64_t N, int C) {
    return (N <= 0 || C <= 0) ? 0 : N * 26 * C * (int64_t)sizeof(float);
}

extern "C" int mumpy_dwconv5_window_bwd(const float* x, const float* w, const float* du, float* dx, float* dw, float* db,
                                        void* workspace, int64_t workspace_bytes, int64_t N, int C, void* stream) {
    MUMPY_REQUIRE(x && w && du && dx && dw && db && workspace, MUMPY_ENULL, "dwconv5_bwd: null pointer");
    MUMPY_REQUIRE(N > 0 && N < (1ll << 31) && C > 0 && C <= 384, MUMPY_EINVAL, "dwconv5_bwd: bad shape (C=%d)", C);
    MUMPY_REQUIRE(workspace_bytes >= mumpy_dwconv5_window_bwd_workspace_bytes(N, C), MUMPY_EINVAL, "dwconv5_bwd: workspace too small");
    float* part = static_cast<float*>(workspace);
    const size_t lds = 2 * WT * C * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv5_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        MUMPY_REQUIRE(e == hipSuccess, (int)e, "dwconv5_bwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(dwconv5_bwd_kernel, dim3((unsigned)N), dim3(256), lds, as_stream(stream), x, w, du, dx, part, C);
    MUMPY_CHECK_LAUNCH("dwconv5_bwd");
    // part rows are [tap 0..24][C] then [bias][C]; dw is (C, 25) like the module's (C,1,5,5) weight: reduce to a (26, C) image,
    // the caller transposes the first 25 rows (mumpy_hip/autograd.py)
    hipLaunchKernelGGL(window_partial_reduce_kernel, dim3((unsigned)((25 * C + 63) / 64)), dim3(1024), 0, as_stream(stream), part, dw, N,
                       (int64_t)26 * C);
    MUMPY_CHECK_LAUNCH("dwconv5_bwd(reduce dw)");
    hipLaunchKernelGGL(window_partial_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, as_stream(stream), part + 25 * C, db,
                       N, (int64_t)26 * C);
    MUMPY_CHECK_LAUNCH("dwconv5_bwd(reduce db)");
    return 0;
}

extern "C" int mumpy_deform_sample_bwd(const float* x2, const float* pos, const float* dsampled, float* dx2, float* dpos_part,
                                       int64_t B2, int C, int nq, void* stream) {
    MUMPY_REQUIRE(x2 && pos && dsampled && dx2 && dpos_part, MUMPY_ENULL, "deform_sample_bwd: null pointer");
    MUMPY_REQUIRE(B2 > 0 && B2 < (1ll << 31) && C > 0 && C % 3 == 0 && C <= 768 && nq > 0, MUMPY_EINVAL, "deform_sample_bwd: bad shape");
    const size_t lds = (2 * WT * (C / 3) + WT * 2) * sizeof(float);
    MUMPY_REQUIRE(lds <= 160 * 1024, MUMPY_ERANGE, "deform_sample_bwd: window does not fit LDS");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(deform_sample_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        MUMPY_REQUIRE(e == hipSuccess, (int)e, "deform_sample_bwd: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(deform_sample_bwd_kernel, dim3((unsigned)B2, 3), dim3(256), lds, as_stream(stream), x2, pos, dsampled, dx2,
                       dpos_part, C, nq);
    MUMPY_CHECK_LAUNCH("deform_sample_bwd");
    return 0;
}
