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
// Round 3: one THREAD per (window, channel) with the channel's 49 pixels and 25 taps in registers and the 7x7x5x5 loop nest fully
// unrolled (bounds are compile-time constants: ~750 FMAs, no LDS, no index arithmetic); loads and stores are coalesced over the
// channel.  (The first version staged the window in LDS and spent its time on address arithmetic and 25 LDS reads per output.)
// grid (N, ceil(C / 128)), 128 threads.
__global__ __launch_bounds__(128) void dwconv5_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ u, int C) {
    const int c = blockIdx.y * 128 + threadIdx.x;
    if (c >= C) return;
    const float* xn = x + (int64_t)blockIdx.x * WT * C + c;
    float xv[WT], wv[25];
#pragma unroll
    for (int p = 0; p < WT; ++p) xv[p] = xn[(int64_t)p * C];
#pragma unroll
    for (int t = 0; t < 25; ++t) wv[t] = w[c * 25 + t];
    const float bias = b[c];
    float* un = u + (int64_t)blockIdx.x * WT * C + c;
#pragma unroll
    for (int y = 0; y < WS; ++y)
#pragma unroll
        for (int xx = 0; xx < WS; ++xx) {
            float acc = bias;
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
                for (int dx = -2; dx <= 2; ++dx) {
                    const int yy = y + dy, xq = xx + dx;
                    if (yy >= 0 && yy < WS && xq >= 0 && xq < WS) acc = fmaf(xv[yy * WS + xq], wv[(dy + 2) * 5 + (dx + 2)], acc);
                }
            un[(int64_t)(y * WS + xx) * C] = acc;
        }
}

// dx[n][p][c] = sum_taps du[n][p - tap][c] w[c][tap];  part[n][26][C] = {dw[tap][c] (25 rows), db[c]} of this window.
// Same thread-per-channel, all-in-registers form (summation orders as in the first version: taps in order, pixels in order).
__global__ __launch_bounds__(128) void dwconv5_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ du, float* __restrict__ dx,
                                                          float* __restrict__ part, int C) {
    const int c = blockIdx.y * 128 + threadIdx.x;
    if (c >= C) return;
    const int64_t base = (int64_t)blockIdx.x * WT * C + c;
    float xv[WT], dv[WT], wv[25];
#pragma unroll
    for (int p = 0; p < WT; ++p) { xv[p] = x[base + (int64_t)p * C]; dv[p] = du[base + (int64_t)p * C]; }
#pragma unroll
    for (int t = 0; t < 25; ++t) wv[t] = w[c * 25 + t];
#pragma unroll
    for (int y = 0; y < WS; ++y)
#pragma unroll
        for (int xx = 0; xx < WS; ++xx) {
            float acc = 0.f;
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
                for (int dx_ = -2; dx_ <= 2; ++dx_) {
                    const int yy = y - dy, xq = xx - dx_;             // output pixel that read this input through tap (dy,dx)
                    if (yy >= 0 && yy < WS && xq >= 0 && xq < WS) acc = fmaf(dv[yy * WS + xq], wv[(dy + 2) * 5 + (dx_ + 2)], acc);
                }
            dx[base + (int64_t)(y * WS + xx) * C] = acc;
        }
    float* pn = part + (int64_t)blockIdx.x * 26 * C + c;
#pragma unroll
    for (int tap = 0; tap < 25; ++tap) {
        const int dy = tap / 5 - 2, dx_ = tap % 5 - 2;
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < WT; ++p) {
            const int y = p / WS, xx = p - y * WS, yy = y + dy, xq = xx + dx_;
            if (yy >= 0 && yy < WS && xq >= 0 && xq < WS) acc = fmaf(dv[p], xv[yy * WS + xq], acc);
        }
        pn[(int64_t)tap * C] = acc;
    }
    float sb = 0.f;
#pragma unroll
    for (int p = 0; p < WT; ++p) sb += dv[p];
    pn[(int64_t)25 * C] = sb;
}

// fixed-order sum over windows: out[i] = sum_n part[n * width + i]   (16 lane groups, as partial_reduce in backward.hip)
__global__ __launch_bounds__(1024) void window_partial_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                     int64_t nparts, int64_t width) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < width && grp < nparts)
        s = ordered_sum(part[grp * width + i], part + (grp + 16) * width + i, 16 * width, (int)((nparts - grp + 15) / 16) - 1);
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

// grid (B2, 3): one block per (kv window, channel group) -- a group's x2 / dsampled slices (2 * 49 * Cg floats) fit LDS.
// Round 3: the corner of every point is computed ONCE and each pixel gets the list of (point, weight) pairs that touch it, in point
// order (the first version recomputed all 49 corners in every (pixel, channel) thread); the position gradient is a wave per point
// with the channels across lanes and a fixed-tree wave sum (it was one THREAD per point walking Cg channels).  178 -> see DESIGN 7c.
constexpr int SB_LIST = WT;                       // worst case: every point lands on one pixel
__global__ __launch_bounds__(256) void deform_sample_bwd_kernel(const float* __restrict__ x2, const float* __restrict__ pos,
                                                                const float* __restrict__ ds, float* __restrict__ dx2,
                                                                float* __restrict__ dpos_part, int C, int nq) {
    extern __shared__ float sm[];     // x2 slice (49*Cg) | dsampled slice (49*Cg) | pos (49*2) | corners (49*4) | counts (49) | lists
    const int b2 = blockIdx.x, g = blockIdx.y, Cg = C / 3;
    float* sx = sm;
    float* sd = sm + WT * Cg;
    float* sp = sm + 2 * WT * Cg;
    float* sc = sp + WT * 2;                                      // per point: y0, x0 (as floats: |values| <= 8), ly, lx
    int* cnt = reinterpret_cast<int*>(sc + WT * 4);
    int* lp = cnt + WT;                                           // [pixel][SB_LIST] point index
    float* lw = reinterpret_cast<float*>(lp + WT * SB_LIST);      // [pixel][SB_LIST] weight
    // 16-byte loads, four per array in flight before the LDS writes (Cg is a multiple of 32): at the late stages the launch is 6-30
    // workgroups, and one 4-byte load per loop step made this staging loop 49 memory latencies long -- most of the kernel
    const int cg4 = Cg >> 2, n4 = WT * cg4;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
        f32x4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i < n4) {
                const int c4 = i % cg4, p = i / cg4;
                const int64_t o = ((int64_t)b2 * WT + p) * C + g * Cg + 4 * c4;
                va[u] = *reinterpret_cast<const f32x4*>(x2 + o);
                vb[u] = *reinterpret_cast<const f32x4*>(ds + o);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u;
            if (i < n4) { reinterpret_cast<f32x4*>(sx)[i] = va[u]; reinterpret_cast<f32x4*>(sd)[i] = vb[u]; }
        }
    }
    const float* pq = pos + ((int64_t)(b2 % nq) * 3 + g) * WT * 2;
    if (threadIdx.x < WT) {
        const int p = threadIdx.x;
        const float py = pq[p * 2], px = pq[p * 2 + 1];
        sp[p * 2] = py; sp[p * 2 + 1] = px;
        const Corner k = corner_of(py, px);
        sc[p * 4] = (float)k.y0; sc[p * 4 + 1] = (float)k.x0; sc[p * 4 + 2] = k.ly; sc[p * 4 + 3] = k.lx;
    }
    __syncthreads();
    if (threadIdx.x < WT) {                                       // pixel -> its (point, weight) pairs, points in order
        const int pix = threadIdx.x, y = pix / WS, x = pix - y * WS;
        int n = 0;
        for (int p = 0; p < WT; ++p) {
            const int y0 = (int)sc[p * 4], x0 = (int)sc[p * 4 + 1];
            const float ly = sc[p * 4 + 2], lx = sc[p * 4 + 3];
            const float wy = (y0 == y ? 1.0f - ly : 0.f) + (y0 + 1 == y ? ly : 0.f);
            const float wx = (x0 == x ? 1.0f - lx : 0.f) + (x0 + 1 == x ? lx : 0.f);
            const float wgt = wy * wx;
            if (wgt != 0.f) { lp[pix * SB_LIST + n] = p; lw[pix * SB_LIST + n] = wgt; ++n; }
        }
        cnt[pix] = n;
    }
    __syncthreads();
    // dx2: thread per (pixel, channel); the pixel's points in order (the same products and order as a walk over all 49)
    for (int i = threadIdx.x; i < WT * Cg; i += 256) {
        const int c = i % Cg, pix = i / Cg, n = cnt[pix];
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc = fmaf(lw[pix * SB_LIST + j], sd[lp[pix * SB_LIST + j] * Cg + c], acc);
        dx2[((int64_t)b2 * WT + pix) * C + g * Cg + c] = acc;
    }
    // dpos: a wave per point, channels across the lanes (lane, lane + 64, ...), fixed-tree wave sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = wave; p < WT; p += 4) {
        const int y0 = (int)sc[p * 4], x0 = (int)sc[p * 4 + 1];
        const float ly = sc[p * 4 + 2], lx = sc[p * 4 + 3];
        const bool in_y0 = (unsigned)y0 < (unsigned)WS, in_y1 = (unsigned)(y0 + 1) < (unsigned)WS;
        const bool in_x0 = (unsigned)x0 < (unsigned)WS, in_x1 = (unsigned)(x0 + 1) < (unsigned)WS;
        const int o00 = (y0 * WS + x0) * Cg, o01 = o00 + Cg, o10 = o00 + WS * Cg, o11 = o10 + Cg;
        float gy = 0.f, gx = 0.f;
        for (int c = lane; c < Cg; c += 64) {
            const float v00 = (in_y0 && in_x0) ? sx[o00 + c] : 0.f, v01 = (in_y0 && in_x1) ? sx[o01 + c] : 0.f;
            const float v10 = (in_y1 && in_x0) ? sx[o10 + c] : 0.f, v11 = (in_y1 && in_x1) ? sx[o11 + c] : 0.f;
            const float d = sd[p * Cg + c];
            gy = fmaf(d, (v10 - v00) * (1.0f - lx) + (v11 - v01) * lx, gy);
            gx = fmaf(d, (v01 - v00) * (1.0f - ly) + (v11 - v10) * ly, gx);
        }
        gy = wave_sum(gy, 64);
        gx = wave_sum(gx, 64);
        if (lane == 0) {
            float* o = dpos_part + (((int64_t)b2 * 3 + g) * WT + p) * 2;
            o[0] = 3.0f * gy;                                    // d f / d pos = 3 (half the 6-pixel span)
            o[1] = 3.0f * gx;
        }
    }
}

}  // namespace

extern "C" int mumpy_dwconv5_window_fwd(const float* x, const float* w, const float* b, float* u, int64_t N, int C, void* stream) {
    if (N == 0) return 0;
    MUMPY_REQUIRE(x && w && b && u, MUMPY_ENULL, "dwconv5: null pointer");
    MUMPY_REQUIRE(N > 0 && N < (1ll << 31) && C > 0 && C <= 384, MUMPY_EINVAL, "dwconv5: bad shape (C=%d)", C);
    hipLaunchKernelGGL(dwconv5_fwd_kernel, dim3((unsigned)N, (unsigned)((C + 127) / 128)), dim3(128), 0, as_stream(stream), x, w, b, u, C);
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
    hipLaunchKernelGGL(dwconv5_bwd_kernel, dim3((unsigned)N, (unsigned)((C + 127) / 128)), dim3(128), 0, as_stream(stream), x, w, du, dx, part, C);
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
    MUMPY_REQUIRE(B2 > 0 && B2 < (1ll << 31) && C > 0 && C % 12 == 0 && C <= 768 && nq > 0, MUMPY_EINVAL, "deform_sample_bwd: bad shape (C %% 12)");
    MUMPY_REQUIRE(aligned16(x2) && aligned16(dsampled), MUMPY_EALIGN, "deform_sample_bwd: x2 and dsampled must be 16-byte aligned");
    const size_t lds = (2 * WT * (C / 3) + WT * 2 + WT * 4 + WT + 2 * WT * WT) * sizeof(float);      // slices, pos, corners, counts, lists
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
