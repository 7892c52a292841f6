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
// partial row per block; ln_param_reduce_kernel sums the partials in a fixed tree.
constexpr int LN_MAXC4 = 8;             // float4 columns per lane: C <= 64 * 4 * 8 = 2048 (PatchMerging's LayerNorm(4C) at C = 512)

// NC4 = float4 columns per lane (C <= 256 * NC4), RU = rows in flight per wave: narrow rows are latency-bound (three
// dependent wave reductions per row), so several independent rows are interleaved
template <int NC4, int RU>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ dy, const float* __restrict__ dx_add,
                                                     float* __restrict__ dx, float* __restrict__ partial, int64_t rows, int C,
                                                     float eps, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int n4 = C >> 2;                                  // float4 columns
    const float invc = 1.0f / (float)C;
    f32x4 dg[NC4], db[NC4], gm[NC4];
#pragma unroll
    for (int i = 0; i < NC4; ++i) {
        dg[i] = f32x4{0, 0, 0, 0}; db[i] = f32x4{0, 0, 0, 0};
        const int c4 = lane + 64 * i;
        gm[i] = c4 < n4 ? reinterpret_cast<const f32x4*>(gamma)[c4] : f32x4{0, 0, 0, 0};
    }
    const int64_t r0 = gw * rows_per_wave;
    const int64_t r1 = (r0 + rows_per_wave < rows) ? r0 + rows_per_wave : rows;
    for (int64_t rb = r0; rb < r1; rb += RU) {
        f32x4 xv[RU][NC4], dv[RU][NC4];
        float s[RU], q[RU], sg[RU], sgx[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int64_t r = (rb + u < r1) ? rb + u : r1 - 1;                 // tail rows recompute the last row (not stored)
            s[u] = 0.f;
#pragma unroll
            for (int i = 0; i < NC4; ++i) {
                const int c4 = lane + 64 * i;
                if (c4 < n4) {
                    xv[u][i] = reinterpret_cast<const f32x4*>(x + r * C)[c4];
                    dv[u][i] = reinterpret_cast<const f32x4*>(dy + r * C)[c4];
                    s[u] += (xv[u][i].x + xv[u][i].y) + (xv[u][i].z + xv[u][i].w);
                } else { xv[u][i] = f32x4{0, 0, 0, 0}; dv[u][i] = f32x4{0, 0, 0, 0}; }
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) s[u] = wave_sum(s[u], 64);               // DPP butterfly (common.h): RU independent chains
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const float mean = s[u] * invc;
            q[u] = 0.f;
#pragma unroll
            for (int i = 0; i < NC4; ++i) {
                const int c4 = lane + 64 * i;
                if (c4 < n4) {
                    xv[u][i] -= mean;
                    q[u] += (xv[u][i].x * xv[u][i].x + xv[u][i].y * xv[u][i].y) + (xv[u][i].z * xv[u][i].z + xv[u][i].w * xv[u][i].w);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) q[u] = wave_sum(q[u], 64);
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const float rstd = rsqrtf(q[u] * invc + eps);
            const bool live = rb + u < r1;
            q[u] = rstd;
            sg[u] = 0.f; sgx[u] = 0.f;
#pragma unroll
            for (int i = 0; i < NC4; ++i) {
                xv[u][i] *= rstd;                                              // xhat
                if (live) { db[i] += dv[u][i]; dg[i] += dv[u][i] * xv[u][i]; }
                dv[u][i] *= gm[i];                                             // g = dy * gamma
                sg[u] += (dv[u][i].x + dv[u][i].y) + (dv[u][i].z + dv[u][i].w);
                sgx[u] += (dv[u][i].x * xv[u][i].x + dv[u][i].y * xv[u][i].y) + (dv[u][i].z * xv[u][i].z + dv[u][i].w * xv[u][i].w);
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) { sg[u] = wave_sum(sg[u], 64); sgx[u] = wave_sum(sgx[u], 64); }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            if (rb + u >= r1) continue;
            const float mg = sg[u] * invc, mgx = sgx[u] * invc, rstd = q[u];
            f32x4* dxr = reinterpret_cast<f32x4*>(dx + (rb + u) * C);
            const f32x4* dar = dx_add ? reinterpret_cast<const f32x4*>(dx_add + (rb + u) * C) : nullptr;    // gradient of the
#pragma unroll                                                                                              // residual branch
            for (int i = 0; i < NC4; ++i) {
                const int c4 = lane + 64 * i;
                if (c4 < n4) {
                    f32x4 g = (dv[u][i] - mg - xv[u][i] * mgx) * rstd;
                    if (dar) g += dar[c4];
                    dxr[c4] = g;
                }
            }
        }
    }
    // one partial row [dgamma | dbeta] per BLOCK: the four waves add their registers into an LDS row in wave order (a fixed order:
    // bitwise reproducible) -- a quarter of the rows the parameter reduce has to read (it was a row per wave)
    extern __shared__ f32x4 lrow[];                          // [2 * n4]
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NC4; ++i) {
                const int c4 = lane + 64 * i;
                if (c4 < n4) {
                    if (w == 0) { lrow[c4] = dg[i]; lrow[n4 + c4] = db[i]; }
                    else { lrow[c4] += dg[i]; lrow[n4 + c4] += db[i]; }
                }
            }
        }
        __syncthreads();
    }
    f32x4* prow = reinterpret_cast<f32x4*>(partial + (int64_t)blockIdx.x * 2 * C);
    for (int i = threadIdx.x; i < 2 * n4; i += 256) prow[i] = lrow[i];
}

// out[c] = sum_p partial[p * stride + c]: 64 columns per block, 16 lane groups each summing every 16th partial, combined
// through LDS in group order -- a fixed summation tree (bitwise reproducible) without a thousands-long serial chain
__global__ __launch_bounds__(1024) void partial_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                              int64_t nparts, int64_t width, int64_t stride) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < width && grp < nparts)
        s = ordered_sum(partial[grp * stride + i], partial + (grp + 16) * stride + i, 16 * stride, (int)((nparts - grp + 15) / 16) - 1);
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < width) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        out[i] = t;
    }
}

// LayerNorm backward: partial rows [dgamma(C) | dbeta(C)] per wave -> dgamma, dbeta (the same fixed tree as
// partial_reduce_kernel), written or ACCUMULATED in place (the caller's gradient buffer: no copies, no add kernels)
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int64_t nparts, int C, int accum) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < 2 * C && grp < nparts)
        s = ordered_sum(partial[(int64_t)grp * 2 * C + i], partial + (int64_t)(grp + 16) * 2 * C + i, (int64_t)32 * C,
                        (int)((nparts - grp + 15) / 16) - 1);
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        float* o = i < C ? dgamma + i : dbeta + (i - C);
        *o = accum ? *o + t : t;
    }
}

// column sums of a (R, C) matrix: stage 1 writes one partial row per block of rows.  A block is 64 columns x 4 row lanes
// (coalesced 256-B row segments, 4 rows in flight per column); the 4 row lanes are combined in order through LDS.
__global__ __launch_bounds__(256) void col_sum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t R,
                                                              int C, int rows_per_block) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int c = blockIdx.x * 64 + col;
    float s = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r0 + rows_per_block && r < R; r += 4) s += x[r * C + c];
    red[rl][col] = s;
    __syncthreads();
    if (rl == 0 && c < C) partial[(int64_t)blockIdx.y * C + c] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
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

// out[b][i] = x[b][i] * scale[b]: stochastic depth (timm DropPath: per-sample Bernoulli mask / keep_prob); its own backward
__global__ __launch_bounds__(256) void scale_samples_kernel(const f32x4* __restrict__ x, const float* __restrict__ scale,
                                                            f32x4* __restrict__ out, int64_t per4) {
    const float sc = scale[blockIdx.y];
    const f32x4* xb = x + (int64_t)blockIdx.y * per4;
    f32x4* ob = out + (int64_t)blockIdx.y * per4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per4; i += (int64_t)gridDim.x * 256) ob[i] = xb[i] * sc;
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

// PatchMerging's 2x2 gather (swin:357-361) as one permutation kernel, both directions: merged (B, H/2, W/2, 4C) with channel block
// q = 0..3 taken from pixel (2i + (q & 1), 2j + (q >> 1)) of x (B, H, W, C).  INV scatters a merged-layout tensor back (the
// backward of the gather: torch's indexing needs 4 x 2 slice_backward = 16 fill/copy launches for it).
template <bool INV>
__global__ __launch_bounds__(256) void patch_gather_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int H, int W, int C4,
                                                           int64_t total) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= total) return;
    const int c = (int)(m % C4);
    int64_t r = m / C4;
    const int q = (int)(r & 3);
    r >>= 2;
    const int W2 = W >> 1, H2 = H >> 1;
    const int j = (int)(r % W2);
    r /= W2;
    const int i = (int)(r % H2);
    const int64_t b = r / H2;
    const int64_t src = ((b * H + 2 * i + (q & 1)) * W + 2 * j + (q >> 1)) * C4 + c;
    if (INV) out[src] = in[m];
    else out[m] = in[src];
}

// The data-gradient convolution's weight: out (Cin, kh, kw, Cout) with out[ci][r][s][co] = w[co][kh-1-r][kw-1-s][ci] from the
// KRSC image w (Cout, kh, kw, Cin) -- per tap a strided 64x64-tiled transpose; grid.z = tap.  (torch: permute + flip + contiguous
// = two launches per convolution backward.)
__global__ __launch_bounds__(256) void conv_weight_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin,
                                                                int taps) {
    __shared__ float tile[64][65];
    const int tap = blockIdx.z, src_tap = taps - 1 - tap;          // (kh-1-r) kw + (kw-1-s) = taps - 1 - (r kw + s)
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;          // r: co, c: ci
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t in_pitch = (int64_t)taps * Cin, out_pitch = (int64_t)taps * Cout;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int co = r0 + ty + 4 * i, ci = c0 + tx;
        if (co < Cout && ci < Cin) tile[ty + 4 * i][tx] = w[co * in_pitch + (int64_t)src_tap * Cin + ci];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ci = c0 + ty + 4 * i, co = r0 + tx;
        if (ci < Cin && co < Cout) out[ci * out_pitch + (int64_t)tap * Cout + co] = tile[tx][ty + 4 * i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm (+ReLU) backward on NHWC (BaselineDecoder blocks, decoder.py:233-271).  With xhat = (z - mean_g) rstd_g,
// pre = xhat gamma_c + beta_c, g = dy * [pre > 0] (ReLU) :
//   dgamma_c = sum g xhat,  dbeta_c = sum g,  dz = rstd_g (g gamma_c - S1_g / n - xhat S2_g / n)
//   S1_g = sum_{c in g} gamma_c T1[c],  S2_g = sum_{c in g} gamma_c T2[c],  T1[c] = sum_pixels g,  T2[c] = sum_pixels g xhat
// pass 1 (grid (nsplit, B)): per-channel T1, T2 of its pixel range -> part[b][split][{T1,T2}][C]; pass 2: dz.
// mean / rstd come from the forward's gn_stats partial sums (same layout, same fixed-order sum as the forward apply).
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                             const float* __restrict__ stats, int nsplit_s,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ part, int64_t HW, int C, int G, int nsplit,
                                                             float eps, int relu) {
    __shared__ float red[256][8];
    __shared__ float gm[32], gr[32];
    const int split = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lpp = C >> 2, ppi = 256 / lpp, c4 = tid % lpp, pl = tid / lpp;
    const int cg = C / G;
    __shared__ double gred[256][2];
    gn_block_stats(stats, nsplit_s, G, b, (double)HW * (double)cg, eps, gm, gr, gred);
    const int64_t per = (HW + nsplit - 1) / nsplit;
    const int64_t p0 = split * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    f32x4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
    if (pl < ppi) {
        const int g = (4 * c4) / cg;
        const float mean = gm[g], rstd = gr[g];
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + 4 * c4), be = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
        const float* zb = z + (int64_t)b * HW * C + 4 * c4;
        const float* db = dy + (int64_t)b * HW * C + 4 * c4;
        for (int64_t p = p0 + pl; p < p1; p += ppi) {
            const f32x4 xh = (*reinterpret_cast<const f32x4*>(zb + p * C) - mean) * rstd;
            f32x4 d = *reinterpret_cast<const f32x4*>(db + p * C);
            if (relu) {                                       // act: 1 = ReLU mask, 2 = sigmoid'(pre) = s (1 - s)
                const f32x4 pre = xh * ga + be;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (relu == 1) { if (!(pre[e] > 0.f)) d[e] = 0.f; }
                    else { const float sg = 1.0f / (1.0f + __expf(-pre[e])); d[e] *= sg * (1.0f - sg); }
                }
            }
            t1 += d;
            t2 += d * xh;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[tid][e] = t1[e]; red[tid][4 + e] = t2[e]; }
    __syncthreads();
    if (tid < lpp) {                                          // fixed order over the pixel slots
        f32x4 a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
        for (int slot = 0; slot < ppi; ++slot)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a1[e] += red[slot * lpp + tid][e]; a2[e] += red[slot * lpp + tid][4 + e]; }
        float* o = part + ((int64_t)b * nsplit + split) * 2 * C;
        *reinterpret_cast<f32x4*>(o + 4 * tid) = a1;
        *reinterpret_cast<f32x4*>(o + C + 4 * tid) = a2;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, int nsplit_s,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ part, float* __restrict__ dz, int64_t HW, int C,
                                                           int G, int nsplit, float eps, int relu) {
    __shared__ float gm[32], gr[32], s1[32], s2[32];
    __shared__ float c1[1024], c2[1024];                 // per channel: gamma * (sum over the splits of T1, T2)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cg = C / G;
    // every block needs the per-group sums of the split partials: the channel sums run on all 256 threads (independent loads, splits
    // in order), then one thread per group adds its channels in order -- the same summation order as one thread per group walking
    // channels x splits, which at 100+ splits took longer than the streaming pass itself
    for (int c = tid; c < C; c += 256) {
        float t1 = 0.f, t2 = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) {
            const float* o = part + ((int64_t)b * nsplit + sp) * 2 * C;
            t1 += o[c]; t2 += o[C + c];
        }
        c1[c] = gamma[c] * t1; c2[c] = gamma[c] * t2;
    }
    __shared__ double red[256][2];
    gn_block_stats(stats, nsplit_s, G, b, (double)HW * (double)cg, eps, gm, gr, red);      // (ends with a barrier: c1 / c2 are complete too)
    if (tid < G) {
        float a1 = 0.f, a2 = 0.f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a1 += c1[c]; a2 += c2[c]; }
        const float n = (float)HW * (float)cg;
        s1[tid] = a1 / n; s2[tid] = a2 / n;
    }
    __syncthreads();
    const int lpp = C >> 2;
    const int64_t total = HW * lpp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % lpp);
        const int64_t p = i / lpp;
        const int g = (4 * c4) / cg;
        const float mean = gm[g], rstd = gr[g];
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + 4 * c4), be = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
        const int64_t off = ((int64_t)b * HW + p) * C + 4 * c4;
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(z + off) - mean) * rstd;
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + off);
        if (relu) {
            const f32x4 pre = xh * ga + be;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (relu == 1) { if (!(pre[e] > 0.f)) d[e] = 0.f; }
                else { const float sg = 1.0f / (1.0f + __expf(-pre[e])); d[e] *= sg * (1.0f - sg); }
            }
        }
        *reinterpret_cast<f32x4*>(dz + off) = (d * ga - s1[g] - xh * s2[g]) * rstd;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward of the bilinear x2 / x4 upsample (nn.Upsample(scale_factor=s, mode="bilinear"), NHWC): every input pixel gathers
// from the output pixels that read it in the forward, with the forward's own index/weight function (deterministic).
__device__ __forceinline__ void up_src(int o, int in, int out, int scale, int align, int& i0, int& i1, float& l0, float& l1) {
    float src;
    if (align) src = (out > 1) ? ((float)(in - 1) / (float)(out - 1)) * (float)o : 0.f;
    else { src = (1.0f / (float)scale) * ((float)o + 0.5f) - 0.5f; if (src < 0.f) src = 0.f; }
    i0 = (int)src;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C,
                                                           int scale, int align) {
    const int b = blockIdx.y;
    const int lpp = C >> 2;
    const int Ho = scale * H, Wo = scale * W;
    const int64_t total = (int64_t)H * W * lpp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % lpp);
        const int64_t p = i / lpp;
        const int x = (int)(p % W), y = (int)(p / W);
        f32x4 acc = {0, 0, 0, 0};
        // candidate outputs: source coordinate within (y-1, y+1) -> o in [scale*(y-1) - scale, scale*(y+1) + scale]
        for (int oy = scale * (y - 2); oy <= scale * (y + 2); ++oy) {
            if (oy < 0 || oy >= Ho) continue;
            int y0, y1; float ly0, ly1;
            up_src(oy, H, Ho, scale, align, y0, y1, ly0, ly1);
            const float wy = (y0 == y ? ly0 : 0.f) + (y1 == y ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = scale * (x - 2); ox <= scale * (x + 2); ++ox) {
                if (ox < 0 || ox >= Wo) continue;
                int x0, x1; float lx0, lx1;
                up_src(ox, W, Wo, scale, align, x0, x1, lx0, lx1);
                const float wx = (x0 == x ? lx0 : 0.f) + (x1 == x ? lx1 : 0.f);
                if (wx == 0.f) continue;
                acc += *reinterpret_cast<const f32x4*>(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + 4 * c4) * (wy * wx);
            }
        }
        *reinterpret_cast<f32x4*>(dx + (((int64_t)b * H + y) * W + x) * C + 4 * c4) = acc;
    }
}

}  // namespace

// waves of the launch (four per block, one partial row per block).  A wave walks its rows RU at a time and every step is three
// dependent wave reductions (~2 us): with 8 rows per wave the B = 2 shapes of config 5 (a few thousand rows) ran 60 blocks for
// 19 us; two steps per wave fill the chip and finish in a third of that.  Capped where the rows alone fill the chip many times over.
static int64_t ln_bwd_waves(int64_t rows, int C) {
    const int ru = C <= 256 ? 4 : C <= 512 ? 2 : 1;
    int64_t w = (rows + 2 * ru - 1) / (2 * ru);
    if (w > 4096) w = 4096;
    if (w < 4) w = 4;
    return (w + 3) / 4 * 4;
}

extern "C" int64_t mumpy_layernorm_bwd_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return (ln_bwd_waves(rows, C) / 4 + 1) * 2 * C * (int64_t)sizeof(float);      // one partial [dgamma | dbeta] row per block (+ one spare)
}

extern "C" int mumpy_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* dx_add, float* dx,
                                   float* dgamma, float* dbeta, void* workspace, int64_t workspace_bytes, int64_t rows, int C,
                                   float eps, int accumulate, void* stream) {
    if (rows == 0) return 0;
    MUMPY_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && workspace, MUMPY_ENULL, "layernorm_bwd: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(gamma) && aligned16(dy) && aligned16(dx) && aligned16(dx_add) && aligned16(workspace),
                  MUMPY_EALIGN, "layernorm_bwd: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(accumulate == 0 || accumulate == 1, MUMPY_EINVAL, "layernorm_bwd: accumulate must be 0 or 1");
    MUMPY_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXC4, MUMPY_EINVAL, "layernorm_bwd: unsupported C=%d", C);
    MUMPY_REQUIRE(workspace_bytes >= mumpy_layernorm_bwd_workspace_bytes(rows, C), MUMPY_EINVAL, "layernorm_bwd: workspace too small");
    const int64_t waves = ln_bwd_waves(rows, C);
    const int rpw = (int)((rows + waves - 1) / waves);
    float* partial = static_cast<float*>(workspace);
#define MUMPY_LN_BWD(NC4_, RU_)                                                                                     \
    hipLaunchKernelGGL((ln_bwd_kernel<NC4_, RU_>), dim3((unsigned)(waves / 4)), dim3(256), (size_t)2 * C * sizeof(float), as_stream(stream), \
                       x, gamma, dy, dx_add, dx, partial, rows, C, eps, rpw)
    if (C <= 256) MUMPY_LN_BWD(1, 4);
    else if (C <= 512) MUMPY_LN_BWD(2, 2);
    else if (C <= 1024) MUMPY_LN_BWD(4, 1);
    else MUMPY_LN_BWD(8, 1);
#undef MUMPY_LN_BWD
    MUMPY_CHECK_LAUNCH("layernorm_bwd");
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(1024), 0, as_stream(stream), partial, dgamma,
                       dbeta, waves / 4, C, accumulate);
    MUMPY_CHECK_LAUNCH("layernorm_bwd(reduce)");
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

extern "C" int mumpy_patch_gather_fwd(const float* in, float* out, int64_t B, int H, int W, int C, int inverse, void* stream) {
    if (B == 0) return 0;
    MUMPY_REQUIRE(in && out, MUMPY_ENULL, "patch_gather: null pointer");
    MUMPY_REQUIRE(aligned16(in) && aligned16(out), MUMPY_EALIGN, "patch_gather: buffers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, MUMPY_EINVAL,
                  "patch_gather: need even H, W and C %% 4 == 0 (got %d x %d x %d)", H, W, C);
    const int64_t total = B * H * W * (C / 4);
    MUMPY_REQUIRE((total + 255) / 256 < (1ll << 31), MUMPY_ERANGE, "patch_gather: too many elements");
    const dim3 grid((unsigned)((total + 255) / 256));
    if (inverse)
        hipLaunchKernelGGL(patch_gather_kernel<true>, grid, dim3(256), 0, as_stream(stream), (const f32x4*)in, (f32x4*)out, H, W, C / 4, total);
    else
        hipLaunchKernelGGL(patch_gather_kernel<false>, grid, dim3(256), 0, as_stream(stream), (const f32x4*)in, (f32x4*)out, H, W, C / 4, total);
    MUMPY_CHECK_LAUNCH("patch_gather");
    return 0;
}

extern "C" int mumpy_conv_weight_dgrad_fwd(const float* w_krsc, float* out, int Cout, int Cin, int kh, int kw, void* stream) {
    MUMPY_REQUIRE(w_krsc && out, MUMPY_ENULL, "conv_weight_dgrad: null pointer");
    MUMPY_REQUIRE(Cout > 0 && Cin > 0 && kh > 0 && kw > 0 && kh * kw < 65536 && (Cout + 63) / 64 < 65536, MUMPY_EINVAL, "conv_weight_dgrad: bad shape");
    hipLaunchKernelGGL(conv_weight_dgrad_kernel, dim3((unsigned)((Cin + 63) / 64), (unsigned)((Cout + 63) / 64), (unsigned)(kh * kw)), dim3(256), 0,
                       as_stream(stream), w_krsc, out, Cout, Cin, kh * kw);
    MUMPY_CHECK_LAUNCH("conv_weight_dgrad");
    return 0;
}

static int64_t col_sum_blocks(int64_t R) {             // row blocks: >= 64 rows each, enough of them to fill the chip
    int64_t b = (R + 63) / 64;
    if (b > 2048) b = 2048;
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
    hipLaunchKernelGGL(col_sum_partial_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)nb), dim3(256), 0, as_stream(stream), x,
                       partial, R, C, rpb);
    MUMPY_CHECK_LAUNCH("col_sum(partial)");
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, as_stream(stream), partial, out, nb,
                       (int64_t)C, (int64_t)C);
    MUMPY_CHECK_LAUNCH("col_sum(reduce)");
    return 0;
}

static int gn_bwd_splits(int64_t HW, int C) {       // 64 KB of the image per workgroup (as ops.gn_stats): at B = 2 the 256 KB rule
    int64_t s = (HW * C) / 16384;                   // left 24-128 workgroups, each walking 64 dependent loads (37 us per launch)
    if (s > 256) s = 256;
    return s < 1 ? 1 : (int)s;
}

extern "C" int64_t mumpy_gn_bwd_workspace_bytes(int B, int64_t HW, int C) {
    if (B <= 0 || HW <= 0 || C <= 0) return 0;
    return (int64_t)B * gn_bwd_splits(HW, C) * 2 * C * (int64_t)sizeof(float);
}

extern "C" int mumpy_gn_bwd_nhwc(const float* z, const float* stats_partial, int nsplit_stats, const float* gamma,
                                 const float* beta, const float* dy, float* dz, float* dgamma, float* dbeta, void* workspace,
                                 int64_t workspace_bytes, int B, int64_t HW, int C, int G, float eps, int relu, void* stream) {
    MUMPY_REQUIRE(z && stats_partial && gamma && beta && dy && dz && dgamma && dbeta && workspace, MUMPY_ENULL, "gn_bwd: null pointer");
    MUMPY_REQUIRE(aligned16(z) && aligned16(dy) && aligned16(dz) && aligned16(gamma) && aligned16(beta) && aligned16(workspace),
                  MUMPY_EALIGN, "gn_bwd: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && B <= 65535 && HW > 0 && nsplit_stats > 0, MUMPY_EINVAL, "gn_bwd: bad sizes");
    MUMPY_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && G > 0 && G <= 32 && C % G == 0 && (C / G) % 4 == 0, MUMPY_EINVAL,
                  "gn_bwd: unsupported C=%d G=%d", C, G);
    MUMPY_REQUIRE(workspace_bytes >= mumpy_gn_bwd_workspace_bytes(B, HW, C), MUMPY_EINVAL, "gn_bwd: workspace too small");
    const int accumulate = (relu & MUMPY_GN_ACCUMULATE) ? 1 : 0;
    relu &= ~MUMPY_GN_ACCUMULATE;
    MUMPY_REQUIRE(relu >= 0 && relu <= 2, MUMPY_EINVAL, "gn_bwd: activation code %d", relu);
    const int ns = gn_bwd_splits(HW, C);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(ns, B), dim3(256), 0, as_stream(stream), z, dy, stats_partial, nsplit_stats, gamma,
                       beta, part, HW, C, G, ns, eps, relu);
    MUMPY_CHECK_LAUNCH("gn_bwd(partial)");
    int64_t grid = (HW * (C / 4) + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((unsigned)grid, B), dim3(256), 0, as_stream(stream), z, dy, stats_partial,
                       nsplit_stats, gamma, beta, part, dz, HW, C, G, ns, eps, relu);
    MUMPY_CHECK_LAUNCH("gn_bwd(apply)");
    // dbeta[c] = sum over (b, split) of T1, dgamma[c] of T2: partial rows are [T1(C) | T2(C)] -- the LayerNorm reduce's layout with
    // the roles swapped; ONE launch for both, written or accumulated in place
    hipLaunchKernelGGL(ln_param_reduce_kernel, dim3((unsigned)((2 * C + 63) / 64)), dim3(1024), 0, as_stream(stream), part, dbeta, dgamma,
                       (int64_t)B * ns, C, accumulate);
    MUMPY_CHECK_LAUNCH("gn_bwd(reduce)");
    return 0;
}

extern "C" int mumpy_upsample_bwd_nhwc(const float* dy, float* dx, int B, int H, int W, int C, int scale, int align_corners,
                                       void* stream) {
    MUMPY_REQUIRE(dy && dx, MUMPY_ENULL, "upsample_bwd: null pointer");
    MUMPY_REQUIRE(aligned16(dy) && aligned16(dx), MUMPY_EALIGN, "upsample_bwd: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (scale == 2 || scale == 4), MUMPY_EINVAL,
                  "upsample_bwd: bad shape or scale %d", scale);
    int64_t grid = ((int64_t)H * W * (C / 4) + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)grid, B), dim3(256), 0, as_stream(stream), dy, dx, H, W, C, scale,
                       align_corners);
    MUMPY_CHECK_LAUNCH("upsample_bwd");
    return 0;
}

extern "C" int mumpy_scale_samples_fwd(const float* x, const float* scale, float* out, int B, int64_t per_sample, void* stream) {
    if (B == 0 || per_sample == 0) return 0;
    MUMPY_REQUIRE(x && scale && out, MUMPY_ENULL, "scale_samples: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(out) && per_sample % 4 == 0, MUMPY_EALIGN,
                  "scale_samples: need 16-byte aligned buffers and per_sample %% 4 == 0");
    MUMPY_REQUIRE(B > 0 && B <= 65535 && per_sample > 0, MUMPY_EINVAL, "scale_samples: bad shape");
    int64_t grid = (per_sample / 4 + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(scale_samples_kernel, dim3((unsigned)grid, B), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const f32x4*>(x), scale, reinterpret_cast<f32x4*>(out), per_sample / 4);
    MUMPY_CHECK_LAUNCH("scale_samples");
    return 0;
}
