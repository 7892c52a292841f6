// deform.hip — the non-GEMM parts of SwinDAttention (deform:324-405) and of its caller's residual wiring
// (mTVE:138, 280-286): offset network, bilinear sampling, and the "scrambled" combine.
// The attention + r-tuple aggregation lives in window_attention.hip (win_attn_cross_kernel); q/k/v/out projections
// and `pre` are mumpy_linear_fwd.
#include <stdlib.h>
#include "common.h"
using namespace mumpy;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// offsets: one block per (q window, group).  depthwise 5x5 (pad 2) -> LayerNorm over the group's channels -> GELU ->
// 1x1 conv to (dy,dx) -> tanh * (1/7) * 2 + reference point (deform:334-349, 311-322).
// LDS: the window's group tile [49][Cg] and the conv result [49][Cg]; lanes run over channels (conflict-free).
template <int Cg>
__global__ __launch_bounds__(256) void deform_offsets_kernel(const float* __restrict__ q, const float* __restrict__ dw_w,
                                                             const float* __restrict__ dw_b,
                                                             const float* __restrict__ ln_g,
                                                             const float* __restrict__ ln_b,
                                                             const float* __restrict__ pw_w, float* __restrict__ pos,
                                                             int H, int W, int C, int nWx, int nWf) {
    constexpr int CP = Cg + 4;      // conv row pitch: 16 pixels x 4 lanes read 64 distinct banks in the LayerNorm phase
    __shared__ __attribute__((aligned(16))) float sm[WT * Cg + WT * CP + 25 * Cg + 4 * Cg];   // static: up to 131 KB at Cg = 256
    float* tile = sm;                       // [49][Cg]
    float* conv = sm + WT * Cg;             // [49][CP]
    float* wsm = conv + WT * CP;            // [Cg][25] depthwise taps
    float* par = wsm + 25 * Cg;             // LayerNorm gamma | beta | 1x1 weights (dy) | (dx)
    const int bw = blockIdx.x, g = blockIdx.y;
    const int b = bw / nWf, n = bw - b * nWf;
    const int wy = n / nWx, wx = n - wy * nWx;
    const int tid = threadIdx.x;
    const int64_t L = (int64_t)H * W;
    const float* qb = q + (int64_t)b * L * C + g * Cg;
    const int cg4 = Cg >> 2;
    // staging: every 16-byte load of a thread is issued before its first LDS write (at the late stages a launch is 24-96 workgroups
    // and its duration is one workgroup's dependent chain: a load per loop step would add a memory latency per step)
    {
        constexpr int NT = WT * (Cg / 4), PT = (NT + 255) / 256;
        f32x4 v[PT];
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int idx = tid + 256 * u;
            if (idx < NT) {
                const int p = idx / cg4, c4 = idx - p * cg4;
                v[u] = *reinterpret_cast<const f32x4*>(qb + (int64_t)window_token(wy, wx, p, H, W, 0) * C + 4 * c4);
            }
        }
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int idx = tid + 256 * u;
            if (idx < NT) { const int p = idx / cg4, c4 = idx - p * cg4; *reinterpret_cast<f32x4*>(tile + p * Cg + 4 * c4) = v[u]; }
        }
        // the group's 25 taps per channel (shared by the groups) and, below, the per-channel parameters of the LayerNorm / 1x1 stage
        constexpr int NW = 25 * Cg / 4, PW = (NW + 255) / 256;
        f32x4 t[PW];
#pragma unroll
        for (int u = 0; u < PW; ++u) { const int idx = tid + 256 * u; if (idx < NW) t[u] = reinterpret_cast<const f32x4*>(dw_w)[idx]; }
#pragma unroll
        for (int u = 0; u < PW; ++u) { const int idx = tid + 256 * u; if (idx < NW) reinterpret_cast<f32x4*>(wsm)[idx] = t[u]; }
    }
    for (int idx = tid; idx < Cg; idx += 256) {
        par[idx] = ln_g[idx];
        par[Cg + idx] = ln_b[idx];
        par[2 * Cg + idx] = pw_w[idx];
        par[3 * Cg + idx] = pw_w[Cg + idx];
    }
    // 256 % Cg == 0 (Cg in {32,64,128,256}): a thread keeps one channel for all its pixels
    const int cc = tid % Cg;
    const float breg = dw_b[cc];
    __syncthreads();
    float wreg[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) wreg[i] = wsm[cc * 25 + i];          // 25 is odd: lanes hit distinct banks
    for (int p = tid / Cg; p < WT; p += 256 / Cg) {
        const int py = p / WS, px = p - py * WS;
        // branch-free taps: out-of-window taps read the centre pixel and contribute fmaf(0, w, acc) == acc, so the 25 LDS
        // reads of a pixel are independent and issue back to back
        float acc = breg;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const int yy = py + dy - 2;
            const bool vy = (unsigned)yy < (unsigned)WS;
            const int yc = vy ? yy : py;
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) {
                const int xx = px + dx - 2;
                const bool ok = vy && (unsigned)xx < (unsigned)WS;
                const int xc = ok ? xx : px;
                const float t = tile[(yc * WS + xc) * Cg + cc];
                acc = fmaf(ok ? t : 0.f, wreg[dy * 5 + dx], acc);
            }
        }
        conv[p * CP + cc] = acc;
    }
    __syncthreads();
    // LayerNorm over the group's channels -> GELU -> 1x1 conv to (dy, dx): FOUR lanes per pixel, each walking a quarter of the
    // channels (lane qd takes channels qd, qd + 4, ...), partial sums combined with two quad shuffles (DPP).  The first version
    // gave a whole wave to a pixel and reduced across 64 lanes four times per pixel: 13 pixels x 24 cross-lane steps per wave
    // were ~2/3 of the kernel's 24-44 us.
    const int p = tid >> 2, qd = tid & 3;
    if (p < WT) {
        const float* cv = conv + p * CP;
        float s = 0.f;
#pragma unroll 8
        for (int ch = qd; ch < Cg; ch += 4) s += cv[ch];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mean = s / (float)Cg;
        float qq = 0.f;
#pragma unroll 8
        for (int ch = qd; ch < Cg; ch += 4) { const float d = cv[ch] - mean; qq = fmaf(d, d, qq); }
        qq += __shfl_xor(qq, 1);
        qq += __shfl_xor(qq, 2);
        const float rstd = rsqrtf(qq / (float)Cg + 1e-5f);
        float oy = 0.f, ox = 0.f;
#pragma unroll 4
        for (int ch = qd; ch < Cg; ch += 4) {
            const float a = gelu_erf((cv[ch] - mean) * rstd * par[ch] + par[Cg + ch]);
            oy = fmaf(a, par[2 * Cg + ch], oy);
            ox = fmaf(a, par[3 * Cg + ch], ox);
        }
        oy += __shfl_xor(oy, 1);
        oy += __shfl_xor(oy, 2);
        ox += __shfl_xor(ox, 1);
        ox += __shfl_xor(ox, 2);
        if (qd == 0) {
            const int py = p / WS, px = p - py * WS;
            const float ry = ((0.5f + (float)py) / 7.0f) * 2.0f - 1.0f;
            const float rx = ((0.5f + (float)px) / 7.0f) * 2.0f - 1.0f;
            const float rng = 1.0f / 7.0f;
            float* o = pos + (((int64_t)bw * 3 + g) * WT + p) * 2;
            o[0] = tanhf(oy) * rng * 2.0f + ry;
            o[1] = tanhf(ox) * rng * 2.0f + rx;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sampling: one block per kv window, a thread per (point, 4 channels); lanes run over channels, so each of the four
// corner reads is a contiguous row segment.  The 49xC window tile (<= 150 KB) is served by L1/L2 after its first touch:
// HBM sees x2 once and the sampled map once.  Window decode is scalar (per block) and the channel count is a template
// constant, so the per-thread index math is two constant divisions (the first version spent its time in 64-bit
// runtime divisions: 3.6 TB/s).
template <int C>
__global__ __launch_bounds__(256) void deform_sample_kernel(const float* __restrict__ x2, const float* __restrict__ pos,
                                                            float* __restrict__ out, int Hs2, int W, int nWx, int nW2,
                                                            int nq) {
    constexpr int C4N = C / 4, CG = C / 3;
    const int b2 = blockIdx.x;
    const int b = b2 / nW2, n = b2 - b * nW2;
    const int wy = n / nWx, wx = n - wy * nWx;
    const float* base = x2 + ((int64_t)b * Hs2 * W + (int64_t)wy * WS * W + wx * WS) * C;   // window's top-left token
    const float* pw = pos + (int64_t)(b2 % nq) * 3 * WT * 2;
    float* ob = out + (int64_t)b2 * WT * C;
    for (int idx = threadIdx.x; idx < WT * C4N; idx += 256) {
        const int p = idx / C4N, c4 = idx - p * C4N;
        const int g = (4 * c4) / CG;
        const float gy = pw[(g * WT + p) * 2], gx = pw[(g * WT + p) * 2 + 1];
        // grid_sample, align_corners=True: pixel = (g + 1) / 2 * (size - 1)
        const float iy = ((gy + 1.0f) * 0.5f) * 6.0f;
        const float ix = ((gx + 1.0f) * 0.5f) * 6.0f;
        const float y0f = floorf(iy), x0f = floorf(ix);
        const int y0 = (int)y0f, x0 = (int)x0f;
        const float wnw = (x0f + 1.0f - ix) * (y0f + 1.0f - iy);
        const float wne = (ix - x0f) * (y0f + 1.0f - iy);
        const float wsw = (x0f + 1.0f - ix) * (iy - y0f);
        const float wse = (ix - x0f) * (iy - y0f);
        const float* src = base + 4 * c4;
        auto corner = [&](int yy, int xx) -> f32x4 {
            if (yy < 0 || yy >= WS || xx < 0 || xx >= WS) return f32x4{0, 0, 0, 0};   // zeros padding
            return *reinterpret_cast<const f32x4*>(src + (yy * W + xx) * C);
        };
        f32x4 r = corner(y0, x0) * wnw;
        r += corner(y0, x0 + 1) * wne;
        r += corner(y0 + 1, x0) * wsw;
        r += corner(y0 + 1, x0 + 1) * wse;
        *reinterpret_cast<f32x4*>(ob + idx * 4) = r;
    }
}

// LDS-staged form (the one the entry point launches): one block per (kv window, 96-channel slab).  The slab of the window
// tile, 49 tokens x 96 channels = 18.4 KB, is read ONCE with coalesced 16-byte loads (7 row segments of 7 x 384 B) into LDS and
// all four corner reads of every point come from there, so HBM sees exactly the algorithmic bytes: the direct form above
// fetched ~1.5x the tile (PMC FETCH_SIZE: corner rows touched a second time after leaving L2) -- profiles/r01_pmc_traffic.md.
// Eight blocks per CU fit (147 KB of LDS), lanes run over channels so a point's corner read is one contiguous 384-B LDS row.
template <int C>
__global__ __launch_bounds__(256) void deform_sample_lds_kernel(const float* __restrict__ x2, const float* __restrict__ pos,
                                                                float* __restrict__ out, int Hs2, int W, int nWx, int nW2,
                                                                int nq) {
    constexpr int SL = 96, S4 = SL / 4, CG = C / 3;
    __shared__ __attribute__((aligned(16))) float tile[WT * SL];
    const int b2 = blockIdx.x, slab = blockIdx.y;
    const int b = b2 / nW2, n = b2 - b * nW2;
    const int wy = n / nWx, wx = n - wy * nWx;
    const float* base = x2 + ((int64_t)b * Hs2 * W + (int64_t)wy * WS * W + wx * WS) * C + slab * SL;
    for (int idx = threadIdx.x; idx < WT * S4; idx += 256) {
        const int tok = idx / S4, j = idx - tok * S4;
        const int ty = tok / WS, tx = tok - ty * WS;
        *reinterpret_cast<f32x4*>(&tile[tok * SL + 4 * j]) = *reinterpret_cast<const f32x4*>(base + (ty * W + tx) * C + 4 * j);
    }
    __syncthreads();
    const float* pw = pos + (int64_t)(b2 % nq) * 3 * WT * 2;
    float* ob = out + (int64_t)b2 * WT * C + slab * SL;
    for (int idx = threadIdx.x; idx < WT * S4; idx += 256) {
        const int p = idx / S4, c4 = idx - p * S4;
        const int g = (slab * SL + 4 * c4) / CG;
        const float gy = pw[(g * WT + p) * 2], gx = pw[(g * WT + p) * 2 + 1];
        const float iy = ((gy + 1.0f) * 0.5f) * 6.0f;        // grid_sample, align_corners=True: pixel = (g + 1) / 2 * (size - 1)
        const float ix = ((gx + 1.0f) * 0.5f) * 6.0f;
        const float y0f = floorf(iy), x0f = floorf(ix);
        const int y0 = (int)y0f, x0 = (int)x0f;
        const float wnw = (x0f + 1.0f - ix) * (y0f + 1.0f - iy);
        const float wne = (ix - x0f) * (y0f + 1.0f - iy);
        const float wsw = (x0f + 1.0f - ix) * (iy - y0f);
        const float wse = (ix - x0f) * (iy - y0f);
        auto corner = [&](int yy, int xx) -> f32x4 {
            if (yy < 0 || yy >= WS || xx < 0 || xx >= WS) return f32x4{0, 0, 0, 0};   // zeros padding
            return *reinterpret_cast<const f32x4*>(&tile[(yy * WS + xx) * SL + 4 * c4]);
        };
        f32x4 r = corner(y0, x0) * wnw;                       // same order of operations as the direct form: bitwise equal
        r += corner(y0, x0 + 1) * wne;
        r += corner(y0 + 1, x0) * wsw;
        r += corner(y0 + 1, x0 + 1) * wse;
        *reinterpret_cast<f32x4*>(ob + p * C + 4 * c4) = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// combine: out[b, n*49+p, c] = x1[b, n*49+p, c] + x1[b, raster(n,p), c] + Yt[bw]^T flat[p*C + c]
// One block per (window, 32-channel slab of Yt): the slab [49][32] is transposed through LDS (row stride 33) and
// lands on 1568 CONTIGUOUS output floats, because flat index f = c'*49 + p' of the (C,49) image is output (p,c) with
// p*C + c = f (deform:403 reshapes (B,C,H,W) to (B,HW,C) without a permute).
__global__ __launch_bounds__(256) void deform_combine_kernel(const float* __restrict__ x1, const float* __restrict__ Yt,
                                                             float* __restrict__ out, int H, int W, int C, int nWx,
                                                             int nWf) {
    __shared__ float tile[WT * 33];
    const int bw = blockIdx.x, cb = blockIdx.y;
    const int b = bw / nWf, n = bw - b * nWf;
    const int wy = n / nWx, wx = n - wy * nWx;
    const int tid = threadIdx.x;
    const float* ys = Yt + (int64_t)bw * WT * C + cb * 32;
    for (int idx = tid; idx < WT * 32; idx += 256) {
        const int p = idx >> 5, cc = idx & 31;
        tile[p * 33 + cc] = ys[(int64_t)p * C + cc];
    }
    __syncthreads();
    const int64_t img = (int64_t)b * H * W;
    const int64_t win_base = (img + (int64_t)n * WT) * C;      // window-major run of 49*C floats inside batch b
    const int f0 = cb * 32 * WT;
    for (int idx = tid; idx < WT * 32; idx += 256) {
        const int f = f0 + idx;
        const int cl = idx / WT, pp = idx - cl * WT;           // source channel (local) and source pixel of Yt
        const int po = f / C, co = f - po * C;                 // destination token-in-window and channel
        const int tok = window_token(wy, wx, po, H, W, 0);
        out[win_base + f] = x1[win_base + f] + x1[(img + tok) * C + co] + tile[pp * 33 + cl];
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ o, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        reinterpret_cast<f32x4*>(o)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

}  // namespace

extern "C" int mumpy_deform_offsets_fwd(const float* q, const float* dw_w, const float* dw_b, const float* ln_g,
                                        const float* ln_b, const float* pw_w, float* pos, int B, int H, int W, int C,
                                        void* stream) {
    MUMPY_REQUIRE(q && dw_w && dw_b && ln_g && ln_b && pw_w && pos, MUMPY_ENULL, "deform_offsets: null pointer");
    MUMPY_REQUIRE(aligned16(q) && aligned16(dw_w), MUMPY_EALIGN, "deform_offsets: q and the depthwise weights must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H % WS == 0 && W % WS == 0 && H > 0 && W > 0, MUMPY_EINVAL, "deform_offsets: bad grid (%d,%d)", H, W);
    const int Cg = C / 3;
    MUMPY_REQUIRE(C % 3 == 0 && (Cg == 32 || Cg == 64 || Cg == 128 || Cg == 256), MUMPY_EINVAL,
                  "deform_offsets: group width C/3=%d must be 32, 64, 128 or 256", Cg);
    const int nWx = W / WS, nWf = (H / WS) * nWx;
#define MUMPY_OFFS(CG)                                                                                              \
    hipLaunchKernelGGL(deform_offsets_kernel<CG>, dim3(B * nWf, 3), dim3(256), 0, as_stream(stream), q, dw_w, dw_b, ln_g, \
                       ln_b, pw_w, pos, H, W, C, nWx, nWf)
    switch (Cg) {
        case 32: MUMPY_OFFS(32); break;
        case 64: MUMPY_OFFS(64); break;
        case 128: MUMPY_OFFS(128); break;
        default: MUMPY_OFFS(256); break;
    }
#undef MUMPY_OFFS
    MUMPY_CHECK_LAUNCH("deform_offsets");
    return 0;
}

extern "C" int mumpy_deform_sample_fwd(const float* x2, const float* pos, float* out, int B, int Hs2, int W, int C,
                                       int nq, void* stream) {
    MUMPY_REQUIRE(x2 && pos && out, MUMPY_ENULL, "deform_sample: null pointer");
    MUMPY_REQUIRE(aligned16(x2) && aligned16(out), MUMPY_EALIGN, "deform_sample: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && Hs2 > 0 && W > 0 && Hs2 % WS == 0 && W % WS == 0 && nq > 0, MUMPY_EINVAL,
                  "deform_sample: bad grid (%d,%d) / nq=%d", Hs2, W, nq);
    MUMPY_REQUIRE(C == 96 || C == 192 || C == 384 || C == 768, MUMPY_EINVAL,
                  "deform_sample: C=%d is not one of the encoder widths 96/192/384/768", C);
    const int nWx = W / WS, nW2 = (Hs2 / WS) * nWx;
    const int64_t nwin = (int64_t)B * nW2;
    MUMPY_REQUIRE(nwin < (1ll << 31), MUMPY_ERANGE, "deform_sample: too many windows");
    static const bool direct = tune_int("MUMPY_SAMPLE_DIRECT", 0) != 0;   // A/B hook
#define MUMPY_SAMPLE(C_)                                                                                             \
    if (direct)                                                                                                      \
        hipLaunchKernelGGL(deform_sample_kernel<C_>, dim3((unsigned)nwin), dim3(256), 0, as_stream(stream), x2, pos, out,  \
                           Hs2, W, nWx, nW2, nq);                                                                    \
    else                                                                                                             \
        hipLaunchKernelGGL(deform_sample_lds_kernel<C_>, dim3((unsigned)nwin, C_ / 96), dim3(256), 0, as_stream(stream), x2, \
                           pos, out, Hs2, W, nWx, nW2, nq)
    switch (C) {
        case 96: MUMPY_SAMPLE(96); break;
        case 192: MUMPY_SAMPLE(192); break;
        case 384: MUMPY_SAMPLE(384); break;
        default: MUMPY_SAMPLE(768); break;
    }
#undef MUMPY_SAMPLE
    MUMPY_CHECK_LAUNCH("deform_sample");
    return 0;
}

extern "C" int mumpy_deform_combine_fwd(const float* x1, const float* Yt, float* out, int B, int H, int W, int C,
                                        void* stream) {
    MUMPY_REQUIRE(x1 && Yt && out, MUMPY_ENULL, "deform_combine: null pointer");
    MUMPY_REQUIRE(x1 != out, MUMPY_EINVAL, "deform_combine: out must not alias x1");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && H % WS == 0 && W % WS == 0 && C % 32 == 0, MUMPY_EINVAL,
                  "deform_combine: bad shape (%d,%d,%d)", H, W, C);
    const int nWx = W / WS, nWf = (H / WS) * nWx;
    hipLaunchKernelGGL(deform_combine_kernel, dim3(B * nWf, C / 32), dim3(256), 0, as_stream(stream), x1, Yt, out, H, W,
                       C, nWx, nWf);
    MUMPY_CHECK_LAUNCH("deform_combine");
    return 0;
}

extern "C" int mumpy_add_fwd(const float* a, const float* b, float* out, int64_t n, void* stream) {
    MUMPY_REQUIRE(a && b && out, MUMPY_ENULL, "add: null pointer");
    MUMPY_REQUIRE(aligned16(a) && aligned16(b) && aligned16(out) && n % 4 == 0 && n >= 0, MUMPY_EALIGN,
                  "add: need 16-byte aligned pointers and n %% 4 == 0");
    if (n == 0) return 0;
    int64_t grid = (n / 4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a, b, out, n / 4);
    MUMPY_CHECK_LAUNCH("add");
    return 0;
}
