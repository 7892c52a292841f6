// decoder.hip — the non-convolution work of the multi-pyramid decoder (decoder.py:67-225) as two NHWC kernels.
//
// Everything between two convolutions of the decoder is a chain of per-pixel / per-channel ops:
//   GroupNorm -> ReLU | Sigmoid -> [PixelShuffle(2) + AvgPool(2) == mean over 4 adjacent channels] ->
//   bilinear x2 / x4 (align_corners True in decoder_2..5, False in SEB / upsample2 / upsample4) ->
//   [ + a*b  |  * a ]   (the "+ gcn*freq", "* freq0", "x1 * upsample(conv(x2))" wiring of decoder.py:204-221, 14)
// The reference materialises every link (torch's upsample alone was 4.7 ms of a 47 ms forward).  Here:
//   mumpy_gn_stats_nhwc_fwd         per (sample, group) partial sums, one coalesced pass over the conv output
//   mumpy_gn_apply_resample_nhwc_fwd  normalise + activation + channel-mean + resample + epilogue, written straight into the
//                                     consumer's (possibly channel-concatenated) NHWC buffer
// DAP commutes with the upsample (both linear, DAP mixes channels only): decoder_5's (B,128,224,224) and DAP's
// (B,32,448,448) intermediates are never formed — the kernel averages 4 channels per tap and writes (B,32,224,224).
// NHWC: lanes run over channels (16-B accesses), a tap is one contiguous channel vector.  HBM-bound streaming.
#include "common.h"
using namespace mumpy;

namespace {

// ---------------------------------------------------------------------------------------------------------------
// x (B, HW, C) NHWC.  grid (nsplit, B).  partial[((b*nsplit + s)*G + g)*2 + {0,1}] = {sum, sum of squares}
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                       int64_t HW, int C, int G, int nsplit) {
    __shared__ float red[256][2];
    const int split = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lpp = C >> 2;                 // lanes per pixel
    const int ppi = 256 / lpp;              // pixels per iteration
    const int c4 = tid % lpp, pl = tid / lpp;
    const int64_t per = (HW + nsplit - 1) / nsplit;
    const int64_t p0 = split * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    const float* xb = x + (int64_t)b * HW * C + 4 * c4;
    float s = 0.f, q = 0.f;
    if (pl < ppi)
        for (int64_t p = p0 + pl; p < p1; p += ppi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + p * C);
            s += (v.x + v.y) + (v.z + v.w);
            q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
    red[tid][0] = s;
    red[tid][1] = q;
    __syncthreads();
    if (tid < G) {                          // fixed summation order: reproducible
        const int cg4 = (C / G) >> 2;       // lanes per group inside a pixel slot
        float ss = 0.f, qq = 0.f;
        for (int slot = 0; slot < ppi; ++slot)
            for (int l = 0; l < cg4; ++l) {
                const int t = slot * lpp + tid * cg4 + l;
                ss += red[t][0];
                qq += red[t][1];
            }
        float* o = partial + (((int64_t)b * nsplit + split) * G + tid) * 2;
        o[0] = ss;
        o[1] = qq;
    }
}

struct ApplyArgs {
    const float* x;        // (B,H,W,C)
    const float* partial;  // gn partial sums or null (identity pre-op)
    const float* gamma;
    const float* beta;
    const float* ep_a;     // epilogue operands, dense (B,Ho,Wo,Cout) or null
    const float* ep_b;
    float* out;            // (B,Ho,Wo,out_ctot), written at channel offset out_coff
    int H, W, C, G, nsplit, act, mean4, scale, align, ep_mode, out_ctot, out_coff;
    float eps;
};

__device__ __forceinline__ float activate(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.0f / (1.0f + __expf(-v));
    return v;
}

__device__ __forceinline__ void src_index(int o, int in, int out, int scale, int align, int& i0, int& i1, float& l0, float& l1) {
    if (scale == 1) { i0 = i1 = o; l0 = 1.f; l1 = 0.f; return; }
    float src;
    if (align) {
        src = (out > 1) ? ((float)(in - 1) / (float)(out - 1)) * (float)o : 0.f;
    } else {
        src = (1.0f / (float)scale) * ((float)o + 0.5f) - 0.5f;      // area_pixel_compute_source_index
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// grid (blocks, B); a thread produces 4 output channels of one output pixel
__global__ __launch_bounds__(256) void gn_apply_resample_kernel(ApplyArgs a) {
    __shared__ float sc[256], sh[256];
    __shared__ float gm[32], gr[32];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int C = a.C;
    if (a.partial) {
        if (tid < a.G) {
            double s = 0.0, q = 0.0;
            const float* p = a.partial + ((int64_t)b * a.nsplit * a.G + tid) * 2;
            for (int k = 0; k < a.nsplit; ++k) {
                s += (double)p[(int64_t)k * a.G * 2];
                q += (double)p[(int64_t)k * a.G * 2 + 1];
            }
            const double n = (double)a.H * a.W * (C / a.G);
            const double mean = s / n;
            double var = q / n - mean * mean;
            if (var < 0.0) var = 0.0;
            gm[tid] = (float)mean;
            gr[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            const int g = c / (C / a.G);
            const float s = gr[g] * a.gamma[c];
            sc[c] = s;
            sh[c] = a.beta[c] - gm[g] * s;
        }
    } else {
        for (int c = tid; c < C; c += 256) { sc[c] = 1.f; sh[c] = 0.f; }
    }
    __syncthreads();
    const int Ho = a.H * a.scale, Wo = a.W * a.scale;
    const int Cout = a.mean4 ? C >> 2 : C;
    const int co4n = Cout >> 2;
    const int64_t total = (int64_t)Ho * Wo * co4n;
    const float* xb = a.x + (int64_t)b * a.H * a.W * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + tid; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int co4 = (int)(idx % co4n);
        const int64_t pix = idx / co4n;
        const int ox = (int)(pix % Wo), oy = (int)(pix / Wo);
        int y0, y1, x0, x1;
        float ly0, ly1, lx0, lx1;
        src_index(oy, a.H, Ho, a.scale, a.align, y0, y1, ly0, ly1);
        src_index(ox, a.W, Wo, a.scale, a.align, x0, x1, lx0, lx1);
        auto tap = [&](int yy, int xx) -> f32x4 {
            const float* p = xb + ((int64_t)yy * a.W + xx) * C;
            f32x4 r;
            if (a.mean4) {                     // DAP: mean of channels 4co .. 4co+3 (PixelShuffle(2) + AvgPool(2))
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c0 = (4 * co4 + e) * 4;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(p + c0);
                    float m = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) m += activate(v[k] * sc[c0 + k] + sh[c0 + k], a.act);
                    r[e] = m * 0.25f;
                }
            } else {
                const int c0 = 4 * co4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + c0);
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = activate(v[k] * sc[c0 + k] + sh[c0 + k], a.act);
            }
            return r;
        };
        f32x4 v;
        if (a.scale == 1) {
            v = tap(oy, ox);
        } else {
            const f32x4 v00 = tap(y0, x0), v01 = tap(y0, x1), v10 = tap(y1, x0), v11 = tap(y1, x1);
            v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);   // upsample_bilinear2d's own form
        }
        const int64_t opix = ((int64_t)b * Ho + oy) * Wo + ox;
        if (a.ep_mode == 1) {
            const f32x4 ea = *reinterpret_cast<const f32x4*>(a.ep_a + opix * Cout + 4 * co4);
            const f32x4 eb = *reinterpret_cast<const f32x4*>(a.ep_b + opix * Cout + 4 * co4);
            v = v + ea * eb;
        } else if (a.ep_mode == 2) {
            v = v * *reinterpret_cast<const f32x4*>(a.ep_a + opix * Cout + 4 * co4);
        }
        *reinterpret_cast<f32x4*>(a.out + opix * a.out_ctot + a.out_coff + 4 * co4) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// final_out: Conv2d(32 -> 1, 3x3, pad 1) on the NHWC feature map, fused with the eval tail (sigmoid -> > thr -> uint8,
// test.py:100-108).  8 lanes per output pixel, each owning 4 channels of the 9 taps (a tap is one 128-B line), reduced
// with three lane shuffles.  HBM-bound: the 51 MB feature map is read once, neighbours come from L1/L2.
__global__ __launch_bounds__(256) void final_conv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ logits,
                                                         uint8_t* __restrict__ mask, int H, int W, int64_t npix, float thr) {
    const int sub = threadIdx.x & 7;
    f32x4 wr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = *reinterpret_cast<const f32x4*>(w + t * 32 + 4 * sub);
    const float b0 = bias[0];
    for (int64_t pix = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; pix < npix; pix += ((int64_t)gridDim.x * 256) >> 3) {
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        float acc = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                if ((unsigned)(yy + dy) >= (unsigned)H || (unsigned)(xx + dx) >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (pix + dy * W + dx) * 32 + 4 * sub);
                const f32x4 k = wr[(dy + 1) * 3 + dx + 1];
                acc += (v.x * k.x + v.y * k.y) + (v.z * k.z + v.w * k.w);
            }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        if (sub == 0) {
            const float z = acc + b0;
            logits[pix] = z;
            if (mask) mask[pix] = (1.0f / (1.0f + __expf(-z)) > thr) ? 1 : 0;
        }
    }
}

// same conv for C = 32*k input channels (BaselineDecoder.final_out is Conv2d(256 -> 1), decoder:275): the 8 lanes of a
// pixel walk the channel blocks, weights come from L1 instead of registers.
__global__ __launch_bounds__(256) void final_conv_wide_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ logits,
                                                              uint8_t* __restrict__ mask, int H, int W, int C, int64_t npix,
                                                              float thr) {
    const int sub = threadIdx.x & 7;
    const float b0 = bias[0];
    for (int64_t pix = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; pix < npix; pix += ((int64_t)gridDim.x * 256) >> 3) {
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        float acc = 0.f;
        for (int cb = 0; cb < C; cb += 32) {
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if ((unsigned)(yy + dy) >= (unsigned)H || (unsigned)(xx + dx) >= (unsigned)W) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (pix + dy * W + dx) * C + cb + 4 * sub);
                    const f32x4 k = *reinterpret_cast<const f32x4*>(w + ((dy + 1) * 3 + dx + 1) * C + cb + 4 * sub);
                    acc += (v.x * k.x + v.y * k.y) + (v.z * k.z + v.w * k.w);
                }
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 4);
        if (sub == 0) {
            const float z = acc + b0;
            logits[pix] = z;
            if (mask) mask[pix] = (1.0f / (1.0f + __expf(-z)) > thr) ? 1 : 0;
        }
    }
}

}  // namespace

extern "C" int mumpy_final_conv_fwd(const float* x, const float* w_krsc, const float* bias, float* logits, uint8_t* mask,
                                    int B, int H, int W, int C, float thr, void* stream) {
    MUMPY_REQUIRE(x && w_krsc && bias && logits, MUMPY_ENULL, "final_conv: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(w_krsc), MUMPY_EALIGN, "final_conv: x and w must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0, MUMPY_EINVAL,
                  "final_conv: bad shape (C=%d must be a multiple of 32)", C);
    const int64_t npix = (int64_t)B * H * W;
    int64_t grid = (npix * 8 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (C == 32)
        hipLaunchKernelGGL(final_conv_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, w_krsc, bias, logits,
                           mask, H, W, npix, thr);
    else
        hipLaunchKernelGGL(final_conv_wide_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, w_krsc, bias,
                           logits, mask, H, W, C, npix, thr);
    MUMPY_CHECK_LAUNCH("final_conv");
    return 0;
}

extern "C" int mumpy_gn_stats_nhwc_fwd(const float* x, float* partial, int B, int64_t HW, int C, int G, int nsplit,
                                       void* stream) {
    MUMPY_REQUIRE(x && partial, MUMPY_ENULL, "gn_stats: null pointer");
    MUMPY_REQUIRE(aligned16(x), MUMPY_EALIGN, "gn_stats: x must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && HW > 0 && nsplit > 0 && nsplit <= 1024, MUMPY_EINVAL, "gn_stats: bad sizes");
    MUMPY_REQUIRE(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0 && G > 0 && G <= 32 && C % G == 0 && (C / G) % 4 == 0,
                  MUMPY_EINVAL, "gn_stats: unsupported C=%d G=%d", C, G);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nsplit, B), dim3(256), 0, as_stream(stream), x, partial, HW, C, G, nsplit);
    MUMPY_CHECK_LAUNCH("gn_stats");
    return 0;
}

extern "C" int mumpy_gn_apply_resample_nhwc_fwd(const float* x, const float* partial, int nsplit, const float* gamma,
                                                const float* beta, int G, float eps, int act, int mean4, int scale,
                                                int align_corners, int ep_mode, const float* ep_a, const float* ep_b,
                                                float* out, int out_ctot, int out_coff, int B, int H, int W, int C,
                                                void* stream) {
    MUMPY_REQUIRE(x && out, MUMPY_ENULL, "gn_apply: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(out) && aligned16(ep_a) && aligned16(ep_b), MUMPY_EALIGN,
                  "gn_apply: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C <= 256 && C % 4 == 0, MUMPY_EINVAL, "gn_apply: bad shape C=%d", C);
    MUMPY_REQUIRE(scale == 1 || scale == 2 || scale == 4, MUMPY_EINVAL, "gn_apply: scale must be 1, 2 or 4");
    MUMPY_REQUIRE(act >= 0 && act <= 2 && ep_mode >= 0 && ep_mode <= 2, MUMPY_EINVAL, "gn_apply: bad act/epilogue");
    MUMPY_REQUIRE(!mean4 || C % 16 == 0, MUMPY_EINVAL, "gn_apply: DAP mean needs C %% 16 == 0");
    if (partial) {
        MUMPY_REQUIRE(gamma && beta && G > 0 && G <= 32 && C % G == 0 && nsplit > 0, MUMPY_EINVAL, "gn_apply: bad GroupNorm args");
    }
    MUMPY_REQUIRE(ep_mode == 0 || ep_a, MUMPY_ENULL, "gn_apply: epilogue operand missing");
    MUMPY_REQUIRE(ep_mode != 1 || ep_b, MUMPY_ENULL, "gn_apply: epilogue operand b missing");
    const int Cout = mean4 ? C / 4 : C;
    MUMPY_REQUIRE(out_ctot >= out_coff + Cout && out_ctot % 4 == 0 && out_coff % 4 == 0, MUMPY_EINVAL,
                  "gn_apply: output channel slice [%d,+%d) does not fit %d", out_coff, Cout, out_ctot);
    ApplyArgs a;
    a.x = x; a.partial = partial; a.gamma = gamma; a.beta = beta; a.ep_a = ep_a; a.ep_b = ep_b; a.out = out;
    a.H = H; a.W = W; a.C = C; a.G = G; a.nsplit = nsplit; a.act = act; a.mean4 = mean4; a.scale = scale;
    a.align = align_corners; a.ep_mode = ep_mode; a.out_ctot = out_ctot; a.out_coff = out_coff; a.eps = eps;
    const int64_t total = (int64_t)H * scale * W * scale * (Cout / 4);
    int64_t grid = (total + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(gn_apply_resample_kernel, dim3((unsigned)grid, B), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("gn_apply_resample");
    return 0;
}
