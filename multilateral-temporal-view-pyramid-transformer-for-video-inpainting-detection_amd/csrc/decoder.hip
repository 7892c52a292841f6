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
        __shared__ double red[256][2];
        gn_block_stats(a.partial, a.nsplit, a.G, b, (double)a.H * a.W * (C / a.G), a.eps, gm, gr, red);
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
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  With pixel runs dealt in launch order the rows
    // above and below a run were fetched by OTHER XCDs, so every tap row came from beyond L2 again: 156 MB read for a 51 MB map
    // (PMC, profiles/r02_pmc_bench_traffic.md).  Renumbered XCD-major, an XCD sweeps one contiguous band of image rows and finds
    // its neighbour rows in its own L2.
    const unsigned G = gridDim.x, xcd = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
    const int64_t vb = (xcd < r8 ? (int64_t)xcd * (q8 + 1) : (int64_t)r8 * (q8 + 1) + (int64_t)(xcd - r8) * q8) + (blockIdx.x >> 3);
    const int64_t per = (npix + G - 1) / G;                      // contiguous pixels per workgroup
    const int64_t pend = (vb + 1) * per < npix ? (vb + 1) * per : npix;
    for (int64_t pix = vb * per + (threadIdx.x >> 3); pix < pend; pix += 32) {
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

// Backward of that convolution (config 5's training step; decoder.py:95,225 under loss.backward()) in one streaming pass:
//   dx[p][c]        = sum_taps w[tap][c] dy[p - tap]          (the pixel that read p through the tap)
//   dw[tap][c]      = sum_p dy[p] x[p + tap][c],   db = sum_p dy[p]
// Same thread layout as the forward (8 lanes x 4 channels per pixel, 32 pixels per pass, contiguous runs per workgroup, XCD-major).
// A thread keeps the nine 4-channel weight-gradient sums of its pixels in registers; a workgroup combines its 32 pixel lanes in a
// fixed order (three shuffles, then the four waves through LDS) and writes ONE partial row [9][32] + [1]; partial_reduce sums the
// rows in a fixed tree.  The first version embedded dy in a 32-channel image for the generic convolution (dx) and ran nine
// M = 1, K = B H W GEMMs on shifted, transposed copies of x (130 us each) for dw: 1.6 ms of a 42 ms step.
constexpr int FCB_ROW = 9 * 32 + 4;                   // partial row: dw (288) | db | 3 pad
__global__ __launch_bounds__(256) void final_conv_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ dy, float* __restrict__ dx,
                                                             float* __restrict__ part, int H, int W, int64_t npix) {
    __shared__ f32x4 red[4][9][8];
    __shared__ float redb[4];
    const int sub = threadIdx.x & 7, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 wr[9], gw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { wr[t] = *reinterpret_cast<const f32x4*>(w + t * 32 + 4 * sub); gw[t] = f32x4{0, 0, 0, 0}; }
    float gb = 0.f;
    const unsigned G = gridDim.x, xcd = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
    const int64_t vb = (xcd < r8 ? (int64_t)xcd * (q8 + 1) : (int64_t)r8 * (q8 + 1) + (int64_t)(xcd - r8) * q8) + (blockIdx.x >> 3);
    const int64_t per = (npix + G - 1) / G;
    const int64_t pend = (vb + 1) * per < npix ? (vb + 1) * per : npix;
    for (int64_t pix = vb * per + (threadIdx.x >> 3); pix < pend; pix += 32) {
        const int xx = (int)(pix % W);
        const int yy = (int)((pix / W) % H);
        const float d0 = dy[pix];
        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
            for (int kx = -1; kx <= 1; ++kx) {
                const int t = (ky + 1) * 3 + kx + 1;
                // forward: out[y][x] += w[t] . in[y + ky][x + kx]  =>  in[yy][xx] was read by out[yy - ky][xx - kx] through tap t
                if ((unsigned)(yy - ky) < (unsigned)H && (unsigned)(xx - kx) < (unsigned)W) acc += wr[t] * dy[pix - ky * W - kx];
                if ((unsigned)(yy + ky) < (unsigned)H && (unsigned)(xx + kx) < (unsigned)W)
                    gw[t] += *reinterpret_cast<const f32x4*>(x + (pix + ky * W + kx) * 32 + 4 * sub) * d0;
            }
        *reinterpret_cast<f32x4*>(dx + pix * 32 + 4 * sub) = acc;
        if (sub == 0) gb += d0;
    }
    // pixel lanes of a wave: lanes with equal `sub` are 8 apart -> xor 8, 16, 32 (every lane takes part: zeros where no pixel)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = gw[t][e];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            gw[t][e] = v;
        }
    gb += __shfl_xor(gb, 8); gb += __shfl_xor(gb, 16); gb += __shfl_xor(gb, 32);
    if (lane < 8) {
#pragma unroll
        for (int t = 0; t < 9; ++t) red[wave][t][sub] = gw[t];
        if (lane == 0) redb[wave] = gb;
    }
    __syncthreads();
    float* row = part + (int64_t)blockIdx.x * FCB_ROW;
    if (threadIdx.x < 72) {
        const int t = threadIdx.x >> 3, q = threadIdx.x & 7;
        const f32x4 v = (red[0][t][q] + red[1][t][q]) + (red[2][t][q] + red[3][t][q]);
        *reinterpret_cast<f32x4*>(row + t * 32 + 4 * q) = v;
    }
    if (threadIdx.x == 72) *reinterpret_cast<f32x4*>(row + 288) = f32x4{(redb[0] + redb[1]) + (redb[2] + redb[3]), 0.f, 0.f, 0.f};
}

// fixed-tree column sums of the partial rows (64 columns per block, 16 row lanes, combined in order through LDS)
__global__ __launch_bounds__(1024) void fcb_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                          int64_t nparts) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float s = 0.f;
    if (i < 289 && grp < nparts)
        s = ordered_sum(part[(int64_t)grp * FCB_ROW + i], part + (int64_t)(grp + 16) * FCB_ROW + i, (int64_t)16 * FCB_ROW,
                        (int)((nparts - grp + 15) / 16) - 1);
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && i < 289) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        if (i < 288) dw[i] = t; else db[0] = t;
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


// ---------------------------------------------------------------------------------------------------------------
// Small data-movement / wiring kernels of the decoder and of the encoder tail (round 3: these were ATen launches --
// avg_pool2d, cat, strided copy_, mul, add, PixelShuffle -- 19 per forward).

// 2x2 average pooling (nn.AvgPool2d(2, stride=2) of the frequency blocks, decoder.py:149-178) into an NHWC map whose channel count
// is padded with zeros to Cpad (the implicit-GEMM convolution wants Cin % 32 == 0; only the 9-channel DCT input needs it).
// NCHW_IN: x is (B, C, H, W) contiguous (the FAF output, dct:79) -- lanes run along x so a plane row is read coalesced.
template <bool NCHW_IN>
__global__ __launch_bounds__(256) void avgpool2_pad_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int C,
                                                           int Cpad, int64_t total) {
    const int Ho = H >> 1, Wo = W >> 1;
    if (NCHW_IN) {
        // thread = output pixel; writes Cpad floats (C <= 32 here)
        for (int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (int64_t)gridDim.x * 256) {
            const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
            const int64_t b = pix / ((int64_t)Wo * Ho);
            const float* p = x + (b * C * H + 2 * oy) * (int64_t)W + 2 * ox;
            float* o = out + pix * Cpad;
            for (int c0 = 0; c0 < Cpad; c0 += 4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + e;
                    if (c < C) {
                        const float* q = p + (int64_t)c * H * W;
                        v[e] = ((q[0] + q[1]) + (q[W] + q[W + 1])) * 0.25f;       // ATen's avg_pool2d sums the window, then divides
                    }
                }
                *reinterpret_cast<f32x4*>(o + c0) = v;
            }
        }
    } else {
        const int c4n = Cpad >> 2;                              // thread = (output pixel, channel quad)
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total * c4n; idx += (int64_t)gridDim.x * 256) {
            const int c4 = (int)(idx % c4n);
            const int64_t pix = idx / c4n;
            const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
            const int64_t b = pix / ((int64_t)Wo * Ho);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (4 * c4 < C) {
                const float* p = x + ((b * H + 2 * oy) * (int64_t)W + 2 * ox) * C + 4 * c4;
                const f32x4 a = *reinterpret_cast<const f32x4*>(p), bq = *reinterpret_cast<const f32x4*>(p + C);
                const f32x4 c = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C), d = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C + C);
                v = ((a + bq) + (c + d)) * 0.25f;
            }
            *reinterpret_cast<f32x4*>(out + pix * Cpad + 4 * c4) = v;
        }
    }
}

// rows x cols floats from rows of pitch src_stride to rows of pitch dst_stride (channel slices of concatenated NHWC maps:
// decoder.py:197,210,213; the first three temporal slices of the global tokens, mTVE:745)
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, int64_t src_stride, float* __restrict__ dst,
                                                        int64_t dst_stride, int64_t rows, int c4n) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < rows * c4n; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / c4n;
        const int c4 = (int)(idx - r * c4n);
        *reinterpret_cast<f32x4*>(dst + r * dst_stride + 4 * c4) = *reinterpret_cast<const f32x4*>(src + r * src_stride + 4 * c4);
    }
}

// channel merge of the three views for the global embedding (mTVE:710-718, 739): out row (b, site, t) =
// [v1[b, t1 == T ? t : 0, site, :C1] | v2 ... | v3 ...]; view k is (B, t_k * n, C_k) with its frames stacked on rows.
struct MergeArgs { const float* v[3]; int c[3]; int t[3]; };
__global__ __launch_bounds__(256) void merge_views_kernel(MergeArgs a, float* __restrict__ out, int T, int n, int64_t rows) {
    const int ctot = a.c[0] + a.c[1] + a.c[2], c4n = ctot >> 2;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < rows * c4n; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / c4n;                            // (b, site, t)
        int c = 4 * (int)(idx - r * c4n);
        const int t = (int)(r % T);
        const int64_t bs = r / T;
        const int site = (int)(bs % n);
        const int64_t b = bs / n;
        const int k = c < a.c[0] ? 0 : (c < a.c[0] + a.c[1] ? 1 : 2);
        c -= k == 0 ? 0 : (k == 1 ? a.c[0] : a.c[0] + a.c[1]);
        const int tk = a.t[k] == 1 ? 0 : t;
        const float* s = a.v[k] + ((b * a.t[k] + tk) * n + site) * (int64_t)a.c[k] + c;
        *reinterpret_cast<f32x4*>(out + idx * 4) = *reinterpret_cast<const f32x4*>(s);
    }
}

// decoder.py:198-205: z = gcn1 * freq3 + PixelShuffle(2)(g * f) with g, f (B,h,w,4C) and gcn1, freq3, z (B,2h,2w,C), all NHWC.
// PixelShuffle: out[c, 2y+i, 2x+j] = in[4c + 2i + j, y, x].
__global__ __launch_bounds__(256) void trunk_head_kernel(const float* __restrict__ g, const float* __restrict__ f,
                                                         const float* __restrict__ gcn, const float* __restrict__ fr,
                                                         float* __restrict__ z, int h, int w, int C, int64_t total) {
    const int c4n = C >> 2;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(idx % c4n);
        const int64_t pix = idx / c4n;
        const int ox = (int)(pix % (2 * w)), oy = (int)((pix / (2 * w)) % (2 * h));
        const int64_t b = pix / ((int64_t)4 * h * w);
        const int sub = 2 * (oy & 1) + (ox & 1);
        const float* gp = g + ((b * h + (oy >> 1)) * (int64_t)w + (ox >> 1)) * (4 * C);
        const float* fp = f + (gp - g);
        f32x4 ps;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ci = 4 * (4 * c4 + e) + sub;
            ps[e] = gp[ci] * fp[ci];
        }
        const f32x4 a = *reinterpret_cast<const f32x4*>(gcn + idx * 4), bq = *reinterpret_cast<const f32x4*>(fr + idx * 4);
        *reinterpret_cast<f32x4*>(z + idx * 4) = a * bq + ps;
    }
}

}  // namespace

extern "C" int mumpy_avgpool2_pad_nhwc_fwd(const float* x, float* out, int B, int H, int W, int C, int Cpad, int nchw_in,
                                           void* stream) {
    MUMPY_REQUIRE(x && out, MUMPY_ENULL, "avgpool2: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(out), MUMPY_EALIGN, "avgpool2: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && Cpad >= C && Cpad % 4 == 0, MUMPY_EINVAL,
                  "avgpool2: bad shape (%d,%d,%d,%d) -> Cpad %d", B, H, W, C, Cpad);
    MUMPY_REQUIRE(nchw_in || C % 4 == 0, MUMPY_EINVAL, "avgpool2: an NHWC input needs C %% 4 == 0 (got %d)", C);
    const int64_t total = (int64_t)B * (H / 2) * (W / 2);
    int64_t grid = ((nchw_in ? total : total * (Cpad / 4)) + 255) / 256;
    if (grid > 4096) grid = 4096;
    if (nchw_in) hipLaunchKernelGGL(avgpool2_pad_kernel<true>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, out, H, W, C, Cpad, total);
    else hipLaunchKernelGGL(avgpool2_pad_kernel<false>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, out, H, W, C, Cpad, total);
    MUMPY_CHECK_LAUNCH("avgpool2_pad");
    return 0;
}

extern "C" int mumpy_copy_rows_fwd(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows, int cols,
                                   void* stream) {
    if (rows == 0) return 0;
    MUMPY_REQUIRE(src && dst, MUMPY_ENULL, "copy_rows: null pointer");
    MUMPY_REQUIRE(aligned16(src) && aligned16(dst) && src_stride % 4 == 0 && dst_stride % 4 == 0 && cols % 4 == 0, MUMPY_EALIGN,
                  "copy_rows: pointers, strides and the column count must be multiples of 16 bytes");
    MUMPY_REQUIRE(rows > 0 && cols > 0 && src_stride >= cols && dst_stride >= cols, MUMPY_EINVAL, "copy_rows: bad sizes");
    int64_t grid = (rows * (cols / 4) + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), src, src_stride, dst, dst_stride, rows, cols / 4);
    MUMPY_CHECK_LAUNCH("copy_rows");
    return 0;
}

extern "C" int mumpy_merge_views_fwd(const float* v1, const float* v2, const float* v3, float* out, int B, int T, int n, int C1,
                                     int C2, int C3, int t1, int t2, int t3, void* stream) {
    MUMPY_REQUIRE(v1 && v2 && v3 && out, MUMPY_ENULL, "merge_views: null pointer");
    MUMPY_REQUIRE(aligned16(v1) && aligned16(v2) && aligned16(v3) && aligned16(out), MUMPY_EALIGN, "merge_views: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && T > 0 && n > 0 && C1 > 0 && C2 > 0 && C3 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && C3 % 4 == 0, MUMPY_EINVAL,
                  "merge_views: bad sizes");
    MUMPY_REQUIRE((t1 == 1 || t1 == T) && (t2 == 1 || t2 == T) && (t3 == 1 || t3 == T), MUMPY_EINVAL,
                  "merge_views: a view's temporal length must be 1 or T=%d (got %d,%d,%d; mTVE:717 repeats by T / t)", T, t1, t2, t3);
    MergeArgs a;
    a.v[0] = v1; a.v[1] = v2; a.v[2] = v3; a.c[0] = C1; a.c[1] = C2; a.c[2] = C3; a.t[0] = t1; a.t[1] = t2; a.t[2] = t3;
    const int64_t rows = (int64_t)B * n * T;
    int64_t grid = (rows * ((C1 + C2 + C3) / 4) + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(merge_views_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a, out, T, n, rows);
    MUMPY_CHECK_LAUNCH("merge_views");
    return 0;
}

extern "C" int mumpy_trunk_head_fwd(const float* g, const float* f, const float* gcn, const float* freq, float* z, int B, int h,
                                    int w, int C, void* stream) {
    MUMPY_REQUIRE(g && f && gcn && freq && z, MUMPY_ENULL, "trunk_head: null pointer");
    MUMPY_REQUIRE(aligned16(g) && aligned16(f) && aligned16(gcn) && aligned16(freq) && aligned16(z), MUMPY_EALIGN,
                  "trunk_head: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, MUMPY_EINVAL, "trunk_head: bad shape");
    const int64_t total = (int64_t)B * 4 * h * w * (C / 4);
    int64_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(trunk_head_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), g, f, gcn, freq, z, h, w, C, total);
    MUMPY_CHECK_LAUNCH("trunk_head");
    return 0;
}

extern "C" int mumpy_final_conv_fwd(const float* x, const float* w_krsc, const float* bias, float* logits, uint8_t* mask,
                                    int B, int H, int W, int C, float thr, void* stream) {
    MUMPY_REQUIRE(x && w_krsc && bias && logits, MUMPY_ENULL, "final_conv: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(w_krsc), MUMPY_EALIGN, "final_conv: x and w must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0, MUMPY_EINVAL,
                  "final_conv: bad shape (C=%d must be a multiple of 32)", C);
    const int64_t npix = (int64_t)B * H * W;
    int64_t grid = (npix * 8 + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (C == 32) {          // contiguous 128-pixel runs per workgroup (four 32-pixel passes), renumbered XCD-major in the kernel
        grid = (npix + 127) / 128;
        if (grid > 65536) grid = 65536;
    }
    if (C == 32)
        hipLaunchKernelGGL(final_conv_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, w_krsc, bias, logits,
                           mask, H, W, npix, thr);
    else
        hipLaunchKernelGGL(final_conv_wide_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, w_krsc, bias,
                           logits, mask, H, W, C, npix, thr);
    MUMPY_CHECK_LAUNCH("final_conv");
    return 0;
}

static int64_t fcb_blocks(int64_t npix) {            // contiguous runs of >= 128 pixels, at most 2048 workgroups
    int64_t g = (npix + 127) / 128;
    if (g > 2048) g = 2048;
    return g < 1 ? 1 : g;
}

extern "C" int64_t mumpy_final_conv_bwd_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return fcb_blocks((int64_t)B * H * W) * FCB_ROW * (int64_t)sizeof(float);
}

extern "C" int mumpy_final_conv_bwd(const float* x, const float* w_krsc, const float* dy, float* dx, float* dw, float* db,
                                    void* workspace, int64_t workspace_bytes, int B, int H, int W, int C, void* stream) {
    MUMPY_REQUIRE(x && w_krsc && dy && dx && dw && db && workspace, MUMPY_ENULL, "final_conv_bwd: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(w_krsc) && aligned16(dx) && aligned16(workspace), MUMPY_EALIGN,
                  "final_conv_bwd: x, w, dx and the workspace must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && C == 32, MUMPY_EINVAL, "final_conv_bwd: built for C = 32 input channels (got %d)", C);
    MUMPY_REQUIRE(workspace_bytes >= mumpy_final_conv_bwd_workspace_bytes(B, H, W), MUMPY_EINVAL, "final_conv_bwd: workspace too small");
    const int64_t npix = (int64_t)B * H * W, grid = fcb_blocks(npix);
    float* part = static_cast<float*>(workspace);
    hipLaunchKernelGGL(final_conv_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x, w_krsc, dy, dx, part, H, W, npix);
    MUMPY_CHECK_LAUNCH("final_conv_bwd");
    hipLaunchKernelGGL(fcb_reduce_kernel, dim3(5), dim3(1024), 0, as_stream(stream), part, dw, db, grid);
    MUMPY_CHECK_LAUNCH("final_conv_bwd(reduce)");
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
