// cva_fused.hip — the two GEMMs of SwinDAttention (deform:324-405) whose neighbours are pure index work, with that work folded
// into the GEMM's operand loader / epilogue so that the intermediate tensors never exist in HBM (round 3):
//
//   mumpy_deform_sample_kv_fwd    kv = [proj_k | proj_v](grid_sample(x2, pos))                 (deform:353-362)
//       the bilinear sampling IS the A-operand loader: a staged A piece (row m = (kv window, point), 4 channels of a 32-deep
//       chunk) is the weighted sum of the four corner pieces of the window's 49 x C tile, which lives in L2 after its first
//       touch; the sampled map (nkv x 49 x C, 48 MB at stage 0 of config 2) is neither written nor read.  A row's corner
//       offsets and weights are recomputed only when the chunk enters the next offset group (C/3 channels share a position).
//       Tile 64 x 192: at stage 0 the whole [k|v] width (2C = 192) is one tile, so every point is sampled exactly once;
//       deeper stages sample a point once per 192-column tile (2-8 times) out of L2-resident 18-150 KB tiles.
//   mumpy_deform_out_combine_fwd  x1 + x1[window order] + reshape_(C,49)->(49,C)(proj_out(o))      (deform:402-403; mTVE:138,285-286)
//       computed TRANSPOSED -- A = W_out (rows = output channel c'), B = o (rows = (q window, point p')) -- so that a lane of
//       the accumulator owns one p' and the scrambled destination  window_base + c' * 49 + p'  (the reference re-reads the flat
//       (C,49) image as (49,C) without a permute) is consecutive across lanes: coalesced stores, and the two x1 terms of the
//       caller's residual wiring are read at the same (contiguous / row-segment) addresses.  Replaces the proj_out GEMM, its
//       (nq,49,C) output and mumpy_deform_combine_fwd.
//
// Both are 4-wave tiled kernels on v_mfma_f32_32x32x2_f32 (exact fp32 products, as every GEMM of the fp32 path), 32-deep
// chunks, two LDS buffers with 36-dword rows, global -> register -> LDS staging one chunk ahead.  They are mid-size launches
// (0.5-9 GFLOP); what they buy is two launches and two HBM round trips less on the cross-view dependency chain.
#include "common.h"
using namespace mumpy;

namespace {

constexpr int BK = 32, LDR = 36;

// m / 49 and m % 49 for m < 2^31 / 49 (mulhi by ceil(2^37 / 49))
__device__ __forceinline__ void divmod49(unsigned m, unsigned& q, unsigned& r) {
    q = (unsigned)(((uint64_t)m * 2804876602ull) >> 37);
    r = m - 49u * q;
}

// ------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256, 2) void deform_sample_kv_kernel(const float* __restrict__ x2, const float* __restrict__ pos,
                                                                  const float* __restrict__ Wkv, const float* __restrict__ bkv,
                                                                  float* __restrict__ kv, int Hs2, int W, int nWx, int nW2, int nq,
                                                                  int M, unsigned gn) {
    constexpr int BM = 64, BN = 192, TN = 3, N = 2 * C, K = C, CG = C / 3, NK = K / BK, CPG = CG / BK;   // CPG: chunks per group
    constexpr int A_LD = BM / 32, B_LD = BN / 32;
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    // XCD-major renumbering; the N tiles of one row panel are neighbours (they sample the same windows)
    const unsigned nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const unsigned tm = wgid / gn, tn = wgid - tm * gn;
    const int m0 = (int)tm * BM, n0 = (int)tn * BN;
    const int ld_row = tid >> 3, ld_c4 = tid & 7;

    // per staged A row: the window's first token, the point, and (per offset group) four corner offsets + weights
    const float* abase[A_LD];
    const float* apos[A_LD];
    int aoff[A_LD][4];
    float aw[A_LD][4];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int m = m0 + ld_row + 32 * i;
        if (m > M - 1) m = M - 1;                       // rows past the edge: clamped, never stored
        unsigned b2, p;
        divmod49((unsigned)m, b2, p);
        const unsigned b = b2 / (unsigned)nW2, n = b2 - b * (unsigned)nW2;
        const unsigned wy = n / (unsigned)nWx, wx = n - wy * (unsigned)nWx;
        abase[i] = x2 + (((int64_t)b * Hs2 + wy * WS) * W + wx * WS) * C + 4 * ld_c4;
        apos[i] = pos + ((int64_t)(b2 % (unsigned)nq) * 3 * WT + p) * 2;
    }
    auto set_group = [&](int g) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const float gy = apos[i][g * WT * 2], gx = apos[i][g * WT * 2 + 1];
            const float iy = ((gy + 1.0f) * 0.5f) * 6.0f;       // grid_sample, align_corners=True: pixel = (g + 1) / 2 * (size - 1)
            const float ix = ((gx + 1.0f) * 0.5f) * 6.0f;
            const float y0f = floorf(iy), x0f = floorf(ix);
            const int y0 = (int)y0f, x0 = (int)x0f;
            const float wgt[4] = {(x0f + 1.0f - ix) * (y0f + 1.0f - iy), (ix - x0f) * (y0f + 1.0f - iy),
                                  (x0f + 1.0f - ix) * (iy - y0f), (ix - x0f) * (iy - y0f)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int yy = y0 + (e >> 1), xx = x0 + (e & 1);
                const bool ok = (unsigned)yy < (unsigned)WS && (unsigned)xx < (unsigned)WS;      // zeros padding
                aw[i][e] = ok ? wgt[e] : 0.f;
                aoff[i][e] = ok ? (yy * W + xx) * C : 0;
            }
        }
    };
    const float* brow[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        int n = n0 + ld_row + 32 * i;
        if (n > N - 1) n = N - 1;
        brow[i] = Wkv + (int64_t)n * K + 4 * ld_c4;
    }
    f32x4 areg[A_LD], breg[B_LD];
    auto gload = [&](int kc) {
        const int k0 = kc * BK;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const float* s = abase[i] + k0;
            // the order of operations of deform_sample_lds_kernel (nw, ne, sw, se): the sampled values are bitwise the same
            f32x4 r = *reinterpret_cast<const f32x4*>(s + aoff[i][0]) * aw[i][0];
            r += *reinterpret_cast<const f32x4*>(s + aoff[i][1]) * aw[i][1];
            r += *reinterpret_cast<const f32x4*>(s + aoff[i][2]) * aw[i][2];
            r += *reinterpret_cast<const f32x4*>(s + aoff[i][3]) * aw[i][3];
            areg[i] = r;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) breg[i] = *reinterpret_cast<const f32x4*>(brow[i] + k0);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(&lds[buf][(ld_row + 32 * i) * LDR + 4 * ld_c4]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i) *reinterpret_cast<f32x4*>(&lds[buf][(BM + ld_row + 32 * i) * LDR + 4 * ld_c4]) = breg[i];
    };
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto compute = [&](int buf) {
        const float* a_f = &lds[buf][(wm * 32 + c) * LDR + 16 * h];
        const float* b_f = &lds[buf][(BM + wn * 96 + c) * LDR + 16 * h];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(a_f + 4 * q);
            f32x4 fb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b_f + 32 * j * LDR + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[j][e], acc[j], 0, 0, 0);
        }
    };
    set_group(0);
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kc = 0; kc < NK; ++kc) {
        if (kc + 1 < NK) {
            if ((kc + 1) % CPG == 0) set_group((kc + 1) / CPG);
            gload(kc + 1);
        }
        compute(kc & 1);
        if (kc + 1 < NK) lstore((kc + 1) & 1);          // the other buffer: its readers finished before the last barrier
        __syncthreads();
    }
    // D[row][col]: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * 96 + 32 * j + c;
        if (n >= N) continue;
        const float bv = bkv[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m < M) kv[(int64_t)m * N + n] = acc[j][r] + bv;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// out[win_base + f] = x1[win_base + f] + x1[(img + tok(f / C)) C + f % C] + Yt[p'][c'],  f = c' 49 + p'
template <int C>
__global__ __launch_bounds__(256, 2) void deform_out_combine_kernel(const float* __restrict__ o, const float* __restrict__ Wout,
                                                                   const float* __restrict__ bout, const float* __restrict__ x1,
                                                                   float* __restrict__ out, int H, int W, int nWx, int nWf, int MO,
                                                                   unsigned gn) {
    constexpr int BM = 64, BN = 64, K = C, NK = K / BK;          // BM: output channels c' (rows of W_out); BN: rows of o
    constexpr int A_LD = BM / 32, B_LD = BN / 32;
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const unsigned nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const unsigned tn = wgid / gn, tm = wgid - tn * gn;           // the channel tiles of one o row panel are neighbours
    const int c0 = (int)tm * BM, m0 = (int)tn * BN;
    const int ld_row = tid >> 3, ld_c4 = tid & 7;
    const float* arow[A_LD];
    const float* brow[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int cc = c0 + ld_row + 32 * i;
        if (cc > C - 1) cc = C - 1;
        arow[i] = Wout + (int64_t)cc * K + 4 * ld_c4;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        int m = m0 + ld_row + 32 * i;
        if (m > MO - 1) m = MO - 1;
        brow[i] = o + (int64_t)m * K + 4 * ld_c4;
    }
    f32x4 areg[A_LD], breg[B_LD];
    auto gload = [&](int kc) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) areg[i] = *reinterpret_cast<const f32x4*>(arow[i] + kc * BK);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) breg[i] = *reinterpret_cast<const f32x4*>(brow[i] + kc * BK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(&lds[buf][(ld_row + 32 * i) * LDR + 4 * ld_c4]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i) *reinterpret_cast<f32x4*>(&lds[buf][(BM + ld_row + 32 * i) * LDR + 4 * ld_c4]) = breg[i];
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&](int buf) {
        const float* a_f = &lds[buf][(wm * 32 + c) * LDR + 16 * h];
        const float* b_f = &lds[buf][(BM + wn * 32 + c) * LDR + 16 * h];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 fa = *reinterpret_cast<const f32x4*>(a_f + 4 * q), fb = *reinterpret_cast<const f32x4*>(b_f + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kc = 0; kc < NK; ++kc) {
        if (kc + 1 < NK) gload(kc + 1);
        compute(kc & 1);
        if (kc + 1 < NK) lstore((kc + 1) & 1);
        __syncthreads();
    }
    // this lane's column = row m of o = (q window bw, point p'); accumulator rows = output channels c'
    const int m = m0 + wn * 32 + c;
    if (m >= MO) return;
    unsigned bw, pp;
    divmod49((unsigned)m, bw, pp);
    const unsigned b = bw / (unsigned)nWf, n = bw - b * (unsigned)nWf;
    const int wy = (int)(n / (unsigned)nWx), wx = (int)(n - (n / (unsigned)nWx) * (unsigned)nWx);
    const int64_t img = (int64_t)b * H * W;
    const int64_t win_base = (img + (int64_t)n * WT) * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cc = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (cc >= C) continue;
        const int f = cc * WT + (int)pp;
        const int po = f / C, co = f - po * C;                   // destination token-in-window and channel (C is a constant)
        const int tok = window_token(wy, wx, po, H, W, 0);
        out[win_base + f] = x1[win_base + f] + x1[(img + tok) * C + co] + (acc[r] + bout[cc]);
    }
}

}  // namespace

extern "C" int mumpy_deform_sample_kv_fwd(const float* x2, const float* pos, const float* Wkv, const float* bkv, float* kv, int B,
                                          int Hs2, int W, int C, int nq, void* stream) {
    MUMPY_REQUIRE(x2 && pos && Wkv && bkv && kv, MUMPY_ENULL, "deform_sample_kv: null pointer");
    MUMPY_REQUIRE(aligned16(x2) && aligned16(Wkv) && aligned16(kv), MUMPY_EALIGN, "deform_sample_kv: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && Hs2 > 0 && W > 0 && Hs2 % WS == 0 && W % WS == 0 && nq > 0, MUMPY_EINVAL,
                  "deform_sample_kv: bad grid (%d,%d) / nq=%d", Hs2, W, nq);
    MUMPY_REQUIRE(C == 96 || C == 192 || C == 384 || C == 768, MUMPY_EINVAL,
                  "deform_sample_kv: C=%d is not one of the encoder widths 96/192/384/768", C);
    const int nWx = W / WS, nW2 = (Hs2 / WS) * nWx;
    const int64_t nwin = (int64_t)B * nW2, M = nwin * WT;
    MUMPY_REQUIRE(M < (1ll << 31) - 64, MUMPY_ERANGE, "deform_sample_kv: too many windows");
    const unsigned gn = (unsigned)((2 * C + 191) / 192);
    const int64_t grid = ((M + 63) / 64) * gn;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "deform_sample_kv: too many tiles");
#define MUMPY_SKV(C_)                                                                                                     \
    hipLaunchKernelGGL(deform_sample_kv_kernel<C_>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), x2, pos, Wkv, bkv, \
                       kv, Hs2, W, nWx, nW2, nq, (int)M, gn)
    switch (C) {
        case 96: MUMPY_SKV(96); break;
        case 192: MUMPY_SKV(192); break;
        case 384: MUMPY_SKV(384); break;
        default: MUMPY_SKV(768); break;
    }
#undef MUMPY_SKV
    MUMPY_CHECK_LAUNCH("deform_sample_kv");
    return 0;
}

extern "C" int mumpy_deform_out_combine_fwd(const float* o, const float* Wout, const float* bout, const float* x1, float* out, int B,
                                            int H, int W, int C, void* stream) {
    MUMPY_REQUIRE(o && Wout && bout && x1 && out, MUMPY_ENULL, "deform_out_combine: null pointer");
    MUMPY_REQUIRE(x1 != out, MUMPY_EINVAL, "deform_out_combine: out must not alias x1");
    MUMPY_REQUIRE(aligned16(o) && aligned16(Wout), MUMPY_EALIGN, "deform_out_combine: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && H % WS == 0 && W % WS == 0, MUMPY_EINVAL, "deform_out_combine: bad grid (%d,%d)", H, W);
    MUMPY_REQUIRE(C == 96 || C == 192 || C == 384 || C == 768, MUMPY_EINVAL,
                  "deform_out_combine: C=%d is not one of the encoder widths 96/192/384/768", C);
    const int nWx = W / WS, nWf = (H / WS) * nWx;
    const int64_t MO = (int64_t)B * nWf * WT;
    MUMPY_REQUIRE(MO * 49 < (1ll << 31) && MO < (1ll << 31) - 64, MUMPY_ERANGE, "deform_out_combine: too many windows");
    const unsigned gn = (unsigned)((C + 63) / 64);
    const int64_t grid = ((MO + 63) / 64) * gn;
#define MUMPY_OC(C_)                                                                                                      \
    hipLaunchKernelGGL(deform_out_combine_kernel<C_>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), o, Wout, bout, x1, \
                       out, H, W, nWx, nWf, (int)MO, gn)
    switch (C) {
        case 96: MUMPY_OC(96); break;
        case 192: MUMPY_OC(192); break;
        case 384: MUMPY_OC(384); break;
        default: MUMPY_OC(768); break;
    }
#undef MUMPY_OC
    MUMPY_CHECK_LAUNCH("deform_out_combine");
    return 0;
}
