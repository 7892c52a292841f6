// faf.hip — FAF frequency features (dct:56-79): S = D x D^T, three band masks on i+j, y_b = D^T (M_b o S) D,
// for ONE frame per clip (only frame index 1 is consumed, mTVE:734: the reference computes all T frames and
// throws 2/3 of them away).
//
// Both transforms are the same "column-block double product"
//     Out[:, cb] = L . ( mask(In) . R[cb, :]^T )            cb = 32 output columns, 224 = 7 x 32
//   forward:  L = D,   In = frame plane,   R = D     ->  S[:, cb]       (scratch)
//   inverse:  L = D^T, In = S o M_band,    R = D^T   ->  y_band[:, cb]  (D[:, cb] = (D^T[cb, :])^T)
// chosen so that BOTH products are "NT" GEMMs whose operands are contiguous along K in memory: the MFMA fragments are
// 16-byte loads (rows of In / L straight from L2, rows of R[cb] from LDS), and the 224x32 intermediate W is written to
// LDS TRANSPOSED ([n][k]: a lane's 4 consecutive accumulator rows are 4 consecutive k) to become the K-contiguous B
// operand of the second product.  The B fragments of a phase (32 columns x 224 deep) stay in registers across the row
// tiles of the phase, so the inner loop is one 16-byte load per four MFMAs.  Band sparsity: the low / mid bands vanish
// for row, col > hi (K loops and row tiles stop there), the high band (i + j >= 224) skips the all-zero upper-left chunks.
// v_mfma_f32_32x32x2_f32: exact fp32.  Replaces the round-1 row-block kernel (one 4-byte LDS read + one 4-byte global
// read per MFMA: 135 + 169 us per forward at B = 8) -- measured: see DESIGN.md.
#include "common.h"
using namespace mumpy;

namespace {

constexpr int N = 224;
constexpr int NCH = N / 32;      // 32-deep K chunks
constexpr int LDT = 228;         // LDS row stride (dwords): 16-B aligned rows, conflict-free ds_read/write_b128
#ifndef MUMPY_FAF_WAVES
#define MUMPY_FAF_WAVES 8
#endif
constexpr int NW = MUMPY_FAF_WAVES;   // waves per block: one 32-row tile of a phase each (7 tiles); 4 waves took two tiles each

struct FafArgs {
    const float* L;      // left matrix (224x224, row-major)
    const float* R;      // right matrix, used as R[cb rows, :]
    const float* In;     // input planes
    float* Out;          // output planes
    int64_t i_batch_stride, i_plane_stride;   // plane (b, ch) of In at In + b*i_batch_stride + ch*i_plane_stride
    int64_t o_batch_stride, o_plane_stride, o_band_stride;
    int masked;          // 0: forward (no mask, gridDim.z == 1); 1: inverse (blockIdx.z = band)
    int lo_hi, mid_lo, mid_hi;
};

__global__ __launch_bounds__(64 * NW) void faf_colblock_kernel(FafArgs a) {
    __shared__ __attribute__((aligned(16))) float Rs[32 * LDT];
    __shared__ __attribute__((aligned(16))) float Wt[32 * LDT];
    const int cb = blockIdx.x, plane = blockIdx.y, band = blockIdx.z;
    const int b = plane / 3, ch = plane - 3 * b;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const float* In = a.In + b * a.i_batch_stride + ch * a.i_plane_stride;
    float* Out = a.Out + b * a.o_batch_stride + ch * a.o_plane_stride + band * a.o_band_stride;

    int lo = 0, hi = 2 * N, kmax = N;        // keep i + j in [lo, hi]
    if (a.masked) {
        if (band == 0) { lo = 0; hi = a.lo_hi; }
        else if (band == 1) { lo = a.mid_lo; hi = a.mid_hi; }
        else { lo = N; hi = 2 * N; }
        kmax = hi + 1 < N ? hi + 1 : N;      // rows / columns of the masked plane past hi are zero
    }
    const int nch = (kmax + 31) / 32;        // K chunks (phase 1) = row tiles of W that can be non-zero = K chunks of phase 2

    for (int idx = tid; idx < 32 * (N / 4); idx += 64 * NW) {                // R[cb rows] -> LDS, 16-B coalesced
        const int r = idx / (N / 4), q = idx - r * (N / 4);
        *reinterpret_cast<f32x4*>(&Rs[r * LDT + 4 * q]) = *reinterpret_cast<const f32x4*>(a.R + (int64_t)(cb * 32 + r) * N + 4 * q);
    }
    for (int idx = tid; idx < 32 * (LDT / 4); idx += 64 * NW) *reinterpret_cast<f32x4*>(&Wt[4 * idx]) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    f32x4 bf[NCH][4];                        // B fragments of the phase: column c of the block, k = 32 ch + 16 h + 4 q ..
    auto load_b = [&](const float* Ls) {
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
            for (int q = 0; q < 4; ++q) bf[kc][q] = *reinterpret_cast<const f32x4*>(&Ls[c * LDT + 32 * kc + 16 * h + 4 * q]);
    };
    load_b(Rs);

    // fragment loads run one chunk ahead of the MFMAs (register double buffer): a wave has nothing else to hide the L2
    // latency of its 16-byte row loads behind
    auto load_a = [&](const float* arow, int kc, f32x4 (&av)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f32x4*>(arow + 32 * kc + 4 * q);
    };
    auto mma = [&](f32x16& acc, const f32x4 (&av)[4], int kc) {
        // bf is indexed with a compile-time chunk at every call site (switch below): runtime indexing would spill it
#define FAF_MMA_CHUNK(KC)                                                                                            \
    case KC:                                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) _Pragma("unroll") for (int e = 0; e < 4; ++e)                  \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][e], bf[KC][q][e], acc, 0, 0, 0);                        \
        break;
        switch (kc) { FAF_MMA_CHUNK(0) FAF_MMA_CHUNK(1) FAF_MMA_CHUNK(2) FAF_MMA_CHUNK(3) FAF_MMA_CHUNK(4) FAF_MMA_CHUNK(5) FAF_MMA_CHUNK(6) }
#undef FAF_MMA_CHUNK
    };
    // phase 1: W[224 x 32] = mask(In) . R[cb]^T; wave w owns row tile w (7 tiles, 8 waves); W goes to LDS transposed
    for (int t = wave; t < nch; t += NW) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int row = 32 * t + c;
        const float* arow = In + (int64_t)row * N + 16 * h;
        // (high band) chunks whose every element has i + j < lo are all zero: start past them
        int k0 = 0;
        if (a.masked) while (k0 < nch && 32 * (t + k0) + 62 < lo) ++k0;
        f32x4 av0[4], av1[4];
        if (k0 < nch) load_a(arow, k0, av0);
        for (int kc = k0; kc < nch; kc += 2) {
            if (kc + 1 < nch) load_a(arow, kc + 1, av1);
            if (a.masked) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int sidx = row + 32 * kc + 16 * h + 4 * q + e;
                        if (sidx < lo || sidx > hi) av0[q][e] = 0.f;
                    }
            }
            mma(acc, av0, kc);
            if (kc + 1 < nch) {
                if (kc + 2 < nch) load_a(arow, kc + 2, av0);
                if (a.masked) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int sidx = row + 32 * (kc + 1) + 16 * h + 4 * q + e;
                            if (sidx < lo || sidx > hi) av1[q][e] = 0.f;
                        }
                }
                mma(acc, av1, kc + 1);
            }
        }
        // D[i][j]: i = (r&3) + 8 (r>>2) + 4 h (row of the tile), j = c (column of the block)  ->  Wt[j][32 t + i]
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(&Wt[c * LDT + 32 * t + 8 * g + 4 * h]) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
    }
    __syncthreads();
    load_b(Wt);
    // phase 2: Out[:, cb] = L . W   (W is zero past row kmax)
    for (int t = wave; t < NCH; t += NW) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* arow = a.L + (int64_t)(32 * t + c) * N + 16 * h;
        f32x4 av0[4], av1[4];
        load_a(arow, 0, av0);
        for (int kc = 0; kc < nch; kc += 2) {
            if (kc + 1 < nch) load_a(arow, kc + 1, av1);
            mma(acc, av0, kc);
            if (kc + 1 < nch) {
                if (kc + 2 < nch) load_a(arow, kc + 2, av0);
                mma(acc, av1, kc + 1);
            }
        }
        float* o = Out + (int64_t)(32 * t + 4 * h) * N + 32 * cb + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * N] = acc[r];
    }
}

}  // namespace

extern "C" int mumpy_faf_fwd(const float* x, const float* D, const float* Dt, float* scratch, float* out, int B, int T,
                             int frame, int lo_hi, int mid_lo, int mid_hi, void* stream) {
    MUMPY_REQUIRE(x && D && Dt && scratch && out, MUMPY_ENULL, "faf: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(D) && aligned16(Dt) && aligned16(scratch) && aligned16(out), MUMPY_EALIGN,
                  "faf: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && T > 0 && frame >= 0 && frame < T, MUMPY_EINVAL, "faf: frame %d outside clip of %d", frame, T);
    MUMPY_REQUIRE(lo_hi >= 0 && mid_lo >= 0 && mid_hi >= mid_lo && mid_hi < 2 * N, MUMPY_EINVAL, "faf: bad band limits");
    const int64_t P = (int64_t)N * N;
    FafArgs f;
    f.L = D; f.R = D; f.In = x + (int64_t)frame * 3 * P; f.Out = scratch;
    f.i_batch_stride = (int64_t)T * 3 * P; f.i_plane_stride = P;
    f.o_batch_stride = 3 * P; f.o_plane_stride = P; f.o_band_stride = 0;
    f.masked = 0; f.lo_hi = lo_hi; f.mid_lo = mid_lo; f.mid_hi = mid_hi;
    hipLaunchKernelGGL(faf_colblock_kernel, dim3(NCH, B * 3, 1), dim3(64 * NW), 0, as_stream(stream), f);
    MUMPY_CHECK_LAUNCH("faf(forward)");
    FafArgs g;
    g.L = Dt; g.R = Dt; g.In = scratch; g.Out = out;
    g.i_batch_stride = 3 * P; g.i_plane_stride = P;
    g.o_batch_stride = 9 * P; g.o_plane_stride = P; g.o_band_stride = 3 * P;
    g.masked = 1; g.lo_hi = lo_hi; g.mid_lo = mid_lo; g.mid_hi = mid_hi;
    hipLaunchKernelGGL(faf_colblock_kernel, dim3(NCH, B * 3, 3), dim3(64 * NW), 0, as_stream(stream), g);
    MUMPY_CHECK_LAUNCH("faf(inverse)");
    return 0;
}
