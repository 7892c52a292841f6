// faf.hip — FAF frequency features (dct:56-79): X = D x D^T, three band masks on i+j, y_b = D^T (M_b o X) D,
// for ONE frame per clip (only frame index 1 is consumed, mTVE:734: the reference computes all T frames and
// throws 2/3 of them away).
//
// One generic "row-block double product" kernel, launched twice:
//     Out[rb*32 .. +32, :] = ( A[rb*32 .. +32, :] . mask(X) ) . Bm        (224 = 7 x 32: MFMA tiles fit exactly)
//   forward:  A = D,   X = frame plane,      no mask, Bm = D^T  -> spectrum (scratch)
//   inverse:  A = D^T, X = spectrum o M_band,          Bm = D    -> out[b][band*3 + rgb]
// Phase 1 keeps the A row block in LDS (row stride 225: conflict-free column reads) and streams X rows as the MFMA
// B operand straight from L2 (128-B row segments); the 32x224 intermediate goes through LDS once to become the A
// operand of phase 2.  Band sparsity: a band's masked spectrum is zero for rows/cols >= kmax (low: 80, mid: 113),
// so both K loops and the phase-1 column tiles stop at kmax.  v_mfma_f32_32x32x2_f32: exact fp32.
#include "common.h"
using namespace mumpy;

namespace {

constexpr int N = 224;
constexpr int LDT = 225;

struct FafArgs {
    const float* A;      // left matrix (224x224)
    const float* Bm;     // right matrix (224x224)
    const float* X;      // input planes
    float* Out;          // output planes
    int64_t x_batch_stride, x_plane_stride;   // plane (b, ch) of X at X + b*x_batch_stride + ch*x_plane_stride
    int64_t o_batch_stride, o_plane_stride, o_band_stride;
    int masked;          // 0: forward (no mask, gridDim.z == 1); 1: inverse (blockIdx.z = band)
    int lo_hi, mid_lo, mid_hi;
};

__global__ __launch_bounds__(256) void faf_rowblock_kernel(FafArgs a) {
    __shared__ float As[32 * LDT];
    __shared__ float Ts[32 * LDT];
    const int rb = blockIdx.x, plane = blockIdx.y, band = blockIdx.z;
    const int b = plane / 3, ch = plane - 3 * b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const float* X = a.X + b * a.x_batch_stride + ch * a.x_plane_stride;
    float* Out = a.Out + b * a.o_batch_stride + ch * a.o_plane_stride + band * a.o_band_stride;

    int lo = 0, hi = 2 * N, kmax = N;        // keep i+j in [lo,hi]
    if (a.masked) {
        if (band == 0) { lo = 0; hi = a.lo_hi; kmax = a.lo_hi + 1; }
        else if (band == 1) { lo = a.mid_lo; hi = a.mid_hi; kmax = a.mid_hi + 1; }
        else { lo = N; hi = 2 * N; kmax = N; }
        if (kmax > N) kmax = N;
        kmax = (kmax + 1) & ~1;
    }
    for (int idx = tid; idx < 32 * N; idx += 256) {
        const int r = idx / N, k = idx - r * N;
        As[r * LDT + k] = a.A[(rb * 32 + r) * N + k];
    }
    for (int idx = tid; idx < 32 * LDT; idx += 256) Ts[idx] = 0.f;
    __syncthreads();

    // phase 1: T[32][224] = A_rb . mask(X); wave w owns column tiles w and w+4
    for (int nt = wave; nt < 7; nt += 4) {
        if (nt * 32 >= kmax) break;          // those columns of mask(X) are all zero
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int n = nt * 32 + c;
#pragma unroll 8
        for (int kk = 0; kk < kmax / 2; ++kk) {
            const int k = 2 * kk + h;
            float xv = X[k * N + n];
            if (a.masked && (k + n < lo || k + n > hi)) xv = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[c * LDT + k], xv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ts[((r & 3) + 8 * (r >> 2) + 4 * h) * LDT + n] = acc[r];
    }
    __syncthreads();
    // phase 2: Out_rb = T . Bm   (T is zero beyond column kmax)
    for (int nt = wave; nt < 7; nt += 4) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int n = nt * 32 + c;
#pragma unroll 8
        for (int kk = 0; kk < kmax / 2; ++kk) {
            const int k = 2 * kk + h;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ts[c * LDT + k], a.Bm[k * N + n], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Out[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * N + n] = acc[r];
    }
}

}  // namespace

extern "C" int mumpy_faf_fwd(const float* x, const float* D, const float* Dt, float* scratch, float* out, int B, int T,
                             int frame, int lo_hi, int mid_lo, int mid_hi, void* stream) {
    MUMPY_REQUIRE(x && D && Dt && scratch && out, MUMPY_ENULL, "faf: null pointer");
    MUMPY_REQUIRE(B > 0 && T > 0 && frame >= 0 && frame < T, MUMPY_EINVAL, "faf: frame %d outside clip of %d", frame, T);
    MUMPY_REQUIRE(lo_hi >= 0 && mid_lo >= 0 && mid_hi >= mid_lo && mid_hi < 2 * N, MUMPY_EINVAL, "faf: bad band limits");
    const int64_t P = (int64_t)N * N;
    FafArgs f;
    f.A = D; f.Bm = Dt; f.X = x + (int64_t)frame * 3 * P; f.Out = scratch;
    f.x_batch_stride = (int64_t)T * 3 * P; f.x_plane_stride = P;
    f.o_batch_stride = 3 * P; f.o_plane_stride = P; f.o_band_stride = 0;
    f.masked = 0; f.lo_hi = lo_hi; f.mid_lo = mid_lo; f.mid_hi = mid_hi;
    hipLaunchKernelGGL(faf_rowblock_kernel, dim3(7, B * 3, 1), dim3(256), 0, as_stream(stream), f);
    MUMPY_CHECK_LAUNCH("faf(forward)");
    FafArgs g;
    g.A = Dt; g.Bm = D; g.X = scratch; g.Out = out;
    g.x_batch_stride = 3 * P; g.x_plane_stride = P;
    g.o_batch_stride = 9 * P; g.o_plane_stride = P; g.o_band_stride = 3 * P;
    g.masked = 1; g.lo_hi = lo_hi; g.mid_lo = mid_lo; g.mid_hi = mid_hi;
    hipLaunchKernelGGL(faf_rowblock_kernel, dim3(7, B * 3, 3), dim3(256), 0, as_stream(stream), g);
    MUMPY_CHECK_LAUNCH("faf(inverse)");
    return 0;
}
