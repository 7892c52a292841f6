// gemm_rd.hip — "background" GEMM: y = act(x W^T + bias) + residual on v_mfma_f32_32x32x2_f32 with NO LDS at all (round 3).
//
// Why it exists.  The persistent GEMM of gemm_ws.h holds one 768-thread workgroup per CU and that CU's WHOLE LDS (163,840 B), so no
// kernel that allocates even one byte of LDS can become resident beside it: on the two-stream probe tools/coresidency_probe.py the
// 16 KB window-attention kernel, the tiled GEMM (36.8 KB) and the 64x64 persistent GEMM (64 KB) all ran strictly AFTER the big
// GEMM (together = sum), while an LDS-free kernel (LayerNorm) overlapped (profiles/r03_coresidency_probe.txt).  In a pyramid stage
// the chains of views 1 and 2 (mid-size GEMMs: M = B*196 rows at stage 2) are forked beside view 3's chain of persistent GEMMs,
// and of their 2.3 ms only 1.0 ms was hidden (tools/phase_timeline.py).  This kernel is the GEMM for those chains: operands go
// global -> registers in MFMA operand layout (lane (c, h) of a 32-row block reads the 16 B at row c, k = 16 h + 4 q of a 32-deep
// chunk, four such loads per chunk and block -- the scheme of the window-attention kernel), so a workgroup needs registers and
// wave slots only (<= 128 VGPRs: the fourth wave of a SIMD whose other three are the persistent kernel's roles) and can live in
// the matrix-pipe bubbles of the big kernel (which leaves ~25 % of the pipe idle).
// It is NOT the fast GEMM when it runs alone: nothing is shared between waves (each re-reads its operand panels through L1 / L2)
// and one wave per SIMD is latency-bound -- the planner never picks it, the caller asks for it (ops.background()).
// Wave tile 32 x 64, workgroup = 2 x 2 waves = 64 x 128; no barriers, waves past the matrix edge just leave.
#include "common.h"
using namespace mumpy;

namespace {

struct RdArgs {
    const float* X; const float* W; const float* bias; const float* residual; float* Y;
    int M, N, K, act;
    unsigned gn;            // workgroup tiles along N
};

__global__ __launch_bounds__(256, 4) void gemm_rd_kernel(RdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const unsigned tm = blockIdx.x / a.gn, tn = blockIdx.x - tm * a.gn;
    const int m0 = (int)tm * 64 + wm * 32, n0 = (int)tn * 128 + wn * 64;
    if (m0 >= a.M || n0 >= a.N) return;
    const int K = a.K;
    int mr = m0 + c;
    if (mr > a.M - 1) mr = a.M - 1;                 // rows past the edge: clamped, never stored
    const float* arow = a.X + (int64_t)mr * K + 16 * h;
    const float* brow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int nr = n0 + 32 * j + c;
        if (nr > a.N - 1) nr = a.N - 1;
        brow[j] = a.W + (int64_t)nr * K + 16 * h;
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // half-chunk = the two 16-byte pieces q = 2 half, 2 half + 1 of every block: k = k0 + 16 h + 8 half + {0..7}
    f32x4 fa0[2], fb0[2][2], fa1[2], fb1[2][2];
    auto load_half = [&](int k0, int half, f32x4 (&fa)[2], f32x4 (&fb)[2][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            fa[q] = *reinterpret_cast<const f32x4*>(arow + k0 + 8 * half + 4 * q);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j][q] = *reinterpret_cast<const f32x4*>(brow[j] + k0 + 8 * half + 4 * q);
        }
    };
    auto mma_half = [&](const f32x4 (&fa)[2], const f32x4 (&fb)[2][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][e], fb[j][q][e], acc[j], 0, 0, 0);
    };
    load_half(0, 0, fa0, fb0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        load_half(k0, 1, fa1, fb1);
        mma_half(fa0, fb0);
        if (k0 + 32 < K) load_half(k0 + 32, 0, fa0, fb0);
        mma_half(fa1, fb1);
    }
    // D[row][col]: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 32 * j + c;
        if (n >= a.N) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (m >= a.M) continue;
            float v = acc[j][r] + bv;
            if (a.act == MUMPY_ACT_GELU) v = gelu_erf(v);
            const int64_t o = (int64_t)m * a.N + n;
            if (a.residual) v += a.residual[o];
            a.Y[o] = v;
        }
    }
}

}  // namespace

extern "C" int mumpy_linear_rd_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M,
                                   int N, int K, int act, void* stream) {
    if (M == 0) return 0;
    MUMPY_REQUIRE(x && W && y, MUMPY_ENULL, "linear_rd: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(W), MUMPY_EALIGN, "linear_rd: x and W must be 16-byte aligned");
    MUMPY_REQUIRE(M > 0 && M < (1ll << 31) - 64 && N > 0 && K > 0 && K % 32 == 0 && N % 32 == 0, MUMPY_EINVAL,
                  "linear_rd: need K %% 32 == 0 and N %% 32 == 0 (got M=%lld N=%d K=%d)", (long long)M, N, K);
    MUMPY_REQUIRE(act == MUMPY_ACT_NONE || act == MUMPY_ACT_GELU, MUMPY_EINVAL, "linear_rd: fp32 arithmetic only, act %d", act);
    RdArgs a;
    a.X = x; a.W = W; a.bias = bias; a.residual = residual; a.Y = y;
    a.M = (int)M; a.N = N; a.K = K; a.act = act;
    a.gn = (unsigned)((N + 127) / 128);
    const int64_t grid = ((M + 63) / 64) * a.gn;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "linear_rd: too many tiles");
    hipLaunchKernelGGL(gemm_rd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("linear_rd");
    return 0;
}
