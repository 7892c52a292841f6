// gemm.hip — y = act(x W^T + bias) + residual in fp32 on v_mfma_f32_32x32x2_f32 (exact f32 FMA chain).
//
// Replaces every nn.Linear / 1x1 conv on the path (swin:46-49,142,164,365; blocks:27-33,57-71; deform:333,361-362,402;
// mTVE:283,740).  Both operands are K-contiguous (x is (M,K), nn.Linear's W is (N,K)), so A and B fragments are
// read the same way: lane (r = lane&31, h = lane>>5) owns row r of a 32-row tile and, per 32-deep K chunk, the 16
// consecutive k's [16h, 16h+16) -> four ds_read_b128; k-slot h of MFMA step s is k = 16h+s for A and B alike.
//
// Block = 4 waves, tile BM x BN x 32, LDS rows padded to 36 dwords (16 consecutive rows hit 16 distinct 16-B slots
// of the 64-bank row: conflict-free ds_read_b128), two LDS buffers, next chunk's global loads (16 B per lane, 128-B
// row segments) in flight under the MFMAs, one barrier per chunk.  Two tile shapes cover every N of the model
// exactly: 128x128 (waves 2x2, 64x64 each) and 128x96 (waves 4x1, 32x96 each) for the 96*2^s widths of views 1/2.
// The epilogue (bias, exact-erf GELU, residual add) runs on the accumulators; stores are 128-B row segments.
#include "common.h"
using namespace mumpy;

namespace {

constexpr int BK = 32;
constexpr int LDR = 36;  // LDS row stride in dwords

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                     const float* __restrict__ bias, const float* residual,
                                                     float* Y, int64_t M, int N, int K, int act, unsigned gn) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    constexpr int A_LD = BM * 8 / 256;  // float4 loads per thread per chunk
    constexpr int B_LD = BN * 8 / 256;
    __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LDR];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // 1-D grid, XCD-aware: blocks are dealt round-robin over the 8 XCDs, so remap the id such that each XCD owns a
    // contiguous run of tiles, N fastest -> the gn tiles that share an x row-panel (and the W panels, which are
    // small) are served by ONE L2 instead of eight (bijective form, cdna guide T1).
    const unsigned nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int64_t m0 = (int64_t)(wgid / gn) * BM;
    const int n0 = (int)(wgid % gn) * BN;

    const int ld_row = tid >> 3, ld_c4 = tid & 7;
    f32x4 areg[A_LD], breg[B_LD];

    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int64_t m = m0 + ld_row + 32 * i;
            areg[i] = (m < M) ? *reinterpret_cast<const f32x4*>(X + m * K + k0 + 4 * ld_c4) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int n = n0 + ld_row + 32 * i;
            breg[i] = (n < N) ? *reinterpret_cast<const f32x4*>(Wt + (int64_t)n * K + k0 + 4 * ld_c4) : f32x4{0, 0, 0, 0};
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            *reinterpret_cast<f32x4*>(&lds[buf][(ld_row + 32 * i) * LDR + 4 * ld_c4]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            *reinterpret_cast<f32x4*>(&lds[buf][(BM + ld_row + 32 * i) * LDR + 4 * ld_c4]) = breg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = K / BK;
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_global((kc + 1) * BK);
        f32x4 af[TM][4], bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                af[i][q] = *reinterpret_cast<const f32x4*>(&lds[buf][(wm * WM + 32 * i + c) * LDR + 16 * h + 4 * q]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                bf[j][q] = *reinterpret_cast<const f32x4*>(&lds[buf][(BM + wn * WN + 32 * j + c) * LDR + 16 * h + 4 * q]);
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s >> 2][s & 3], bf[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
        if (kc + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + 32 * j + c;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= M) continue;
                float v = acc[i][j][r] + bv;
                if (act == MUMPY_ACT_GELU) v = gelu_erf(v);
                if (residual) v += residual[m * N + n];
                Y[m * N + n] = v;
            }
    }
}

}  // namespace

extern "C" int mumpy_linear_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                                int64_t M, int N, int K, int act, void* stream) {
    if (M == 0) return 0;      // empty batch
    MUMPY_REQUIRE(x && W && y, MUMPY_ENULL, "linear: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(W) && aligned16(y) && aligned16(residual), MUMPY_EALIGN,
                  "linear: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 32 == 0, MUMPY_EINVAL,
                  "linear: need K %% 32 == 0 and N %% 32 == 0 (got M=%lld N=%d K=%d)", (long long)M, N, K);
    MUMPY_REQUIRE(act == MUMPY_ACT_NONE || act == MUMPY_ACT_GELU, MUMPY_EINVAL, "linear: unknown act %d", act);
    if (M == 0) return 0;
    const int64_t gm = (M + 127) / 128;
    const bool n96 = (N % 128 != 0 && N % 96 == 0);
    const unsigned gn = n96 ? N / 96 : (N + 127) / 128;
    MUMPY_REQUIRE(gm * gn < (1ll << 31), MUMPY_ERANGE, "linear: too many tiles");
    hipStream_t s = as_stream(stream);
    if (n96) {
        hipLaunchKernelGGL((linear_kernel<128, 96, 32, 96>), dim3((unsigned)(gm * gn)), dim3(256), 0, s, x, W, bias,
                           residual, y, M, N, K, act, gn);
    } else {
        hipLaunchKernelGGL((linear_kernel<128, 128, 64, 64>), dim3((unsigned)(gm * gn)), dim3(256), 0, s, x, W, bias,
                           residual, y, M, N, K, act, gn);
    }
    MUMPY_CHECK_LAUNCH("linear");
    return 0;
}
