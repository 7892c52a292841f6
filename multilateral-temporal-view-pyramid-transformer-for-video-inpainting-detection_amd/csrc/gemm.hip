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
                                                     float* Y, int64_t M, int N, int K, int act, unsigned gn,
                                                     int ksplit, float* slab, int64_t rpb, int64_t bstride) {
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
    unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int ks = (int)(wgid % (unsigned)ksplit);       // K slice (split-K): slices of one tile run side by side
    wgid /= (unsigned)ksplit;
    const int64_t m0 = (int64_t)(wgid / gn) * BM;
    const int n0 = (int)(wgid % gn) * BN;
    const int kbeg = ks * (K / ksplit);

    const int ld_row = tid >> 3, ld_c4 = tid & 7;
    f32x4 areg[A_LD], breg[B_LD];

    // A rows may be strided in blocks (rows m of block m / rpb start at X + (m / rpb) * bstride): lets a caller feed
    // (B, t, n, C) tokens of one time slice as an (B*n, C) operand without a copy.  Dense: rpb = M.
    const float* arow[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int64_t m = m0 + ld_row + 32 * i;
        arow[i] = (m < M) ? X + (m / rpb) * bstride + (m % rpb) * K + 4 * ld_c4 : nullptr;
    }
    auto load_global = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            areg[i] = arow[i] ? *reinterpret_cast<const f32x4*>(arow[i] + k0) : f32x4{0, 0, 0, 0};
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

    const int nk = K / ksplit / BK;
    load_global(kbeg);
    store_lds(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_global(kbeg + (kc + 1) * BK);
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
    if (ksplit > 1) {       // raw partial sums -> slab[ks][M][N]; bias/act/residual happen in splitk_reduce_kernel
        float* S = slab + (int64_t)ks * M * N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + 32 * j + c;
            if (n >= N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + wm * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < M) S[m * N + n] = acc[i][j][r];
                }
        }
        return;
    }
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

// split-K combine: y = act(sum_s slab[s] + bias) + residual, slices summed in fixed order (bitwise reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                            const float* residual, float* Y, int64_t MN4, int N,
                                                            int ksplit, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < MN4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(slab)[i];
        for (int s = 1; s < ksplit; ++s) v += reinterpret_cast<const f32x4*>(slab)[(int64_t)s * MN4 + i];
        const int n = (int)((i * 4) % N);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (act == MUMPY_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (residual) v += reinterpret_cast<const f32x4*>(residual)[i];
        reinterpret_cast<f32x4*>(Y)[i] = v;
    }
}

struct Plan {
    int tile;    // 0: 128x128, 1: 128x96, 2: 64x64
    int ksplit;
    unsigned gn;
    int64_t gm;
};

// Shape heuristic (measured on MI355X, tools/gemm_shapes.py): 128-wide tiles need >= ~1.5 blocks per CU to pay;
// below that the 64x64 tile quadruples the block count, and if the grid is still small and K is deep, K is split
// so that every CU gets work (slices of >= 384).
Plan make_plan(int64_t M, int N, int K, bool allow_split) {
    Plan p;
    const bool n96 = (N % 128 != 0 && N % 96 == 0);
    const int64_t gm128 = (M + 127) / 128;
    const unsigned gn128 = n96 ? N / 96 : (N + 127) / 128;
    if (gm128 * gn128 >= 384) {
        p.tile = n96 ? 1 : 0; p.ksplit = 1; p.gn = gn128; p.gm = gm128;
        return p;
    }
    p.tile = 2; p.gm = (M + 63) / 64; p.gn = (N + 63) / 64; p.ksplit = 1;
    const int64_t blocks = p.gm * p.gn;
    if (allow_split && blocks < 512 && K >= 768) {
        int s = (int)((768 + blocks - 1) / blocks);
        if (s > K / 384) s = K / 384;
        if (s > 16) s = 16;
        while (s > 1 && (K % (32 * s)) != 0) --s;
        p.ksplit = s < 1 ? 1 : s;
    }
    return p;
}

int launch_linear(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M, int N,
                  int K, int act, float* ws, int64_t ws_bytes, hipStream_t s, int64_t rpb = 0, int64_t bstride = 0) {
    if (rpb <= 0) { rpb = M; bstride = 0; }
    Plan p = make_plan(M, N, K, ws != nullptr);
    if (p.ksplit > 1 && (int64_t)p.ksplit * M * N * (int64_t)sizeof(float) > ws_bytes) p.ksplit = 1;
    const int64_t grid = p.gm * p.gn * p.ksplit;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "linear: too many tiles");
#define MUMPY_GEMM(BM_, BN_, WM_, WN_)                                                                               \
    hipLaunchKernelGGL((linear_kernel<BM_, BN_, WM_, WN_>), dim3((unsigned)grid), dim3(256), 0, s, x, W, bias, residual, \
                       y, M, N, K, act, p.gn, p.ksplit, ws, rpb, bstride)
    if (p.tile == 0) MUMPY_GEMM(128, 128, 64, 64);
    else if (p.tile == 1) MUMPY_GEMM(128, 96, 32, 96);
    else MUMPY_GEMM(64, 64, 32, 32);
#undef MUMPY_GEMM
    MUMPY_CHECK_LAUNCH("linear");
    if (p.ksplit > 1) {
        const int64_t mn4 = M * N / 4;
        int64_t g = (mn4 + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, ws, bias, residual, y, mn4, N,
                           p.ksplit, act);
        MUMPY_CHECK_LAUNCH("linear(split-K reduce)");
    }
    return 0;
}

}  // namespace

static int check_linear_args(const float* x, const float* W, const float* residual, const float* y, int64_t M, int N,
                             int K, int act) {
    MUMPY_REQUIRE(x && W && y, MUMPY_ENULL, "linear: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(W) && aligned16(y) && aligned16(residual), MUMPY_EALIGN,
                  "linear: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 32 == 0, MUMPY_EINVAL,
                  "linear: need K %% 32 == 0 and N %% 32 == 0 (got M=%lld N=%d K=%d)", (long long)M, N, K);
    MUMPY_REQUIRE(act == MUMPY_ACT_NONE || act == MUMPY_ACT_GELU, MUMPY_EINVAL, "linear: unknown act %d", act);
    return 0;
}

extern "C" int mumpy_linear_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                                int64_t M, int N, int K, int act, void* stream) {
    if (M == 0) return 0;      // empty batch
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    return launch_linear(x, W, bias, residual, y, M, N, K, act, nullptr, 0, as_stream(stream));
}

extern "C" int64_t mumpy_linear_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK) return 0;
    const Plan p = make_plan(M, N, K, true);
    return p.ksplit > 1 ? (int64_t)p.ksplit * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int mumpy_linear_ws_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                                   int64_t M, int N, int K, int act, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    if (M == 0) return 0;
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    MUMPY_REQUIRE(aligned16(workspace), MUMPY_EALIGN, "linear: workspace must be 16-byte aligned");
    return launch_linear(x, W, bias, residual, y, M, N, K, act, static_cast<float*>(workspace),
                         workspace ? workspace_bytes : 0, as_stream(stream));
}

extern "C" int mumpy_linear_rows_fwd(const float* x, int64_t rows_per_block, int64_t block_stride, const float* W,
                                     const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    if (M == 0) return 0;
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    MUMPY_REQUIRE(rows_per_block > 0 && M % rows_per_block == 0 && block_stride % 4 == 0, MUMPY_EINVAL,
                  "linear_rows: M=%lld must be a multiple of rows_per_block=%lld and block_stride %% 4 == 0",
                  (long long)M, (long long)rows_per_block);
    return launch_linear(x, W, bias, residual, y, M, N, K, act, static_cast<float*>(workspace),
                         workspace ? workspace_bytes : 0, as_stream(stream), rows_per_block, block_stride);
}
