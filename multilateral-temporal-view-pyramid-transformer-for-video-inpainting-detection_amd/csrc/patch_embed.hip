// patch_embed.hip — tokenizer: Conv3d(3 -> C, kernel = stride = (t,4,4)) + LayerNorm(C) as an implicit GEMM
// (mTVE:605-618).  Non-overlapping patches: the conv is out[m][n] = sum_k patch[m][k] Wt[k][n] with
// k = ((cin*t + dt)*4 + dy)*4 + dx, K = 48 t.  A block owns 64 consecutive tokens: it gathers their patches into LDS
// with 16-byte loads (dx runs 0..3 = one float4; consecutive tokens are consecutive float4s of an image row, so the
// gather is coalesced), multiplies on v_mfma_f32_32x32x2_f32 against W^T streamed from L2, then adds the bias and
// applies the LayerNorm on the tile before one coalesced store.  Output is token-major with frames stacked on the
// token axis: (B, t_out*56*56, C) — the layout every later kernel consumes (mTVE:614, 701-708).
#include "common.h"
using namespace mumpy;

namespace {

constexpr int TM = 64;   // tokens per block

__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out, int T,
                                                          int H, int W, int t, int C, int K, int t_out, int64_t Mtot,
                                                          float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lda = K + 1;          // odd stride: conflict-free column reads
    const int ldo = C + 4;          // output tile pitch: 16-byte aligned rows; 16 tokens x 4 lanes of the statistics pass hit 64 banks
    float* As = sm;                 // [64][K+1]; reused as the output tile [64][C+1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int Hp = H >> 2, Wp = W >> 2;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int kq = K >> 2;          // float4 chunks per patch: (cin, dt, dy)
    // a thread stages ONE token (ml = tid % 64: the stride 256 of the loop is a multiple of 64), so the token's image
    // coordinates are decoded once -- the 64-bit divisions were paid per staged float4 before (15x per thread at t = 5)
    {
        const int ml = tid & (TM - 1);
        const int64_t m = m0 + ml;
        const bool live = m < Mtot;
        const float* src = x;
        if (live) {
            const int hx = (int)(m % Wp);
            int64_t r = m / Wp;
            const int hy = (int)(r % Hp);
            r /= Hp;
            const int to = (int)(r % t_out);
            const int64_t b = r / t_out;
            src = x + (((b * T + to * t) * 3) * H + 4 * hy) * (int64_t)W + 4 * hx;      // (frame to*t, cin 0, row 4 hy)
        }
        const int64_t plane = (int64_t)H * W;                      // one (frame, channel) image
        float* drow = As + ml * lda;
        // loads in batches of up to 5 (all issued before the first LDS write: one memory latency per batch, not per load;
        // padding tokens of the last block re-read token 0's patch and are never stored)
        for (int q0 = tid >> 6; q0 < kq; q0 += 20) {                // q = (cin*t + dt)*4 + dy, this thread's q step is 4
            f32x4 v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int q = q0 + 4 * j;
                const int qq = q < kq ? q : q0;
                const int dy = qq & 3, ct = qq >> 2;
                const int dt = ct % t, cin = ct / t;
                v[j] = *reinterpret_cast<const f32x4*>(src + (dt * 3 + cin) * plane + dy * W);
            }
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int q = q0 + 4 * j;
                if (q < kq) {
                    float* dst = drow + 4 * q;
                    dst[0] = v[j].x; dst[1] = v[j].y; dst[2] = v[j].z; dst[3] = v[j].w;
                }
            }
        }
    }
    __syncthreads();
    const int ntile = C >> 5;       // 3 or 4 column tiles; wave w owns tile w
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (wave < ntile) {
        const int n = wave * 32 + c;
#pragma unroll 4
        for (int kk = 0; kk < K / 2; ++kk) {
            const int k = 2 * kk + h;
            const float wv = Wt[k * C + n];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[c * lda + k], wv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(32 + c) * lda + k], wv, acc[1], 0, 0, 0);
        }
    }
    __syncthreads();                // everyone is done reading As
    if (wave < ntile) {
        const int n = wave * 32 + c;
        const float bv = bias[n];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) As[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * h) * ldo + n] = acc[i][r] + bv;
    }
    __syncthreads();
    // LayerNorm per token.  Statistics: FOUR lanes per token (lane qd walks channels qd, qd + 4, ...), partial sums combined with two
    // quad shuffles; then a coalesced normalise-and-store pass, four channels per lane.  (The first version gave a wave to a token
    // and reduced across its 64 lanes twice per token -- 16 tokens x 12 cross-lane steps per wave were most of the kernel's time.)
    float* st = sm + TM * ldo;      // [64][2]: mean, rstd (behind the output tile)
    {
        const int ml = tid >> 2, qd = tid & 3;
        const float* row = As + ml * ldo;
        float s = 0.f;
        for (int ch = qd; ch < C; ch += 4) s += row[ch];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mean = s / (float)C;
        float q = 0.f;
        for (int ch = qd; ch < C; ch += 4) { const float d = row[ch] - mean; q = fmaf(d, d, q); }
        q += __shfl_xor(q, 1);
        q += __shfl_xor(q, 2);
        if (qd == 0) { st[2 * ml] = mean; st[2 * ml + 1] = rsqrtf(q / (float)C + eps); }
    }
    __syncthreads();
    const int c4n = C >> 2;
    for (int idx = tid; idx < TM * c4n; idx += 256) {
        const int ml = idx / c4n, c4 = idx - ml * c4n;
        const int64_t m = m0 + ml;
        if (m >= Mtot) break;
        const f32x4 v = *reinterpret_cast<const f32x4*>(As + ml * ldo + 4 * c4);
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + 4 * c4), b4 = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
        const float mean = st[2 * ml], rstd = st[2 * ml + 1];
        *reinterpret_cast<f32x4*>(out + m * C + 4 * c4) = (v - mean) * rstd * g4 + b4;
    }
}

}  // namespace

extern "C" int mumpy_patch_embed_fwd(const float* x, const float* Wt, const float* bias, const float* gamma,
                                     const float* beta, float* out, int B, int T, int H, int W, int t, int C, float eps,
                                     void* stream) {
    MUMPY_REQUIRE(x && Wt && bias && gamma && beta && out, MUMPY_ENULL, "patch_embed: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(out) && aligned16(gamma) && aligned16(beta), MUMPY_EALIGN,
                  "patch_embed: x, out, gamma and beta must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && T > 0 && t > 0 && t <= T && H % 4 == 0 && W % 4 == 0, MUMPY_EINVAL,
                  "patch_embed: bad clip shape T=%d t=%d H=%d W=%d", T, t, H, W);
    MUMPY_REQUIRE(C % 32 == 0 && C >= 32 && C <= 128, MUMPY_ERANGE, "patch_embed: C=%d must be 32..128, multiple of 32", C);
    const int K = 48 * t;
    MUMPY_REQUIRE(K <= 1024, MUMPY_ERANGE, "patch_embed: tubelet %d too long", t);
    const int t_out = (T - t) / t + 1;
    const int64_t Mtot = (int64_t)B * t_out * (H / 4) * (W / 4);
    const int ld = (K + 1 > C + 4 ? K + 1 : C + 4);
    const size_t lds = ((size_t)TM * ld + 2 * TM) * sizeof(float);     // patch / output tile + the (mean, rstd) pairs
    MUMPY_REQUIRE(lds <= 160 * 1024, MUMPY_ERANGE, "patch_embed: tile needs %zu B of LDS (t=%d too long)", lds, t);
    if (lds > 64 * 1024) {   // long tubelets (T = 9): raise the dynamic-LDS cap of this kernel (idempotent)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(patch_embed_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        MUMPY_REQUIRE(e == hipSuccess, (int)e, "patch_embed: cannot raise dynamic LDS to %zu B", lds);
    }
    const int64_t grid = (Mtot + TM - 1) / TM;
    hipLaunchKernelGGL(patch_embed_kernel, dim3((unsigned)grid), dim3(256), lds, as_stream(stream), x, Wt, bias, gamma,
                       beta, out, T, H, W, t, C, K, t_out, Mtot, eps);
    MUMPY_CHECK_LAUNCH("patch_embed");
    return 0;
}
