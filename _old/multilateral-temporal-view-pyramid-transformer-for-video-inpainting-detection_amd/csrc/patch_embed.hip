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
    const int ldo = C + 1;
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
    // LayerNorm per token: 16 tokens per wave, lanes over channels (C <= 128: two per lane)
    for (int ml = wave; ml < TM; ml += 4) {
        const int64_t m = m0 + ml;
        if (m >= Mtot) break;
        const float v0 = (lane < C) ? As[ml * ldo + lane] : 0.f;
        const float v1 = (lane + 64 < C) ? As[ml * ldo + lane + 64] : 0.f;
        const float mean = wave_sum(v0 + v1, 64) / (float)C;
        const float d0 = (lane < C) ? v0 - mean : 0.f, d1 = (lane + 64 < C) ? v1 - mean : 0.f;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1, 64) / (float)C + eps);
        float* o = out + m * C;
        if (lane < C) o[lane] = d0 * rstd * gamma[lane] + beta[lane];
        if (lane + 64 < C) o[lane + 64] = d1 * rstd * gamma[lane + 64] + beta[lane + 64];
    }
}

}  // namespace

extern "C" int mumpy_patch_embed_fwd(const float* x, const float* Wt, const float* bias, const float* gamma,
                                     const float* beta, float* out, int B, int T, int H, int W, int t, int C, float eps,
                                     void* stream) {
    MUMPY_REQUIRE(x && Wt && bias && gamma && beta && out, MUMPY_ENULL, "patch_embed: null pointer");
    MUMPY_REQUIRE(aligned16(x), MUMPY_EALIGN, "patch_embed: x must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && T > 0 && t > 0 && t <= T && H % 4 == 0 && W % 4 == 0, MUMPY_EINVAL,
                  "patch_embed: bad clip shape T=%d t=%d H=%d W=%d", T, t, H, W);
    MUMPY_REQUIRE(C % 32 == 0 && C >= 32 && C <= 128, MUMPY_ERANGE, "patch_embed: C=%d must be 32..128, multiple of 32", C);
    const int K = 48 * t;
    MUMPY_REQUIRE(K <= 1024, MUMPY_ERANGE, "patch_embed: tubelet %d too long", t);
    const int t_out = (T - t) / t + 1;
    const int64_t Mtot = (int64_t)B * t_out * (H / 4) * (W / 4);
    const int ld = (K > C ? K : C) + 1;
    const size_t lds = (size_t)TM * ld * sizeof(float);
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
