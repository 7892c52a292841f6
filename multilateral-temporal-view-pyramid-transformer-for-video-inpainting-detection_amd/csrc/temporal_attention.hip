// temporal_attention.hip — attention of the 12 global ViT blocks (blocks:57-71).  Under vmap(in_dims=2) (mTVE:741)
// every one of the B*49 spatial sites is its own sequence of T <= 16 temporal tokens, 12 heads of width 64: the
// score matrix is T x T.  One wave per (site, head), lane = channel: q/k/v rows are 256-B coalesced loads, a score is
// a 64-lane reduction, softmax and P V are per-lane scalar work.  Negligible FLOPs; the point is one launch and no
// (S,heads,T,T) tensors in HBM.
#include "common.h"
using namespace mumpy;

namespace {
constexpr int TMAX = 16;

// Tq <= T: only the first Tq temporal tokens are queries (out is (S, Tq, C)); keys / values are all T tokens.  The encoder tail keeps
// temporal slices 0..2 only (mTVE:745), so the LAST global block needs no output for t >= 3.
__global__ __launch_bounds__(256) void temporal_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int64_t units, int T, int Tq, int C, int heads, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= units) return;
    const int head = (int)(u % heads);
    const int64_t s = u / heads;
    const float* base = qkv + s * T * 3 * C + head * 64 + lane;
    float q[TMAX], k[TMAX], v[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
            const float* r = base + (int64_t)t * 3 * C;
            q[t] = r[0];
            k[t] = r[C];
            v[t] = r[2 * C];
        }
    }
    float* ob = out + s * Tq * C + head * 64 + lane;
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t >= Tq) continue;
        float sc[TMAX];
        float m = -3.0e38f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            if (j < T) {
                sc[j] = wave_sum(q[t] * k[j], 64) * scale;     // scale on the product (blocks:66)
                m = fmaxf(m, sc[j]);
            }
        }
        float sum = 0.f, o = 0.f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j) {
            if (j < T) {
                const float e = __expf(sc[j] - m);
                sum += e;
                o = fmaf(e, v[j], o);
            }
        }
        ob[(int64_t)t * C] = o / sum;
    }
}

// backward: per (site, head) the T x T probabilities are recomputed; with g = P o (dP - rowsum(P o dP)), dP = dO V^T:
//   dQ = scale g K,  dK = scale g^T Q,  dV = P^T dO.   Same one-wave-per-unit layout (lane = channel).
__global__ __launch_bounds__(256) void temporal_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, int64_t units, int T, int C, int heads,
                                                                float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= units) return;
    const int head = (int)(u % heads);
    const int64_t s = u / heads;
    const float* base = qkv + s * T * 3 * C + head * 64 + lane;
    const float* dob = dout + s * T * C + head * 64 + lane;
    float q[TMAX], k[TMAX], v[TMAX], d_o[TMAX], dq[TMAX], dk[TMAX], dv[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        dq[t] = 0.f; dk[t] = 0.f; dv[t] = 0.f;
        if (t < T) {
            const float* r = base + (int64_t)t * 3 * C;
            q[t] = r[0]; k[t] = r[C]; v[t] = r[2 * C];
            d_o[t] = dob[(int64_t)t * C];
        }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
        if (t >= T) continue;
        float p[TMAX], dp[TMAX];
        float m = -3.0e38f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j)
            if (j < T) {
                p[j] = wave_sum(q[t] * k[j], 64) * scale;
                dp[j] = wave_sum(d_o[t] * v[j], 64);
                m = fmaxf(m, p[j]);
            }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j)
            if (j < T) { p[j] = __expf(p[j] - m); sum += p[j]; }
        const float inv = 1.0f / sum;
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < TMAX; ++j)
            if (j < T) { p[j] *= inv; dsum += p[j] * dp[j]; }
#pragma unroll
        for (int j = 0; j < TMAX; ++j)
            if (j < T) {
                const float g = p[j] * (dp[j] - dsum) * scale;
                dq[t] = fmaf(g, k[j], dq[t]);
                dk[j] = fmaf(g, q[t], dk[j]);
                dv[j] = fmaf(p[j], d_o[t], dv[j]);
            }
    }
    float* ob = dqkv + s * T * 3 * C + head * 64 + lane;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) {
            float* r = ob + (int64_t)t * 3 * C;
            r[0] = dq[t]; r[C] = dk[t]; r[2 * C] = dv[t];
        }
}
}  // namespace

extern "C" int mumpy_temporal_attention_bwd(const float* qkv, const float* dout, float* dqkv, int64_t S, int T, int C, int heads,
                                            float scale, void* stream) {
    MUMPY_REQUIRE(qkv && dout && dqkv, MUMPY_ENULL, "temporal_attention_bwd: null pointer");
    MUMPY_REQUIRE(S >= 0 && T >= 1 && T <= TMAX, MUMPY_ERANGE, "temporal_attention_bwd: T=%d outside 1..16", T);
    MUMPY_REQUIRE(heads > 0 && C == heads * 64, MUMPY_EINVAL, "temporal_attention_bwd: need head width 64 (C=%d heads=%d)", C, heads);
    if (S == 0) return 0;
    const int64_t units = S * heads;
    const int64_t grid = (units + 3) / 4;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "temporal_attention_bwd: too many sites");
    hipLaunchKernelGGL(temporal_attn_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), qkv, dout, dqkv, units, T, C,
                       heads, scale);
    MUMPY_CHECK_LAUNCH("temporal_attention_bwd");
    return 0;
}

extern "C" int mumpy_temporal_attention_q_fwd(const float* qkv, float* out, int64_t S, int T, int Tq, int C, int heads, float scale,
                                              void* stream);
extern "C" int mumpy_temporal_attention_fwd(const float* qkv, float* out, int64_t S, int T, int C, int heads,
                                            float scale, void* stream) {
    return mumpy_temporal_attention_q_fwd(qkv, out, S, T, T, C, heads, scale, stream);
}

extern "C" int mumpy_temporal_attention_q_fwd(const float* qkv, float* out, int64_t S, int T, int Tq, int C, int heads, float scale,
                                              void* stream) {
    MUMPY_REQUIRE(qkv && out, MUMPY_ENULL, "temporal_attention: null pointer");
    MUMPY_REQUIRE(S >= 0 && T >= 1 && T <= TMAX, MUMPY_ERANGE, "temporal_attention: T=%d outside 1..16", T);
    MUMPY_REQUIRE(Tq >= 1 && Tq <= T, MUMPY_EINVAL, "temporal_attention: Tq=%d outside 1..T=%d", Tq, T);
    MUMPY_REQUIRE(heads > 0 && C == heads * 64, MUMPY_EINVAL, "temporal_attention: need head width 64 (C=%d heads=%d)", C, heads);
    if (S == 0) return 0;
    const int64_t units = S * heads;
    const int64_t grid = (units + 3) / 4;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "temporal_attention: too many sites");
    hipLaunchKernelGGL(temporal_attn_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), qkv, out, units, T, Tq, C,
                       heads, scale);
    MUMPY_CHECK_LAUNCH("temporal_attention");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Attention MAPS (visualisation path only: `return_attention=True` of blocks.Block / SwinDAttention / CVAModule, blocks:85-87,
// deform:364-396, mTVE:134-137 -- no caller of the forward or training path asks for them, and the fast kernels keep the
// probabilities in registers).  P = softmax(scale * q k^T) per unit (outer index o, head h): Nq, Nk <= 64, head width D <= 64.
// One wave per unit, lane = key: the key row sits in registers, a query's scores are one value per lane, max / sum are wave
// reductions.  q / k are addressed by strides so that the same kernel reads the temporal qkv tensor and the deformable q / kv.
namespace {
struct ProbsArgs {
    const float* q; const float* k; float* out;
    int64_t units;             // outer * heads
    int heads, nq, nk, d;
    int64_t q_outer, q_row, k_outer, k_row;    // strides in floats; head h adds h * d
    int64_t q_mod;             // q outer index = outer % q_mod (deform:330: kv window i pairs with q window i mod B1)
    float scale;
};

__global__ __launch_bounds__(256) void attention_probs_kernel(ProbsArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= a.units) return;
    const int h = (int)(u % a.heads);
    const int64_t o = u / a.heads;
    const float* qb = a.q + (o % a.q_mod) * a.q_outer + (int64_t)h * a.d;
    const float* kb = a.k + o * a.k_outer + (int64_t)h * a.d;
    float kr[64];
    const bool live = lane < a.nk;
#pragma unroll
    for (int c = 0; c < 64; ++c) kr[c] = (live && c < a.d) ? kb[(int64_t)lane * a.k_row + c] : 0.f;
    float* ob = a.out + u * a.nq * a.nk;
    for (int i = 0; i < a.nq; ++i) {
        const float* qr = qb + (int64_t)i * a.q_row;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c)
            if (c < a.d) s = fmaf(qr[c], kr[c], s);             // (qr[c] is wave-uniform: scalar loads)
        s = live ? s * a.scale : -3.0e38f;
        float m = s;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float e = live ? __expf(s - m) : 0.f;
        const float sum = wave_sum(e, 64);
        if (live) ob[(int64_t)i * a.nk + lane] = e / sum;
    }
}
}  // namespace

extern "C" int mumpy_attention_probs_fwd(const float* q, const float* k, float* out, int64_t outer, int heads, int nq, int nk, int d,
                                         int64_t q_outer_stride, int64_t q_row_stride, int64_t k_outer_stride, int64_t k_row_stride,
                                         int64_t q_mod, float scale, void* stream) {
    MUMPY_REQUIRE(q && k && out, MUMPY_ENULL, "attention_probs: null pointer");
    MUMPY_REQUIRE(outer >= 0 && heads > 0 && nq >= 1 && nq <= 64 && nk >= 1 && nk <= 64 && d >= 1 && d <= 64 && q_mod >= 1, MUMPY_EINVAL,
                  "attention_probs: need 1 <= nq, nk, d <= 64 (got %d, %d, %d)", nq, nk, d);
    if (outer == 0) return 0;
    ProbsArgs a{q, k, out, outer * heads, heads, nq, nk, d, q_outer_stride, q_row_stride, k_outer_stride, k_row_stride, q_mod, scale};
    const int64_t grid = (a.units + 3) / 4;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "attention_probs: too many units");
    hipLaunchKernelGGL(attention_probs_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("attention_probs");
    return 0;
}
