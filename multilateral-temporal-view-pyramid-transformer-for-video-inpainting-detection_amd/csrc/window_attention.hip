// window_attention.hip — Swin window attention core and the deformable cross-view attention+aggregation,
// one (window, head) unit per wave, everything in registers, fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces: window_partition / roll / window_reverse (swin:54-83, 273, 295) + the softmax(QK^T + bias + mask)V core
// of WindowAttention.forward (swin:145-163), and the attention + "(b t)->b t" sum of SwinDAttention (deform:360-395).
//
// Data flow per unit (49 tokens x 32 channels per operand, padded to 64 x 32 inside the wave):
//   * Q and K rows are loaded straight into MFMA operand layout: lane (r = lane&31, h = lane>>5) holds, for the
//     rows r and r+32, the 16 consecutive channels [16h, 16h+16) -> k-slot h of MFMA step s is channel 16h+s.
//     The window gather and the cyclic shift are nothing but the row address; nothing is staged or materialised.
//   * S^T = K Q^T is accumulated (key on the MFMA row, query on the lane), so a query's scores live in ONE lane
//     pair (lane, lane^32): softmax is 32 registers + one cross-half exchange, no LDS.
//   * the normalised P stays in the accumulator registers and is fed back as the A operand of P V (the
//     accumulator->operand trick: k-slot h of step (jt,g,e) is key 32jt+8g+4h+e, which is exactly the key the
//     lane's register 4g+e holds); V rows are loaded in that same key order, one dword per lane (128-B rows).
//   * keys >= 49 are masked by the -1e30 columns of the pre-padded bias; queries >= 49 are never stored.
// The unit is HBM-bound (12.25 FLOP/B): no LDS staging, ~25 KB in flight per wave, >= 8 waves per CU.
#include "common.h"
using namespace mumpy;

namespace {

struct SelfArgs {
    const float* qkv;
    float* out;
    const float* bias;      // (nH,64,64)
    const float* mask_tab;  // (nU,64,64) or null
    const int32_t* mask_id; // (n_mask) or null; window bw uses mask_id[bw % n_mask]
    int B, Hs, W, C, nH, shift, nWx, nW, n_mask;
    float scale;
    int64_t units;
};

struct CrossArgs {
    const float* q;        // (B, H*W, C) raster
    const float* kv;       // (B2w, 49, 2C) window-major
    const float* padmask;  // (1,64,64)
    float* out;            // (B1w, 49, C) window-major
    int B, H, W, C, nH, r, nWx, nWf, B1w;
    float scale;
    int64_t units;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// load the 16 channels [16h,16h+16) of one 32-channel head row, or zeros
__device__ __forceinline__ void load_frag(f32x4 (&f)[4], const float* row, bool valid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = valid ? *reinterpret_cast<const f32x4*>(row + 4 * i) : f32x4{0, 0, 0, 0};
}

// S^T += K Q^T for the 2x2 tiles; q already scaled
__device__ __forceinline__ void qk_product(f32x16 (&s)[2][2], const f32x4 (&kf)[2][4], const f32x4 (&qf)[2][4]) {
#pragma unroll
    for (int st = 0; st < 16; ++st) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int it = 0; it < 2; ++it) s[jt][it] = mfma32(kf[jt][st >> 2][st & 3], qf[it][st >> 2][st & 3], s[jt][it]);
    }
}

// add bias (+mask) rows and run the softmax over keys for the two query columns this lane owns
__device__ __forceinline__ void bias_softmax(f32x16 (&s)[2][2], const float* bias_h, const float* mask_w, int c, int h,
                                             float post_scale) {
    constexpr float NEG = -1e30f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int i = 32 * it + c;
        float m = NEG;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (jt == 1 && g == 3) {  // keys 56..63: all padding
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[jt][it][4 * g + e] = NEG;
                    continue;
                }
                const int off = i * 64 + 32 * jt + 8 * g + 4 * h;
                f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + off);
                if (mask_w) b += *reinterpret_cast<const f32x4*>(mask_w + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = s[jt][it][4 * g + e] * post_scale + b[e];
                    s[jt][it][4 * g + e] = v;
                    m = fmaxf(m, v);
                }
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __expf(s[jt][it][r] - m);
                s[jt][it][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[jt][it][r] *= inv;
    }
}

// the 25 (jt,g,e) MFMA steps of P V that can hold a key < 49; key of lane half h is 32jt+8g+4h+e
template <typename F>
__device__ __forceinline__ void for_pv_steps(F&& body) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (jt == 1 && (g == 3 || (g == 2 && e > 0))) continue;
                body(jt, g, e);
            }
}

template <typename VROW>
__device__ __forceinline__ void load_v(float (&vf)[2][16], VROW vrow, int c, int h) {
    for_pv_steps([&](int jt, int g, int e) {
        const int j = 32 * jt + 8 * g + 4 * h + e;
        vf[jt][4 * g + e] = (j < WT) ? vrow(j)[c] : 0.f;
    });
}

__device__ __forceinline__ void pv_product(f32x16 (&o)[2], const f32x16 (&s)[2][2], const float (&vf)[2][16]) {
    for_pv_steps([&](int jt, int g, int e) {
        o[0] = mfma32(s[jt][0][4 * g + e], vf[jt][4 * g + e], o[0]);
        o[1] = mfma32(s[jt][1][4 * g + e], vf[jt][4 * g + e], o[1]);
    });
}

template <typename OROW>
__device__ __forceinline__ void store_o(const f32x16 (&o)[2], OROW orow, int c, int h) {
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (it == 1 && (r >> 2) >= 2 && !((r >> 2) == 2 && (r & 3) == 0)) continue;  // statically >= 49
            if (i < WT) orow(i)[c] = o[it][r];
        }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void win_attn_self_kernel(SelfArgs a) {
    __shared__ int tok_tab[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= a.units) return;
    const int head = (int)(u % a.nH);
    const int64_t bw = u / a.nH;                 // window index over the batch
    const int n = (int)(bw % a.nW);
    const int64_t b = bw / a.nW;
    const int wy = n / a.nWx, wx = n - wy * a.nWx;
    int* tt = tok_tab[wave];
    tt[lane] = (lane < WT) ? window_token(wy, wx, lane, a.Hs, a.W, a.shift) : 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t L = (int64_t)a.Hs * a.W;
    const float* base = a.qkv + b * L * 3 * a.C + head * HD;
    const int rs = 3 * a.C;

    f32x4 qf[2][4], kf[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = 32 * t + c;
        const bool valid = p < WT;
        const float* row = base + (int64_t)tt[p & 63] * rs + 16 * h;
        load_frag(qf[t], row, valid);
        load_frag(kf[t], row + a.C, valid);
    }
    float vf[2][16];
    const float* vbase = base + 2 * a.C;
    load_v(vf, [&](int j) { return vbase + (int64_t)tt[j] * rs; }, c, h);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) qf[t][i] *= a.scale;   // q = q * scale before QK^T (swin:145)

    f32x16 s[2][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[jt][it][r] = 0.f;
    qk_product(s, kf, qf);

    const float* mask_w = nullptr;
    if (a.mask_id) {
        const int id = a.mask_id[(int)(bw % a.n_mask)];
        if (id >= 0) mask_w = a.mask_tab + (int64_t)id * 4096;
    }
    bias_softmax(s, a.bias + (int64_t)head * 4096, mask_w, c, h, 1.0f);

    f32x16 o[2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[it][r] = 0.f;
    pv_product(o, s, vf);
    float* obase = a.out + b * L * a.C + head * HD;
    store_o(o, [&](int i) { return obase + (int64_t)tt[i] * a.C; }, c, h);
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void win_attn_cross_kernel(CrossArgs a) {
    __shared__ int tok_tab[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= a.units) return;
    const int head = (int)(u % a.nH);
    const int64_t b1 = u / a.nH;                 // output window
    int* tt = tok_tab[wave];
    const int64_t L = (int64_t)a.H * a.W;
    f32x16 o[2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[it][r] = 0.f;

    for (int t = 0; t < a.r; ++t) {
        const int64_t b2 = b1 * a.r + t;                  // kv window; adjacent r-tuples are summed (deform:394-395)
        const int qw = (int)(b2 % a.B1w);                 // q window = kv window mod B1 (x1.repeat, deform:330)
        const int qb = qw / a.nWf, qn = qw - qb * a.nWf;
        const int wy = qn / a.nWx, wx = qn - wy * a.nWx;
        __builtin_amdgcn_wave_barrier();
        tt[lane] = (lane < WT) ? window_token(wy, wx, lane, a.H, a.W, 0) : 0;
        __builtin_amdgcn_wave_barrier();
        const float* qbase = a.q + ((int64_t)qb * L) * a.C + head * HD;
        const float* kbase = a.kv + b2 * WT * 2 * a.C + head * HD;
        f32x4 qf[2][4], kf[2][4];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int p = 32 * tl + c;
            const bool valid = p < WT;
            load_frag(qf[tl], qbase + (int64_t)tt[p & 63] * a.C + 16 * h, valid);
            load_frag(kf[tl], kbase + (int64_t)(valid ? p : 0) * 2 * a.C + 16 * h, valid);
        }
        float vf[2][16];
        const float* vbase = kbase + a.C;
        load_v(vf, [&](int j) { return vbase + (int64_t)j * 2 * a.C; }, c, h);

        f32x16 s[2][2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[jt][it][r] = 0.f;
        qk_product(s, kf, qf);
        bias_softmax(s, a.padmask, nullptr, c, h, a.scale);   // scale on the product (deform:364)
        pv_product(o, s, vf);
    }
    float* obase = a.out + b1 * WT * a.C + head * HD;
    store_o(o, [&](int i) { return obase + (int64_t)i * a.C; }, c, h);
}

}  // namespace

extern "C" int mumpy_window_attention_fwd(const float* qkv, float* out, const float* bias, const float* mask_tab,
                                          const int32_t* mask_id, int n_mask, int B, int Hs, int W, int C, int shift,
                                          float scale, void* stream) {
    MUMPY_REQUIRE(qkv && out && bias, MUMPY_ENULL, "window_attention: null pointer");
    MUMPY_REQUIRE((mask_tab == nullptr) == (mask_id == nullptr), MUMPY_ENULL,
                  "window_attention: mask_tab and mask_id must be given together");
    MUMPY_REQUIRE(aligned16(qkv) && aligned16(out) && aligned16(bias) && aligned16(mask_tab), MUMPY_EALIGN,
                  "window_attention: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && Hs > 0 && W > 0 && Hs % WS == 0 && W % WS == 0, MUMPY_EINVAL,
                  "window_attention: grid (%d,%d) not divisible by window 7", Hs, W);
    MUMPY_REQUIRE(C > 0 && C % HD == 0, MUMPY_EINVAL, "window_attention: C=%d not a multiple of head width 32", C);
    MUMPY_REQUIRE(shift >= 0 && shift < WS, MUMPY_EINVAL, "window_attention: shift=%d out of [0,7)", shift);
    MUMPY_REQUIRE(mask_id == nullptr || n_mask > 0, MUMPY_EINVAL, "window_attention: n_mask must be > 0 with a mask");
    SelfArgs a;
    a.qkv = qkv; a.out = out; a.bias = bias; a.mask_tab = mask_tab; a.mask_id = mask_id;
    a.B = B; a.Hs = Hs; a.W = W; a.C = C; a.nH = C / HD; a.shift = shift;
    a.nWx = W / WS; a.nW = (Hs / WS) * (W / WS); a.scale = scale; a.n_mask = n_mask > 0 ? n_mask : 1;
    a.units = (int64_t)B * a.nW * a.nH;
    const int64_t grid = (a.units + 3) / 4;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "window_attention: too many windows");
    hipLaunchKernelGGL(win_attn_self_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("window_attention");
    return 0;
}

extern "C" int mumpy_deform_attention_fwd(const float* q, const float* kv, const float* padmask, float* out, int B,
                                          int H, int W, int C, int r, float scale, void* stream) {
    MUMPY_REQUIRE(q && kv && padmask && out, MUMPY_ENULL, "deform_attention: null pointer");
    MUMPY_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(padmask) && aligned16(out), MUMPY_EALIGN,
                  "deform_attention: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && H % WS == 0 && W % WS == 0 && r >= 1, MUMPY_EINVAL,
                  "deform_attention: bad grid (%d,%d) or ratio %d", H, W, r);
    MUMPY_REQUIRE(C > 0 && C % HD == 0, MUMPY_EINVAL, "deform_attention: C=%d not a multiple of 32", C);
    CrossArgs a;
    a.q = q; a.kv = kv; a.padmask = padmask; a.out = out;
    a.B = B; a.H = H; a.W = W; a.C = C; a.nH = C / HD; a.r = r;
    a.nWx = W / WS; a.nWf = (H / WS) * (W / WS); a.B1w = B * a.nWf; a.scale = scale;
    a.units = (int64_t)a.B1w * a.nH;
    const int64_t grid = (a.units + 3) / 4;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "deform_attention: too many windows");
    hipLaunchKernelGGL(win_attn_cross_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("deform_attention");
    return 0;
}
