// common.h — shared helpers for the libmumpy_hip.so kernels (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mumpy_hip.h"

namespace mumpy {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;
constexpr int WS = 7;        // window side
constexpr int WT = 49;       // tokens per window
constexpr int HD = 32;       // head width of every Swin / deformable head

void set_error(const char* fmt, ...);

// Tuning hooks.  The shipped library reads NO environment variables: tune_int() is the constant default.  A diagnostics build
// (make TUNING=1 -> -DMUMPY_TUNING) reads MUMPY_* variables once per process for A/B runs (tools/gemm_shapes.py & co).
#ifdef MUMPY_TUNING
int tune_int(const char* name, int dflt);
const char* tune_str(const char* name);
#else
inline int tune_int(const char*, int dflt) { return dflt; }
inline const char* tune_str(const char*) { return nullptr; }
#endif

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Every launch goes through this: reports launch-time errors without synchronising.
#define MUMPY_CHECK_LAUNCH(name)                                                     \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            mumpy::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
            return (int)e_;                                                          \
        }                                                                            \
    } while (0)

#define MUMPY_REQUIRE(cond, code, ...)                                               \
    do {                                                                             \
        if (!(cond)) {                                                               \
            mumpy::set_error(__VA_ARGS__);                                           \
            return code;                                                             \
        }                                                                            \
    } while (0)

// Cross-lane all-reduce building blocks.  A butterfly from the low bit up: after the steps for 1 and 2 every lane of a quad
// holds the quad's sum, so the "xor 4" / "xor 8" partners can be ANY lane of the other quad / other half-row -- DPP's
// row_half_mirror and row_mirror, which (like the quad permutes) fold into the v_add itself: one VALU instruction per step
// instead of an address computation + ds_bpermute round trip (~100 cycles) per step.  xor 16 stays inside 32 lanes: ds_swizzle
// (bit-mask mode, no address register); only xor 32 needs the permute.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float swz_xor16(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1f));
}

// sum over aligned groups of `width` lanes (a power of two <= 64), result in every lane of the group; all lanes active
__device__ __forceinline__ float wave_sum(float v, int width) {
    if (width > 1) v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
    if (width > 2) v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
    if (width > 4) v += dpp_f<0x141>(v);       // row_half_mirror
    if (width > 8) v += dpp_f<0x140>(v);       // row_mirror
    if (width > 16) v += swz_xor16(v);
    if (width > 32) v += __shfl_xor(v, 32);
    return v;
}

// first + p[stride] + p[2 stride] + ... (n - 1 more terms), added IN THAT ORDER (the reproducibility contract of every slab / partial
// reduce here), but with the loads of eight terms issued before their adds: written as one load per add with a run-time trip count
// the loop waits out a memory latency per term, which is most of the duration of the small reduce launches of a training step.
template <typename T>
__device__ __forceinline__ T ordered_sum(T first, const T* __restrict__ p, int64_t stride, int n_more) {
    T s = first;
    int z = 0;
    for (; z + 8 <= n_more; z += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(z + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < n_more; ++z) s += p[(int64_t)z * stride];
    return s;
}

// GroupNorm: mean / rstd of every group of image b from the split partials {sum, sum of squares} that gn_stats_kernel wrote
// ([b][split][group][2]).  Every consumer block needs them before it can start; with the partials of up to 256 splits one thread
// per group walking them in sequence costs more than the streaming pass that follows, so all 256 threads take part: thread
// (slice k, group g) adds splits k, k + nslice, ... in double, thread g then adds the slices in order (a fixed order: reproducible).
// Call from ALL threads of a 256-thread block; ends with a barrier.
__device__ __forceinline__ void gn_block_stats(const float* __restrict__ stats, int nsplit, int G, int b, double n, float eps,
                                               float* gm, float* gr, double (*red)[2]) {
    const int tid = threadIdx.x, nslice = 256 / G, g = tid % G, k = tid / G;
    double s = 0.0, q = 0.0;
    if (k < nslice)
        for (int sp = k; sp < nsplit; sp += nslice) {
            const float* o = stats + (((int64_t)b * nsplit + sp) * G + g) * 2;
            s += (double)o[0]; q += (double)o[1];
        }
    red[tid][0] = s; red[tid][1] = q;
    __syncthreads();
    if (tid < G) {
        double ss = 0.0, qq = 0.0;
        for (int j = 0; j < nslice; ++j) { ss += red[j * G + tid][0]; qq += red[j * G + tid][1]; }
        const double m = ss / n;
        double var = qq / n - m * m;
        if (var < 0.0) var = 0.0;
        gm[tid] = (float)m;
        gr[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
}

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 round-off level): 1 rcp + 1 exp + 6 fma instead of
// libm erff's ~40-instruction piecewise path -- the GELU epilogue of the fc1 GEMMs is VALU-bound otherwise.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// raster token of in-window position p of window (wy,wx) on a (Hs,W) grid after roll(-shift)
// (swin:54-66 + 273): shifted[y][x] = src[(y+shift)%Hs][(x+shift)%W].
__device__ __forceinline__ int window_token(int wy, int wx, int p, int Hs, int W, int shift) {
    int py = p / WS, px = p - py * WS;
    int y = wy * WS + py + shift;
    int x = wx * WS + px + shift;
    if (y >= Hs) y -= Hs;
    if (x >= W) x -= W;
    return y * W + x;
}

}  // namespace mumpy
