// norm.hip — LayerNorm and the gather+LayerNorm front half of patch merging.  HBM-bound streaming kernels:
// rows are split over LPR lanes (a power of two <= 64) with 16-byte accesses, several rows per wave when C is small,
// the row is held in registers between the statistics pass and the normalise pass (one read, one write).
#include "common.h"
using namespace mumpy;

namespace {

constexpr int MAXNV = 16;  // float4 per lane: C <= 64 lanes * 16 * 4 = 4096

// SRC: functor giving the address of float4 index v (0..C/4) of virtual row `row`.
// OUT16: y is bf16 (the activation storage of config 3); statistics and arithmetic stay fp32.
template <typename SRC, bool OUT16 = false>
__global__ __launch_bounds__(256) void ln_rows_kernel(SRC src, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y,
                                                      int64_t rows, int C, int lpr, int nv, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int rpw = 64 / lpr;                      // rows per wave
    const int sub = lane / lpr, l = lane % lpr;
    const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * rpw + sub;
    const bool live = row < rows;
    f32x4 v[MAXNV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXNV; ++i) {
        if (i < nv) {
            v[i] = live ? *reinterpret_cast<const f32x4*>(src(row, l + i * lpr)) : f32x4{0, 0, 0, 0};
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    s = wave_sum(s, lpr);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXNV; ++i) {
        if (i < nv) {
            f32x4 d = v[i] - mean;
            q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    q = wave_sum(q, lpr);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
    float* yr = y + row * (int64_t)C;
#pragma unroll
    for (int i = 0; i < MAXNV; ++i) {
        if (i < nv) {
            const int c4 = l + i * lpr;
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c4);
            const f32x4 o = (v[i] - mean) * rstd * g + b;
            if (OUT16) {
                typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
                *reinterpret_cast<bf16x4v*>(reinterpret_cast<__bf16*>(y) + row * (int64_t)C + 4 * c4) = __builtin_convertvector(o, bf16x4v);
            } else {
                *reinterpret_cast<f32x4*>(yr + 4 * c4) = o;
            }
        }
    }
}

struct PlainRows {
    const float* x;
    int C;
    __device__ const float* operator()(int64_t row, int c4) const { return x + row * (int64_t)C + 4 * c4; }
};

// virtual row of patch merging: out token (b, i, j) = concat of x tokens (2i,2j),(2i+1,2j),(2i,2j+1),(2i+1,2j+1)
// (swin:357-361) on the stacked (Hs,W) grid.
struct MergeRows {
    const float* x;
    int Hs, W, C;  // C = per-token channels of x (virtual row has 4C)
    __device__ const float* operator()(int64_t row, int c4) const {
        const int w2 = W >> 1, h2 = Hs >> 1;
        const int j = (int)(row % w2);
        const int64_t t = row / w2;
        const int i = (int)(t % h2);
        const int64_t b = t / h2;
        const int seg = (4 * c4) / C;            // which of the 4 source tokens
        const int c = 4 * c4 - seg * C;
        const int dy = seg & 1, dx = seg >> 1;   // order (0,0),(1,0),(0,1),(1,1)
        const int64_t tok = (b * Hs + (2 * i + dy)) * (int64_t)W + (2 * j + dx);
        return x + tok * C + c;
    }
};

bool pick_split(int C, int* lpr, int* nv) {
    if (C % 4 || C > 4096 || C < 4) return false;
    const int n4 = C / 4;
    int l = 64;                                   // fewest lanes per row that still leave >= 3 float4 per lane in flight
    while (l > 1 && ((n4 % l) || n4 / l < 3)) l >>= 1;   // (measured: 1 float4 per lane ran at 3.7 TB/s)
    if (n4 / l > MAXNV) return false;
    *lpr = l;
    *nv = n4 / l;
    return true;
}

}  // namespace

extern "C" int mumpy_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int64_t rows,
                                   int C, float eps, void* stream) {
    if (rows == 0) return 0;   // empty batch: nothing to do, pointers may be null
    MUMPY_REQUIRE(x && gamma && beta && y, MUMPY_ENULL, "layernorm: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), MUMPY_EALIGN,
                  "layernorm: pointers must be 16-byte aligned");
    int lpr, nv;
    MUMPY_REQUIRE(rows >= 0 && pick_split(C, &lpr, &nv), MUMPY_EINVAL, "layernorm: unsupported C=%d", C);
    if (rows == 0) return 0;
    const int rows_per_block = 4 * (64 / lpr);
    const int64_t grid = (rows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(ln_rows_kernel<PlainRows>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                       PlainRows{x, C}, gamma, beta, y, rows, C, lpr, nv, eps);
    MUMPY_CHECK_LAUNCH("layernorm");
    return 0;
}

extern "C" int mumpy_layernorm_bf16_fwd(const float* x, const float* gamma, const float* beta, void* y, int64_t rows,
                                        int C, float eps, void* stream) {
    if (rows == 0) return 0;
    MUMPY_REQUIRE(x && gamma && beta && y, MUMPY_ENULL, "layernorm_bf16: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), MUMPY_EALIGN,
                  "layernorm_bf16: pointers must be 16-byte aligned");
    int lpr, nv;
    MUMPY_REQUIRE(rows >= 0 && pick_split(C, &lpr, &nv), MUMPY_EINVAL, "layernorm_bf16: unsupported C=%d", C);
    const int rows_per_block = 4 * (64 / lpr);
    const int64_t grid = (rows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL((ln_rows_kernel<PlainRows, true>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                       PlainRows{x, C}, gamma, beta, static_cast<float*>(y), rows, C, lpr, nv, eps);
    MUMPY_CHECK_LAUNCH("layernorm_bf16");
    return 0;
}

extern "C" int mumpy_patch_merge_ln_fwd(const float* x, const float* gamma, const float* beta, float* out, int B,
                                        int Hs, int W, int C, float eps, void* stream) {
    MUMPY_REQUIRE(x && gamma && beta && out, MUMPY_ENULL, "patch_merge_ln: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(out) && aligned16(gamma) && aligned16(beta), MUMPY_EALIGN,
                  "patch_merge_ln: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && Hs > 0 && W > 0 && (Hs % 2 == 0) && (W % 2 == 0), MUMPY_EINVAL,
                  "patch_merge_ln: grid (%d,%d) must be even (swin:353)", Hs, W);
    int lpr, nv;
    MUMPY_REQUIRE(C % 4 == 0 && pick_split(4 * C, &lpr, &nv), MUMPY_EINVAL, "patch_merge_ln: unsupported C=%d", C);
    const int64_t rows = (int64_t)B * (Hs / 2) * (W / 2);
    const int rows_per_block = 4 * (64 / lpr);
    const int64_t grid = (rows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(ln_rows_kernel<MergeRows>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream),
                       MergeRows{x, Hs, W, C}, gamma, beta, out, rows, 4 * C, lpr, nv, eps);
    MUMPY_CHECK_LAUNCH("patch_merge_ln");
    return 0;
}

// ------------------------------------------------------------------ eval tail: sigmoid -> threshold -> uint8
namespace {
__global__ __launch_bounds__(256) void sigmoid_thr_kernel(const float* __restrict__ z, uint8_t* __restrict__ m,
                                                          int64_t n, float thr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) m[i] = (1.0f / (1.0f + __expf(-z[i])) > thr) ? 1 : 0;
}
}  // namespace

extern "C" int mumpy_sigmoid_threshold_fwd(const float* logits, uint8_t* mask, int64_t n, float thr, void* stream) {
    MUMPY_REQUIRE(logits && mask, MUMPY_ENULL, "sigmoid_threshold: null pointer");
    MUMPY_REQUIRE(n >= 0, MUMPY_EINVAL, "sigmoid_threshold: n < 0");
    if (n == 0) return 0;
    int64_t grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(sigmoid_thr_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), logits, mask, n, thr);
    MUMPY_CHECK_LAUNCH("sigmoid_threshold");
    return 0;
}

// ------------------------------------------------------------------ input staging
// ToTensor + Normalize of the eval pipeline (test.py:22-25; torchvision semantics: u8/255, then (v - mean)/std per
// channel) fused with the HWC -> CHW transpose: frames (N,H,W,3) uint8 -> clip tensor (N,3,H,W) fp32.
// A thread converts 4 consecutive pixels of one channel: 16-B coalesced store; the 12 source bytes come from L1.
namespace {
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                           int64_t nframes, int64_t HW, float m0, float m1, float m2,
                                                           float s0, float s1, float s2) {
    const int64_t q4 = HW >> 2;
    const int64_t total = nframes * 3 * q4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t p4 = i % q4;
        const int64_t t = i / q4;
        const int ch = (int)(t % 3);
        const int64_t f = t / 3;
        const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
        const float stdv = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
        const uint8_t* s = src + (f * HW + 4 * p4) * 3 + ch;
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = ((float)s[3 * k] / 255.0f - mean) / stdv;
        *reinterpret_cast<f32x4*>(dst + (f * 3 + ch) * HW + 4 * p4) = v;
    }
}
}  // namespace

// Same staging with the loader's resize in front (universaldataset.py:75-79: `img.resize(self.inputRes)` with no filter
// argument; the reference pins pillow==4.0.0, whose default filter is NEAREST).  Pillow's NEAREST resize
// (Geometry.c, ImagingScaleAffine) walks a double accumulator: xo = 0.5 * a; for each output x { xin = (int)xo; xo += a; }
// with a = src / dst -- the ACCUMULATED value, not (x + 0.5) * a, decides exact-integer ties (1920 -> 224: x = 3 lands on
// 30.0 or 29.999...), so the source-index tables are built on the host with exactly that loop (mumpy_resize_nearest_table)
// and the kernel gathers through them.  How footage that is not 224x224 (432x240 DVI clips, config 4) enters the model.
extern "C" int mumpy_resize_nearest_table(int src, int dst, int32_t* table_host) {
    MUMPY_REQUIRE(table_host, MUMPY_ENULL, "resize_nearest_table: null pointer");
    MUMPY_REQUIRE(src > 0 && dst > 0, MUMPY_EINVAL, "resize_nearest_table: bad sizes %d -> %d", src, dst);
    const double a = (double)src / dst;
    double xo = a * 0.5;
    for (int x = 0; x < dst; ++x) {
        int xin = xo < 0.0 ? -1 : (int)xo;
        if (xin > src - 1) xin = src - 1;          // cannot happen for a pure scale; keeps the gather in bounds regardless
        table_host[x] = xin;
        xo += a;
    }
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void resize_normalize_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                                  const int32_t* __restrict__ ytab, const int32_t* __restrict__ xtab,
                                                                  int64_t nframes, int Hs, int Ws, int H, int W, float m0, float m1,
                                                                  float m2, float s0, float s1, float s2) {
    const int q4 = W >> 2;
    const int64_t total = nframes * 3 * H * q4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x4 = (int)(i % q4);
        int64_t t = i / q4;
        const int y = (int)(t % H); t /= H;
        const int ch = (int)(t % 3);
        const int64_t f = t / 3;
        const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
        const float stdv = ch == 0 ? s0 : (ch == 1 ? s1 : s2);
        const uint8_t* row = src + ((f * Hs + ytab[y]) * (int64_t)Ws) * 3 + ch;
        const int4 xi = *reinterpret_cast<const int4*>(xtab + 4 * x4);
        f32x4 v;
        v[0] = ((float)row[3 * (int64_t)xi.x] / 255.0f - mean) / stdv;
        v[1] = ((float)row[3 * (int64_t)xi.y] / 255.0f - mean) / stdv;
        v[2] = ((float)row[3 * (int64_t)xi.z] / 255.0f - mean) / stdv;
        v[3] = ((float)row[3 * (int64_t)xi.w] / 255.0f - mean) / stdv;
        *reinterpret_cast<f32x4*>(dst + ((f * 3 + ch) * H + y) * (int64_t)W + 4 * x4) = v;
    }
}
}  // namespace

extern "C" int mumpy_resize_normalize_u8_fwd(const uint8_t* frames, float* out, const int32_t* ytab, const int32_t* xtab,
                                             int64_t nframes, int Hs, int Ws, int H, int W, const float* mean3,
                                             const float* std3, void* stream) {
    MUMPY_REQUIRE(frames && out && mean3 && std3 && ytab && xtab, MUMPY_ENULL, "resize_normalize_u8: null pointer");
    MUMPY_REQUIRE(nframes >= 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0 && W % 4 == 0, MUMPY_EINVAL,
                  "resize_normalize_u8: bad sizes %dx%d -> %dx%d (output width must be a multiple of 4)", Hs, Ws, H, W);
    MUMPY_REQUIRE(aligned16(out) && aligned16(xtab), MUMPY_EALIGN, "resize_normalize_u8: out and xtab must be 16-byte aligned");
    MUMPY_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, MUMPY_EINVAL, "resize_normalize_u8: zero std");
    if (nframes == 0) return 0;
    const int64_t total = nframes * 3 * (int64_t)H * (W / 4);
    int64_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(resize_normalize_u8_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), frames, out, ytab, xtab,
                       nframes, Hs, Ws, H, W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    MUMPY_CHECK_LAUNCH("resize_normalize_u8");
    return 0;
}

extern "C" int mumpy_normalize_u8_fwd(const uint8_t* frames, float* out, int64_t nframes, int H, int W, const float* mean3,
                                      const float* std3, void* stream) {
    MUMPY_REQUIRE(frames && out && mean3 && std3, MUMPY_ENULL, "normalize_u8: null pointer");
    MUMPY_REQUIRE(nframes >= 0 && H > 0 && W > 0 && ((int64_t)H * W) % 4 == 0, MUMPY_EINVAL, "normalize_u8: H*W must be a multiple of 4");
    MUMPY_REQUIRE(aligned16(out), MUMPY_EALIGN, "normalize_u8: out must be 16-byte aligned");
    MUMPY_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, MUMPY_EINVAL, "normalize_u8: zero std");
    if (nframes == 0) return 0;
    const int64_t total = nframes * 3 * ((int64_t)H * W / 4);
    int64_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(normalize_u8_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), frames, out, nframes,
                       (int64_t)H * W, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    MUMPY_CHECK_LAUNCH("normalize_u8");
    return 0;
}
