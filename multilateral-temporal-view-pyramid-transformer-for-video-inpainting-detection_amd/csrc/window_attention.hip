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
// The unit is below the fp32 ridge (12.25 FLOP/B): no LDS staging, ~25 KB in flight per wave, 12 waves per CU (166 VGPRs).
// Measured on the largest launch of the B=8,T=5 forward (40 frames 56x56, C=128: 10,240 units, 257 MB of q/k/v/o;
// MUMPY_WA_DBG ablation, MI355X): 65.9 us = 30 % of the 157 TFLOP/s datasheet peak in useful 49x49 FLOPs, 46 % MFMA-busy
// by SQ_VALU_MFMA_BUSY_CYCLES (114 MFMAs per unit incl. the 49->64 padding; profiles/r01_pmc_mfma.md).  Floors: 41 us of
// HBM time at the 6.3 TB/s this part sustains, 38.5 us of MFMA issue at the 124 TFLOP/s a bare fp32 MFMA loop reaches.
// Ablation: loads only 33 us, MFMAs + softmax only 46 us (80 % of the practical MFMA rate), loads + stores 41 us.
// What moved it (79.8 -> 65.9 us):
//   * addressing: wave-uniform bases in SGPRs + pre-multiplied 32-bit byte offsets from the token tables; the pointer form
//     spent 66 v_mad_u64 + 130 v_mul_lo_u32 (quarter-rate) per unit, ~40 % of the MFMA time (79.8 -> 67.8 us);
//   * one branch per unit on the mask pointer (unmasked windows run branch-free, masked ones batch their loads), no
//     exp / max / scale work on the statically padded key slots (67.8 -> 65.9 us).
// Tried and measured, not kept: staggered block starts, s_setprio per resident block or around the MFMA phase (no
// change: a per-wave s_memtime trace shows the SIMD busy in some wave's compute phase ~all the time; the remaining gap
// is the memory phase of a unit not overlapping its own wave's compute); cross-unit REGISTER prefetch at 2 waves/SIMD
// (spills) and at 1 wave/SIMD (no overlap: hipcc's waitcnt insertion drains loop-carried prefetches); row-coalesced q/k
// address pattern (-5 %, needs an LDS transpose).  Next step: K/V of the next unit through an LDS-DMA ring at 2 waves/SIMD.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
using namespace mumpy;

namespace {

struct SelfArgs {
    const float* qkv;
    float* out;
    const float* bias;      // (nH,64,64)
    const float* mask_tab;  // (nU,64,64) or null
    const int32_t* mask_id; // (n_mask) or null; window bw uses mask_id[bw % n_mask]
    int B, Hs, W, C, nH, shift, nWx, nW, n_mask, groups, stagger;
    int dbg;               // diagnostic ablation mask (MUMPY_WA_DBG): 1 skip q/k/v loads, 2 skip MFMAs+softmax, 4 skip stores
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

// load the 16 channels [16h,16h+16) of one 32-channel head row.  Padded rows (token slot >= 49) are CLAMPED to slot 48 by
// the token table instead of predicated: the duplicates are finite, their scores are overwritten with -1e30 (keys) or
// never stored (queries), and branch-free loads keep the compiler's vmcnt bookkeeping exact, which the cross-unit
// prefetch depends on (an exec-masked load made it wait vmcnt(0) and drain the prefetch).
__device__ __forceinline__ void load_frag(f32x4 (&f)[4], const float* row, bool valid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = valid ? *reinterpret_cast<const f32x4*>(row + 4 * i) : f32x4{0, 0, 0, 0};
}
__device__ __forceinline__ void load_frag_nb(f32x4 (&f)[4], const float* row) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const f32x4*>(row + 4 * i);
}

// S^T[jt] += K[jt] Q^T for ONE query tile (32 queries on the lanes); q already scaled
__device__ __forceinline__ void qk_product(f32x16 (&s)[2], const f32x4 (&kf)[2][4], const f32x4 (&qf)[4]) {
#pragma unroll
    for (int st = 0; st < 16; ++st) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) s[jt] = mfma32(kf[jt][st >> 2][st & 3], qf[st >> 2][st & 3], s[jt]);
    }
}

// add bias (+mask) rows and run the softmax over keys for the query column this lane owns (query i = 32*it + c).
// MASKED is a compile-time switch: the caller branches once per unit on the (wave-uniform) mask pointer, so unmasked
// windows run branch-free and a masked window issues its 7 mask loads back to back (one wait) instead of load-wait pairs.
template <bool MASKED, typename BIAS>
__device__ __forceinline__ void bias_softmax(f32x16 (&s)[2], BIAS bias_at, const float* mask_w, int i, int h,
                                             float post_scale, float* m_out = nullptr, float* inv_out = nullptr) {
    constexpr float NEG = -1e30f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (jt == 1 && g == 3) continue;                             // keys 56..63: all padding
            const f32x4 b = bias_at(jt, g);                              // keys 32jt+8g+4h .. +3 of query i
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (jt == 1 && g == 2 && e > 0) continue;                // keys 49..51 / 53..55: padding in both halves
                s[jt][4 * g + e] = s[jt][4 * g + e] * post_scale + b[e];
            }
        }
    if (MASKED) {                                                        // (s + bias) + mask, as swin:153-157
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            f32x4 mk[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (!(jt == 1 && g == 3)) mk[g] = *reinterpret_cast<const f32x4*>(mask_w + i * 64 + 32 * jt + 8 * g + 4 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (jt == 1 && g == 3) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (jt == 1 && g == 2 && e > 0) continue;
                    s[jt][4 * g + e] += mk[g][e];
                }
            }
        }
    }
    float m = NEG;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (jt == 1 && r >= 9) continue;
            m = fmaxf(m, s[jt][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 32));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (jt == 1 && r >= 9) { s[jt][r] = 0.f; continue; }        // padded keys: exp(-1e30 - m) == 0 exactly
            const float e = __expf(s[jt][r] - m);
            s[jt][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (jt == 1 && r >= 9) continue;
            s[jt][r] *= inv;
        }
    if (m_out) { *m_out = m; *inv_out = inv; }
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
__device__ __forceinline__ void load_v_nb(float (&vf)[2][16], VROW vrow, int c, int h) {      // vrow clamps j itself
    for_pv_steps([&](int jt, int g, int e) { vf[jt][4 * g + e] = vrow(32 * jt + 8 * g + 4 * h + e)[c]; });
}

template <typename VROW>
__device__ __forceinline__ void load_v(float (&vf)[2][16], VROW vrow, int c, int h) {
    for_pv_steps([&](int jt, int g, int e) {
        const int j = 32 * jt + 8 * g + 4 * h + e;
        vf[jt][4 * g + e] = (j < WT) ? vrow(j)[c] : 0.f;
    });
}

__device__ __forceinline__ void pv_product(f32x16& o, const f32x16 (&s)[2], const float (&vf)[2][16]) {
    for_pv_steps([&](int jt, int g, int e) { o = mfma32(s[jt][4 * g + e], vf[jt][4 * g + e], o); });
}

template <typename OROW>
__device__ __forceinline__ void store_o(const f32x16& o, int it, OROW orow, int c, int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (it == 1 && (r >> 2) >= 2 && !((r >> 2) == 2 && (r & 3) == 0)) continue;  // statically >= 49
        if (i < WT) orow(i)[c] = o[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int BLD = 68;   // LDS row stride of the staged bias table: 68 floats -> conflict-free ds_read_b128 across rows

// byte-offset addressing: base is wave-uniform (SGPR pair), the per-lane part a 32-bit byte offset from the token tables,
// so every access is "global_* v, v_off, s[base]" with one v_add at most -- the 64-bit token*stride products the
// pointer form needs (2 v_mul_lo + v_mad_u64 + ... per access, all quarter-rate) cost ~40 % of the MFMA time of a unit.
__device__ __forceinline__ const f32x4* at16(const char* base, uint32_t off) {
    return reinterpret_cast<const f32x4*>(base + off);
}

// DBG = false is the shipped instantiation: the MUMPY_WA_DBG ablation switches (skip loads / MFMAs / stores) exist only in
// the diagnostic instantiation, which the launcher selects when that variable is set.
// IO16: qkv and out are bf16 in memory (config 3's activation storage); the arithmetic is the same fp32 MFMA flow.
// NOLDS: the "background" instantiation (round 3, see gemm_rd.hip): no LDS allocation at all, so that the kernel can be resident on
// a CU whose whole LDS belongs to the persistent GEMM.  The per-wave token tables live in registers and are read through
// ds_bpermute (which uses the LDS crossbar but no LDS memory); the bias rows come from global memory / L1.  Slower per unit (the
// row-strided bias reads cost L1 tag cycles), used only for the small launches of views 1 / 2 that run beside view 3's GEMMs.
template <bool DBG, bool IO16 = false, bool NOLDS = false>
__global__ __launch_bounds__(256, (NOLDS ? 4 : 3)) void win_attn_self_kernel(SelfArgs a) {
    __shared__ uint32_t tok_in[4][64];    // token * (3C*4): byte offset of the token's qkv row
    __shared__ __attribute__((aligned(16))) uint32_t tok_out[4][64];   // token * (C*4):  byte offset of the token's out row
    __shared__ __attribute__((aligned(16))) float bias_s[WT * BLD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    // a block = ONE head x 4 consecutive windows: the head's 49x49 bias table is staged once in LDS (the per-lane
    // row-strided reads of it would otherwise cost as many L1 tag cycles as the MFMAs); a (window, head) unit owns its
    // 128-B q/k/v row segments exclusively, so grouping by head costs no extra HBM or L2 traffic.
    const int head = blockIdx.x % a.nH;
    const int slot = blockIdx.x / a.nH;
    const int64_t nwin = (int64_t)a.B * a.nW;
    if (!NOLDS) {   // bias table: staged ONCE per persistent block
        const float* bsrc = a.bias + (int64_t)head * 4096;
        for (int idx = threadIdx.x; idx < WT * 16; idx += 256) {
            const int row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4*>(&bias_s[row * BLD + 4 * c4]) = *reinterpret_cast<const f32x4*>(bsrc + row * 64 + 4 * c4);
        }
        __syncthreads();
    }
    for (int d = 0; d < (slot % 3) * a.stagger; ++d) __builtin_amdgcn_s_sleep(127);
    const int64_t L = (int64_t)a.Hs * a.W;
    const uint32_t rsb = (IO16 ? 6u : 12u) * a.C, rob = (IO16 ? 2u : 4u) * a.C;                      // row strides in bytes
    uint32_t* ti = NOLDS ? nullptr : tok_in[wave];
    uint32_t* to = NOLDS ? nullptr : tok_out[wave];
    uint32_t ti_reg = 0, to_reg = 0;                 // NOLDS: lane l holds table entry l
    auto TI = [&](int i) -> uint32_t { return NOLDS ? (uint32_t)__shfl((int)ti_reg, i) : ti[i]; };
    auto TO = [&](int i) -> uint32_t { return NOLDS ? (uint32_t)__shfl((int)to_reg, i) : to[i]; };
    for (int64_t bw = (int64_t)slot * 4 + wave; bw < nwin; bw += (int64_t)a.groups * 4) {   // all scalar
    const int n = (int)(bw % a.nW);
    const int64_t b = bw / a.nW;
    const int wy = n / a.nWx, wx = n - wy * a.nWx;
    {
        const uint32_t tok = (uint32_t)window_token(wy, wx, lane < WT ? lane : WT - 1, a.Hs, a.W, a.shift);   // padded slots -> slot 48
        if (NOLDS) { ti_reg = tok * rsb; to_reg = tok * rob; }
        else { ti[lane] = tok * rsb; to[lane] = tok * rob; }
    }
    __builtin_amdgcn_wave_barrier();
    const char* base = IO16 ? reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(a.qkv) + b * L * 3 * a.C + head * HD)
                            : reinterpret_cast<const char*>(a.qkv + b * L * 3 * a.C + head * HD);

    // q/k/v go straight to registers (MFMA operand layout); branch-free
    f32x4 qf[2][4], kf[2][4];
    float vf[2][16];
    const int dbg = DBG ? a.dbg : 0;
    if (!(dbg & 1)) {
        if (IO16) {
            // a lane's 16 channels are 32 bytes: two 16-byte loads of 8 bf16, widened to fp32 by a shift
            typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
            auto widen = [](u32x4v w, f32x4& lo, f32x4& hi) {
                lo = f32x4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
                hi = f32x4{__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u)};
            };
            const char* kbase = base + 2 * a.C;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint32_t off = TI(32 * t + c) + 32u * h;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    widen(*reinterpret_cast<const u32x4v*>(base + (off + 16u * i)), qf[t][2 * i], qf[t][2 * i + 1]);
                    widen(*reinterpret_cast<const u32x4v*>(kbase + (off + 16u * i)), kf[t][2 * i], kf[t][2 * i + 1]);
                }
            }
            const char* vbase = base + 4 * a.C;
            for_pv_steps([&](int jt, int g, int e) {
                const uint32_t w = *reinterpret_cast<const uint16_t*>(vbase + (TI(32 * jt + 8 * g + 4 * h + e) + 2u * c));
                vf[jt][4 * g + e] = __uint_as_float(w << 16);
            });
        } else {
        const char* kbase = base + 4 * a.C;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t off = TI(32 * t + c) + 64u * h;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qf[t][i] = *at16(base, off + 16u * i);
                kf[t][i] = *at16(kbase, off + 16u * i);
            }
        }
        const char* vbase = base + 8 * a.C;
        for_pv_steps([&](int jt, int g, int e) {
            vf[jt][4 * g + e] = *reinterpret_cast<const float*>(vbase + (TI(32 * jt + 8 * g + 4 * h + e) + 4u * c));
        });
        }
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) { load_frag(qf[t], a.qkv, false); load_frag(kf[t], a.qkv, false); }
        for_pv_steps([&](int jt, int g, int e) { vf[jt][4 * g + e] = 1.f; });
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) qf[t][i] *= a.scale;   // q = q * scale before QK^T (swin:145)

    const float* mask_w = nullptr;
    if (a.mask_id) {
        const int id = a.mask_id[bw % a.n_mask];   // scalar load
        if (id >= 0) mask_w = a.mask_tab + (int64_t)id * 4096;
    }
    char* obase = IO16 ? reinterpret_cast<char*>(reinterpret_cast<__bf16*>(a.out) + b * L * a.C + head * HD)
                       : reinterpret_cast<char*>(a.out + b * L * a.C + head * HD);
    // the two 32-query tiles go one after the other: S needs 32 accumulator registers instead of 64
    auto tiles = [&](auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 s[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[jt][r] = 0.f;
            if (!(dbg & 2)) qk_product(s, kf, qf[it]);
            else { s[0][0] = kf[0][0][0] + qf[it][0][0]; s[1][3] = kf[1][1][1] * qf[it][2][1]; }
            const int qi = 32 * it + c;
            const float* brow = NOLDS ? a.bias + (int64_t)head * 4096 + (qi < WT ? qi : WT - 1) * 64 + 4 * h
                                      : &bias_s[(qi < WT ? qi : WT - 1) * BLD + 4 * h];    // padded queries re-read row 48
            if (!(dbg & 2))
                bias_softmax<MASKED>(s, [&](int jt, int g) {
                    f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 32 * jt + 8 * g);
                    if (jt == 1 && g == 2 && h) bv.x = -1e30f;                     // key 52 is padding (key 48 is real)
                    return bv;
                }, mask_w, qi, h, 1.0f);
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            if (!(dbg & 2)) pv_product(o, s, vf);
            else { o[0] = s[0][0] + vf[0][0]; o[5] = s[1][2] * vf[1][8]; }
            if (!(dbg & 4)) {
                // row offsets of the 4 consecutive queries a lane's register group g holds: one 16-byte table read
                typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
                u32x4v to4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (!(it == 1 && g == 3)) {
                        if (NOLDS) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) to4[g][e] = TO(32 * it + 8 * g + 4 * h + e);
                        } else {
                            to4[g] = *reinterpret_cast<const u32x4v*>(&to[32 * it + 8 * g + 4 * h]);
                        }
                    }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (it == 1 && (r >> 2) >= 2 && !((r >> 2) == 2 && (r & 3) == 0)) continue;  // statically >= 49
                    const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (i < WT) {
                        if (IO16) *reinterpret_cast<__bf16*>(obase + (to4[r >> 2][r & 3] + 2u * c)) = (__bf16)o[r];
                        else *reinterpret_cast<float*>(obase + (to4[r >> 2][r & 3] + 4u * c)) = o[r];
                    }
                }
            } else if (o[0] == 1234.5f && o[5] == 77.f) obase[0] = 1;
        }
    };
    if (mask_w) tiles(std::true_type{}); else tiles(std::false_type{});
    __builtin_amdgcn_wave_barrier();   // the token tables are rewritten by the next unit
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void win_attn_cross_kernel(CrossArgs a) {
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

#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 s[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[jt][r] = 0.f;
            qk_product(s, kf, qf[it]);
            bias_softmax<false>(s, [&](int jt, int g) {                        // no bias: only the 49->64 key padding
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (jt == 1 && g == 2 && h) bv.x = -1e30f;                      // key 52 is padding (key 48 is real)
                return bv;
            }, nullptr, 32 * it + c, h, a.scale);                               // scale on the product (deform:364)
            pv_product(o[it], s, vf);
        }
    }
    float* obase = a.out + b1 * WT * a.C + head * HD;
    store_o(o[0], 0, [&](int i) { return obase + (int64_t)i * a.C; }, c, h);
    store_o(o[1], 1, [&](int i) { return obase + (int64_t)i * a.C; }, c, h);
}

// ===============================================================================================================
// BACKWARD of the window attention core ("backward HIP kernels for row 5").  Given dO, the unit's
// P = softmax(q k^T + bias + mask) is recomputed (nothing but q/k/v is kept from the forward) and
//   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - rowsum(P o dP)),   dQ = scale dS K,   dK = dS^T (scale Q),   dBias += dS.
// Two kernels, one per MFMA orientation, so that every product gets its A operand straight from accumulator registers
// (the forward's accumulator->operand trick) and nothing is transposed through LDS:
//   bwd_q : lane = QUERY (S^T = K Q^T as in the forward): softmax statistics, D = rowsum(P o dP), dS^T, dQ = dS K, and the
//           per-wave running sum of dS for the bias gradient; writes {m, 1/l, D} per query for the second kernel.
//   bwd_kv: lane = KEY (S = Q K^T, the same fragments with the MFMA operands swapped): P and dS rebuilt from the saved
//           statistics, dV = P^T dO and dK = dS^T Q accumulated over the queries.
// One wave per (window, head) unit, persistent blocks of 4 waves per head as in the forward; 1 wave per SIMD (the
// operand sets of a unit need ~300 VGPRs).  Deterministic: per-wave dBias partials are reduced in a fixed order.
struct BwdArgs {
    const float* qkv; const float* dout; const float* bias; const float* mask_tab; const int32_t* mask_id;
    float* dqkv; float* stats; float* dbias_part;
    int B, Hs, W, C, nH, shift, nWx, nW, n_mask, groups;
    float scale;
};

// 16 consecutive channels [16h, 16h+16) of a 32-channel head row (MFMA A/B fragment layout of the forward)
__device__ __forceinline__ void load_frag16(f32x4 (&f)[4], const char* base, uint32_t off) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const f32x4*>(base + (off + 16u * i));
}

__global__ __launch_bounds__(256, 1) void win_attn_bwd_q_kernel(BwdArgs a) {
    __shared__ uint32_t tok_in[4][64];
    __shared__ uint32_t tok_out[4][64];
    __shared__ __attribute__((aligned(16))) float bias_s[WT * BLD];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int head = blockIdx.x % a.nH;
    const int slot = blockIdx.x / a.nH;
    const int64_t nwin = (int64_t)a.B * a.nW;
    {
        const float* bsrc = a.bias + (int64_t)head * 4096;
        for (int idx = threadIdx.x; idx < WT * 16; idx += 256) {
            const int row = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4*>(&bias_s[row * BLD + 4 * c4]) = *reinterpret_cast<const f32x4*>(bsrc + row * 64 + 4 * c4);
        }
    }
    __syncthreads();
    const int64_t L = (int64_t)a.Hs * a.W;
    const uint32_t rsb = 12u * a.C, rob = 4u * a.C;
    uint32_t* ti = tok_in[wave];
    uint32_t* to = tok_out[wave];
    f32x16 dsum[2][2];                                                    // running sum of dS^T over this wave's units
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dsum[it][jt][r] = 0.f;

    for (int64_t bw = (int64_t)slot * 4 + wave; bw < nwin; bw += (int64_t)a.groups * 4) {
        const int n = (int)(bw % a.nW);
        const int64_t b = bw / a.nW;
        const int wy = n / a.nWx, wx = n - wy * a.nWx;
        {
            const uint32_t tok = (uint32_t)window_token(wy, wx, lane < WT ? lane : WT - 1, a.Hs, a.W, a.shift);
            ti[lane] = tok * rsb;
            to[lane] = tok * rob;
        }
        __builtin_amdgcn_wave_barrier();
        const char* qb = reinterpret_cast<const char*>(a.qkv + b * L * 3 * a.C + head * HD);
        const char* kb = qb + 4 * a.C;
        const char* vb = qb + 8 * a.C;
        const char* dob = reinterpret_cast<const char*>(a.dout + b * L * a.C + head * HD);
        f32x4 qf[2][4], kf[2][4], vkf[2][4], dof[2][4];
        float kv[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t off = ti[32 * t + c] + 64u * h;
            load_frag16(qf[t], qb, off);
            load_frag16(kf[t], kb, off);
            load_frag16(vkf[t], vb, off);
            load_frag16(dof[t], dob, to[32 * t + c] + 64u * h);
        }
        for_pv_steps([&](int jt, int g, int e) {
            kv[jt][4 * g + e] = *reinterpret_cast<const float*>(kb + (ti[32 * jt + 8 * g + 4 * h + e] + 4u * c));
        });
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) qf[t][i] *= a.scale;
        const float* mask_w = nullptr;
        if (a.mask_id) {
            const int id = a.mask_id[bw % a.n_mask];
            if (id >= 0) mask_w = a.mask_tab + (int64_t)id * 4096;
        }
        char* dqb = reinterpret_cast<char*>(a.dqkv + b * L * 3 * a.C + head * HD);
        float* st = a.stats + (bw * a.nH + head) * 192;                   // {m[64], inv[64], D[64]} of this unit
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x16 s[2], dp[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[jt][r] = 0.f; dp[jt][r] = 0.f; }
            qk_product(s, kf, qf[it]);                                    // S^T = K Q^T
            const int qi = 32 * it + c;
            const float* brow = &bias_s[(qi < WT ? qi : WT - 1) * BLD + 4 * h];
            auto bias_at = [&](int jt, int g) {
                f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 32 * jt + 8 * g);
                if (jt == 1 && g == 2 && h) bv.x = -1e30f;
                return bv;
            };
            float m, inv;
            if (mask_w) bias_softmax<true>(s, bias_at, mask_w, qi, h, 1.0f, &m, &inv);
            else bias_softmax<false>(s, bias_at, nullptr, qi, h, 1.0f, &m, &inv);
            qk_product(dp, vkf, dof[it]);                                 // dP^T = V dO^T
            float d = 0.f;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (jt == 1 && r >= 9) continue;                      // P == 0 on padded keys
                    d += s[jt][r] * dp[jt][r];
                }
            d += __shfl_xor(d, 32);
            const bool qvalid = qi < WT;
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (jt == 1 && r >= 9) { s[jt][r] = 0.f; continue; }
                    const float ds = qvalid ? s[jt][r] * (dp[jt][r] - d) : 0.f;    // padded queries contribute nothing
                    s[jt][r] = ds;
                    dsum[it][jt][r] += ds;
                }
            if (h == 0 && qvalid) { st[qi] = m; st[64 + qi] = inv; st[128 + qi] = d; }
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            pv_product(o, s, kv);                                         // dQ = dS K   (rows = queries, lanes = channels)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (it == 1 && (r >> 2) >= 2 && !((r >> 2) == 2 && (r & 3) == 0)) continue;
                const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i < WT) *reinterpret_cast<float*>(dqb + (ti[i] + 4u * c)) = o[r] * a.scale;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    float* part = a.dbias_part + ((int64_t)blockIdx.x * 4 + wave) * 4096;  // [it][jt][r][lane]
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[((it * 2 + jt) * 16 + r) * 64 + lane] = dsum[it][jt][r];
}

__global__ __launch_bounds__(256, 1) void win_attn_bwd_kv_kernel(BwdArgs a) {
    __shared__ uint32_t tok_in[4][64];
    __shared__ uint32_t tok_out[4][64];
    __shared__ __attribute__((aligned(16))) float biasT_s[WT * BLD];      // bias^T: row = key j, column = query i
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int head = blockIdx.x % a.nH;
    const int slot = blockIdx.x / a.nH;
    const int64_t nwin = (int64_t)a.B * a.nW;
    {
        const float* bsrc = a.bias + (int64_t)head * 4096;
        for (int idx = threadIdx.x; idx < WT * WT; idx += 256) {
            const int i = idx / WT, j = idx - i * WT;
            biasT_s[j * BLD + i] = bsrc[i * 64 + j];
        }
        for (int idx = threadIdx.x; idx < WT * (64 - WT); idx += 256) {   // query columns 49..63 of every key row: finite filler
            const int j = idx / (64 - WT), i = WT + idx % (64 - WT);
            biasT_s[j * BLD + i] = 0.f;
        }
    }
    __syncthreads();
    const int64_t L = (int64_t)a.Hs * a.W;
    const uint32_t rsb = 12u * a.C, rob = 4u * a.C;
    uint32_t* ti = tok_in[wave];
    uint32_t* to = tok_out[wave];
    for (int64_t bw = (int64_t)slot * 4 + wave; bw < nwin; bw += (int64_t)a.groups * 4) {
        const int n = (int)(bw % a.nW);
        const int64_t b = bw / a.nW;
        const int wy = n / a.nWx, wx = n - wy * a.nWx;
        {
            const uint32_t tok = (uint32_t)window_token(wy, wx, lane < WT ? lane : WT - 1, a.Hs, a.W, a.shift);
            ti[lane] = tok * rsb;
            to[lane] = tok * rob;
        }
        __builtin_amdgcn_wave_barrier();
        const char* qb = reinterpret_cast<const char*>(a.qkv + b * L * 3 * a.C + head * HD);
        const char* kb = qb + 4 * a.C;
        const char* vb = qb + 8 * a.C;
        const char* dob = reinterpret_cast<const char*>(a.dout + b * L * a.C + head * HD);
        f32x4 qf[2][4], kf[2][4], vkf[2][4], dof[2][4];
        float qv[2][16], dov[2][16];                                      // query-order operands: [it][4g+e] = row 32it+8g+4h+e, lane = channel
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t off = ti[32 * t + c] + 64u * h;
            load_frag16(qf[t], qb, off);
            load_frag16(kf[t], kb, off);
            load_frag16(vkf[t], vb, off);
            load_frag16(dof[t], dob, to[32 * t + c] + 64u * h);
        }
        for_pv_steps([&](int it, int g, int e) {
            const int i = 32 * it + 8 * g + 4 * h + e;
            qv[it][4 * g + e] = *reinterpret_cast<const float*>(qb + (ti[i] + 4u * c)) * a.scale;
            dov[it][4 * g + e] = *reinterpret_cast<const float*>(dob + (to[i] + 4u * c));
        });
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) qf[t][i] *= a.scale;
        const float* mask_w = nullptr;
        if (a.mask_id) {
            const int id = a.mask_id[bw % a.n_mask];
            if (id >= 0) mask_w = a.mask_tab + (int64_t)id * 4096;
        }
        char* dkb = reinterpret_cast<char*>(a.dqkv + b * L * 3 * a.C + head * HD) + 4 * a.C;
        char* dvb = dkb + 4 * a.C;
        const float* st = a.stats + (bw * a.nH + head) * 192;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            f32x16 s[2], dp[2];                                           // [query tile it]: lane = key 32jt+c, rows = queries
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[it][r] = 0.f; dp[it][r] = 0.f; }
            qk_product(s, qf, kf[jt]);                                    // S = Q K^T   (A = q rows, B = k rows)
            qk_product(dp, dof, vkf[jt]);                                 // dP = dO V^T
            const int kj = 32 * jt + c;
            const int kjc = kj < WT ? kj : WT - 1;
            const float* brow = &biasT_s[kjc * BLD + 4 * h];
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (it == 1 && g == 3) {                              // queries 56..63: padding
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s[it][4 * g + e] = 0.f; dp[it][4 * g + e] = 0.f; }
                        continue;
                    }
                    const int i0 = 32 * it + 8 * g + 4 * h;               // this lane's 4 consecutive queries i0 .. i0+3
                    f32x4 bv = *reinterpret_cast<const f32x4*>(brow + 32 * it + 8 * g);
                    if (mask_w) bv += *reinterpret_cast<const f32x4*>(mask_w + kjc * 64 + i0);   // mask is symmetric in (i, j)
                    const f32x4 mv = *reinterpret_cast<const f32x4*>(st + i0);
                    const f32x4 iv = *reinterpret_cast<const f32x4*>(st + 64 + i0);
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(st + 128 + i0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool valid = (i0 + e < WT) && (kj < WT);
                        const float pr = valid ? __expf(s[it][4 * g + e] + bv[e] - mv[e]) * iv[e] : 0.f;
                        s[it][4 * g + e] = pr;                                             // P
                        dp[it][4 * g + e] = valid ? pr * (dp[it][4 * g + e] - dv[e]) : 0.f;   // dS
                    }
                }
            f32x16 ov, ok;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ov[r] = 0.f; ok[r] = 0.f; }
            pv_product(ov, s, dov);                                       // dV = P^T dO   (sum over queries)
            pv_product(ok, dp, qv);                                       // dK = dS^T (scale Q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (jt == 1 && (r >> 2) >= 2 && !((r >> 2) == 2 && (r & 3) == 0)) continue;
                const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (j < WT) {
                    *reinterpret_cast<float*>(dvb + (ti[j] + 4u * c)) = ov[r];
                    *reinterpret_cast<float*>(dkb + (ti[j] + 4u * c)) = ok[r];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward of the deformable cross-view attention core (deform:360-395 in window form): unit = (kv window b2, head);
// q from q window b2 % B1w (x1.repeat, deform:330), k/v from kv[b2], dO from output window b2 / r (the r-tuple sum,
// deform:394-395, hands the same dO to its r members).  Same two-orientation scheme as the self-attention backward;
// no bias (only the 49 -> 64 key padding), scale on the product.  dq_part holds each kv window's contribution to its q
// window; the caller sums the r contributions per q window.
struct CrossBwdArgs {
    const float* q; const float* kv; const float* dout;
    float* dq_part; float* dkv; float* stats;
    int C, nH, r, B1w;
    float scale;
    int64_t units;
};

__device__ __forceinline__ f32x4 pad_bias(int jt, int g, int h) {       // 0 on real keys, -1e30 on key 52 (keys 49..51 / 53.. are skipped)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (jt == 1 && g == 2 && h) bv.x = -1e30f;
    return bv;
}

__global__ __launch_bounds__(256, 1) void deform_attn_bwd_q_kernel(CrossBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= a.units) return;
    const int head = (int)(u % a.nH);
    const int64_t b2 = u / a.nH;
    const int64_t qw = b2 % a.B1w, b1 = b2 / a.r;
    const uint32_t rq = 4u * a.C, rk = 8u * a.C;                          // row pitches in bytes
    const char* qb = reinterpret_cast<const char*>(a.q + qw * WT * a.C + head * HD);
    const char* kb = reinterpret_cast<const char*>(a.kv + b2 * WT * 2 * a.C + head * HD);
    const char* vb = kb + 4 * a.C;
    const char* dob = reinterpret_cast<const char*>(a.dout + b1 * WT * a.C + head * HD);
    auto row = [](int p) { return (uint32_t)(p < WT ? p : WT - 1); };     // padded slots re-read row 48
    f32x4 qf[2][4], kf[2][4], vkf[2][4], dof[2][4];
    float kv[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t p = row(32 * t + c);
        load_frag16(qf[t], qb, p * rq + 64u * h);
        load_frag16(kf[t], kb, p * rk + 64u * h);
        load_frag16(vkf[t], vb, p * rk + 64u * h);
        load_frag16(dof[t], dob, p * rq + 64u * h);
    }
    for_pv_steps([&](int jt, int g, int e) {
        kv[jt][4 * g + e] = *reinterpret_cast<const float*>(kb + (row(32 * jt + 8 * g + 4 * h + e) * rk + 4u * c));
    });
    char* dqb = reinterpret_cast<char*>(a.dq_part + b2 * WT * a.C + head * HD);
    float* st = a.stats + u * 192;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        f32x16 s[2], dp[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[jt][r] = 0.f; dp[jt][r] = 0.f; }
        qk_product(s, kf, qf[it]);
        const int qi = 32 * it + c;
        float m, inv;
        bias_softmax<false>(s, [&](int jt, int g) { return pad_bias(jt, g, h); }, nullptr, qi, h, a.scale, &m, &inv);
        qk_product(dp, vkf, dof[it]);
        float d = 0.f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (jt == 1 && r >= 9) continue;
                d += s[jt][r] * dp[jt][r];
            }
        d += __shfl_xor(d, 32);
        const bool qvalid = qi < WT;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (jt == 1 && r >= 9) { s[jt][r] = 0.f; continue; }
                s[jt][r] = qvalid ? s[jt][r] * (dp[jt][r] - d) : 0.f;
            }
        if (h == 0 && qvalid) { st[qi] = m; st[64 + qi] = inv; st[128 + qi] = d; }
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        pv_product(o, s, kv);                                             // dQ = dS K
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (it == 1 && (r >> 2) >= 2 && !((r >> 2) == 2 && (r & 3) == 0)) continue;
            const int i = 32 * it + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (i < WT) *reinterpret_cast<float*>(dqb + ((uint32_t)i * rq + 4u * c)) = o[r] * a.scale;
        }
    }
}

__global__ __launch_bounds__(256, 1) void deform_attn_bwd_kv_kernel(CrossBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t u = (int64_t)blockIdx.x * 4 + wave;
    if (u >= a.units) return;
    const int head = (int)(u % a.nH);
    const int64_t b2 = u / a.nH;
    const int64_t qw = b2 % a.B1w, b1 = b2 / a.r;
    const uint32_t rq = 4u * a.C, rk = 8u * a.C;
    const char* qb = reinterpret_cast<const char*>(a.q + qw * WT * a.C + head * HD);
    const char* kb = reinterpret_cast<const char*>(a.kv + b2 * WT * 2 * a.C + head * HD);
    const char* vb = kb + 4 * a.C;
    const char* dob = reinterpret_cast<const char*>(a.dout + b1 * WT * a.C + head * HD);
    auto row = [](int p) { return (uint32_t)(p < WT ? p : WT - 1); };
    f32x4 qf[2][4], kf[2][4], vkf[2][4], dof[2][4];
    float qv[2][16], dov[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t p = row(32 * t + c);
        load_frag16(qf[t], qb, p * rq + 64u * h);
        load_frag16(kf[t], kb, p * rk + 64u * h);
        load_frag16(vkf[t], vb, p * rk + 64u * h);
        load_frag16(dof[t], dob, p * rq + 64u * h);
    }
    for_pv_steps([&](int it, int g, int e) {
        const uint32_t i = row(32 * it + 8 * g + 4 * h + e);
        qv[it][4 * g + e] = *reinterpret_cast<const float*>(qb + (i * rq + 4u * c));
        dov[it][4 * g + e] = *reinterpret_cast<const float*>(dob + (i * rq + 4u * c));
    });
    char* dkb = reinterpret_cast<char*>(a.dkv + b2 * WT * 2 * a.C + head * HD);
    char* dvb = dkb + 4 * a.C;
    const float* st = a.stats + u * 192;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        f32x16 s[2], dp[2];
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[it][r] = 0.f; dp[it][r] = 0.f; }
        qk_product(s, qf, kf[jt]);                                        // S = Q K^T (unscaled; scale applied below)
        qk_product(dp, dof, vkf[jt]);                                     // dP = dO V^T
        const int kj = 32 * jt + c;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (it == 1 && g == 3) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s[it][4 * g + e] = 0.f; dp[it][4 * g + e] = 0.f; }
                    continue;
                }
                const int i0 = 32 * it + 8 * g + 4 * h;
                const f32x4 mv = *reinterpret_cast<const f32x4*>(st + i0);
                const f32x4 iv = *reinterpret_cast<const f32x4*>(st + 64 + i0);
                const f32x4 dv = *reinterpret_cast<const f32x4*>(st + 128 + i0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool valid = (i0 + e < WT) && (kj < WT);
                    const float pr = valid ? __expf(s[it][4 * g + e] * a.scale - mv[e]) * iv[e] : 0.f;
                    s[it][4 * g + e] = pr;
                    dp[it][4 * g + e] = valid ? pr * (dp[it][4 * g + e] - dv[e]) * a.scale : 0.f;   // scale: dS/d(q k)
                }
            }
        f32x16 ov, ok;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ov[r] = 0.f; ok[r] = 0.f; }
        pv_product(ov, s, dov);                                           // dV = P^T dO
        pv_product(ok, dp, qv);                                           // dK = scale dS^T Q
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (jt == 1 && (r >> 2) >= 2 && !((r >> 2) == 2 && (r & 3) == 0)) continue;
            const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (j < WT) {
                *reinterpret_cast<float*>(dvb + ((uint32_t)j * rk + 4u * c)) = ov[r];
                *reinterpret_cast<float*>(dkb + ((uint32_t)j * rk + 4u * c)) = ok[r];
            }
        }
    }
}

// dbias_full[head][i][j] = sum over that head's wave partials (fixed order: 4 lane groups take every 4th partial, then
// the groups are combined in order); partial layout [it][jt][r][lane]
__global__ __launch_bounds__(256) void win_attn_dbias_reduce_kernel(const float* __restrict__ part, float* __restrict__ full, int nH,
                                                                    int nblocks) {
    __shared__ float red[4][64];
    const int head = blockIdx.y;
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + col;                                // element of the 4096-float partial
    const int nparts = ((nblocks - head + nH - 1) / nH) * 4;             // this head's blocks x 4 waves
    // lane group grp takes wave grp of every block of this head, blocks in order: terms nH * 4 * 4096 floats apart
    float s = 0.f;
    if (grp < nparts)
        s = ordered_sum(part[((int64_t)head * 4 + grp) * 4096 + idx], part + ((int64_t)(head + nH) * 4 + grp) * 4096 + idx,
                        (int64_t)nH * 4 * 4096, nparts / 4 - 1);
    red[grp][col] = s;
    __syncthreads();
    if (grp) return;
    s = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    const int lane = idx & 63, r = (idx >> 6) & 15, jt = (idx >> 10) & 1, it = idx >> 11;
    const int i = 32 * it + (lane & 31), j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    full[(int64_t)head * 4096 + i * 64 + j] = s;
}

// dtable[t][head] = sum over the (i, j) pairs with relative_position_index[i][j] == t: one wave per (t, head), lanes
// stride over the 2401 pairs in order, then a fixed-order wave reduction
__global__ __launch_bounds__(64) void win_attn_dtable_kernel(const float* __restrict__ full, const int32_t* __restrict__ rel_index,
                                                             float* __restrict__ dtable, int nH, int ntab, int accum) {
    const int t = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
    float s = 0.f;
    for (int p = lane; p < WT * WT; p += 64) {
        const int i = p / WT, j = p - i * WT;
        if (rel_index[p] == t) s += full[(int64_t)head * 4096 + i * 64 + j];
    }
    s = wave_sum(s, 64);
    if (lane == 0) dtable[(int64_t)t * nH + head] = accum ? dtable[(int64_t)t * nH + head] + s : s;
}

// the same through the inverse of relative_position_index (built once by the caller): csr = [ptr (ntab + 1) | pairs (49*49)], the pairs
// p = 49 i + j of table entry t are pairs[ptr[t] .. ptr[t+1]) in increasing p.  A wave per (t, head) reads ITS <= 49 values (the scan
// above walks all 2401 index entries in every one of the 169 x nH waves: 20 us per Swin block of the training step, 60 blocks).
__global__ __launch_bounds__(256) void win_attn_dtable_csr_kernel(const float* __restrict__ full, const int32_t* __restrict__ csr,
                                                                 float* __restrict__ dtable, int nH, int ntab, int accum) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6), head = blockIdx.y;
    if (t >= ntab) return;
    const int p0 = csr[t], p1 = csr[t + 1];
    float s = 0.f;
    for (int e = p0 + lane; e < p1; e += 64) {
        const int p = csr[ntab + 1 + e], i = p / WT, j = p - i * WT;
        s += full[(int64_t)head * 4096 + i * 64 + j];
    }
    s = wave_sum(s, 64);
    if (lane == 0) dtable[(int64_t)t * nH + head] = accum ? dtable[(int64_t)t * nH + head] + s : s;
}

// relative_position_bias_table (169, nH) + relative_position_index (49*49, int32) -> padded bias (nH, 64, 64) [head][query][key]:
// rows >= 49 zero, key columns >= 49 = -1e30 (the 49 -> 64 padding mask of the attention kernels; swin:148-151)
__global__ __launch_bounds__(256) void relpos_bias_expand_kernel(const float* __restrict__ table, const int32_t* __restrict__ rel_index,
                                                                 float* __restrict__ out, int nH) {
    // one element per thread (grid (nH, 16)): the two dependent loads of an element are the whole latency of the kernel -- sixteen
    // elements per thread in sequence made this 4096-element gather take 10 us, 60 times per training step
    const int head = blockIdx.x, e = blockIdx.y * 256 + threadIdx.x;
    const int i = e >> 6, j = e & 63;
    float v = 0.f;
    if (j >= WT) v = -1e30f;
    else if (i < WT) v = table[(int64_t)rel_index[i * WT + j] * nH + head];
    out[(int64_t)head * 4096 + e] = v;
}

}  // namespace

static int window_attention_launch(int kind, const float* qkv, float* out, const float* bias, const float* mask_tab,
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
    static const int dbgmask = tune_int("MUMPY_WA_DBG", 0);
    a.dbg = dbgmask;
    a.units = (int64_t)B * a.nW * a.nH;
    // persistent grid: ~3 resident blocks per CU (3 waves/SIMD); each block walks its head's window quads
    const int64_t quads = ((int64_t)B * a.nW + 3) / 4;
    static const int wa_blocks = tune_int("MUMPY_WA_BLOCKS", 768);
    static const int wa_stagger = tune_int("MUMPY_WA_STAGGER", 0);
    int64_t groups = (wa_blocks + a.nH - 1) / a.nH;
    if (groups > quads) groups = quads;
    a.groups = (int)groups; a.stagger = wa_stagger;
    const int64_t grid = groups * a.nH;
    const bool io16 = kind == 1;
    if (kind == 2) hipLaunchKernelGGL((win_attn_self_kernel<false, false, true>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    else if (io16) hipLaunchKernelGGL((win_attn_self_kernel<false, true>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    else if (dbgmask) hipLaunchKernelGGL((win_attn_self_kernel<true, false>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((win_attn_self_kernel<false, false>), dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("window_attention");
    return 0;
}

extern "C" int mumpy_window_attention_fwd(const float* qkv, float* out, const float* bias, const float* mask_tab,
                                          const int32_t* mask_id, int n_mask, int B, int Hs, int W, int C, int shift,
                                          float scale, void* stream) {
    return window_attention_launch(0, qkv, out, bias, mask_tab, mask_id, n_mask, B, Hs, W, C, shift, scale, stream);
}

// "Background" form: identical arithmetic and results, NO LDS allocation (token tables in registers via ds_bpermute, bias rows from
// L1), so that the launch can be resident beside the persistent GEMM, which owns every CU's whole LDS (see gemm_rd.hip).
extern "C" int mumpy_window_attention_bg_fwd(const float* qkv, float* out, const float* bias, const float* mask_tab,
                                             const int32_t* mask_id, int n_mask, int B, int Hs, int W, int C, int shift,
                                             float scale, void* stream) {
    return window_attention_launch(2, qkv, out, bias, mask_tab, mask_id, n_mask, B, Hs, W, C, shift, scale, stream);
}

// bf16 STORAGE: qkv (B, Hs*W, 3C) and out (B, Hs*W, C) are bf16; bias / mask tables fp32; same arithmetic.
extern "C" int mumpy_window_attention_bf16_fwd(const void* qkv, void* out, const float* bias, const float* mask_tab,
                                               const int32_t* mask_id, int n_mask, int B, int Hs, int W, int C, int shift,
                                               float scale, void* stream) {
    return window_attention_launch(1, static_cast<const float*>(qkv), static_cast<float*>(out), bias, mask_tab, mask_id, n_mask,
                                   B, Hs, W, C, shift, scale, stream);
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

static int64_t wa_bwd_groups(int B, int nW, int nH) {
    const int64_t quads = ((int64_t)B * nW + 3) / 4;
    static const int target = tune_int("MUMPY_WA_BWD_BLOCKS", 256);
    int64_t groups = (target + nH - 1) / nH;              // ~one 4-wave block per CU (1 wave per SIMD)
    return groups > quads ? quads : groups;
}

extern "C" int64_t mumpy_window_attention_bwd_workspace_bytes(int B, int Hs, int W, int C) {
    if (B <= 0 || Hs <= 0 || W <= 0 || C <= 0 || Hs % WS || W % WS || C % HD) return 0;
    const int nW = (Hs / WS) * (W / WS), nH = C / HD;
    const int64_t stats = (int64_t)B * nW * nH * 192;
    const int64_t part = wa_bwd_groups(B, nW, nH) * nH * 4 * 4096;
    return (stats + part + (int64_t)nH * 4096) * (int64_t)sizeof(float);
}

static int window_attention_bwd_impl(const float* qkv, const float* dout, const float* bias, const float* mask_tab,
                                          const int32_t* mask_id, int n_mask, const int32_t* rel_index, const int32_t* rel_csr, float* dqkv,
                                          float* dtable, void* workspace, int64_t workspace_bytes, int B, int Hs, int W, int C,
                                          int shift, float scale, int accumulate, void* stream) {
    MUMPY_REQUIRE(qkv && dout && bias && rel_index && dqkv && dtable && workspace, MUMPY_ENULL, "window_attention_bwd: null pointer");
    MUMPY_REQUIRE(accumulate == 0 || accumulate == 1, MUMPY_EINVAL, "window_attention_bwd: accumulate must be 0 or 1");
    MUMPY_REQUIRE((mask_tab == nullptr) == (mask_id == nullptr), MUMPY_ENULL,
                  "window_attention_bwd: mask_tab and mask_id must be given together");
    MUMPY_REQUIRE(aligned16(qkv) && aligned16(dout) && aligned16(bias) && aligned16(mask_tab) && aligned16(dqkv) &&
                      aligned16(workspace), MUMPY_EALIGN, "window_attention_bwd: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && Hs > 0 && W > 0 && Hs % WS == 0 && W % WS == 0, MUMPY_EINVAL,
                  "window_attention_bwd: grid (%d,%d) not divisible by window 7", Hs, W);
    MUMPY_REQUIRE(C > 0 && C % HD == 0 && shift >= 0 && shift < WS, MUMPY_EINVAL, "window_attention_bwd: bad C=%d / shift=%d", C, shift);
    MUMPY_REQUIRE(mask_id == nullptr || n_mask > 0, MUMPY_EINVAL, "window_attention_bwd: n_mask must be > 0 with a mask");
    MUMPY_REQUIRE(workspace_bytes >= mumpy_window_attention_bwd_workspace_bytes(B, Hs, W, C), MUMPY_EINVAL,
                  "window_attention_bwd: workspace too small");
    BwdArgs a;
    a.qkv = qkv; a.dout = dout; a.bias = bias; a.mask_tab = mask_tab; a.mask_id = mask_id; a.dqkv = dqkv;
    a.B = B; a.Hs = Hs; a.W = W; a.C = C; a.nH = C / HD; a.shift = shift; a.nWx = W / WS; a.nW = (Hs / WS) * (W / WS);
    a.n_mask = n_mask > 0 ? n_mask : 1; a.scale = scale;
    a.groups = (int)wa_bwd_groups(B, a.nW, a.nH);
    float* ws = static_cast<float*>(workspace);
    a.stats = ws;
    a.dbias_part = ws + (int64_t)B * a.nW * a.nH * 192;
    float* full = a.dbias_part + (int64_t)a.groups * a.nH * 4 * 4096;
    const unsigned grid = (unsigned)(a.groups * a.nH);
    hipLaunchKernelGGL(win_attn_bwd_q_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("window_attention_bwd(q)");
    hipLaunchKernelGGL(win_attn_bwd_kv_kernel, dim3(grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("window_attention_bwd(kv)");
    hipLaunchKernelGGL(win_attn_dbias_reduce_kernel, dim3(64, a.nH), dim3(256), 0, as_stream(stream), a.dbias_part, full, a.nH,
                       (int)grid);
    MUMPY_CHECK_LAUNCH("window_attention_bwd(dbias reduce)");
    const int ntab = (2 * WS - 1) * (2 * WS - 1);
    if (rel_csr)
        hipLaunchKernelGGL(win_attn_dtable_csr_kernel, dim3((ntab + 3) / 4, a.nH), dim3(256), 0, as_stream(stream), full, rel_csr, dtable,
                           a.nH, ntab, accumulate);
    else
        hipLaunchKernelGGL(win_attn_dtable_kernel, dim3(ntab, a.nH), dim3(64), 0, as_stream(stream), full, rel_index, dtable, a.nH,
                           ntab, accumulate);
    MUMPY_CHECK_LAUNCH("window_attention_bwd(dtable)");
    return 0;
}

extern "C" int mumpy_window_attention_bwd(const float* qkv, const float* dout, const float* bias, const float* mask_tab,
                                          const int32_t* mask_id, int n_mask, const int32_t* rel_index, float* dqkv,
                                          float* dtable, void* workspace, int64_t workspace_bytes, int B, int Hs, int W, int C,
                                          int shift, float scale, int accumulate, void* stream) {
    return window_attention_bwd_impl(qkv, dout, bias, mask_tab, mask_id, n_mask, rel_index, nullptr, dqkv, dtable, workspace,
                                     workspace_bytes, B, Hs, W, C, shift, scale, accumulate, stream);
}

extern "C" int mumpy_window_attention_bwd_csr(const float* qkv, const float* dout, const float* bias, const float* mask_tab,
                                              const int32_t* mask_id, int n_mask, const int32_t* rel_index, const int32_t* rel_csr,
                                              float* dqkv, float* dtable, void* workspace, int64_t workspace_bytes, int B, int Hs, int W,
                                              int C, int shift, float scale, int accumulate, void* stream) {
    MUMPY_REQUIRE(rel_csr, MUMPY_ENULL, "window_attention_bwd_csr: null inverse index");
    return window_attention_bwd_impl(qkv, dout, bias, mask_tab, mask_id, n_mask, rel_index, rel_csr, dqkv, dtable, workspace,
                                     workspace_bytes, B, Hs, W, C, shift, scale, accumulate, stream);
}

extern "C" int mumpy_relpos_bias_expand_fwd(const float* table, const int32_t* rel_index, float* out, int nH, void* stream) {
    MUMPY_REQUIRE(table && rel_index && out, MUMPY_ENULL, "relpos_bias_expand: null pointer");
    MUMPY_REQUIRE(nH > 0, MUMPY_EINVAL, "relpos_bias_expand: bad head count %d", nH);
    hipLaunchKernelGGL(relpos_bias_expand_kernel, dim3((unsigned)nH, 16), dim3(256), 0, as_stream(stream), table, rel_index, out, nH);
    MUMPY_CHECK_LAUNCH("relpos_bias_expand");
    return 0;
}

extern "C" int64_t mumpy_deform_attention_bwd_workspace_bytes(int64_t B2w, int C) {
    return (B2w <= 0 || C <= 0) ? 0 : B2w * (C / HD) * 192 * (int64_t)sizeof(float);
}

extern "C" int mumpy_deform_attention_bwd(const float* q, const float* kv, const float* dout, float* dq_part, float* dkv,
                                          void* workspace, int64_t workspace_bytes, int64_t B1w, int r, int C, float scale,
                                          void* stream) {
    MUMPY_REQUIRE(q && kv && dout && dq_part && dkv && workspace, MUMPY_ENULL, "deform_attention_bwd: null pointer");
    MUMPY_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(dout) && aligned16(dq_part) && aligned16(dkv) && aligned16(workspace),
                  MUMPY_EALIGN, "deform_attention_bwd: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B1w > 0 && r >= 1 && C > 0 && C % HD == 0, MUMPY_EINVAL, "deform_attention_bwd: bad shape");
    const int64_t B2w = B1w * r;
    MUMPY_REQUIRE(workspace_bytes >= mumpy_deform_attention_bwd_workspace_bytes(B2w, C), MUMPY_EINVAL,
                  "deform_attention_bwd: workspace too small");
    CrossBwdArgs a;
    a.q = q; a.kv = kv; a.dout = dout; a.dq_part = dq_part; a.dkv = dkv; a.stats = static_cast<float*>(workspace);
    a.C = C; a.nH = C / HD; a.r = r; a.B1w = (int)B1w; a.scale = scale; a.units = B2w * a.nH;
    const int64_t grid = (a.units + 3) / 4;
    MUMPY_REQUIRE(grid < (1ll << 31) && B1w < (1ll << 31), MUMPY_ERANGE, "deform_attention_bwd: too many windows");
    hipLaunchKernelGGL(deform_attn_bwd_q_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("deform_attention_bwd(q)");
    hipLaunchKernelGGL(deform_attn_bwd_kv_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
    MUMPY_CHECK_LAUNCH("deform_attention_bwd(kv)");
    return 0;
}
