// gemm_ws64.h — the persistent wave-specialised fp32 GEMM of gemm_ws.h at 64x64 tiles, TWO workgroups per CU, for the
// mid-size nn.Linear shapes (a few hundred to ~2000 rows: views 1 / 2 of stage 2, the global blocks' projections, the
// stage-0/1/3 layers -- swin:46-49,142,164).  Why: on those shapes the one-role tiled kernel of gemm.hip spends a third of a
// block's life in its epilogue (the fp32 MFMA stream of the co-resident blocks leaves a bias + GELU + store sequence one vector
// instruction per ~49 cycles: tools/micro/coissue.hip), and 128x128 tiles are too few to fill the chip.  Same three roles,
// same stage image, swizzle, barrier protocol and tile order as gemm_ws.h (whose header explains them); what differs:
//   * matrix waves own 32x32 of the tile: per 8-deep sub-step 2 ds_read_b128 feed 4 MFMAs (twice the reads per MFMA of the
//     128x128 kernel -- the price of the small tile);
//   * loader waves issue 4 LDS-DMA pieces per chunk (2 x 8 rows of x, 2 x 8 rows of W each);
//   * the epilogue waves need 4 passes per tile (16 rows x 64 columns each);
//   * 3 stages x 128 rows x 128 B + the 64x64 accumulator image = 65,536 B of LDS: two workgroups per CU, so a CU's four
//     matrix pipes see two matrix waves each and one workgroup's tile change or tail hides behind the other's MFMAs.
// Whole tiles only (no split schedule), fp32 only.
#pragma once
#include "gemm_ws.h"

namespace mumpy {
namespace gemm_ws64 {

using gemm_ws::BK;
using gemm_ws::OOB;
using gemm_ws::Params;
using gemm_ws::tile_coords;
using gemm_ws::u32x4;
using gemm_ws::ws_barrier;

constexpr int T = 64;
constexpr int STAGE_DW = 2 * T * BK;
constexpr int NSTAGE = 3;
constexpr int E_DW = T * T;
constexpr int E_OFF_DW = NSTAGE * STAGE_DW;
constexpr int LDS_BYTES = (E_OFF_DW + E_DW) * 4;
constexpr int PASSES = 4;                   // 16 rows (256 epilogue lanes x 16 B = 16 rows x 64 columns) per pass

__device__ __forceinline__ void matrix_role(const Params& p, float* lds, int n_chunks, int wave, int lane) {
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int sw = (c >> 1) & 7;
    int a_off[4], b_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a_off[q] = (32 * wm + c) * BK + 4 * ((4 * h + q) ^ sw);
        b_off[q] = (T + 32 * wn + c) * BK + 4 * ((4 * h + q) ^ sw);
    }
    float* const E = lds + E_OFF_DW + (32 * wm + 4 * h) * T + 32 * wn + c;
    f32x16 acc;
    f32x4 fa0, fb0, fa1, fb1;
    auto rd = [&](const float* st, int q, f32x4& fa, f32x4& fb) {
        fa = *reinterpret_cast<const f32x4*>(st + a_off[q]);
        fb = *reinterpret_cast<const f32x4*>(st + b_off[q]);
    };
    auto mm = [&](const f32x4& fa, const f32x4& fb, int e) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0); };
    auto mm4 = [&](const f32x4& fa, const f32x4& fb) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mm(fa, fb, e);
    };
    auto dump = [&]() {          // D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h
#pragma unroll
        for (int r = 0; r < 16; ++r) E[((r & 3) + 8 * (r >> 2)) * T] = acc[r];
    };
    auto zero = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    };
    zero();
    ws_barrier();                                   // chunk 0 is staged
    rd(lds, 0, fa0, fb0);
    int kc = 0, stage = 0;
    for (int i = 0; i < n_chunks; ++i) {
        const float* st = lds + stage * STAGE_DW;
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        if (kc == 0 && i > 0) { dump(); zero(); }
        __builtin_amdgcn_sched_barrier(0);
        rd(st, 1, fa1, fb1);
        mm4(fa0, fb0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        rd(st, 2, fa0, fb0);
        mm4(fa1, fb1);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        rd(st, 3, fa1, fb1);
        mm4(fa0, fb0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        mm(fa1, fb1, 0);
        mm(fa1, fb1, 1);
        __builtin_amdgcn_sched_barrier(0);
        ws_barrier();                               // chunk i+1 is staged
        __builtin_amdgcn_sched_barrier(0);
        rd(lds + stage * STAGE_DW, 0, fa0, fb0);    // (after the last chunk: a harmless read of an idle stage)
        mm(fa1, fb1, 2);
        mm(fa1, fb1, 3);
        __builtin_amdgcn_sched_barrier(0);
        if (++kc == p.nk) kc = 0;
    }
    dump();
    ws_barrier();                                   // the last tile's accumulators are in LDS
}

// waves 4-7: 2 + 2 LDS-DMA pieces per chunk and wave (8 tile rows x 128 B each), three stages, counted vmcnt(4)
__device__ __forceinline__ void loader_role(const Params& p, float* lds, unsigned tile0, int n_chunks, int hl) {
    const int lane = hl & 63, lw = __builtin_amdgcn_readfirstlane(hl >> 6);
    const int prow = lane >> 3;
    uint32_t aoff[2], boff[2];
    auto set_tile = [&](unsigned t) {
        unsigned tm, tn;
        tile_coords(p, t, tm, tn);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = 32 * q + 8 * lw + prow;
            const uint32_t ch = (uint32_t)((lane & 7) ^ ((r >> 1) & 7));
            int m = (int)tm * T + r;
            if (m > p.M - 1) m = p.M - 1;           // rows past the edge are clamped: their products are never stored
            aoff[q] = (uint32_t)m * (uint32_t)p.K * 4u + 16u * ch;
            int n = (int)tn * T + r;
            if (n > p.N - 1) n = p.N - 1;
            boff[q] = (uint32_t)n * (uint32_t)p.K * 4u + 16u * ch;
        }
    };
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)0x7fffffff, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)0x7fffffff, 0x00020000);
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](int kc, int stage) {
        const int so = kc * 128;
        float* st = lds + stage * STAGE_DW + 8 * lw * BK;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(st + 32 * q * BK), 16, aoff[q], so, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(st + (T + 32 * q) * BK), 16, boff[q], so, 0, 0);
    };
    unsigned ld_tile = tile0;
    int ld_kc = 0, ld_idx = 0;
    auto advance = [&]() {
        if (ld_idx + 1 < n_chunks) {
            ++ld_idx;
            if (++ld_kc == p.nk) { ld_kc = 0; ++ld_tile; set_tile(ld_tile); }
        }
    };
    set_tile(ld_tile);
    dma(ld_kc, 0);
    advance();
    dma(ld_kc, 1);
    advance();
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            // chunk 0 has landed
    ws_barrier();
    int stage = 2;
    for (int i = 0; i < n_chunks; ++i) {
        dma(ld_kc, stage);                                       // chunk i+2 -> the stage chunk i-1 was read from
        advance();
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        // chunk i+1 has landed
        ws_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no DMA may outlive the workgroup's LDS allocation
    ws_barrier();
}

// waves 8-11: the previous tile's image -> bias, exact-erf GELU, residual, 16-B stores; P passes per chunk
template <int P>
__device__ __forceinline__ void epilogue_role(const Params& p, float* lds, unsigned v0, unsigned n_tiles, int hl) {
    const float* const E = lds + E_OFF_DW;
    const int e_row = hl >> 4, e_c4 = hl & 15;
    const int nk = p.nk;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)((int64_t)p.M * p.N * 4), 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.Y), 0,
                                                        p.residual ? (int)((int64_t)p.M * p.N * 4) : 0, 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.Y), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    constexpr int STEPS = PASSES / P;
    f32x4 rv[PASSES], bias4;
    uint32_t yo[PASSES];
    const uint32_t row16 = 16u * (uint32_t)p.N * 4u;
    auto begin_tile = [&](unsigned t) {
        unsigned tm, tn;
        tile_coords(p, t, tm, tn);
        const int n = (int)tn * T + 4 * e_c4;
        const uint32_t ybase = n < p.N ? (((uint32_t)tm * T + e_row) * (uint32_t)p.N + (uint32_t)n) * 4u : OOB;
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            yo[e] = ybase + (uint32_t)e * row16;
            rv[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, yo[e], 0, 0));
        }
        bias4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, n < p.N ? (uint32_t)n * 4u : OOB, 0, 0));
    };
    auto pass = [&](int e) {
        f32x4 v = *reinterpret_cast<const f32x4*>(E + (16 * e + e_row) * T + 4 * e_c4) + bias4;
        if (p.act == MUMPY_ACT_GELU) {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = gelu_erf(v[x]);
        }
        v += rv[e];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, yo[e], 0, 0);
    };
    ws_barrier();                                   // chunk 0 is staged
    for (int c = 0; c < nk; ++c) ws_barrier();      // first tile: nothing to write out yet
    for (unsigned t = 1; t < n_tiles; ++t) {
        begin_tile(v0 + t - 1);
        ws_barrier();                               // chunk 0 of tile t: the previous tile's image is being dumped
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {        // chunks 1 .. STEPS: P passes each
#pragma unroll
            for (int k = 0; k < P; ++k) pass(st * P + k);
            ws_barrier();
        }
        for (int c = 1 + STEPS; c < nk; ++c) ws_barrier();
    }
    begin_tile(v0 + n_tiles - 1);
    ws_barrier();                                   // the last tile's accumulators are in LDS
#pragma unroll
    for (int e = 0; e < PASSES; ++e) pass(e);
}

template <int P>
__global__ __launch_bounds__(768, 6) void gemm_ws64_kernel(Params p) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned G = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = G >> 3, r8 = G & 7;
    const unsigned b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const unsigned cnt = b < p.tiles ? (p.tiles - b + G - 1) / G : 0;     // whole tiles, dealt round-robin (gemm_ws.h)
    if (cnt == 0) return;
    const unsigned v0 = b * p.rr_cnt;
    const int n_chunks = (int)cnt * p.nk;
    if (wave < 4) matrix_role(p, lds, n_chunks, wave, lane);
    else {
        __builtin_amdgcn_s_setprio(3);
        if (wave < 8) loader_role(p, lds, v0, n_chunks, tid - 256);
        else epilogue_role<P>(p, lds, v0, cnt, tid - 512);
    }
}

inline bool eligible(int64_t M, int N, int K) { return gemm_ws::eligible(M, N, K); }

// per_cu: workgroups per CU (2, or 1: the launch then claims the CU's LDS so that a second workgroup cannot join -- with a
// GELU epilogue one matrix wave per SIMD leaves the epilogue waves enough issue slots, two do not)
inline int launch(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M, int N, int K,
                  int act, int num_cu, hipStream_t s, int per_cu = 2) {
    Params p;
    p.cv_H = p.cv_W = p.cv_C = p.cv_kh = p.cv_kw = p.cv_cpc = 0;
    p.cv_mhw = p.cv_shw = p.cv_mw = p.cv_sw = 0;
    p.X = x; p.W = W; p.bias = bias; p.residual = residual; p.Y = y;
    p.M = (int)M; p.N = N; p.K = K; p.act = act; p.nk = K / BK;
    p.gm = (unsigned)((M + T - 1) / T); p.gn = (unsigned)((N + T - 1) / T);
    p.tiles = p.gm * p.gn;
    p.units = p.tiles * (unsigned)p.nk;
#ifdef MUMPY_WS_STAMP
    p.stamps = nullptr;
#endif
    const int need = (PASSES + p.nk - 2) / (p.nk - 1);           // passes per chunk so that a tile's epilogue fits under the next tile
    const int P = need <= 1 ? 1 : need <= 2 ? 2 : 4;
    p.lmin = 0;
    const unsigned slots = (unsigned)per_cu * (unsigned)num_cu;
    const int lds_bytes = per_cu == 1 ? 2 * LDS_BYTES + 1024 : LDS_BYTES;
    const unsigned grid = p.tiles < slots ? p.tiles : slots;
    p.st_w = (p.gn % 4 == 0) ? 4u : (p.gn % 2 == 0) ? 2u : 1u;
    p.rr_G = grid;
    p.rr_cnt = (p.tiles + grid - 1) / grid;
    p.flags = nullptr; p.slabs = nullptr;
#define MUMPY_WS64_LAUNCH(P_)                                                                                           \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws64_kernel<P_>),                     \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_BYTES + 1024);       \
            if (e != hipSuccess) { set_error("gemm_ws64: cannot reserve %d B of LDS: %s", 2 * LDS_BYTES + 1024, hipGetErrorString(e)); return (int)e; } \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL(gemm_ws64_kernel<P_>, dim3(grid), dim3(768), lds_bytes, s, p);                               \
    } while (0)
    if (P == 1) MUMPY_WS64_LAUNCH(1);
    else if (P == 2) MUMPY_WS64_LAUNCH(2);
    else MUMPY_WS64_LAUNCH(4);
#undef MUMPY_WS64_LAUNCH
    return 0;
}

}  // namespace gemm_ws64
}  // namespace mumpy
