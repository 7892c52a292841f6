// gemm_ws.h — persistent, wave-specialised fp32 GEMM for the large dense nn.Linear shapes of the path
// (swin:46-49,142,164; blocks:27-33,57-71): y = act(x W^T + bias) + residual on v_mfma_f32_32x32x2_f32.
//
// One 512-thread workgroup per CU, looping over 128x128 output tiles:
//   * waves 0-3 ("matrix waves", one per SIMD, 64x64 of the tile each) do nothing but ds_read_b128 fragment reads and
//     MFMAs: per 8-deep K sub-step 4 reads feed 16 MFMAs (1024 matrix-pipe cycles), so the pipe is issued back to back by
//     ONE wave per SIMD; the only synchronisation is one s_barrier per 32-deep chunk, placed in the middle of the chunk's
//     last 16 MFMAs (the MFMA ahead of it is still executing while the wave sits in the barrier);
//   * waves 4-7 ("helper waves") stage the operands (global -> registers -> LDS, three register sets, so two chunks of
//     loads are always in flight) and run the epilogue of the PREVIOUS tile from an LDS image of its accumulators: bias,
//     exact-erf GELU, residual, 16-B row-contiguous stores -- all under the next tile's MFMAs.  The matrix waves pay ~300
//     cycles per tile to dump their 64 accumulator registers to LDS; prologue, epilogue and store latency leave the matrix
//     pipe's critical path, which is what held the one-role kernels of gemm.hip at ~70 % matrix-pipe occupancy.
// The helper's loop body is STRAIGHT-LINE code: every load and store is issued every iteration, predicated by an
// out-of-range buffer offset instead of a branch, and the residual / bias operands of an epilogue pass are fetched one
// iteration ahead.  vmcnt retires in issue order, so one wait on a freshly issued load (or a conservative vmcnt(0) at a
// control-flow join, which is what hipcc emits after a branch that contains a memory operation) would drain the two
// chunks of staging loads in flight and put the memory latency on the barrier the matrix waves wait at: measured, the
// branchy first version ran 84 TFLOP/s where its matrix waves alone reach 126.
// LDS: 3 stages x 256 rows x 32 dwords, XOR-swizzled 16-B chunks (98,304 B) + the 128x128 accumulator image (65,536 B)
// = 163,840 B, the whole CU.
#pragma once
#include "common.h"

namespace mumpy {
namespace gemm_ws {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int STAGE_DW = (BM + BN) * BK;
constexpr int NSTAGE = 3;
constexpr int E_DW = BM * BN;
constexpr int E_OFF_DW = NSTAGE * STAGE_DW;
constexpr int LDS_BYTES = (E_OFF_DW + E_DW) * 4;
constexpr int PASSES = BM / 8;              // epilogue passes per tile: 8 rows (4 helper waves x 2 rows) each
constexpr unsigned NG = 4;                  // N-tiles per group of the tile order
#ifndef MUMPY_WS_DBG
#define MUMPY_WS_DBG 0      // harness diagnostics, compile time (a runtime switch would put branches around the loads): 1 = no operand loads, 2 = no epilogue, 4 = no LDS staging writes, 8 = no priority
#endif
constexpr int DBG = MUMPY_WS_DBG;
constexpr uint32_t OOB = 0x80000000u;       // buffer offset past every buffer: the access is dropped by the bounds check

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Params {
    const float* X;
    const float* W;
    const float* bias;
    const float* residual;
    float* Y;
    int M, N, K, act;
    int nk;                // K / 32
    unsigned gm, gn;       // tiles along M, N
    unsigned tiles;        // gm * gn
    unsigned sched_S, sched_SN;   // experiment: > 0 = lockstep 8x4 super-tiles per XCD (S per XCD, SN across N)
#ifdef MUMPY_WS_STAMP
    unsigned long long* stamps;   // diagnostics build: [block][8] cycle sums
#endif
};

// workgroup barrier that does NOT drain the vector-memory counter (the helper waves keep two chunks of loads in
// flight across it); "memory" pins the compiler's LDS accesses on their side of it
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void tile_coords(const Params& p, unsigned t, unsigned& tm, unsigned& tn) {
    if (p.sched_S) {       // virtual id = b' * S + s: workgroup b' = 32 x + j is tile j of the s-th super-tile of XCD x
        const unsigned b = t / p.sched_S, sq = t - b * p.sched_S, x = b >> 5, j = b & 31, id = sq * 8 + x;
        const unsigned sm = id / p.sched_SN, sn = id - sm * p.sched_SN;
        tm = sm * 8 + (j >> 2);
        tn = sn * 4 + (j & 3);
        return;
    }
    // order (N-group, M, N-in-group): the tiles one XCD works on share a few x row panels and a narrow slice of W
    const unsigned full = (p.gn / NG) * NG;
    if (t < p.gm * full) {
        const unsigned grp = t / (p.gm * NG), rem = t - grp * p.gm * NG;
        tm = rem / NG;
        tn = grp * NG + rem % NG;
    } else {
        const unsigned wdt = p.gn - full, rem = t - p.gm * full;
        tm = rem / wdt;
        tn = full + rem % wdt;
    }
}

// ------------------------------------------------------------------------------------------------ matrix waves
// LDS stage image: [row][32 floats], 16-B chunk x of row R stored at chunk x ^ ((R >> 1) & 7): 16 consecutive rows read
// the same logical chunk from 16 distinct 16-B slots of the 256-B bank row (conflict-free ds_read_b128), and the 8 lanes
// that write one row cover its 128 bytes.
__device__ __forceinline__ void matrix_role(const Params& p, float* lds, int n_chunks, int wave, int lane) {
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int sw = (c >> 1) & 7;
    int a_off[4], b_off[4];                          // dword offsets of this lane's chunk (4h + q) in tile row c
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        a_off[q] = (64 * wm + c) * BK + 4 * ((4 * h + q) ^ sw);
        b_off[q] = (BM + 64 * wn + c) * BK + 4 * ((4 * h + q) ^ sw);
    }
    float* const E = lds + E_OFF_DW + (64 * wm + 4 * h) * BN + 64 * wn + c;
    f32x16 acc[2][2];
    f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
    auto rd = [&](const float* st, int q, f32x4 (&fa)[2], f32x4 (&fb)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(st + a_off[q] + 32 * i * BK);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(st + b_off[q] + 32 * j * BK);
    };
    auto mm = [&](const f32x4 (&fa)[2], const f32x4 (&fb)[2], int e) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    };
    auto mm4 = [&](const f32x4 (&fa)[2], const f32x4 (&fb)[2]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mm(fa, fb, e);
    };
    auto dump = [&]() {          // D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) E[(32 * i + (r & 3) + 8 * (r >> 2)) * BN + 32 * j] = acc[i][j][r];
    };
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero();
#ifdef MUMPY_WS_STAMP
    unsigned long long t_bar = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    ws_barrier();                                   // chunk 0 is staged
#ifdef MUMPY_WS_STAMP
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    rd(lds, 0, fa0, fb0);
    int kc = 0, stage = 0;
    // issue order, pinned: each fragment read sits behind one MFMA of the previous sub-step, so its latency is covered
    // by the 15 MFMAs (960 matrix-pipe cycles) that follow; hipcc's own order put the reads at the END of a sub-step,
    // one MFMA ahead of their first use
#define WS_INTERLEAVE()                                            \
    do {                                                           \
        _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     \
        }                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);        \
    } while (0)
    for (int i = 0; i < n_chunks; ++i) {
        const float* st = lds + stage * STAGE_DW;
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        if (kc == 0 && i > 0) { dump(); zero(); }
        __builtin_amdgcn_sched_barrier(0);
        rd(st, 1, fa1, fb1);
        mm4(fa0, fb0);
        WS_INTERLEAVE();
        rd(st, 2, fa0, fb0);
        mm4(fa1, fb1);
        WS_INTERLEAVE();
        rd(st, 3, fa1, fb1);
        mm4(fa0, fb0);
        WS_INTERLEAVE();
        mm(fa1, fb1, 0);
        mm(fa1, fb1, 1);
        __builtin_amdgcn_sched_barrier(0);
#ifdef MUMPY_WS_STAMP
        const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
        ws_barrier();                               // chunk i+1 is staged
#ifdef MUMPY_WS_STAMP
        t_bar += __builtin_amdgcn_s_memtime() - tb0;
#endif
        __builtin_amdgcn_sched_barrier(0);
        rd(lds + stage * STAGE_DW, 0, fa0, fb0);    // (after the last chunk: a harmless read of an idle stage)
        mm(fa1, fb1, 2);
        mm(fa1, fb1, 3);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (++kc == p.nk) kc = 0;
    }
#undef WS_INTERLEAVE
#ifdef MUMPY_WS_STAMP
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (wave == 0 && lane == 0) {
        unsigned long long* o = p.stamps + 8 * blockIdx.x;
        o[0] = t_loop - t_begin; o[1] = t_end - t_loop; o[2] = t_bar; o[3] = (unsigned long long)n_chunks;
    }
#endif
    dump();
    ws_barrier();                                   // the last tile's accumulators are in LDS
}

// ------------------------------------------------------------------------------------------------ loader waves
// Waves 4-7: operand chunks by LDS-DMA (buffer_load ... lds, 16 B per lane; a wave-instruction fills 8 tile rows), three
// stages: at iteration i the DMA of chunk i+2 goes into the stage chunk i-1 vacated, and the wave then waits -- counted,
// vmcnt(8): the 8 pieces just issued stay in flight -- for chunk i+1 before the workgroup barrier.  No vector ALU work and
// no registers: beside a dense fp32 MFMA stream another wave of the SIMD gets ~1 vector instruction per 50 cycles
// (tools/micro/coissue.hip), so these waves issue nothing but the DMA itself.
// The LDS image is lane-linear per wave-instruction, so the XOR swizzle is applied to the SOURCE address: lane l of a piece
// covers tile row r0 + l/8, physical chunk l%8, and fetches logical chunk (l%8) ^ ((row >> 1) & 7).
__device__ __forceinline__ void loader_role(const Params& p, float* lds, unsigned tile0, int n_chunks, int hl) {
    const int lane = hl & 63, lw = __builtin_amdgcn_readfirstlane(hl >> 6);
    const int prow = lane >> 3;                                  // row inside a piece
    uint32_t aoff[4], boff[4];                                   // byte offsets of this lane's source chunk, per piece
    auto set_tile = [&](unsigned t) {
        unsigned tm, tn;
        tile_coords(p, t, tm, tn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 32 * q + 8 * lw + prow;                // tile row of this lane in piece q (A and B alike)
            const uint32_t ch = (uint32_t)((lane & 7) ^ ((r >> 1) & 7));
            int m = (int)tm * BM + r;
            if (m > p.M - 1) m = p.M - 1;           // rows past the edge are clamped: their products are never stored
            aoff[q] = ((uint32_t)m * (uint32_t)p.K + 4u * ch) * 4u;
            int n = (int)tn * BN + r;
            if (n > p.N - 1) n = p.N - 1;
            boff[q] = ((uint32_t)n * (uint32_t)p.K + 4u * ch) * 4u;
        }
    };
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)0x7fffffff, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)0x7fffffff, 0x00020000);
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto dma = [&](int kc, int stage) {
        if (DBG & 1) return;
        const int so = kc * 128;
        float* st = lds + stage * STAGE_DW + 8 * lw * BK;        // this wave's 8 rows of piece 0
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(st + 32 * q * BK), 16, aoff[q], so, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(st + (BM + 32 * q) * BK), 16, boff[q], so, 0, 0);
    };
    // cursor (two chunks ahead of the matrix waves).  Past the workgroup's last chunk it stays put: the DMA is
    // unconditional (the counted wait needs a fixed number of pieces per iteration), a duplicate lands in an idle stage.
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
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // chunk 0 has landed
    ws_barrier();
    int stage = 2;                                               // stage of chunk i+2
#ifdef MUMPY_WS_STAMP
    unsigned long long t_work = 0, t_wait = 0, t_prev = __builtin_amdgcn_s_memtime();
#endif
    for (int i = 0; i < n_chunks; ++i) {
        dma(ld_kc, stage);                                       // chunk i+2 -> the stage chunk i-1 was read from
        advance();
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // chunk i+1 has landed
#ifdef MUMPY_WS_STAMP
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        ws_barrier();
#ifdef MUMPY_WS_STAMP
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_work += t1 - t_prev; t_wait += t2 - t1; t_prev = t2;
#endif
    }
#ifdef MUMPY_WS_STAMP
    if (hl == 0) { p.stamps[8 * blockIdx.x + 4] = t_work; p.stamps[8 * blockIdx.x + 5] = t_wait; }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no DMA may outlive the workgroup's LDS allocation
    ws_barrier();
}

// ------------------------------------------------------------------------------------------------ epilogue waves
// Waves 8-11: y tile = act(image + bias) + residual.  During chunk 0 of the next tile (while the matrix waves dump) they
// issue the 16 residual loads + the bias load of the finished tile; during chunk 1 they pull the whole accumulator image
// into registers (16 x ds_read_b128 per lane: rows 8e + (hl >> 5), columns 4 (hl & 31) .. + 3); then P passes per chunk
// (P >= ceil(16 / (nk - 1))): bias, exact-erf GELU, residual, one 16-B store per lane -- ~60 vector instructions per
// GELU pass, which fit the issue slots the MFMA stream leaves.  The pass code is unrolled with static register indices;
// the chunks that remain of a tile only join the barrier.
template <int P>
__device__ __forceinline__ void epilogue_role(const Params& p, float* lds, unsigned tile0, int n_chunks, int hl) {
    const float* const E = lds + E_OFF_DW;
    const int e_row = hl >> 5, e_c4 = hl & 31;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)((int64_t)p.M * p.N * 4), 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.Y), 0,
                                                        p.residual ? (int)((int64_t)p.M * p.N * 4) : 0, 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.Y), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    constexpr int STEPS = PASSES / P;
    f32x4 rv[PASSES], bias4;
    uint32_t ybase = OOB;                           // byte offset of (row e_row, column 4 e_c4) of the tile in flight in y
    int yrows = 0;                                  // its valid rows
    const uint32_t row8 = 8u * (uint32_t)p.N * 4u;  // byte pitch of 8 rows of y
    auto begin_tile = [&](unsigned t) {             // the tile whose image is being dumped: fetch its residual rows + bias
        unsigned tm, tn;
        tile_coords(p, t, tm, tn);
        const int n = (int)tn * BN + 4 * e_c4;
        yrows = p.M - (int)tm * BM;
        ybase = n < p.N ? (((uint32_t)tm * BM + e_row) * (uint32_t)p.N + (uint32_t)n) * 4u : OOB;
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            const uint32_t off = (ybase != OOB && 8 * e + e_row < yrows) ? ybase + (uint32_t)e * row8 : OOB;
            rv[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, off, 0, 0));
        }
        bias4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, ybase != OOB ? (uint32_t)n * 4u : OOB, 0, 0));
    };
    auto pass = [&](int e) {                        // e is a compile-time constant at every call site
        f32x4 v = *reinterpret_cast<const f32x4*>(E + (8 * e + e_row) * BN + 4 * e_c4) + bias4;
        if (p.act == MUMPY_ACT_GELU) {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = gelu_erf(v[x]);
        }
        v += rv[e];
        const uint32_t off = (ybase != OOB && 8 * e + e_row < yrows) ? ybase + (uint32_t)e * row8 : OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, off, 0, 0);
    };
    ws_barrier();
    const int n_tiles = n_chunks / p.nk;
    for (int kc = 0; kc < p.nk; ++kc) ws_barrier();            // first tile: nothing to write out yet
    if (DBG & 2) {
        for (int i = p.nk; i < n_chunks; ++i) ws_barrier();
        ws_barrier();
        return;
    }
    for (int t = 1; t < n_tiles; ++t) {
        begin_tile(tile0 + t - 1);                  // chunk 0: the previous tile is being dumped
        ws_barrier();
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {           // chunks 1 .. STEPS: P passes each
#pragma unroll
            for (int k = 0; k < P; ++k) pass(s * P + k);
            ws_barrier();
        }
        for (int kc = 1 + STEPS; kc < p.nk; ++kc) ws_barrier();
    }
    begin_tile(tile0 + n_tiles - 1);
    ws_barrier();                                   // the last tile's accumulators are in LDS
#pragma unroll
    for (int e = 0; e < PASSES; ++e) pass(e);
}

template <int P>
__global__ __launch_bounds__(768, 3) void gemm_ws_kernel(Params p) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent schedule: workgroup b' owns the contiguous tile range [b' T / G, (b'+1) T / G); b' is the XCD-major
    // renumbering of blockIdx.x (workgroups are dealt round-robin over the 8 XCDs), so an XCD's L2 sees neighbouring tiles
    const unsigned G = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = G >> 3, r8 = G & 7;
    const unsigned b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    unsigned t0 = (unsigned)(((uint64_t)b * p.tiles) / G), t1 = (unsigned)(((uint64_t)(b + 1) * p.tiles) / G);
    if (p.sched_S) { t0 = b * p.sched_S; t1 = t0 + p.sched_S; }
    const int n_chunks = (int)(t1 - t0) * p.nk;
    if (n_chunks == 0) return;
    if (wave < 4) matrix_role(p, lds, n_chunks, wave, lane);
    else {
        // the other roles' vector instructions must not queue behind the matrix wave's MFMA stream on the shared SIMD
        // (issue is arbitrated by priority, then age, and the matrix waves are the oldest): at equal priority a helper
        // iteration took ~6,300 cycles against the 4,096 of a chunk's MFMAs, and the matrix waves waited at the barrier
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(3);
        if (wave < 8) loader_role(p, lds, t0, n_chunks, tid - 256);
        else epilogue_role<P>(p, lds, t0, n_chunks, tid - 512);
    }
}

// eligibility of a shape for this kernel (the caller falls back to the tiled kernels of gemm.hip otherwise)
inline bool eligible(int64_t M, int N, int K) {
    return K % BK == 0 && K >= 3 * BK && N % 4 == 0 && M >= 1 && M * (int64_t)K * 4 < (1ll << 31) &&
           (int64_t)N * K * 4 < (1ll << 31) && M * (int64_t)N * 4 < (1ll << 31);
}

inline int launch(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M, int N,
                  int K, int act, int num_cu, hipStream_t s, void* stamps = nullptr, int lockstep = 0) {
    Params p;
    p.X = x; p.W = W; p.bias = bias; p.residual = residual; p.Y = y;
    p.M = (int)M; p.N = N; p.K = K; p.act = act; p.nk = K / BK;
    p.gm = (unsigned)((M + BM - 1) / BM); p.gn = (unsigned)((N + BN - 1) / BN);
    p.tiles = p.gm * p.gn;
#ifdef MUMPY_WS_STAMP
    p.stamps = static_cast<unsigned long long*>(stamps);
#endif
    unsigned grid = p.tiles < (unsigned)num_cu ? p.tiles : (unsigned)num_cu;
    p.sched_S = p.sched_SN = 0;
    if (lockstep && p.gm % 8 == 0 && p.gn % 4 == 0 && (p.tiles / 32) % 8 == 0) { p.sched_SN = p.gn / 4; p.sched_S = p.tiles / 32 / 8; grid = 256; }
    const int need = (PASSES + p.nk - 2) / (p.nk - 1);
#define MUMPY_WS_LAUNCH(P_)                                                                                             \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<P_>),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                  \
            if (e != hipSuccess) { set_error("gemm_ws: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return (int)e; } \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL(gemm_ws_kernel<P_>, dim3(grid), dim3(768), LDS_BYTES, s, p);                                 \
    } while (0)
    if (need <= 1) MUMPY_WS_LAUNCH(1);
    else if (need <= 2) MUMPY_WS_LAUNCH(2);
    else if (need <= 4) MUMPY_WS_LAUNCH(4);
    else MUMPY_WS_LAUNCH(8);
#undef MUMPY_WS_LAUNCH
    return 0;
}

}  // namespace gemm_ws
}  // namespace mumpy
