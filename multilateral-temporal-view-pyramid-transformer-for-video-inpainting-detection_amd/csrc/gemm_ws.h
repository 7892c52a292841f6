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
constexpr int NSTAGE = 2;
constexpr int E_DW = BM * BN;
constexpr int PMAX = 4;                   // most epilogue passes per chunk (residual ring: 2 slots x PMAX passes x 4 KB)
constexpr int E_OFF_DW = NSTAGE * STAGE_DW;
constexpr int R_OFF_DW = E_OFF_DW + E_DW;
constexpr int LDS_BYTES = (R_OFF_DW + 2 * PMAX * 1024) * 4;
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
        stage ^= 1;
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
// Waves 4-7 issue EVERY global load of the workgroup: the operand chunks (global -> registers -> LDS, three register
// sets: the loads of chunk i+3 are issued while chunk i+1, loaded two iterations ago, is written to its stage, so two
// chunks are in flight at any time) and, riding in the same register sets, the residual rows of the epilogue passes that
// run three iterations later, which they park in a small LDS ring.  Only loads, straight-line code: hipcc's counted
// vmcnt waits are exact, and no wave ever waits for a store to be acknowledged.
template <int P>
__device__ __forceinline__ void loader_role(const Params& p, float* lds, unsigned tile0, int n_chunks, int hl) {
    const int ld_row = hl >> 3, ld_c4 = hl & 7;
    const int e_row = hl >> 5, e_c4 = hl & 31;      // the epilogue lane this lane fetches residual rows for
    uint32_t aoff[4], boff[4];                      // byte offsets of this lane's 4 + 4 staged rows
    uint32_t rbase = OOB;                           // byte offset of (row e_row, column 4 e_c4) of the PREVIOUS tile in y
    int rrows = 0;                                  // valid rows of that tile
    auto set_tile = [&](unsigned t) {
        unsigned tm, tn;
        tile_coords(p, t, tm, tn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int m = (int)tm * BM + ld_row + 32 * q;
            if (m > p.M - 1) m = p.M - 1;           // rows past the edge are clamped: their products are never stored
            aoff[q] = ((uint32_t)m * (uint32_t)p.K + 4u * ld_c4) * 4u;
            int n = (int)tn * BN + ld_row + 32 * q;
            if (n > p.N - 1) n = p.N - 1;
            boff[q] = ((uint32_t)n * (uint32_t)p.K + 4u * ld_c4) * 4u;
        }
        if (t > tile0) {
            tile_coords(p, t - 1, tm, tn);
            const int n = (int)tn * BN + 4 * e_c4;
            rrows = p.M - (int)tm * BM;
            rbase = n < p.N ? (((uint32_t)tm * BM + e_row) * (uint32_t)p.N + (uint32_t)n) * 4u : OOB;
        }
    };
    // buffer form: per-lane row offset in voffset (changes per tile), the chunk's byte offset in soffset (scalar) -- no
    // per-load address arithmetic on the vector unit, which these waves share with the MFMA stream
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)0x7fffffff, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)0x7fffffff, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.X), 0,
                                                        p.residual ? (int)((int64_t)p.M * p.N * 4) : 0, 0x00020000);
    auto gload = [&](int kc, f32x4 (&r)[8 + P]) {
        if (DBG & 1) return;
        const int so = kc * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, aoff[q], so, 0));
#pragma unroll
        for (int q = 0; q < 4; ++q) r[4 + q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, boff[q], so, 0));
        // residual rows of the passes that run while chunk kc of this tile is in the MFMAs: e = (kc - 1) P + k of the
        // previous tile (none during chunk 0, none past pass 15, none past row M: out-of-range offset, the load returns 0)
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int e = (kc - 1) * P + k, row = 8 * e + e_row;
            const uint32_t off = (kc >= 1 && e < PASSES && row < rrows && rbase != OOB) ? rbase + (uint32_t)(8 * e) * (uint32_t)p.N * 4u : OOB;
            r[8 + k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, off, 0, 0));
        }
    };
    const int st_off = ld_row * BK + 4 * (ld_c4 ^ ((ld_row >> 1) & 7));     // (row + 32 q keeps the swizzle key)
    float* const R = lds + R_OFF_DW + 4 * hl;
    auto lstore = [&](int stage, const f32x4 (&r)[8 + P]) {
        float* st = lds + stage * STAGE_DW + st_off;
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(st + 32 * q * BK) = r[q];
#pragma unroll
        for (int k = 0; k < P; ++k) *reinterpret_cast<f32x4*>(R + (stage * P + k) * 1024) = r[8 + k];
    };
    // cursor (three chunks ahead of the matrix waves).  Past the workgroup's last chunk it stays put: loads and LDS writes
    // are unconditional, a duplicate of the last chunk lands in the idle stage.
    unsigned ld_tile = tile0;
    int ld_kc = 0, ld_idx = 0;
    auto advance = [&]() {
        if (ld_idx + 1 < n_chunks) {
            ++ld_idx;
            if (++ld_kc == p.nk) { ld_kc = 0; ++ld_tile; set_tile(ld_tile); }
        }
    };
    f32x4 s0[8 + P], s1[8 + P], s2[8 + P];
    set_tile(ld_tile);
    gload(ld_kc, s0);
    advance();
    gload(ld_kc, s1);
    advance();
    gload(ld_kc, s2);
    advance();
    lstore(0, s0);
    ws_barrier();
    int stage = 1;                                  // LDS stage chunk i+1 goes to
#ifdef MUMPY_WS_STAMP
    unsigned long long t_work = 0, t_wait = 0, t_prev = __builtin_amdgcn_s_memtime();
#endif
    auto step = [&](f32x4 (&r_free)[8 + P], f32x4 (&r_next)[8 + P]) {
        // chunk i is in the MFMAs.  r_free held chunk i (in LDS since the last iteration); r_next holds chunk i+1.
        gload(ld_kc, r_free);
        advance();
        if (!(DBG & 4)) lstore(stage, r_next);
        stage ^= 1;
#ifdef MUMPY_WS_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        ws_barrier();
#ifdef MUMPY_WS_STAMP
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_work += t1 - t_prev; t_wait += t2 - t1; t_prev = t2;
#endif
    };
    int i = 0;
    for (; i + 2 < n_chunks; i += 3) {
        step(s0, s1);
        step(s1, s2);
        step(s2, s0);
    }
    if (i < n_chunks) step(s0, s1);
    if (i + 1 < n_chunks) step(s1, s2);
#ifdef MUMPY_WS_STAMP
    if (hl == 0) { p.stamps[8 * blockIdx.x + 4] = t_work; p.stamps[8 * blockIdx.x + 5] = t_wait; }
#endif
    ws_barrier();
}

// ------------------------------------------------------------------------------------------------ epilogue waves
// Waves 8-11: the previous tile's accumulator image -> y.  P = passes per chunk: the image of a tile is free during the
// nk - 1 chunks that follow its dump, so P >= ceil(16 / (nk - 1)).  Pass e = image rows 8e + (hl >> 5), columns
// 4 (hl & 31) .. + 3.  In the loop these waves touch global memory with stores only (predicated by an out-of-range
// buffer offset), so they never wait on the memory counter; the one load, the tile's bias, is issued a chunk ahead.
template <int P>
__device__ __forceinline__ void epilogue_role(const Params& p, float* lds, unsigned tile0, int n_chunks, int hl) {
    const float* const E = lds + E_OFF_DW;
    const float* const R = lds + R_OFF_DW + 4 * hl;
    const int e_row = hl >> 5, e_c4 = hl & 31;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)((int64_t)p.M * p.N * 4), 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.Y), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    unsigned cur_tile = tile0;                      // tile / chunk-in-tile of iteration i
    int kc = 0, stage = 0;
    uint32_t ybase = OOB;                           // byte offset of (row e_row, column 4 e_c4) of the previous tile in y
    int yrows = 0;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, bias_next = {0.f, 0.f, 0.f, 0.f};
    ws_barrier();
#ifdef MUMPY_WS_STAMP
    unsigned long long t_work = 0, t_wait = 0, t_prev = __builtin_amdgcn_s_memtime();
#endif
    for (int i = 0; i < n_chunks; ++i) {
        if (DBG & 2) {
        } else if (kc == 0) {
            // chunk 0 of a tile: its predecessor's image is being dumped.  Fetch that tile's bias now (used from the next
            // chunk on: the only wait on a load in this role, once per tile, a chunk after the issue).
            ybase = OOB;
            uint32_t boff = OOB;
            if (cur_tile > tile0 && !(DBG & 2)) {
                unsigned tm, tn;
                tile_coords(p, cur_tile - 1, tm, tn);
                const int n = (int)tn * BN + 4 * e_c4;
                yrows = p.M - (int)tm * BM;
                if (n < p.N) { ybase = (((uint32_t)tm * BM + e_row) * (uint32_t)p.N + (uint32_t)n) * 4u; boff = (uint32_t)n * 4u; }
            }
            bias_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, boff, 0, 0));
        } else {
            if (kc == 1) bias4 = bias_next;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int e = (kc - 1) * P + k, row = 8 * e + e_row;
                const int ec = e < PASSES ? e : PASSES - 1;      // (pass past the image: any row, the store is dropped)
                f32x4 v = *reinterpret_cast<const f32x4*>(E + (8 * ec + e_row) * BN + 4 * e_c4) + bias4;
                if (p.act == MUMPY_ACT_GELU) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = gelu_erf(v[x]);
                }
                v += *reinterpret_cast<const f32x4*>(R + (stage * P + k) * 1024);
                const uint32_t off = (e < PASSES && row < yrows && ybase != OOB) ? ybase + (uint32_t)(8 * e) * (uint32_t)p.N * 4u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, off, 0, 0);
            }
        }
#ifdef MUMPY_WS_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        ws_barrier();
#ifdef MUMPY_WS_STAMP
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        t_work += t1 - t_prev; t_wait += t2 - t1; t_prev = t2;
#endif
        stage ^= 1;
        if (++kc == p.nk) { kc = 0; ++cur_tile; }
    }
#ifdef MUMPY_WS_STAMP
    if (hl == 0) { p.stamps[8 * blockIdx.x + 6] = t_work; p.stamps[8 * blockIdx.x + 7] = t_wait; }
#endif
    ws_barrier();                                   // the last tile's accumulators are in LDS
    if (DBG & 2) return;
    {   // tail: the whole image of the last tile, nothing left to overlap with
        unsigned tm, tn;
        tile_coords(p, cur_tile - (kc == 0 ? 1u : 0u), tm, tn);
        const int n = (int)tn * BN + 4 * e_c4;
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && n < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
        for (int e = 0; e < PASSES; ++e) {
            const int r = 8 * e + e_row, m = (int)tm * BM + r;
            if (m < p.M && n < p.N) {
                f32x4 v = *reinterpret_cast<const f32x4*>(E + r * BN + 4 * e_c4) + b4;
                if (p.act == MUMPY_ACT_GELU) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = gelu_erf(v[x]);
                }
                const int64_t o = (int64_t)m * p.N + n;
                if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + o);
                *reinterpret_cast<f32x4*>(p.Y + o) = v;
            }
        }
    }
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
        if (wave < 8) loader_role<P>(p, lds, t0, n_chunks, tid - 256);
        else epilogue_role<P>(p, lds, t0, n_chunks, tid - 512);
    }
}

// eligibility of a shape for this kernel (the caller falls back to the tiled kernels of gemm.hip otherwise)
inline bool eligible(int64_t M, int N, int K) {
    return K % BK == 0 && K >= 5 * BK && N % 4 == 0 && M >= 1 && M * (int64_t)K * 4 < (1ll << 31) &&
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
    else MUMPY_WS_LAUNCH(4);
#undef MUMPY_WS_LAUNCH
    return 0;
}

}  // namespace gemm_ws
}  // namespace mumpy
