// gemm_ws.h — persistent, wave-specialised fp32 GEMM for the large dense nn.Linear shapes of the path
// (swin:46-49,142,164; blocks:27-33,57-71): y = act(x W^T + bias) + residual on v_mfma_f32_32x32x2_f32.
//
// What shaped it (all measured on MI355X, tools/micro/):
//   * the fp32 MFMA executes on the vector ALU: in ONE wave every other instruction is serialised with it -- a v_fma costs
//     ~5 cycles of the 64-cycle MFMA period, a v_exp ~8.5, a ds_read_b128 ~10-15 (samewave.hip: 64.0 / 73 / 86 / 106 / 152
//     cycles per MFMA with 0 / 1 / 4 / 8 / 16 v_fma behind each) -- there are no free issue slots to hide staging or
//     epilogue work in, which is why the one-role kernels of gemm.hip sit at ~70 % matrix-pipe occupancy;
//   * ANOTHER wave of the same SIMD does not slow a back-to-back MFMA stream at all (still 64.0 cycles per MFMA), but it
//     only gets the bubbles: ~1 vector instruction per 49 cycles, ~1 LDS read per 150, whatever its priority (coissue.hip).
// So: one 768-thread workgroup per CU, three roles, one wave of each per SIMD, looping over 128x128 output tiles:
//   * waves 0-3, "matrix": 64x64 of the tile each; nothing but ds_read_b128 fragment reads and MFMAs (per 8-deep K
//     sub-step 4 reads feed 16 MFMAs) and ONE s_barrier per 32-deep chunk, placed in the middle of the chunk's last 16
//     MFMAs.  4,350 cycles per chunk against 4,096 of pure MFMA issue.  At a tile's end they dump the 64 accumulator
//     registers to an LDS image (~300 cycles) and go on with the next tile;
//   * waves 4-7, "loader": operand chunks by LDS-DMA (buffer_load ... lds), three stages, two chunks in flight, a counted
//     vmcnt(8) before the barrier.  No vector-ALU work, no registers;
//   * waves 8-11, "epilogue": the previous tile's image -> bias, exact-erf GELU, residual, 16-B row-contiguous stores,
//     a few passes per chunk under the next tile's MFMAs (their ~60 vector instructions per GELU pass live in the
//     bubbles); residual rows and bias are fetched a chunk ahead.  Rows / columns past the matrix edge are predicated by
//     the buffer range check, not by branches.
// Schedule: persistent; workgroup b' (XCD-major renumbering) owns a contiguous run of the (tile, chunk) sequence.  Whole
// tiles, or -- "split" -- an even share of chunks each: a tile cut by a boundary is finished by the workgroup that holds
// its FIRST chunks (its last segment), the others write raw partial images to per-workgroup slabs in the caller's
// workspace and raise a flag (agent-scope release); the owner acquires, adds the slabs in chunk order (bitwise
// reproducible) and runs the epilogue.  A part is always the first segment of its workgroup, which depends on nothing, so
// the owner's wait cannot deadlock.
// Measured (one device, bare MFMA loop 137-141 TFLOP/s; the tiled kernels of gemm.hip in brackets): M=7840 N=2048 K=512
// +GELU 146 us = 113 TFLOP/s [176-180 us]; N=512 K=2048 +residual 149 us [159-165]; N=1536 113 us [132]; M=125440 N=512
// K=128 +GELU 155 us [215]; M=1960 N=3072 K=768 (1.5 rounds of tiles, split) 93 us [104].  Matrix waves alone 116-120.
// LDS: 3 stages x 256 rows x 32 dwords, XOR-swizzled 16-B chunks (98,304 B) + the 128x128 accumulator image (65,536 B)
// = 163,840 B, the whole CU.
// The same skeleton runs the decoder's convolutions (loader_role<CONV>: implicit GEMM, taps by address arithmetic) and
// config 3's bf16-storage GEMMs (IO = 1 / 2: bf16 stages -- a 128-byte tile row is 64 bf16 -- on
// v_mfma_f32_32x32x16_bf16, bf16 or fp32 output): M=7840 N=512 K=2048 24.8 us [tiled bf16 kernel 65.2], N=2048 K=512 +GELU
// 33.0 [43.4], N=512 K=512 11.6 [35.7]; with 512 matrix-pipe cycles per chunk those are bound by the epilogue waves.
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
#ifndef MUMPY_WS_DBG
#define MUMPY_WS_DBG 0      // harness diagnostics, compile time (a runtime switch would put branches around the loads): 1 = no operand loads, 2 = no epilogue, 4 = no LDS staging writes, 8 = no priority
#endif
constexpr int DBG = MUMPY_WS_DBG;
#ifndef MUMPY_WS_STORE_AUX
#define MUMPY_WS_STORE_AUX 0   // cache policy bits of the output stores (gfx950: 1 = sc0, 2 = nt, 16 = sc1)
#endif
constexpr uint32_t OOB = 0x80000000u;       // buffer offset past every buffer: the access is dropped by the bounds check
constexpr int LN_GUARD_WORD = 1022;         // sticky: a folded LayerNorm met a row with |mean| > LN_GUARD_RATIO sigma (reduced accuracy)
constexpr float LN_GUARD_RATIO = 256.f;
constexpr int STATUS_WORD = 1023;           // last word of the flag page: sticky "a split part never arrived" status (0 = fine)

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
    unsigned rr_cnt, rr_G; // whole-tile schedule: tiles per workgroup (max) and grid size; 0 = split schedule (virtual id = position)
    unsigned st_w;         // super-tile width in tiles (4, 2 or 1)
    unsigned walk;         // whole-tile schedule, 8 x 32 workgroups: an XCD walks ALONG N inside one strip of SH tile rows (see tile_coords)
    unsigned units;        // tiles * nk: the workgroups split this chunk sequence evenly (split tiles: "stream-K")
    int lmin;              // shortest allowed head part of a split tile (chunks)
    // implicit-GEMM convolution (loader_role<true>): x is an NHWC image batch, row m = output pixel (img, y, x), K index =
    // (tap, channel); all zero for a plain GEMM
    int cv_H, cv_W, cv_C, cv_kh, cv_kw, cv_cpc;             // image size, channels, taps, chunks per tap (Cin / 32)
    unsigned cv_mhw, cv_shw, cv_mw, cv_sw;                  // magic numbers: m / (H W) and rem / W as mulhi + shift
    unsigned* flags;       // [grid] arrival flags of the partial slabs (zeroed by the launcher), or null: whole tiles only
    float* slabs;          // [grid][128*128] partial accumulator images of split tiles
    // LayerNorm folded into the GEMMs either side of it (epilogue_role<.., LN>; swin:266,305, blocks:86-88):
    //   LN = 1, producer (a residual GEMM whose output is the next LayerNorm's input): the epilogue also writes, per output row
    //           and 128-column tile, {mean_t, M2_t} of the values it stores (two-pass inside the tile: no cancellation);
    //   LN = 2, consumer (y = act(LayerNorm(x) W^T + b)): X is the RAW x, W is W diag(gamma), `bias` is W beta + b, and the
    //           epilogue finishes  rstd (acc - mean * colsum) + bias  with mean / rstd combined (Chan) from the producer's partials.
    float* stats_out;      // LN = 1: [M][gn][2]
    const float* ln_stats; // LN = 2: [M][ln_gn][2] from the producer (ln_gn = its column tiles, ln_C = its N = this K)
    const float* ln_colsum;  // LN = 2: [N] sum_k W[n][k] gamma[k]
    int ln_gn, ln_C;
    float ln_eps;
    unsigned* ln_guard;    // LN = 2: sticky precision-guard word of the workspace's flag page, or null
#ifdef MUMPY_WS_STAMP
    unsigned long long* stamps;   // diagnostics build: [block][8] cycle sums
#endif
};

// workgroup barrier that does NOT drain the vector-memory counter (the helper waves keep two chunks of loads in
// flight across it); "memory" pins the compiler's LDS accesses on their side of it
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Tile sequence.  Position q -> (tm, tn): "super-tiles" of 32 tiles (SH rows x SW columns of tiles, SW = 4, 2 or 1 --
// the widest that divides the tile columns), super-tiles row-major, tiles row-major inside; the last, shorter super-row is
// compacted.  Whole-tile schedule: workgroup b' takes positions b', b' + G, b' + 2G, ... -- so in every round the 32
// workgroups that share an XCD (b' is the XCD-major renumbering of blockIdx.x) hold ONE super-tile and walk K in step: an
// x row panel is fetched into the XCD's L2 once and hit by the SW - 1 other workgroups that need it, a W panel by SH - 1.
// (With each workgroup on its own contiguous run of tiles the PMC passes showed 173 MB fetched per launch for ~20 MB of
// operands on M=7840 N=2048 K=512: every tile re-read its x panel from beyond L2 -- profiles/r02_pmc_bench_traffic.md.)
// The roles number tiles by a "virtual id" v whose chunks [v nk, (v+1) nk) are contiguous per workgroup: v = q under the
// split schedule, v = b' * rr_cnt + round under the whole-tile one.
__device__ __forceinline__ void tile_coords(const Params& p, unsigned v, unsigned& tm, unsigned& tn) {
    unsigned q = v;
    if (p.rr_cnt) {
        const unsigned bq = v / p.rr_cnt, rnd = v - bq * p.rr_cnt;
        q = bq + rnd * p.rr_G;
    }
    const unsigned SW = p.st_w, SH = 32u / SW, SN = p.gn / SW, row_tiles = SN * 32u;
    const unsigned full_rows = p.gm / SH, full = full_rows * row_tiles;
    if (p.walk && q < (full_rows >> 3) * 8u * row_tiles) {
        // groups of 8 strips (one per XCD): in round r the 32 workgroups of XCD x hold super-tile (strip 8 g + x, column group r),
        // so the strip's SH x row panels are re-used from the XCD's L2 round after round and only W streams
        const unsigned grp = 8u * row_tiles, g = q / grp, l = q - g * grp, s = l >> 5, j = l & 31u;
        tm = (g * 8u + (s & 7u)) * SH + j / SW;
        tn = (s >> 3) * SW + j % SW;
    } else if (q < full) {
        const unsigned sm = q / row_tiles, r = q - sm * row_tiles, sn = r >> 5, j = r & 31u;
        tm = sm * SH + j / SW;
        tn = sn * SW + j % SW;
    } else {
        const unsigned per = (p.gm - full_rows * SH) * SW, r = q - full, sn = r / per, j = r - sn * per;   // (per > 0 here)
        tm = full_rows * SH + j / SW;
        tn = sn * SW + j % SW;
    }
}

// Split schedule: first chunk of workgroup b (b = G: one past the end) -- an even share of the chunk sequence, moved to the
// tile boundary when it would leave a head part shorter than lmin chunks (the epilogue of the tile before it needs that many
// chunks to run under) or a tail part of one or two chunks (not worth a slab).
__device__ __host__ __forceinline__ unsigned first_chunk(unsigned b, unsigned G, unsigned units, int nk, int lmin) {
    unsigned u = (unsigned)(((uint64_t)b * units) / G);
    const unsigned r = u % (unsigned)nk;
    if (r != 0 && r < (unsigned)lmin) u -= r;
    else if (r != 0 && (unsigned)nk - r < 3u) u += (unsigned)nk - r;
    return u;
}

// ------------------------------------------------------------------------------------------------ matrix waves
// LDS stage image: [row][32 floats], 16-B chunk x of row R stored at chunk x ^ ((R >> 1) & 7): 16 consecutive rows read
// the same logical chunk from 16 distinct 16-B slots of the 256-B bank row (conflict-free ds_read_b128), and the 8 lanes
// that write one row cover its 128 bytes.
__device__ __forceinline__ void matrix_role(const Params& p, float* lds, int kc0, int n_chunks, int wave, int lane) {
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
    int kc = kc0, stage = 0;
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

// bf16 operands (config 3's storage): the same stage image -- a tile row is 128 bytes = 64 bf16, so a chunk is 64 deep --
// and v_mfma_f32_32x32x16_bf16: lane (c, h) supplies k = 8 h .. 8 h + 7 of a 16-deep step, i.e. the 16-byte chunk 2 t + h of
// its row for step t: one ds_read_b128 per operand and MFMA, 16 MFMAs (512 matrix-pipe cycles) per chunk and wave.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void matrix_role16(const Params& p, float* lds, int kc0, int n_chunks, int wave, int lane) {
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int sw = (c >> 1) & 7;
    int a_off[4], b_off[4];                          // dword offsets of this lane's chunk (2 t + h) in tile row c
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a_off[t] = (64 * wm + c) * BK + 4 * ((2 * t + h) ^ sw);
        b_off[t] = (BM + 64 * wn + c) * BK + 4 * ((2 * t + h) ^ sw);
    }
    float* const E = lds + E_OFF_DW + (64 * wm + 4 * h) * BN + 64 * wn + c;
    f32x16 acc[2][2];
    bf16x8 fa0[2], fb0[2], fa1[2], fb1[2];
    auto rd = [&](const float* st, int t, bf16x8 (&fa)[2], bf16x8 (&fb)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(st + a_off[t] + 32 * i * BK);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(st + b_off[t] + 32 * j * BK);
    };
    auto mm_row = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[2], int i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    auto dump = [&]() {
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
    ws_barrier();                                   // chunk 0 is staged
    rd(lds, 0, fa0, fb0);
    int kc = kc0, stage = 0;
    for (int i = 0; i < n_chunks; ++i) {
        const float* st = lds + stage * STAGE_DW;
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
        if (kc == 0 && i > 0) { dump(); zero(); }
        rd(st, 1, fa1, fb1);
        mm_row(fa0, fb0, 0); mm_row(fa0, fb0, 1);
        rd(st, 2, fa0, fb0);
        mm_row(fa1, fb1, 0); mm_row(fa1, fb1, 1);
        rd(st, 3, fa1, fb1);
        mm_row(fa0, fb0, 0); mm_row(fa0, fb0, 1);
        mm_row(fa1, fb1, 0);
        ws_barrier();                               // chunk i+1 is staged
        rd(lds + stage * STAGE_DW, 0, fa0, fb0);    // (after the last chunk: a harmless read of an idle stage)
        mm_row(fa1, fb1, 1);
        if (++kc == p.nk) kc = 0;
    }
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
//
// CONV (implicit GEMM, NHWC, stride 1, zero "same" padding): the A row of output pixel m for chunk (tap (r, s), channels
// c0 .. c0+31) is the 128 contiguous bytes of the input pixel (y + r - ph, x + s - pw) -- or zeros.  Per tile each lane
// decodes its four rows once (two magic-number divisions) and keeps a bit mask of the taps that fall inside the image;
// per chunk that is one bit test, one add of the chunk's (wave-uniform) byte displacement and one select of the OOB offset
// per piece: out-of-image taps are dropped by the buffer range check and the DMA writes zeros.
template <bool CONV, bool IN16 = false>
__device__ __forceinline__ void loader_role(const Params& p, float* lds, unsigned tile0, int kc0, int n_chunks, int hl) {
    constexpr uint32_t ESZ = IN16 ? 2u : 4u;                     // operand element size: a tile row is 128 bytes either way
    const int lane = hl & 63, lw = __builtin_amdgcn_readfirstlane(hl >> 6);
    const int prow = lane >> 3;                                  // row inside a piece
    uint32_t aoff[4], boff[4];                                   // byte offsets of this lane's source chunk, per piece
    uint32_t amask[4];                                           // CONV: bit (r kw + s) = tap (r, s) of this row is inside the image
    auto set_tile = [&](unsigned t) {
        unsigned tm, tn;
        tile_coords(p, t, tm, tn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 32 * q + 8 * lw + prow;                // tile row of this lane in piece q (A and B alike)
            const uint32_t ch = (uint32_t)((lane & 7) ^ ((r >> 1) & 7));
            int m = (int)tm * BM + r;
            if (m > p.M - 1) m = p.M - 1;           // rows past the edge are clamped: their products are never stored
            if (CONV) {
                const uint32_t img = __umulhi((uint32_t)m, p.cv_mhw) >> p.cv_shw;
                const uint32_t rem = (uint32_t)m - img * (uint32_t)(p.cv_H * p.cv_W);
                const uint32_t yy = __umulhi(rem, p.cv_mw) >> p.cv_sw, xx = rem - yy * (uint32_t)p.cv_W;
                uint32_t colbits = 0, mask = 0;
                for (int sx = 0; sx < p.cv_kw; ++sx)
                    colbits |= (uint32_t)((uint32_t)((int)xx + sx - (p.cv_kw >> 1)) < (uint32_t)p.cv_W) << sx;
                for (int ry = 0; ry < p.cv_kh; ++ry)
                    if ((uint32_t)((int)yy + ry - (p.cv_kh >> 1)) < (uint32_t)p.cv_H) mask |= colbits << (ry * p.cv_kw);
                amask[q] = mask;
                aoff[q] = ((uint32_t)m * (uint32_t)p.cv_C + 4u * ch) * 4u;
            } else {
                aoff[q] = (uint32_t)m * (uint32_t)p.K * ESZ + 16u * ch;
            }
            int n = (int)tn * BN + r;
            if (n > p.N - 1) n = p.N - 1;
            boff[q] = (uint32_t)n * (uint32_t)p.K * ESZ + 16u * ch;
        }
    };
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), 0, (int)0x7fffffff, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.W), 0, (int)0x7fffffff, 0x00020000);
    typedef __attribute__((address_space(3))) void* lptr_t;
    // CONV cursor state (wave-uniform): tap row / column and chunk inside the tap of the chunk the cursor points at
    int cv_r = 0, cv_s = 0, cv_c = 0;
    if (CONV) {
        const int tap0 = kc0 / p.cv_cpc;
        cv_c = kc0 - tap0 * p.cv_cpc;
        cv_r = tap0 / p.cv_kw;
        cv_s = tap0 - cv_r * p.cv_kw;
    }
    auto dma = [&](int kc, int stage) {
        if (DBG & 1) return;
        const int so = kc * 128;
        float* st = lds + stage * STAGE_DW + 8 * lw * BK;        // this wave's 8 rows of piece 0
        if (CONV) {
            const int tap = cv_r * p.cv_kw + cv_s;
            const int delta = (((cv_r - (p.cv_kh >> 1)) * p.cv_W + (cv_s - (p.cv_kw >> 1))) * p.cv_C + 32 * cv_c) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t off = ((amask[q] >> tap) & 1u) ? aoff[q] + (uint32_t)delta : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(st + 32 * q * BK), 16, off, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(st + 32 * q * BK), 16, aoff[q], so, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(st + (BM + 32 * q) * BK), 16, boff[q], so, 0, 0);
    };
    // cursor (two chunks ahead of the matrix waves).  Past the workgroup's last chunk it stays put: the DMA is
    // unconditional (the counted wait needs a fixed number of pieces per iteration), a duplicate lands in an idle stage.
    unsigned ld_tile = tile0;
    int ld_kc = kc0, ld_idx = 0;
    auto advance = [&]() {
        if (ld_idx + 1 < n_chunks) {
            ++ld_idx;
            if (++ld_kc == p.nk) {
                ld_kc = 0; ++ld_tile; set_tile(ld_tile);
                if (CONV) { cv_r = 0; cv_s = 0; cv_c = 0; }
            } else if (CONV && ++cv_c == p.cv_cpc) {
                cv_c = 0;
                if (++cv_s == p.cv_kw) { cv_s = 0; ++cv_r; }
            }
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
// OUT16: y is bf16 (config 3's activation storage; no residual then): a pass packs its four values and stores 8 bytes.
// sum over the 32 lanes of this lane's half-wave (the lanes that share an epilogue row), result in all of them: four DPP adds and
// one ds_swizzle (common.h)
__device__ __forceinline__ float half_sum32(float x) { return wave_sum(x, 32); }

template <int P, bool OUT16 = false, int LN = 0>
__device__ __forceinline__ void epilogue_role(const Params& p, float* lds, unsigned b, unsigned G, unsigned u0, unsigned u1, int hl) {
    constexpr uint32_t OSZ = OUT16 ? 2u : 4u;
    const float* const E = lds + E_OFF_DW;
    const int e_row = hl >> 5, e_c4 = hl & 31;
    const unsigned nk = (unsigned)p.nk;
    const bool split = p.flags != nullptr;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)((int64_t)p.M * p.N * OSZ), 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual ? p.residual : p.Y), 0,
                                                        p.residual ? (int)((int64_t)p.M * p.N * 4) : 0, 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias ? p.bias : p.Y), 0, p.bias ? p.N * 4 : 0, 0x00020000);
    // this workgroup's slab (a raw 128x128 image, rows of 512 B): where a part that does not own its tile goes
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(split ? p.slabs + (size_t)b * E_DW : p.Y, 0, split ? E_DW * 4 : 0, 0x00020000);
    constexpr int STEPS = PASSES / P;
    f32x4 rv[PASSES], bias4;
    uint32_t yo[PASSES];                            // byte offsets of this lane's 16 output rows in y (tile in flight)
    const uint32_t row8 = 8u * (uint32_t)p.N * OSZ; // byte pitch of 8 rows of y
    // LayerNorm folding (see Params)
    const auto rs_st = __builtin_amdgcn_make_buffer_rsrc(LN == 1 ? p.stats_out : p.Y, 0, LN == 1 ? (int)((int64_t)p.M * p.gn * 8) : 0, 0x00020000);
    const auto rs_ln = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LN == 2 ? p.ln_stats : p.Y), 0,
                                                         LN == 2 ? (int)((int64_t)p.M * p.ln_gn * 8) : 0, 0x00020000);
    const auto rs_cs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(LN == 2 ? p.ln_colsum : p.Y), 0, LN == 2 ? p.N * 4 : 0, 0x00020000);
    uint32_t st_off = 0;                            // LN = 1: byte offset of {mean_t, M2_t} of this lane's first row in this tile
    float st_inv = 0.f;                             //         1 / (valid columns of this tile)
    bool st_valid = false;                          //         this lane's four columns are inside the matrix
    float ln_mean = 0.f, ln_rstd = 0.f;             // LN = 2: statistics of row 8 (e_c4 & 15) + e_row of this tile (lane e holds row e's)
    f32x4 cs4 = {0.f, 0.f, 0.f, 0.f};               //         column sums of W gamma for this lane's four columns
    const int ln_lane4 = ((hl & 32)) * 4;           //         bpermute byte address of lane 0 of this half-wave
    // Rows past M need no predicate: their offsets are past the end of the buffer (num_records = M N 4) and the access is
    // dropped by the range check; columns past N start from the OOB offset.
    auto begin_tile = [&](unsigned t) {             // the tile whose image is being dumped: fetch its residual rows + bias
        unsigned tm, tn;
        tile_coords(p, t, tm, tn);
        const int n = (int)tn * BN + 4 * e_c4;
        const uint32_t ybase = n < p.N ? (((uint32_t)tm * BM + e_row) * (uint32_t)p.N + (uint32_t)n) * OSZ : OOB;
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            yo[e] = ybase + (uint32_t)e * row8;
            rv[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, yo[e], 0, 0));
        }
        bias4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, n < p.N ? (uint32_t)n * 4u : OOB, 0, 0));
        if (LN == 1) {
            const int nt = p.N - (int)tn * BN < BN ? p.N - (int)tn * BN : BN;
            st_inv = 1.0f / (float)nt;
            st_valid = n < p.N;
            st_off = ((((uint32_t)tm * BM + e_row) * p.gn) + tn) * 8u;
        }
        if (LN == 2) {
            cs4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cs, n < p.N ? (uint32_t)n * 4u : OOB, 0, 0));
            // lane j of a half-wave combines the partials of row 8 (j & 15) + e_row (Chan et al.: n, mean, M2 per column tile)
            const uint32_t row = (uint32_t)tm * BM + 8u * (uint32_t)(e_c4 & 15) + (uint32_t)e_row;
            const uint32_t base = row * (uint32_t)p.ln_gn * 8u;
            float msum = 0.f;
            for (int t = 0; t < p.ln_gn; ++t) {
                const float nt = (float)(p.ln_C - t * BN < BN ? p.ln_C - t * BN : BN);
                msum += nt * __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ln, base + 8u * t, 0, 0));
            }
            const float mean = msum / (float)p.ln_C;
            float m2 = 0.f;
            for (int t = 0; t < p.ln_gn; ++t) {
                const float nt = (float)(p.ln_C - t * BN < BN ? p.ln_C - t * BN : BN);
                const float mt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ln, base + 8u * t, 0, 0));
                const float qt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ln, base + 8u * t + 4u, 0, 0));
                const float d = mt - mean;
                m2 += qt + nt * d * d;
            }
            ln_mean = mean;
            ln_rstd = rsqrtf(m2 / (float)p.ln_C + p.ln_eps);
            // precision guard: the product is taken on the UN-centred x, so acc - mean colsum loses ~log2(|mean| / sigma) bits
            // (measured: 1.2e-5 of the output scale at |mean| = 30 sigma against 1e-6 for the two-launch route).  Rows beyond
            // 256 sigma (> 1e-4) raise a sticky word of the workspace; ops.check_workspaces() reports it and turns folding off.
            if (fabsf(mean) * ln_rstd > LN_GUARD_RATIO && row < (uint32_t)p.M && p.ln_guard)
                __hip_atomic_store(p.ln_guard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto pass = [&](int e) {                        // e is a compile-time constant at every call site
        f32x4 v = *reinterpret_cast<const f32x4*>(E + (8 * e + e_row) * BN + 4 * e_c4);
        if (LN == 2) {                              // LayerNorm(x) W^T = rstd (x (W gamma)^T - mean colsum) + (W beta + b)
            const float mean = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(ln_lane4 + 4 * e, __builtin_bit_cast(int, ln_mean)));
            const float rstd = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(ln_lane4 + 4 * e, __builtin_bit_cast(int, ln_rstd)));
            v = rstd * (v - mean * cs4);
        }
        v += bias4;
        if (p.act == MUMPY_ACT_GELU) {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = gelu_erf(v[x]);
        }
        v += rv[e];
        if (LN == 1) {                              // statistics of the stored values: mean over the tile's columns, then M2
            const float mean_t = half_sum32(st_valid ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f) * st_inv;
            const f32x4 d = v - mean_t;
            const float m2 = half_sum32(st_valid ? (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]) : 0.f);
            if (e_c4 == 0) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 st = {__builtin_bit_cast(uint32_t, mean_t), __builtin_bit_cast(uint32_t, m2)};
                __builtin_amdgcn_raw_buffer_store_b64(st, rs_st, st_off + (uint32_t)e * (8u * p.gn * 8u), 0, 0);
            }
        }
        if (OUT16) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};      // round to nearest even
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs_y, yo[e], 0, MUMPY_WS_STORE_AUX);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_y, yo[e], 0, MUMPY_WS_STORE_AUX);
        }
    };
    const uint32_t soff = ((uint32_t)e_row * BN + 4u * e_c4) * 4u;
    auto pass_part = [&](int e) {                   // raw partial sums -> this workgroup's slab (always in range)
        const f32x4 v = *reinterpret_cast<const f32x4*>(E + (8 * e + e_row) * BN + 4 * e_c4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_s, soff, e * (8 * BN * 4), 0);
    };
    // publish the slab: every storing wave has waited for its stores and passed a workgroup barrier before this is called
    // by ONE lane; agent-scope release (L2 write-back), then the flag (cdna guide, Guideline 16: plain payload + release
    // fence + relaxed agent flag)
    auto publish = [&]() {
        if (hl == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.flags + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    ws_barrier();
    // segments of this workgroup: [u0, end of u0's tile), whole tiles, [start of u1's tile, u1)
    unsigned prev_t = u0 / nk;                      // the segment whose image is (about to be) in LDS
    bool prev_part = (u0 % nk) != 0;                // it starts inside its tile: another workgroup owns the tile
    unsigned u = (prev_t + 1) * nk < u1 ? (prev_t + 1) * nk : u1;
    for (unsigned c = u0; c < u; ++c) ws_barrier(); // first segment: nothing to write out yet
    if (DBG & 2) {
        for (; u < u1; ++u) ws_barrier();
        ws_barrier();
        return;
    }
    // every further segment is >= 1 + STEPS chunks long (first_chunk's rule): room for the previous segment's epilogue.
    // Only the FIRST segment can be a part of a tile owned elsewhere: its write-out (raw, to the slab) is peeled.
    if (u < u1 && prev_part) {
        const unsigned end = u + nk < u1 ? u + nk : u1;
        ws_barrier();                               // chunk 0: the previous segment is being dumped
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
#pragma unroll
            for (int k = 0; k < P; ++k) pass_part(st * P + k);
            if (st == STEPS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ws_barrier();
        }
        publish();
        for (unsigned c = u + 1 + STEPS; c < end; ++c) ws_barrier();
        prev_t = u / nk;
        prev_part = false;
        u = end;
    }
    while (u < u1) {
        const unsigned end = u + nk < u1 ? u + nk : u1;
        begin_tile(prev_t);
        ws_barrier();
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {       // chunks 1 .. STEPS: P passes each
#pragma unroll
            for (int k = 0; k < P; ++k) pass(st * P + k);
            ws_barrier();
        }
        for (unsigned c = u + 1 + STEPS; c < end; ++c) ws_barrier();
        prev_t = u / nk;
        u = end;
    }
    // tail: the last segment, nothing left to overlap with
    const bool head = split && !prev_part && (u1 % nk) != 0;     // this workgroup owns a tile whose later chunks ran elsewhere
    if (!prev_part) begin_tile(prev_t);
    ws_barrier();                                   // the last segment's accumulators are in LDS
    if (prev_part) {
#pragma unroll
        for (int e = 0; e < PASSES; ++e) {
            pass_part(e);
            if (e % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // (the other roles have left: 4 waves)
        publish();
        return;
    }
    if (head) {
        // the other parts of the tile are the FIRST segments of the following workgroups (they finished long ago: a
        // first segment depends on nothing, so this wait cannot deadlock; bounded all the same).  One lane polls, then an
        // agent-scope acquire makes the slabs visible to this CU; the flags are put back to 0 for the next launch.
        const unsigned tile_end = (prev_t + 1) * nk;
        if (hl == 0) {
            for (unsigned b2 = b + 1; b2 < G; ++b2) {
                const unsigned f0 = first_chunk(b2, G, p.units, p.nk, p.lmin);
                if (f0 >= tile_end) break;
                if (first_chunk(b2 + 1, G, p.units, p.nk, p.lmin) == f0) continue;        // empty workgroup
                unsigned spins = 0;
                while (__hip_atomic_load(p.flags + b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++spins < (1u << 24))
                    __builtin_amdgcn_s_sleep(8);
                if (spins >= (1u << 24)) {
                    // the part never arrived (cannot happen while every workgroup of the launch runs: a first segment depends on
                    // nothing).  Do not pretend: raise the STICKY status word of the workspace (the host turns it into an error,
                    // mumpy_workspace_status) and leave the flag alone -- a late arrival must not be mistaken for the next launch's.
                    __hip_atomic_store(p.flags + STATUS_WORD, 1u + b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    continue;
                }
                __hip_atomic_store(p.flags + b2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // sum the parts in chunk order into the image (fixed order: bitwise reproducible)
        float* const Ew = lds + E_OFF_DW;
        for (unsigned b2 = b + 1; b2 < G; ++b2) {
            const unsigned f0 = first_chunk(b2, G, p.units, p.nk, p.lmin);
            if (f0 >= tile_end) break;
            if (first_chunk(b2 + 1, G, p.units, p.nk, p.lmin) == f0) continue;
            const float* sl = p.slabs + (size_t)b2 * E_DW + e_row * BN + 4 * e_c4;
#pragma unroll 4
            for (int e = 0; e < PASSES; ++e) {
                f32x4* d = reinterpret_cast<f32x4*>(Ew + (8 * e + e_row) * BN + 4 * e_c4);
                *d += *reinterpret_cast<const f32x4*>(sl + 8 * e * BN);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int e = 0; e < PASSES; ++e) {
        pass(e);
        if (e % 4 == 3) __builtin_amdgcn_sched_barrier(0);       // (keeps hipcc from hoisting all 16 image reads: spills)
    }
}

// IO: 0 = fp32 operands and output; 1 = bf16 operands, bf16 output; 2 = bf16 operands, fp32 output (+ fp32 residual)
template <int P, bool CONV = false, int IO = 0, int LN = 0>
__global__ __launch_bounds__(768, 3) void gemm_ws_kernel(Params p) {
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent schedule: workgroup b' owns a contiguous run of the (tile, chunk) sequence; b' is the XCD-major
    // renumbering of blockIdx.x (workgroups are dealt round-robin over the 8 XCDs), so an XCD's L2 sees neighbouring tiles
    const unsigned G = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = G >> 3, r8 = G & 7;
    const unsigned b = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const bool split = p.flags != nullptr;
    unsigned u0, u1;
    if (split) {
        u0 = first_chunk(b, G, p.units, p.nk, p.lmin);
        u1 = first_chunk(b + 1, G, p.units, p.nk, p.lmin);
    } else {                                        // whole tiles, dealt round-robin: virtual ids b rr_cnt .. + own count
        const unsigned cnt = b < p.tiles ? (p.tiles - b + G - 1) / G : 0;
        u0 = b * p.rr_cnt * (unsigned)p.nk;
        u1 = u0 + cnt * (unsigned)p.nk;
    }
    const int n_chunks = (int)(u1 - u0);
    if (n_chunks == 0) return;
    const unsigned t0 = u0 / (unsigned)p.nk;
    const int kc0 = (int)(u0 - t0 * (unsigned)p.nk);
    if (wave < 4) {
        if (IO == 0) matrix_role(p, lds, kc0, n_chunks, wave, lane);
        else matrix_role16(p, lds, kc0, n_chunks, wave, lane);
    } else {
        // (priority: no measurable effect either way beside an fp32 MFMA stream -- tools/micro/coissue.hip; kept so that
        // the few instructions of these roles are not additionally delayed by arbitration)
        if (!(DBG & 8)) __builtin_amdgcn_s_setprio(3);
        if (wave < 8) loader_role<CONV, IO != 0>(p, lds, t0, kc0, n_chunks, tid - 256);
        else epilogue_role<P, IO == 1, LN>(p, lds, b, G, u0, u1, tid - 512);
    }
}

// eligibility of a shape for this kernel (the caller falls back to the tiled kernels of gemm.hip otherwise)
inline bool eligible(int64_t M, int N, int K) {
    return K % BK == 0 && K >= 3 * BK && N % 4 == 0 && M >= 1 && M * (int64_t)K * 4 < (1ll << 31) &&
           (int64_t)N * K * 4 < (1ll << 31) && M * (int64_t)N * 4 < (1ll << 31);
}

// convolution geometry for the implicit-GEMM mode (stride 1, odd taps, zero "same" padding; x NHWC, W [Cout][kh][kw][Cin])
struct Conv { int H, W, C, kh, kw; };
inline bool conv_eligible(int64_t M, int N, const Conv& c) {
    const int64_t K = (int64_t)c.kh * c.kw * c.C;
    return c.C % BK == 0 && c.kh * c.kw <= 32 && K >= 3 * BK && K < (1 << 24) && N % 4 == 0 && M >= 1 && c.W >= 2 && c.H * (int64_t)c.W < (1 << 30) &&
           M * (int64_t)c.C * 4 < (1ll << 31) && (int64_t)N * K * 4 < (1ll << 31) && M * (int64_t)N * 4 < (1ll << 31);
}
// n / d for every n < 2^31 as mulhi(n, magic) >> shift (round-up method, N = 31 bits; d >= 2)
inline void magic_div(unsigned d, unsigned& magic, unsigned& shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    magic = (unsigned)(((1ull << (31 + s)) + d - 1) / d);
    shift = s - 1;
}

// workspace for the split schedule: arrival flags + one slab per workgroup
inline int64_t workspace_bytes(int num_cu) { return 4096 + (int64_t)num_cu * E_DW * 4; }

// LayerNorm folding (Params): exactly one of stats_out (producer) / ln_stats (consumer) is set
struct LnArgs {
    float* stats_out;
    const float* ln_stats;
    const float* ln_colsum;
    int ln_gn, ln_C;
    float ln_eps;
};

inline int launch(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M, int N,
                  int K, int act, int num_cu, hipStream_t s, void* ws = nullptr, int64_t ws_bytes = 0, int force_split = -1,
                  void* stamps = nullptr, bool ws_clean = false, const Conv* cv = nullptr, const LnArgs* ln = nullptr) {
    Params p;
    p.stats_out = ln ? ln->stats_out : nullptr;
    p.ln_stats = ln ? ln->ln_stats : nullptr;
    p.ln_colsum = ln ? ln->ln_colsum : nullptr;
    p.ln_gn = ln ? ln->ln_gn : 0; p.ln_C = ln ? ln->ln_C : 0; p.ln_eps = ln ? ln->ln_eps : 0.f;
    p.ln_guard = (ln && ln->ln_stats && ws && ws_bytes >= 4096) ? static_cast<unsigned*>(ws) + LN_GUARD_WORD : nullptr;
    const int ln_mode = !ln ? 0 : (ln->stats_out ? 1 : 2);
    p.cv_H = p.cv_W = p.cv_C = p.cv_kh = p.cv_kw = p.cv_cpc = 0;
    p.cv_mhw = p.cv_shw = p.cv_mw = p.cv_sw = 0;
    if (cv) {
        p.cv_H = cv->H; p.cv_W = cv->W; p.cv_C = cv->C; p.cv_kh = cv->kh; p.cv_kw = cv->kw; p.cv_cpc = cv->C / BK;
        magic_div((unsigned)(cv->H * cv->W), p.cv_mhw, p.cv_shw);
        magic_div((unsigned)cv->W, p.cv_mw, p.cv_sw);
    }
    p.X = x; p.W = W; p.bias = bias; p.residual = residual; p.Y = y;
    p.M = (int)M; p.N = N; p.K = K; p.act = act; p.nk = K / BK;
    p.gm = (unsigned)((M + BM - 1) / BM); p.gn = (unsigned)((N + BN - 1) / BN);
    p.tiles = p.gm * p.gn;
    p.units = p.tiles * (unsigned)p.nk;
#ifdef MUMPY_WS_STAMP
    p.stamps = static_cast<unsigned long long*>(stamps);
#endif
    const int need = (PASSES + p.nk - 2) / (p.nk - 1);
    int P = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    // split tiles over workgroups when whole tiles would leave the last round of CUs under-used (and the caller gave a
    // workspace): every workgroup then gets the same number of chunks, at the price of one slab round trip per split tile
    const double rounds = (double)p.tiles / num_cu;
    const double eff = rounds / (double)((p.tiles + num_cu - 1) / num_cu);           // CU utilisation of whole-tile rounds
    bool split = ws && ws_bytes >= workspace_bytes(num_cu) && eff < 0.93 && p.units >= 16u * (unsigned)num_cu;
    if (force_split >= 0) split = force_split && ws && ws_bytes >= workspace_bytes(num_cu);
    // a head part must leave room for the previous tile's epilogue (1 + 16 / P chunks): more passes per chunk keep that
    // short, so that the even split is not rounded away
    if (split && P < 4) P = 4;
    p.lmin = 1 + PASSES / P;
    unsigned grid = p.tiles < (unsigned)num_cu ? p.tiles : (unsigned)num_cu;
    p.st_w = (p.gn % 4 == 0) ? 4u : (p.gn % 2 == 0) ? 2u : 1u;
    p.rr_G = grid;
    p.rr_cnt = (p.tiles + grid - 1) / grid;
    p.walk = (grid == 256u && !split && tune_int("MUMPY_WS_WALK", 0)) ? 1u : 0u;
    p.flags = nullptr; p.slabs = nullptr;
    if (split) {
        p.rr_cnt = 0;
        grid = (unsigned)num_cu;
        p.flags = static_cast<unsigned*>(ws);
        p.slabs = reinterpret_cast<float*>(static_cast<char*>(ws) + 4096);
        // the owners put every flag they consume back to 0, so a workspace whose flag page was zero before a launch is zero
        // after it: a caller that keeps such a workspace (ws_clean) saves the memset node in front of every launch
        if (!ws_clean) {
            hipError_t e = hipMemsetAsync(p.flags, 0, 4096, s);
            if (e != hipSuccess) { set_error("gemm_ws: flag reset failed: %s", hipGetErrorString(e)); return (int)e; }
        }
    }
#define MUMPY_WS_LAUNCH(P_)                                                                                             \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<P_>),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                  \
            if (e != hipSuccess) { set_error("gemm_ws: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return (int)e; } \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        static bool attr_set_cv = false;                                                                                \
        if (cv && !attr_set_cv) {                                                                                       \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<P_, true>),                 \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                  \
            if (e != hipSuccess) { set_error("gemm_ws: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return (int)e; } \
            attr_set_cv = true;                                                                                         \
        }                                                                                                               \
        static bool attr_set_ln1 = false, attr_set_ln2 = false;                                                         \
        if (ln_mode == 1 && !attr_set_ln1) {                                                                            \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<P_, false, 0, 1>),          \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                  \
            if (e != hipSuccess) { set_error("gemm_ws: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return (int)e; } \
            attr_set_ln1 = true;                                                                                        \
        }                                                                                                               \
        if (ln_mode == 2 && !attr_set_ln2) {                                                                            \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<P_, false, 0, 2>),          \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                  \
            if (e != hipSuccess) { set_error("gemm_ws: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return (int)e; } \
            attr_set_ln2 = true;                                                                                        \
        }                                                                                                               \
        if (cv) hipLaunchKernelGGL((gemm_ws_kernel<P_, true>), dim3(grid), dim3(768), LDS_BYTES, s, p);                 \
        else if (ln_mode == 1) hipLaunchKernelGGL((gemm_ws_kernel<P_, false, 0, 1>), dim3(grid), dim3(768), LDS_BYTES, s, p); \
        else if (ln_mode == 2) hipLaunchKernelGGL((gemm_ws_kernel<P_, false, 0, 2>), dim3(grid), dim3(768), LDS_BYTES, s, p); \
        else hipLaunchKernelGGL((gemm_ws_kernel<P_, false>), dim3(grid), dim3(768), LDS_BYTES, s, p);                   \
    } while (0)
    if (P == 1) MUMPY_WS_LAUNCH(1);
    else if (P == 2) MUMPY_WS_LAUNCH(2);
    else if (P == 4) MUMPY_WS_LAUNCH(4);
    else MUMPY_WS_LAUNCH(8);
#undef MUMPY_WS_LAUNCH
    return 0;
}

// ---- bf16 operands (config 3's storage).  x (M,K) and W (N,K) bf16 in memory, bias / residual fp32, y bf16 (no residual)
// or fp32.  Whole tiles only (these launches are short: no slab round trip).  A chunk is 64 deep.
inline bool eligible16(int64_t M, int N, int K) {
    return K % 64 == 0 && K >= 192 && N % 4 == 0 && M >= 1 && M * (int64_t)K * 2 < (1ll << 31) && (int64_t)N * K * 2 < (1ll << 31) &&
           M * (int64_t)N * 4 < (1ll << 31);
}

inline int launch16(const void* x16, const void* W16, const float* bias, const float* residual, void* y, int64_t M, int N, int K,
                    int act, bool out_bf16, int num_cu, hipStream_t s) {
    Params p;
    p.cv_H = p.cv_W = p.cv_C = p.cv_kh = p.cv_kw = p.cv_cpc = 0;
    p.cv_mhw = p.cv_shw = p.cv_mw = p.cv_sw = 0;
    p.X = static_cast<const float*>(x16); p.W = static_cast<const float*>(W16); p.bias = bias; p.residual = residual;
    p.Y = static_cast<float*>(y);
    p.stats_out = nullptr; p.ln_stats = nullptr; p.ln_colsum = nullptr; p.ln_gn = p.ln_C = 0; p.ln_eps = 0.f; p.ln_guard = nullptr;
    p.M = (int)M; p.N = N; p.K = K; p.act = act; p.nk = K / 64;
    p.gm = (unsigned)((M + BM - 1) / BM); p.gn = (unsigned)((N + BN - 1) / BN);
    p.tiles = p.gm * p.gn;
    p.units = p.tiles * (unsigned)p.nk;
#ifdef MUMPY_WS_STAMP
    p.stamps = nullptr;
#endif
    const int need = (PASSES + p.nk - 2) / (p.nk - 1);
    const int P = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
    p.lmin = 1 + PASSES / P;
    const unsigned grid = p.tiles < (unsigned)num_cu ? p.tiles : (unsigned)num_cu;
    p.st_w = (p.gn % 4 == 0) ? 4u : (p.gn % 2 == 0) ? 2u : 1u;
    p.rr_G = grid;
    p.rr_cnt = (p.tiles + grid - 1) / grid;
    p.walk = 0u;
    p.flags = nullptr; p.slabs = nullptr;
#define MUMPY_WS_LAUNCH16(P_, IO_)                                                                                      \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<P_, false, IO_>),           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                  \
            if (e != hipSuccess) { set_error("gemm_ws: cannot reserve %d B of LDS: %s", LDS_BYTES, hipGetErrorString(e)); return (int)e; } \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((gemm_ws_kernel<P_, false, IO_>), dim3(grid), dim3(768), LDS_BYTES, s, p);                   \
    } while (0)
#define MUMPY_WS_PICK16(IO_)                                                                                            \
    do {                                                                                                                \
        if (P == 1) MUMPY_WS_LAUNCH16(1, IO_);                                                                          \
        else if (P == 2) MUMPY_WS_LAUNCH16(2, IO_);                                                                     \
        else if (P == 4) MUMPY_WS_LAUNCH16(4, IO_);                                                                     \
        else MUMPY_WS_LAUNCH16(8, IO_);                                                                                 \
    } while (0)
    if (out_bf16) MUMPY_WS_PICK16(1);
    else MUMPY_WS_PICK16(2);
#undef MUMPY_WS_PICK16
#undef MUMPY_WS_LAUNCH16
    return 0;
}

}  // namespace gemm_ws
}  // namespace mumpy
