// gemm.hip — y = act(x W^T + bias) + residual in fp32 on v_mfma_f32_32x32x2_f32 (exact f32 FMA chain).
//
// Replaces every nn.Linear / 1x1 conv on the path (swin:46-49,142,164,365; blocks:27-33,57-71; deform:333,361-362,402;
// mTVE:283,740).  Both operands are K-contiguous (x is (M,K), nn.Linear's W is (N,K)), so A and B fragments are
// read the same way: lane (r = lane&31, h = lane>>5) owns row r of a 32-row tile and, per 32-deep K chunk, the 16
// consecutive k's [16h, 16h+16) -> four ds_read_b128; k-slot h of MFMA step s is k = 16h+s for A and B alike.
//
// Tiles: 128x128x32 with 8 waves (2x4, 64x32 per wave, one block per CU) and 64x64x32 with 4 waves (2x2, 32x32 per wave,
// four blocks per CU).  LDS rows are padded to 36 dwords (16 consecutive rows hit 16 distinct 16-B slots of the 64-bank
// row: conflict-free ds_read_b128), two LDS buffers.  The main loop is software-pipelined with ONE barrier per 32-deep
// chunk: MFMAs of chunk k issue from one register fragment set while the wave reads chunk k+1's fragments from LDS into
// the other, writes chunk k+2 to LDS and issues the global loads (16 B per lane, 128-B row segments) of chunk k+3.
// The epilogue (bias, exact-erf GELU, residual add) runs on the accumulators; stores are 128-B row segments.
// Measured on MI355X (tools/kernel_micro.py, MUMPY_GEMM_DBG ablation, tools/micro/mfma_peak.hip, s_memtime/s_memrealtime
// stamps in diagnostic builds):
//   * a bare loop of this MFMA sustains 154-155 TFLOP/s at 2.38 GHz (random operands, 1-4 waves/SIMD, 1 dependent
//     accumulator is enough); inside the GEMM the chip holds ~2.05 GHz (-> a 134 TFLOP/s ceiling), the kernel with every
//     memory phase ablated runs at 125 and the full kernel at 85-117 depending on the shape.
//   * per-CU timeline of the 64x64 kernel on M=7840 N=2048 K=512 (3936 blocks): a block lives 40 us = 1.9 prologue +
//     26.8 main loop + 10.8 epilogue; on average 2.4 of the 4 resident blocks are inside their main loop and the matrix
//     pipe is ~70 % busy.  The epilogue crawls (0.6 us per predicated row store) because its neighbours' MFMA streams
//     hold the issue slots; a predicate-free interior path cut it to 4.5 us and the time reappeared in the neighbours'
//     main loops (zero-sum, not kept).  Main-loop iterations themselves are MFMA-paced (4132 cycles per chunk for 4096
//     cycles of MFMA on the SIMD).
//   * tried and not kept: persistent blocks (+3-5 % on the 128x128 tile, spills on the 128-VGPR 64x64 tile), staggered
//     first-round starts and s_setprio in the epilogue (no change), a 64x64-per-wave DMA variant with register-pipelined
//     fragments (spills at 2 waves/SIMD, slower at 1), and a from-scratch structure after the cdna guide's "pipelining
//     across barriers" (4-stage LDS-DMA ring with counted vmcnt + raw s_barrier, ping-pong wave groups, persistent,
//     next item's prologue under the epilogue): bit-correct, MFMA-only 125 us / + fragment reads 137 / + DMA 154-163 on
//     M=7840 N=512 K=2048, i.e. it TIES the kernels here on the large shapes (162 vs 159 us, 188 vs 186 us) and loses on
//     small grids.  Structurally different kernels converging on the same rate says the limiter is the clock the chip
//     holds under fp32-MFMA-plus-operand-traffic load, so the remaining lever is energy per MFMA (bytes moved per MFMA).
//   * reference point (tools/gemm_shapes.py with MUMPY_COMPARE_TORCH=1): the vendor library's fp32 kernels (hipBLASLt via
//     torch.addmm: Tensile MT256x256x32 / 128x128 macro tiles, 128x128 per wave, v_mfma_f32_16x16x4_f32, accumulators in
//     AGPRs) run the large shapes at 117-140 TFLOP/s back to back, 1.15-1.35x this kernel (18.0 vs 20.8 ms over the model's
//     shapes) -- so the 125-134 "ceiling" above is this kernel family's, not the chip's: a 128x128 wave tile reads a
//     quarter of the LDS bytes per MFMA.  Two follow-ups, both negative: (i) this template instantiated as a 256x256 block of
//     four 128x128 waves (LDS-DMA, 256 AGPRs + 256 VGPRs, one block per CU) is correct but no faster (172 us on M=7840 N=2048
//     K=512 vs 177 here and 141 in the library): at one wave per SIMD every stall is exposed and hipcc's schedule is not
//     the library's hand-placed one; (ii) swapping the library in for every large non-GELU GEMM of the forward (experiment
//     through torch.addmm under the same hipGraph) moved the whole forward from 24.32 to 24.07 ms: in situ -- cold weights,
//     co-scheduled branches, bias + residual no longer fused -- the isolated-kernel advantage does not carry over.
#include <stdlib.h>
#include "common.h"
#include "gemm_ws.h"
#include "gemm_ws64.h"
using namespace mumpy;

namespace {

constexpr int BK = 32;
constexpr int LDR = 36;  // LDS row stride in dwords

// Implicit-GEMM convolution (NHWC, stride 1, "same" zero padding): out pixel m = (b,y,x), K index = (tap, c) with
// tap = dy*kw + dx -- exactly the GEMM below with the A row address shifted by a per-chunk tap offset and a border
// predicate.  The weight is the channels_last (KRSC) image of the nn.Conv2d kernel == an (N, K) row-major matrix.
struct ConvGeom {
    int H, W, Cin, kh, kw, ph, pw;
    // plain GEMM with a SEGMENTED contraction index (rows mode only): k = (segment j, c), c < kseg, and segment j of a row lives
    // kstride floats after segment j - 1 -- the (B, T, n, C) token tensor as the (B n) x (T C) operand of a Conv3d(k = s = (T,1,1))
    // head (decoder.py:62-66) without a copy.  kseg = 0: dense rows of K floats.  kseg % 32 == 0 (a chunk stays in a segment).
    int kseg, kstride;
};

// waves per SIMD the register allocator must leave room for: 8 waves (one block) per CU for the 128x128 tile,
// 16 waves (four blocks) per CU for the 64x64 tile
// GLDS: stage tiles with global_load_lds (LDS-DMA, 16 B per lane) instead of global_load + ds_write.  A wave-instruction
// writes 1 KiB linearly = 8 tile rows x 128 B, so the LDS image is UNPADDED [row][32 floats]; bank conflicts are avoided by
// an XOR swizzle of the 16-B chunk index, chunk' = chunk ^ ((row >> 1) & 7), applied on the SOURCE address (which is
// per-lane) and again on the fragment reads (cdna guide rule 21: swizzle both sides or neither).  Out-of-image conv taps
// are fetched from a page of zeros.
__device__ __attribute__((aligned(128))) float g_zero_page[32];

// epilogue shared by the fp32 and bf16-math kernels.  D[row][col]: col = lane&31, row = (r&3) + 8*(r>>2) + 4*h.
// Addresses are a wave-uniform sub-tile base (SGPRs) plus a 32-bit element offset: row * N is a scalar multiply and the
// lane part is computed once.  (The 64-bit m * N form costs two quarter-rate v_mul_lo_u32 and a v_mad_u64 per stored
// row: 5-15 % of a short-K tile.)  ksplit > 1: raw partial sums -> slab[ks][M][N]; bias/act/residual happen in
// splitk_reduce_kernel.
// OUT16: y is bf16 (config 3's activation storage); bias / residual / partial slabs stay fp32.
template <int TM, int TN, int WM, int WN, bool OUT16 = false>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[TM][TN], int64_t m0, int n0, int wm, int wn, int c, int h,
                                           int64_t M, int N, const float* __restrict__ bias, const float* residual,
                                           float* Y, int act, int ksplit, int ks, float* slab) {
    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);
    const int64_t mw = m0 + wm_u * WM;                                   // first row of this wave's sub-tile
    const int64_t wbase = mw * N + n0 + wn_u * WN;
    const int rows_left = (int)((M - mw) < (int64_t)WM ? (M - mw) : (int64_t)WM);
    const uint32_t nb = 4u * (uint32_t)N;                                // row pitch in bytes (sub-tile extent < 4 GiB)
    const uint32_t lane_off = 4u * h * nb + 4u * c;
    if (ksplit > 1) {
        char* S = reinterpret_cast<char*>(slab + (int64_t)ks * M * N + wbase);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (n0 + wn_u * WN + 32 * j + c >= N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * i + (r & 3) + 8 * (r >> 2);
                    if (row + 4 * h < rows_left) *reinterpret_cast<float*>(S + ((uint32_t)row * nb + lane_off + 128u * j)) = acc[i][j][r];
                }
        }
        return;
    }
    char* Yw = OUT16 ? reinterpret_cast<char*>(reinterpret_cast<__bf16*>(Y) + wbase) : reinterpret_cast<char*>(Y + wbase);
    const char* Rw = residual ? reinterpret_cast<const char*>(residual + wbase) : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn_u * WN + 32 * j + c;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + (r & 3) + 8 * (r >> 2);
                if (row + 4 * h >= rows_left) continue;
                const uint32_t off = (uint32_t)row * nb + lane_off + 128u * j;
                float v = acc[i][j][r] + bv;
                if (act == MUMPY_ACT_GELU) v = gelu_erf(v);
                if (Rw) v += *reinterpret_cast<const float*>(Rw + off);
                if (OUT16) *reinterpret_cast<__bf16*>(Yw + (off >> 1)) = (__bf16)v;      // round to nearest even
                else *reinterpret_cast<float*>(Yw + off) = v;
            }
    }
}

// PIPE: double-buffer the MFMA fragments in registers (read chunk k+1 while chunk k is in the MFMAs).  PIPE = false
// (with GLDS) is the big-wave-tile variant: one fragment set, the DMA runs two chunks ahead, two barriers per chunk.
template <int BM, int BN, int WM, int WN, bool CONV, bool GLDS, bool PIPE = true>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), (BM * BN > 64 * 64) ? 2 : (CONV ? 3 : 4)) void linear_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                     const float* __restrict__ bias, const float* residual,
                                                     float* Y, int64_t M, int N, int K, int act, unsigned gn,
                                                     int ksplit, float* slab, int64_t rpb, int64_t bstride, ConvGeom cg) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = 64 * (BM / WM) * (BN / WN);   // threads per block (4 or 8 waves)
    constexpr int RPI = NT / 8;                      // tile rows staged per pass (8 lanes x 16 B per 32-float row)
    constexpr int A_LD = BM / RPI;                   // float4 loads per thread per chunk
    constexpr int B_LD = BN / RPI;
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows must divide over the staging passes");
    constexpr int LD = GLDS ? 32 : LDR;              // LDS row stride in dwords
    __shared__ __attribute__((aligned(1024))) float lds[2][(BM + BN) * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // 1-D grid, XCD-aware: blocks are dealt round-robin over the 8 XCDs, so remap the id such that each XCD owns a
    // contiguous run of the tile order (bijective form, cdna guide T1).  The tile order itself is (N-group, M, N-in-group)
    // with NG tiles per group: the blocks resident on an XCD then share a few x row-panels and ONE narrow slice of W, which
    // stay in the 4 MiB L2.  (PMC before this ordering, M=7840 N=2048 K=512: 240 MB fetched for 20 MB of operands --
    // W was re-read from the Infinity Cache for every row panel; profiles/r01_pmc_traffic.md.)
    constexpr unsigned NG = (BN >= 128) ? 4 : 8;
    const unsigned nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int ks = (int)(wgid % (unsigned)ksplit);       // K slice (split-K): slices of one tile run side by side
    wgid /= (unsigned)ksplit;
    const unsigned gm = (unsigned)((M + BM - 1) / BM);
    const unsigned full = (gn / NG) * NG;                // tiles in complete N-groups
    unsigned tm, tn;
    if (wgid < gm * full) {
        const unsigned grp = wgid / (gm * NG), rem = wgid - grp * gm * NG;
        tm = rem / NG;
        tn = grp * NG + rem % NG;
    } else {                                             // the last, narrower N-group
        const unsigned wdt = gn - full, rem = wgid - gm * full;
        tm = rem / wdt;
        tn = full + rem % wdt;
    }
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = (int)tn * BN;
    const int kbeg = ks * (K / ksplit);

    const int ld_row = tid >> 3, ld_c4 = tid & 7;
    f32x4 areg[A_LD], breg[B_LD];

    // A rows may be strided in blocks (rows m of block m / rpb start at X + (m / rpb) * bstride): lets a caller feed
    // (B, t, n, C) tokens of one time slice as an (B*n, C) operand without a copy.  Dense: rpb = M.
    // Rows past M (and W rows past N) are CLAMPED to the last valid row instead of predicated: their products land in
    // accumulator rows/columns the epilogue never stores, and the staging loads stay branch-free.
    const float* arow[A_LD];
    const float* brow[B_LD];
    int ayx[A_LD];                                               // conv: (y << 16) | x of the staged output pixel
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int64_t m = m0 + ld_row + RPI * i;
        if (m > M - 1) m = M - 1;
        if (CONV) {
            const unsigned mu = (unsigned)m, qx = mu / (unsigned)cg.W;       // M < 2^31 (checked on the host): 32-bit division
            const int x = (int)(mu - qx * (unsigned)cg.W);
            const int y = (int)(qx % (unsigned)cg.H);
            ayx[i] = (y << 16) | x;
            arow[i] = X + m * cg.Cin + 4 * ld_c4;
        } else {
            ayx[i] = 0;
            if (rpb >= M && !cg.kseg) arow[i] = X + m * K + 4 * ld_c4;       // dense
            else {
                const unsigned mu = (unsigned)m, blk = mu / (unsigned)rpb;
                arow[i] = X + (int64_t)blk * bstride + (int64_t)(mu - blk * (unsigned)rpb) * (cg.kseg ? cg.kseg : K) + 4 * ld_c4;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        int n = n0 + ld_row + RPI * i;
        if (n > N - 1) n = N - 1;
        brow[i] = Wt + (int64_t)n * K + 4 * ld_c4;
    }
    // GLDS: per-lane source chunk (swizzled) for each staged row; the LDS destination of a wave-instruction is linear
    int asw[A_LD], bsw[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) asw[i] = 4 * ((ld_c4 ^ (((ld_row + RPI * i) >> 1) & 7)) - ld_c4);
#pragma unroll
    for (int i = 0; i < B_LD; ++i) bsw[i] = 4 * ((ld_c4 ^ (((ld_row + RPI * i) >> 1) & 7)) - ld_c4);
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto gdma = [&](int k0, int buf) {                           // global -> LDS directly (GLDS)
        const int wave_row = (tid >> 6) * 8;                     // this wave's 8 rows inside a staging pass
        if (CONV) {
            const int tap = k0 / cg.Cin, c0 = k0 - tap * cg.Cin;
            const int dy = tap / cg.kw - cg.ph, dx = tap % cg.kw - cg.pw;
            const int off = (dy * cg.W + dx) * cg.Cin + c0;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int yy = (ayx[i] >> 16) + dy, xx = (ayx[i] & 0xffff) + dx;
                const bool ok = (unsigned)yy < (unsigned)cg.H && (unsigned)xx < (unsigned)cg.W;
                const float* src = ok ? arow[i] + off + asw[i] : g_zero_page + 4 * ld_c4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&lds[buf][(wave_row + RPI * i) * LD], 16, 0, 0);
            }
        } else {
            const int ka = cg.kseg ? (k0 / cg.kseg) * cg.kstride + k0 % cg.kseg : k0;     // (wave-uniform)
#pragma unroll
            for (int i = 0; i < A_LD; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(arow[i] + ka + asw[i]), (lptr_t)&lds[buf][(wave_row + RPI * i) * LD], 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(brow[i] + k0 + bsw[i]), (lptr_t)&lds[buf][(BM + wave_row + RPI * i) * LD], 16, 0, 0);
    };
    auto gload = [&](int k0) {                                   // global -> staging registers (16 B per lane)
        if (CONV) {
            const int tap = k0 / cg.Cin, c0 = k0 - tap * cg.Cin;     // wave-uniform: scalar ALU
            const int dy = tap / cg.kw - cg.ph, dx = tap % cg.kw - cg.pw;
            const int off = (dy * cg.W + dx) * cg.Cin + c0;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int yy = (ayx[i] >> 16) + dy, xx = (ayx[i] & 0xffff) + dx;
                const bool ok = (unsigned)yy < (unsigned)cg.H && (unsigned)xx < (unsigned)cg.W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(arow[i] + (ok ? off : 0));   // own pixel when outside
                areg[i] = ok ? v : f32x4{0, 0, 0, 0};
            }
        } else {
            const int ka = cg.kseg ? (k0 / cg.kseg) * cg.kstride + k0 % cg.kseg : k0;     // (wave-uniform)
#pragma unroll
            for (int i = 0; i < A_LD; ++i) areg[i] = *reinterpret_cast<const f32x4*>(arow[i] + ka);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) breg[i] = *reinterpret_cast<const f32x4*>(brow[i] + k0);
    };
    auto lstore = [&](int buf) {                                 // staging registers -> LDS tile
#pragma unroll
        for (int i = 0; i < A_LD; ++i)
            *reinterpret_cast<f32x4*>(&lds[buf][(ld_row + RPI * i) * LD + 4 * ld_c4]) = areg[i];
#pragma unroll
        for (int i = 0; i < B_LD; ++i)
            *reinterpret_cast<f32x4*>(&lds[buf][(BM + ld_row + RPI * i) * LD + 4 * ld_c4]) = breg[i];
    };
    auto fread = [&](int buf, f32x4 (&af)[TM][4], f32x4 (&bf)[TN][4]) {   // LDS -> MFMA operand fragments
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int R = wm * WM + 32 * i + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = GLDS ? 4 * ((4 * h + q) ^ ((R >> 1) & 7)) : 16 * h + 4 * q;
                af[i][q] = *reinterpret_cast<const f32x4*>(&lds[buf][R * LD + col]);
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int R = wn * WN + 32 * j + c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = GLDS ? 4 * ((4 * h + q) ^ ((R >> 1) & 7)) : 16 * h + 4 * q;
                bf[j][q] = *reinterpret_cast<const f32x4*>(&lds[buf][(BM + R) * LD + col]);
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma = [&](const f32x4 (&af)[TM][4], const f32x4 (&bf)[TN][4]) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s >> 2][s & 3], bf[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);
    };

    // Software pipeline, one barrier per chunk.  While the MFMAs of chunk k issue from fragment set A (B), the wave
    //   - reads chunk k+1's fragments from LDS into set B (A),
    //   - writes chunk k+2 (global-loaded one step ago) into the LDS buffer chunk k vacated,
    //   - issues the global loads of chunk k+3,
    // so LDS latency, the staging writes and the HBM/L2 latency all sit under MFMA issue.
    const int nk = K / ksplit / BK;
    f32x4 afA[TM][4], bfA[TN][4];
    if (GLDS && !PIPE) {
        gdma(kbeg, 0);
        if (nk > 1) gdma(kbeg + BK, 1);
        for (int kc = 0; kc < nk; ++kc) {
            __syncthreads();                                     // chunk kc has landed (hipcc drains the DMA: vmcnt(0))
            fread(kc & 1, afA, bfA);
            __syncthreads();                                     // every wave holds chunk kc in registers: buffer is free
            if (kc + 2 < nk) gdma(kbeg + (kc + 2) * BK, kc & 1);
            mma(afA, bfA);
        }
    } else if (GLDS) {
        f32x4 afB[TM][4], bfB[TN][4];
        // DMA pipeline, one barrier per chunk: while chunk k's MFMAs issue, chunk k+1's fragments are read from the other
        // LDS buffer and chunk k+2 is DMA-ed into the buffer chunk k vacated (no staging registers, no ds_write).
        // __syncthreads() drains the wave's outstanding DMA (hipcc emits vmcnt(0) for it) before the barrier.
        gdma(kbeg, 0);
        if (nk > 1) gdma(kbeg + BK, 1);
        __syncthreads();
        fread(0, afA, bfA);
        __syncthreads();                                         // everyone has chunk 0's fragments: buffer 0 is free
        if (nk > 2) gdma(kbeg + 2 * BK, 0);
        // the pair loop has no exit in the middle (an odd last chunk is peeled): with a mid-loop `break` hipcc ping-pongs
        // the accumulators between two register sets (s_nop 15 + 8 v_mov_b64 per pair of chunks)
        int kc = 0;
        for (; kc + 1 < nk; kc += 2) {
            fread(1, afB, bfB);
            mma(afA, bfA);
            __syncthreads();                                     // chunk kc+2 landed; buffer 1 readers done
            if (kc + 3 < nk) gdma(kbeg + (kc + 3) * BK, 1);
            if (kc + 2 < nk) fread(0, afA, bfA);
            mma(afB, bfB);
            __syncthreads();
            if (kc + 4 < nk) gdma(kbeg + (kc + 4) * BK, 0);
        }
        if (kc < nk) mma(afA, bfA);
    } else {
    f32x4 afB[TM][4], bfB[TN][4];
    gload(kbeg);
    lstore(0);
    if (nk > 1) gload(kbeg + BK);
    __syncthreads();
    fread(0, afA, bfA);
    if (nk > 1) lstore(1);
    if (nk > 2) gload(kbeg + 2 * BK);
    __syncthreads();
    int kc = 0;
    for (; kc + 1 < nk; kc += 2) {                                // no mid-loop exit (see the DMA variant above)
        fread(1, afB, bfB);
        mma(afA, bfA);
        if (kc + 2 < nk) lstore(0);
        if (kc + 3 < nk) gload(kbeg + (kc + 3) * BK);
        __syncthreads();
        if (kc + 2 < nk) fread(0, afA, bfA);
        mma(afB, bfB);
        if (kc + 3 < nk) lstore(1);
        if (kc + 4 < nk) gload(kbeg + (kc + 4) * BK);
        __syncthreads();
    }
    if (kc < nk) mma(afA, bfA);                                   // odd last chunk: its fragments are already in set A
    }
    store_tile<TM, TN, WM, WN>(acc, m0, n0, wm, wn, c, h, M, N, bias, residual, Y, act, ksplit, ks, slab);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 MATRIX-MATH mode (MUMPY_MATH_BF16): tensors stay fp32 in HBM; x and W are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) while they are staged into LDS, products accumulate in fp32 on v_mfma_f32_32x32x16_bf16 (16x the
// fp32 MFMA rate), bias / GELU / residual stay fp32.  This is config 3's arithmetic ("bf16 operands, fp32 accumulate")
// applied to the GEMMs and convolutions only; it turns them from MFMA-bound into staging/HBM-bound.
// LDS rows hold 32 bf16 (16 dwords) padded to 20 dwords: conflict-free ds_read_b128 (one 8-element fragment per lane
// per 16-deep MFMA step).  Same tiles / planner / split-K / implicit-GEMM addressing as the fp32 kernel.
//
// SPLIT-PRECISION mode (MUMPY_MATH_BF16X3, NP = 3): fp32 products on the bf16 matrix pipe.  Each fp32 operand is split
// while being staged into three bf16 pieces, v = p0 + p1 + p2 with p0 = bf16(v), p1 = bf16(v - p0), p2 = bf16(v - p0 - p1)
// (round-to-nearest-even; both subtractions are exact in fp32), i.e. 24+ mantissa bits are kept, and the product is
// accumulated in fp32 from the six piece products of weight >= 2^-16: a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0), small
// terms first.  The three dropped products are <= 2^-24 |a||b| each -- the size of ONE fp32 rounding -- so the result
// carries fp32-level error (measured against an fp64 product: same max / rms error as the v_mfma_f32_32x32x2_f32
// kernel, tests/test_hip_parity.py::test_linear_bf16x3_math) at 6/16 of the fp32 MFMA time.  Each piece has its own LDS
// plane [piece][row][LDH]; the wide tile keeps ONE LDS buffer (three planes x 256 rows = 60 KB, two blocks per CU)
// and refills it between two barriers while the other resident block's MFMAs run.
// Measured (MI355X, M=8192 N=2048 K=512): 107 us = 160 TFLOP/s of fp32-equivalent work (the fp32-MFMA kernel: 181 us), i.e.
// ~42 % of the bf16 matrix pipe's issue slots.  Tried without gain: (i) a software-pipelined form after the fp32 kernel
// (16-deep chunks, two LDS buffers, fragment AND staging register double-buffering, one barrier per chunk, every phase in
// one basic block so the scheduler interleaves split VALU / LDS / loads with the MFMA stream): 112 us; (ii) removing the
// W-side split arithmetic altogether (an upper bound for pre-split weights): 108 us.  Two different structures and half
// the VALU landing on one rate points at what they share -- the MFMA count at the clock the chip holds under bf16-MFMA
// load plus the LDS bytes per MFMA (three planes per operand).  Confirmed by (iii): with the split VALU kept but the piece
// stores to LDS dropped (wrong results, timing only) the same launch takes 81 us instead of 114 on the box of that run --
// the VGPR->LDS store path (6 bytes per staged element at ~80 B/clk/CU) is the largest single cost after the MFMAs.
// Splitting AFTER the LDS (raw fp32 tiles by LDS-DMA, split per wave) would remove it but doubles the split VALU
// (each tile row is consumed by two waves), which then saturates vector issue.  Dropping only the W-side piece stores
// (what pre-split weights fetched by LDS-DMA would remove) gives 110 -> 107.6 us: the cost is not store throughput but the
// split -> store -> barrier -> fragment-read chain of a chunk, which half the stores leave in place.
// (v) a WAVE-SPECIALISED form was built and measured (bit-correct on the whole bf16x3 test set, not kept): 8 waves per 128x128
// tile, one block per CU, two LDS buffers of three planes (120 KB); waves 0-3 only read fragments and issue MFMAs, waves
// 4-7 only load, split and store the other buffer, raw s_barrier hand-over per chunk (lgkmcnt-only wait, so the split
// waves' prefetch survives the barrier).  M=7840 N=512 K=2048 (248 tiles, one round): 102 us vs 117 us for the form kept here
// -- but the K=512 shapes lose (135 vs 115 us: prologue and epilogue are exposed at one block per CU; a persistent tile loop
// would be the next step).  Its ablation locates the costs: matrix waves alone, no split work and no fragment reads, 68 us
// (= 48 MFMAs x 32 cycles per chunk at the ~1.64 GHz the chip holds under this load: the floor); + fragment reads 81.5 us;
// + split work on the partner wave of each SIMD 107 us.  So once the chain is off the critical path the limiter is vector
// ISSUE: per MFMA slot the SIMD must also issue ~4 split/LDS instructions of the partner wave.  Moving the hand-over
// barrier to mid-chunk (fragment reads fully under MFMAs) and v_pk_add_f32 for the remainders both measured slower.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int LDH = 20;

// IN16: x and W are bf16 in memory (bf16 STORAGE, config 3 as written: the staging traffic halves and there is nothing to
// convert); OUT16: y is written as bf16.  Both only with NP == 1 and dense rows.
template <int BM, int BN, int WM, int WN, bool CONV, int NP = 1, int NBUF = 2, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(256, 2) void linear_bf16_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                            const float* __restrict__ bias, const float* residual,
                                                            float* Y, int64_t M, int N, int K, int act, unsigned gn,
                                                            int ksplit, float* slab, int64_t rpb, int64_t bstride,
                                                            ConvGeom cg) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    constexpr int RPI = 32, A_LD = BM / RPI, B_LD = BN / RPI;
    constexpr int PL = (BM + BN) * LDH;              // dwords per piece plane
    static_assert(NBUF * NP * PL * 4 <= 65536, "LDS tile exceeds the static limit");
    __shared__ __attribute__((aligned(16))) uint32_t lds[NBUF][NP * PL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    constexpr unsigned NG = (BN >= 128) ? 4 : 8;
    const unsigned nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int ks = (int)(wgid % (unsigned)ksplit);
    wgid /= (unsigned)ksplit;
    const unsigned gm = (unsigned)((M + BM - 1) / BM);
    const unsigned full = (gn / NG) * NG;
    unsigned tm, tn;
    if (wgid < gm * full) {
        const unsigned grp = wgid / (gm * NG), rem = wgid - grp * gm * NG;
        tm = rem / NG; tn = grp * NG + rem % NG;
    } else {
        const unsigned wdt = gn - full, rem = wgid - gm * full;
        tm = rem / wdt; tn = full + rem % wdt;
    }
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = (int)tn * BN;
    const int kbeg = ks * (K / ksplit);
    // (Tried: lanes 8..15 of every 16-lane store group on row +4 instead of row +1, which makes the ds_write_b64 piece stores
    // conflict-free at this 20-dword pitch -- SQ_LDS_BANK_CONFLICT 8.1 M -> 1.8 M cycles per launch -- with no change in the
    // kernel's duration: the LDS is not what the waves wait on.)
    const int ld_row = tid >> 3, ld_c4 = tid & 7;
    static_assert(!(IN16 || OUT16) || (NP == 1 && !CONV), "bf16 storage: plain bf16 products on dense operands only");
    f32x4 areg[A_LD], breg[B_LD];
    uint2 areg16[IN16 ? A_LD : 1], breg16[IN16 ? B_LD : 1];      // IN16: 4 bf16 per lane per row pass, as they come
    const float* arow[A_LD];
    const float* brow[B_LD];
    int ayx[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        int64_t m = m0 + ld_row + RPI * i;
        if (m > M - 1) m = M - 1;
        if (IN16) {                                                         // element offsets halve: arow counts in floats
            ayx[i] = 0;
            arow[i] = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(X) + m * K + 4 * ld_c4);
            continue;
        }
        if (CONV) {
            const unsigned mu = (unsigned)m, qx = mu / (unsigned)cg.W;       // M < 2^31 (checked on the host): 32-bit division
            ayx[i] = ((int)(qx % (unsigned)cg.H) << 16) | (int)(mu - qx * (unsigned)cg.W);
            arow[i] = X + m * cg.Cin + 4 * ld_c4;
        } else {
            ayx[i] = 0;
            if (rpb >= M && !cg.kseg) arow[i] = X + m * K + 4 * ld_c4;       // dense
            else {
                const unsigned mu = (unsigned)m, blk = mu / (unsigned)rpb;
                arow[i] = X + (int64_t)blk * bstride + (int64_t)(mu - blk * (unsigned)rpb) * (cg.kseg ? cg.kseg : K) + 4 * ld_c4;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        int n = n0 + ld_row + RPI * i;
        if (n > N - 1) n = N - 1;
        if (IN16) brow[i] = reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(Wt) + (int64_t)n * K + 4 * ld_c4);
        else brow[i] = Wt + (int64_t)n * K + 4 * ld_c4;
    }
    auto gload = [&](int k0) {
        if (IN16) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) areg16[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(arow[i]) + k0);
#pragma unroll
            for (int i = 0; i < B_LD; ++i) breg16[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(brow[i]) + k0);
            return;
        }
        if (CONV) {
            const int tap = k0 / cg.Cin, c0 = k0 - tap * cg.Cin;
            const int dy = tap / cg.kw - cg.ph, dx = tap % cg.kw - cg.pw;
            const int off = (dy * cg.W + dx) * cg.Cin + c0;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int yy = (ayx[i] >> 16) + dy, xx = (ayx[i] & 0xffff) + dx;
                const bool ok = (unsigned)yy < (unsigned)cg.H && (unsigned)xx < (unsigned)cg.W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(arow[i] + (ok ? off : 0));
                areg[i] = ok ? v : f32x4{0, 0, 0, 0};
            }
        } else {
            const int ka = cg.kseg ? (k0 / cg.kseg) * cg.kstride + k0 % cg.kseg : k0;     // (wave-uniform)
#pragma unroll
            for (int i = 0; i < A_LD; ++i) areg[i] = *reinterpret_cast<const f32x4*>(arow[i] + ka);
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) breg[i] = *reinterpret_cast<const f32x4*>(brow[i] + k0);
    };
    // piece p of the split: RNE to bf16 (v_cvt_pk_bf16_f32), the remainder (exact in fp32) goes on to the next piece
    auto lstore_row = [&](uint32_t* dst, f32x4 r) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bf16x2 lo = __builtin_convertvector(f32x2{r.x, r.y}, bf16x2), hi = __builtin_convertvector(f32x2{r.z, r.w}, bf16x2);
            const uint2 u = {*reinterpret_cast<const uint32_t*>(&lo), *reinterpret_cast<const uint32_t*>(&hi)};
            *reinterpret_cast<uint2*>(dst + p * PL) = u;
            if (p + 1 < NP)
                r -= f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u)};
        }
    };
    auto lstore = [&](int buf) {
        if (IN16) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) *reinterpret_cast<uint2*>(&lds[buf][(ld_row + RPI * i) * LDH + 2 * ld_c4]) = areg16[i];
#pragma unroll
            for (int i = 0; i < B_LD; ++i) *reinterpret_cast<uint2*>(&lds[buf][(BM + ld_row + RPI * i) * LDH + 2 * ld_c4]) = breg16[i];
            return;
        }
#pragma unroll
        for (int i = 0; i < A_LD; ++i) lstore_row(&lds[buf][(ld_row + RPI * i) * LDH + 2 * ld_c4], areg[i]);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) lstore_row(&lds[buf][(BM + ld_row + RPI * i) * LDH + 2 * ld_c4], breg[i]);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // piece products, smallest weight first: (a2 b0) (a0 b2) (a1 b1) | (a1 b0) (a0 b1) | (a0 b0)
    // NP == 2 (MUMPY_MATH_BF16X2): two pieces per operand = 16 mantissa bits, products (a1 b0) (a0 b1) (a0 b0)
    constexpr int NPROD = (NP == 3) ? 6 : (NP == 2) ? 3 : 1;
    constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
    const int nk = K / ksplit / BK;
    bf16x8 af[TM][2][NP], bf[TN][2][NP];
    auto fread = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int st = 0; st < 2; ++st)      // MFMA step st: k = 16 st + 8 h + j (A/B operand maps of 32x32x16 bf16)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    af[i][st][p] = *reinterpret_cast<const bf16x8*>(&lds[buf][p * PL + (wm * WM + 32 * i + c) * LDH + 8 * st + 4 * h]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bf[j][st][p] = *reinterpret_cast<const bf16x8*>(&lds[buf][p * PL + (BM + wn * WN + 32 * j + c) * LDH + 8 * st + 4 * h]);
    };
    auto mma = [&]() {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int q = 0; q < NPROD; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][st][NP > 1 ? PA[q] : 0],
                                                                            bf[j][st][NP > 1 ? PB[q] : 0], acc[i][j], 0, 0, 0);
    };
    const int klast = kbeg + (nk - 1) * BK;
    if (NBUF == 2) {
        // chunk k+1 is loaded under chunk k's MFMAs and split into the other LDS buffer after them (one barrier per chunk)
        gload(kbeg);
        lstore(0);
        __syncthreads();
        for (int kc = 0; kc + 1 < nk; ++kc) {
            gload(kbeg + (kc + 1) * BK);
            __builtin_amdgcn_sched_barrier(0);       // keep the loads at the top: hipcc otherwise sinks them below the MFMAs,
                                                     // next to their use in lstore, and the wave waits out their whole latency
            fread(kc & 1);
            mma();
            lstore((kc & 1) ^ 1);
            __syncthreads();
        }
        fread((nk - 1) & 1);
        mma();
    } else {
        // One LDS buffer; the staging registers run a chunk further ahead.  Per chunk: read chunk k's fragments | barrier |
        // split chunk k+1 (loaded an iteration ago) into the buffer, issue chunk k+2's loads, chunk k's MFMAs | barrier.
        // Split VALU, LDS writes and MFMAs sit in ONE basic block (the last chunk is peeled, loads past the end are
        // clamped to the last chunk instead of predicated) so they interleave: the split hides under the matrix pipe.
        gload(kbeg);
        lstore(0);
        gload(nk > 1 ? kbeg + BK : kbeg);
        __syncthreads();
        for (int kc = 0; kc + 1 < nk; ++kc) {
            fread(0);
            __syncthreads();                     // every wave holds chunk kc in registers: the buffer may be refilled
            lstore(0);
            const int knext = kbeg + (kc + 2) * BK;
            gload(knext < klast ? knext : klast);
            mma();
            __syncthreads();
        }
        fread(0);
        mma();
    }
    store_tile<TM, TN, WM, WN, OUT16>(acc, m0, n0, wm, wn, c, h, M, N, bias, residual, Y, act, ksplit, ks, slab);
}

// split-K combine: y = act(sum_s slab[s] + bias) + residual, slices summed in fixed order (bitwise reproducible)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                            const float* residual, float* Y, int64_t MN4, int N,
                                                            int ksplit, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < MN4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = ordered_sum(reinterpret_cast<const f32x4*>(slab)[i], reinterpret_cast<const f32x4*>(slab) + MN4 + i, MN4, ksplit - 1);
        const int n = (int)((i * 4) % N);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (act == MUMPY_ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (residual) v += reinterpret_cast<const f32x4*>(residual)[i];
        reinterpret_cast<f32x4*>(Y)[i] = v;
    }
}

struct Plan {
    int tile;    // 0: 128x128 (8 waves, pipelined), 1: 64x128 (tuning only), 2: 64x64 (4 waves), 3: 128x128 (4 waves, LDS-DMA)
    int ksplit;
    unsigned gn;
    int64_t gm;
};

// Shape planner, fitted to tools/gemm_shapes.py timings on MI355X.  Blocks are dispatched dynamically, so a launch costs
//   ceil(blocks / 256 CUs) x (time of one block at its residency).
// 128x128 (one block per CU): unit time, but its prologue/epilogue are exposed (~2 chunk-times on top of K/32 chunks).
// 64x64 (four co-resident per CU): 0.272 of the unit each (a quarter of the work at ~92 % efficiency: twice the staging
// traffic per FLOP), never below 0.357 (a lone block cannot fill the CU); co-resident blocks hide each other's
// prologue/epilogue.  When even the 64-tile grid leaves CUs idle and K is deep, K is split (slices >= 384).
constexpr int NUM_CU = 256;

// Split-precision (bf16x3) kernels, fitted to the same sweep (gpurun tools/gemm_shapes.py with MUMPY_MATH=bf16x3 and
// MUMPY_GEMM_FORCE): the 128x128 tile (two co-resident blocks per CU = 512 slots) wins whenever K-splitting can bring its
// grid to ~half the slots or more; below that the 64x64 tile (LDS-bound at three planes per operand) is the lesser evil.
Plan make_plan_x3(int64_t M, int N, int K, bool allow_split) {
    Plan p;
    const int64_t gm128 = (M + 127) / 128, gm64 = (M + 63) / 64;
    const unsigned gn128 = (N + 127) / 128, gn64 = (N + 63) / 64;
    const int64_t b128 = gm128 * gn128, b64 = gm64 * gn64;
    auto fit = [&](int ks) {                         // largest split <= ks with slices >= 384 deep and whole chunks
        if (!allow_split) return 1;
        if (ks > K / 384) ks = K / 384;
        if (ks > 16) ks = 16;
        while (ks > 1 && (K % (32 * ks)) != 0) --ks;
        return ks < 1 ? 1 : ks;
    };
    const int ksw = fit((int)((448 + b128 / 2) / b128));
    if (b128 * ksw >= 240) {
        p.tile = 0; p.ksplit = ksw; p.gm = gm128; p.gn = gn128;
    } else {
        p.tile = 2; p.ksplit = (b64 < 2 * 256 && K >= 768) ? fit((int)((3 * 256 + b64 - 1) / b64)) : 1;
        p.gm = gm64; p.gn = gn64;
    }
    return p;
}

Plan make_plan(int64_t M, int N, int K, bool allow_split, bool x3 = false) {
    static const char* force = tune_str("MUMPY_GEMM_FORCE");      // tuning hook: "tile,ksplit"
    if (x3 && !force) return make_plan_x3(M, N, K, allow_split);
    Plan p;
    const int64_t gm128 = (M + 127) / 128, gm64 = (M + 63) / 64;
    const unsigned gn128 = (N + 127) / 128, gn64 = (N + 63) / 64;
    const int64_t b128 = gm128 * gn128, b64 = gm64 * gn64;
    const double nk = (double)K / BK;
    const double t128 = (double)((b128 + NUM_CU - 1) / NUM_CU) * (nk + 2.0) / nk;
    double t64 = (double)((b64 + NUM_CU - 1) / NUM_CU) * 0.272;
    if (t64 < 0.357) t64 = 0.357;
    t64 *= (nk + 0.5) / nk;
    int tile = (t128 <= t64) ? 0 : 2;
    // with more than one wide tile queued per CU and a deep enough K, the 4-wave 64x64-per-wave DMA variant (two
    // co-resident blocks, lowest LDS-read ratio) measured 4-9 % faster than the 8-wave pipelined one
    if (tile == 0 && b128 > NUM_CU && K >= 512) tile = 3;
    int ks = 1;
    static const int min_slice = tune_int("MUMPY_GEMM_MINSLICE", 384), max_ks = tune_int("MUMPY_GEMM_MAXKS", 16),
                     min_k = tune_int("MUMPY_GEMM_SPLIT_MINK", 768);
    if (tile == 2 && allow_split && b64 < 2 * NUM_CU && K >= min_k) {
        ks = (int)((3 * NUM_CU + b64 - 1) / b64);
        if (ks > K / min_slice) ks = K / min_slice;
        if (ks > max_ks) ks = max_ks;
        while (ks > 1 && (K % (32 * ks)) != 0) --ks;
        if (ks < 1) ks = 1;
    }
    if (force) {
        int ft = -1, fk = -1;
        if (sscanf(force, "%d,%d", &ft, &fk) >= 1 && ft >= 0 && ft <= 3) {
            tile = ft;
            if (fk >= 1 && allow_split && K % (32 * fk) == 0) ks = fk; else if (fk >= 1) ks = 1;
        }
    }
    p.tile = tile; p.ksplit = ks;
    p.gm = (tile == 0 || tile == 3) ? gm128 : gm64;
    p.gn = (tile == 2) ? gn64 : gn128;
    return p;
}

int device_cus() {
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
        else num_cu = NUM_CU;
    }
    return num_cu;
}

// Which schedule of the persistent wave-specialised kernel (gemm_ws.h) an eligible fp32 shape takes: 0 = none (the 64x64
// persistent kernel or the tiled kernels), 1 = whole tiles, 2 = split ("stream-K").  MUMPY_GEMM_WS=0 disables it, =1 forces it
// for every eligible shape, =3 forces the split schedule wherever a workspace is given (both: tuning).  Default, fitted to
// same-device A/B runs of tools/gemm_shapes.py over the model's shapes: take it when whole 128x128 tiles fill >= 86 % of the
// rounds they need (K >= 128), or -- with a workspace -- when an even split of the chunk sequence gives every CU >= 24 chunks
// and cuts a tile into <= 3 parts; leave the deep-K shapes with two or more tiles per CU to the tiled kernels (two co-resident
// workgroups hide each other's prologue and epilogue there: 121 TFLOP/s).
int ws_plan(int64_t M, int N, int K, bool have_ws, bool conv) {
    static const int ws_mode = tune_int("MUMPY_GEMM_WS", 2);
    const int num_cu = device_cus();
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int nk = K / 32;
    const double rounds = (double)tiles / num_cu, eff = rounds / (double)((tiles + num_cu - 1) / num_cu);
    const double per_cu = (double)tiles * nk / num_cu;                 // chunks per CU under an even split
    int how = 0;
    if (ws_mode == 1) how = 1;
    else if (ws_mode == 3) how = have_ws ? 2 : 1;
    else if (ws_mode == 2 && conv) {
        // convolutions (decoder: N = 128 / 256, 196 or 784 tiles = 0.77 of the rounds they need): the even split first,
        // whole tiles down to 75 % round utilisation (profiles/r02_conv_shapes.txt; the tiled kernels sit at 74-80 TFLOP/s)
        if (have_ws && eff < 0.86 && per_cu >= 24.0 && (double)nk / per_cu <= 3.0) how = 2;
        else if (tiles >= (int64_t)(0.75 * num_cu) && eff >= 0.75 && K >= 128) how = 1;
    } else if (ws_mode == 2) {
        if (tiles >= (int64_t)(0.75 * num_cu) && eff >= 0.86 && K >= 128 && !(K >= 1024 && rounds >= 1.8)) how = 1;
        else if (have_ws && eff < 0.86 && per_cu >= 24.0 && (double)nk / per_cu <= 3.0 && !(K >= 1024 && rounds >= 1.8)) how = 2;
    }
    return how;
}

int launch_linear(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M, int N,
                  int K, int act, float* ws, int64_t ws_bytes, hipStream_t s, int64_t rpb = 0, int64_t bstride = 0,
                  const ConvGeom* conv = nullptr, bool ws_clean = false, const gemm_ws::LnArgs* ln = nullptr, int kseg = 0,
                  int kstride = 0) {
    if (rpb <= 0) { rpb = M; bstride = 0; }
    ConvGeom cg = conv ? *conv : ConvGeom{0, 0, 0, 0, 0, 0, 0, 0, 0};
    cg.kseg = kseg; cg.kstride = kstride;
    const bool math_bf16 = (act & MUMPY_MATH_BF16) != 0;
    const bool math_x2 = (act & MUMPY_MATH_BF16X2) != 0;
    const bool math_x3 = (act & MUMPY_MATH_BF16X3) != 0 || math_x2;      // the two-piece mode shares the three-piece planner
    act &= 0xff;
    // Large dense fp32 shapes: the persistent wave-specialised kernel (gemm_ws.h).  MUMPY_GEMM_WS=0 disables it, =1 forces
    // it for every eligible shape, =3 forces its split ("stream-K") schedule wherever a workspace is given (both: tuning).
    // Default, fitted to same-device A/B runs of tools/gemm_shapes.py over the model's shapes: take it when whole 128x128
    // tiles fill >= 86 % of the rounds they need (K >= 128), or -- with a workspace -- when an even split of the chunk
    // sequence gives every CU >= 24 chunks and cuts a tile into <= 3 parts; leave the deep-K shapes with two or more tiles
    // per CU to the tiled kernels (two co-resident workgroups hide each other's prologue and epilogue there: 121 TFLOP/s).
    const gemm_ws::Conv cvd{cg.H, cg.W, cg.Cin, cg.kh, cg.kw};
    if (ln && (rpb < M || math_bf16 || math_x3 || conv || !gemm_ws::eligible(M, N, K) ||
               !ws_plan(M, N, K, ws && ws_bytes >= gemm_ws::workspace_bytes(device_cus()), false))) {
        set_error("linear: LayerNorm folding needs a shape the persistent 128x128 kernel takes (mumpy_linear_ln_tiles) in fp32 mode");
        return MUMPY_EINVAL;
    }
    if (rpb >= M && !kseg && !math_bf16 && !math_x3 && (conv ? gemm_ws::conv_eligible(M, N, cvd) : gemm_ws::eligible(M, N, K))) {
        const int num_cu = device_cus();
        const bool have_ws = ws && ws_bytes >= gemm_ws::workspace_bytes(num_cu);
        const int how = ws_plan(M, N, K, have_ws, conv != nullptr);           // 0: tiled kernels, 1: whole tiles, 2: split
        const int nk = K / 32;
        // Mid-size shapes without a GELU epilogue: the same design at 64x64 tiles, two workgroups per CU (gemm_ws64.h).  Fitted
        // to same-device runs of tools/gemm_shapes.py with MUMPY_GEMM_WS64 = 0 / 1 / 2 (profiles/r02_gemm_ws64_shapes.txt): it
        // wins up to 32 chunks deep when the tiles fill at most one round of the 2 x CUs slots or at least 2.5, and for the
        // short-K shapes (<= 12 chunks) in between; deeper K wants the tiled kernels' split-K, a GELU epilogue is bound by the
        // epilogue waves (two matrix waves per SIMD leave them even fewer issue slots).  =0 disables it, =2 forces it.
        static const int ws64_mode = tune_int("MUMPY_GEMM_WS64", 1);
        if (!how && !conv && ws64_mode) {
            const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64);
            const double r64 = (double)t64 / (2.0 * num_cu);
            const bool fits = ws64_mode == 2 || (act != MUMPY_ACT_GELU && nk <= 32 && t64 >= 64 && (r64 <= 1.0 || r64 >= 2.5 || nk <= 12));
            if (fits) {
                if (int rc = gemm_ws64::launch(x, W, bias, residual, y, M, N, K, act, num_cu, s)) return rc;
                MUMPY_CHECK_LAUNCH("linear(ws64)");
                return 0;
            }
        }
        if (how) {
            if (int rc = gemm_ws::launch(x, W, bias, residual, y, M, N, K, act, num_cu, s, ws, ws_bytes, how == 2 ? 1 : 0, nullptr, ws_clean,
                                         conv ? &cvd : nullptr, ln)) return rc;
            MUMPY_CHECK_LAUNCH("linear(ws)");
            return 0;
        }
    }
    Plan p = make_plan(M, N, K, ws != nullptr, math_x3);
    // the split-K slabs start one page into the workspace: its first 4096 bytes are the persistent kernel's arrival flags,
    // which a kept workspace (mumpy_linear_wsz_fwd) promises to leave zero
    if (ws) { ws += 1024; ws_bytes -= 4096; }
    if (p.ksplit > 1 && (int64_t)p.ksplit * M * N * (int64_t)sizeof(float) > ws_bytes) p.ksplit = 1;
    const int64_t grid = p.gm * p.gn * p.ksplit;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "linear: too many tiles");
#define MUMPY_GEMM(BM_, BN_, WM_, WN_, CV_)                                                                        \
    if (use_glds)                                                                                                  \
        hipLaunchKernelGGL((linear_kernel<BM_, BN_, WM_, WN_, CV_, true>), dim3((unsigned)grid),                     \
                           dim3(64 * (BM_ / WM_) * (BN_ / WN_)), 0, s, x, W, bias, residual, y, M, N, K, act, p.gn,  \
                           p.ksplit, ws, rpb, bstride, cg);                                               \
    else                                                                                                           \
    hipLaunchKernelGGL((linear_kernel<BM_, BN_, WM_, WN_, CV_, false>), dim3((unsigned)grid),                        \
                       dim3(64 * (BM_ / WM_) * (BN_ / WN_)), 0, s, x, W, bias, residual, y, M, N, K, act, p.gn, p.ksplit, \
                       ws, rpb, bstride, cg)
    static const bool use_glds = tune_int("MUMPY_GEMM_GLDS", 0) != 0;
    if (math_x3) {
        const bool wide = (p.tile == 0 || p.tile == 3);
#define MUMPY_GEMM_X3(BM_, BN_, WM_, WN_, CV_, NB_)                                                                    \
    hipLaunchKernelGGL((linear_bf16_kernel<BM_, BN_, WM_, WN_, CV_, 3, NB_>), dim3((unsigned)grid), dim3(256), 0, s, x, W, \
                       bias, residual, y, M, N, K, act, p.gn, p.ksplit, ws, rpb, bstride, cg)
#define MUMPY_GEMM_X2(BM_, BN_, WM_, WN_, CV_, NB_)                                                                    \
    hipLaunchKernelGGL((linear_bf16_kernel<BM_, BN_, WM_, WN_, CV_, 2, NB_>), dim3((unsigned)grid), dim3(256), 0, s, x, W, \
                       bias, residual, y, M, N, K, act, p.gn, p.ksplit, ws, rpb, bstride, cg)
        if (math_x2) {
            if (wide && conv) MUMPY_GEMM_X2(128, 128, 64, 64, true, 1);
            else if (wide) MUMPY_GEMM_X2(128, 128, 64, 64, false, 1);
            else if (conv) MUMPY_GEMM_X2(64, 64, 32, 32, true, 2);
            else MUMPY_GEMM_X2(64, 64, 32, 32, false, 2);
        }
        else if (wide && conv) MUMPY_GEMM_X3(128, 128, 64, 64, true, 1);
        else if (wide) MUMPY_GEMM_X3(128, 128, 64, 64, false, 1);
        else if (conv) MUMPY_GEMM_X3(64, 64, 32, 32, true, 2);
        else MUMPY_GEMM_X3(64, 64, 32, 32, false, 2);
#undef MUMPY_GEMM_X3
#undef MUMPY_GEMM_X2
    } else if (math_bf16) {
        const bool wide = (p.tile == 0 || p.tile == 3);
#define MUMPY_GEMM_H(BM_, BN_, WM_, WN_, CV_)                                                                          \
    hipLaunchKernelGGL((linear_bf16_kernel<BM_, BN_, WM_, WN_, CV_>), dim3((unsigned)grid), dim3(256), 0, s, x, W, bias,   \
                       residual, y, M, N, K, act, p.gn, p.ksplit, ws, rpb, bstride, cg)
        if (wide && conv) MUMPY_GEMM_H(128, 128, 64, 64, true);
        else if (wide) MUMPY_GEMM_H(128, 128, 64, 64, false);
        else if (conv) MUMPY_GEMM_H(64, 64, 32, 32, true);
        else MUMPY_GEMM_H(64, 64, 32, 32, false);
#undef MUMPY_GEMM_H
    } else if (p.tile == 3) {
        if (conv)
            hipLaunchKernelGGL((linear_kernel<128, 128, 64, 64, true, true, false>), dim3((unsigned)grid), dim3(256), 0, s, x, W,
                               bias, residual, y, M, N, K, act, p.gn, p.ksplit, ws, rpb, bstride, cg);
        else
            hipLaunchKernelGGL((linear_kernel<128, 128, 64, 64, false, true, false>), dim3((unsigned)grid), dim3(256), 0, s, x, W,
                               bias, residual, y, M, N, K, act, p.gn, p.ksplit, ws, rpb, bstride, cg);
    } else if (conv) {
        if (p.tile == 0) MUMPY_GEMM(128, 128, 64, 32, true);
        else if (p.tile == 1) MUMPY_GEMM(64, 128, 32, 64, true);
        else MUMPY_GEMM(64, 64, 32, 32, true);
    } else {
        if (p.tile == 0) MUMPY_GEMM(128, 128, 64, 32, false);
        else if (p.tile == 1) MUMPY_GEMM(64, 128, 32, 64, false);
        else MUMPY_GEMM(64, 64, 32, 32, false);
    }
#undef MUMPY_GEMM
    MUMPY_CHECK_LAUNCH("linear");
    if (p.ksplit > 1) {
        const int64_t mn4 = M * N / 4;
        int64_t g = (mn4 + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, ws, bias, residual, y, mn4, N,
                           p.ksplit, act);
        MUMPY_CHECK_LAUNCH("linear(split-K reduce)");
    }
    return 0;
}

}  // namespace

static int check_linear_args(const float* x, const float* W, const float* residual, const float* y, int64_t M, int N,
                             int K, int act) {
    MUMPY_REQUIRE(x && W && y, MUMPY_ENULL, "linear: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(W) && aligned16(y) && aligned16(residual), MUMPY_EALIGN,
                  "linear: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(M < (1ll << 31) - 256 && (int64_t)N * 4 * 128 < (1ll << 32), MUMPY_ERANGE,
                  "linear: M=%lld rows / N=%d columns beyond the 32-bit row index / tile byte offsets", (long long)M, N);
    MUMPY_REQUIRE(M >= 0 && N > 0 && K > 0 && K % BK == 0 && N % 32 == 0, MUMPY_EINVAL,
                  "linear: need K %% 32 == 0 and N %% 32 == 0 (got M=%lld N=%d K=%d)", (long long)M, N, K);
    MUMPY_REQUIRE((act & 0xff) == MUMPY_ACT_NONE || (act & 0xff) == MUMPY_ACT_GELU, MUMPY_EINVAL, "linear: unknown act %d", act);
    constexpr int math_bits = MUMPY_MATH_BF16 | MUMPY_MATH_BF16X3 | MUMPY_MATH_BF16X2;
    MUMPY_REQUIRE((act & ~(0xff | math_bits)) == 0, MUMPY_EINVAL, "linear: unknown flag bits in act 0x%x", act);
    MUMPY_REQUIRE(((act & math_bits) & ((act & math_bits) - 1)) == 0, MUMPY_EINVAL, "linear: the MUMPY_MATH_* modes are exclusive");
    return 0;
}

extern "C" int mumpy_linear_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                                int64_t M, int N, int K, int act, void* stream) {
    if (M == 0) return 0;      // empty batch
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    return launch_linear(x, W, bias, residual, y, M, N, K, act, nullptr, 0, as_stream(stream));
}

extern "C" int64_t mumpy_linear_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK) return 0;
    const Plan p = make_plan(M, N, K, true), q = make_plan(M, N, K, true, true);     // either matrix-math mode
    const int ks = p.ksplit > q.ksplit ? p.ksplit : q.ksplit;
    int64_t bytes = ks > 1 ? (int64_t)ks * M * N * (int64_t)sizeof(float) + 4096 : 0;
    // the persistent kernel's split schedule (gemm_ws.h): flags + one 64-KB slab per workgroup, when the shape may take it
    if (gemm_ws::eligible(M, N, K)) {
        const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
        const double per_cu = (double)tiles * (K / 32) / NUM_CU;
        const int64_t need = gemm_ws::workspace_bytes(2 * NUM_CU);          // (room for devices with more CUs than the model)
        if (per_cu >= 16.0 && bytes < need) bytes = need;
    }
    return bytes;
}

extern "C" int mumpy_linear_ws_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                                   int64_t M, int N, int K, int act, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    if (M == 0) return 0;
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    MUMPY_REQUIRE(aligned16(workspace), MUMPY_EALIGN, "linear: workspace must be 16-byte aligned");
    return launch_linear(x, W, bias, residual, y, M, N, K, act, static_cast<float*>(workspace),
                         workspace ? workspace_bytes : 0, as_stream(stream));
}

// bf16 STORAGE (config 3 as written): x (M,K) and W (N,K) bf16, bias / residual fp32, y bf16 (out_bf16 != 0) or fp32.
extern "C" int mumpy_linear_bf16s_fwd(const void* x, const void* W, const float* bias, const float* residual, void* y,
                                      int64_t M, int N, int K, int act, int out_bf16, void* stream) {
    if (M == 0) return 0;
    MUMPY_REQUIRE(x && W && y, MUMPY_ENULL, "linear_bf16s: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(W) && aligned16(y) && aligned16(residual), MUMPY_EALIGN,
                  "linear_bf16s: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(M > 0 && M < (1ll << 31) - 256 && (int64_t)N * 4 * 128 < (1ll << 32), MUMPY_ERANGE, "linear_bf16s: M=%lld / N=%d out of range", (long long)M, N);
    MUMPY_REQUIRE(N > 0 && K > 0 && K % BK == 0 && N % 32 == 0, MUMPY_EINVAL, "linear_bf16s: need K %% 32 == 0 and N %% 32 == 0 (got N=%d K=%d)", N, K);
    MUMPY_REQUIRE(act == MUMPY_ACT_NONE || act == MUMPY_ACT_GELU, MUMPY_EINVAL, "linear_bf16s: unknown act %d", act);
    const float* xf = static_cast<const float*>(x);
    const float* wf = static_cast<const float*>(W);
    float* yf = static_cast<float*>(y);
    const ConvGeom cg{0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t s = as_stream(stream);
    MUMPY_REQUIRE(!(out_bf16 && residual), MUMPY_EINVAL, "linear_bf16s: a bf16 output takes no residual");
    // every eligible shape (K % 64 == 0, K >= 192): the persistent wave-specialised kernel with bf16 stages (gemm_ws.h).  With
    // 512 matrix-pipe cycles per chunk even a launch of 40 tiles is latency-bound, and the DMA pipeline of that kernel beats
    // the tiled kernel's load -> convert -> LDS loop everywhere (forward of config 3: 725 -> 797 clips/s against taking it
    // only for >= 0.75 of a round of tiles).  MUMPY_GEMM_WS16=0 disables it, =1 restricts it to the large shapes (A/B runs).
    {
        static const int ws16 = tune_int("MUMPY_GEMM_WS16", 2);
        static int num_cu = 0;
        if (!num_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
            else num_cu = NUM_CU;
        }
        const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
        const double rounds = (double)tiles / num_cu, eff = rounds / (double)((tiles + num_cu - 1) / num_cu);
        if (ws16 && gemm_ws::eligible16(M, N, K) && (ws16 == 2 || (tiles >= (int64_t)(0.75 * num_cu) && eff >= 0.75))) {
            if (int rc = gemm_ws::launch16(x, W, bias, residual, y, M, N, K, act, out_bf16 != 0, num_cu, s)) return rc;
            MUMPY_CHECK_LAUNCH("linear_bf16s(ws)");
            return 0;
        }
    }
    const int64_t gm128 = (M + 127) / 128, gm64 = (M + 63) / 64;
    const unsigned gn128 = (N + 127) / 128, gn64 = (N + 63) / 64;
    // the products take 1/16 of the fp32 MFMA time: these launches are staging-bound, so prefer the wide tile (half the
    // operand traffic per FLOP) as soon as it yields about a round of workgroups
    const bool wide = gm128 * gn128 >= 200;
    const unsigned gn = wide ? gn128 : gn64;
    const int64_t grid = (wide ? gm128 : gm64) * gn;
    MUMPY_REQUIRE(grid < (1ll << 31), MUMPY_ERANGE, "linear_bf16s: too many tiles");
#define MUMPY_GEMM_S(BM_, BN_, WM_, WN_, O16_)                                                                             \
    hipLaunchKernelGGL((linear_bf16_kernel<BM_, BN_, WM_, WN_, false, 1, 2, true, O16_>), dim3((unsigned)grid), dim3(256), 0, s, \
                       xf, wf, bias, residual, yf, M, N, K, act, gn, 1, nullptr, M, (int64_t)0, cg)
    if (wide && out_bf16) MUMPY_GEMM_S(128, 128, 64, 64, true);
    else if (wide) MUMPY_GEMM_S(128, 128, 64, 64, false);
    else if (out_bf16) MUMPY_GEMM_S(64, 64, 32, 32, true);
    else MUMPY_GEMM_S(64, 64, 32, 32, false);
#undef MUMPY_GEMM_S
    MUMPY_CHECK_LAUNCH("linear_bf16s");
    return 0;
}

// Same as mumpy_linear_ws_fwd for a workspace the caller KEEPS: its first 4096 bytes are zero on entry (zero it once when it is
// allocated) and the library leaves them zero on exit, which saves the flag-reset node in front of the persistent kernel's
// split schedule.  Not to be shared by launches that may overlap (one per stream).
extern "C" int mumpy_linear_wsz_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y,
                                    int64_t M, int N, int K, int act, void* workspace, int64_t workspace_bytes,
                                    void* stream) {
    if (M == 0) return 0;
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    MUMPY_REQUIRE(aligned16(workspace), MUMPY_EALIGN, "linear: workspace must be 16-byte aligned");
    return launch_linear(x, W, bias, residual, y, M, N, K, act, static_cast<float*>(workspace),
                         workspace ? workspace_bytes : 0, as_stream(stream), 0, 0, nullptr, true);
}

// Sticky status of a kept workspace (blocking: copies one word back).  0 = fine; b + 1 = the owner of a split tile gave up waiting
// for the part of workgroup b (gemm_ws.h) -- the launch's output is incomplete and the flag page may hold a stale arrival: the
// caller must discard the results and re-zero the workspace before using it again.
extern "C" int mumpy_workspace_status(const void* workspace, int* status) {
    MUMPY_REQUIRE(workspace && status, MUMPY_ENULL, "workspace_status: null pointer");
    unsigned w[2] = {0, 0};                                  // [LN_GUARD_WORD, STATUS_WORD] are neighbours
    static_assert(gemm_ws::LN_GUARD_WORD + 1 == gemm_ws::STATUS_WORD, "status words must be adjacent");
    hipError_t e = hipMemcpy(w, static_cast<const unsigned*>(workspace) + gemm_ws::LN_GUARD_WORD, sizeof(w), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("workspace_status: %s", hipGetErrorString(e)); return (int)e; }
    *status = w[1] ? (int)w[1] : (w[0] ? -1 : 0);
    return 0;
}

// ---- LayerNorm folded into the GEMMs either side of it (gemm_ws.h, epilogue_role<.., LN>) ------------------------------------
// Column tiles (ceil(N / 128)) of a shape that an fp32 launch WITH a kept workspace runs on the persistent 128x128 kernel -- the
// only kernel whose epilogue can emit / consume the per-tile row statistics -- or 0.
extern "C" int mumpy_linear_ln_tiles(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || K % BK || N % 32 || !gemm_ws::eligible(M, N, K)) return 0;
    return ws_plan(M, N, K, true, false) ? (N + 127) / 128 : 0;
}

extern "C" int mumpy_linear_lnx_fwd(const float* x, const float* W, const float* bias, const float* residual, float* y, int64_t M,
                                    int N, int K, int act, void* workspace, int64_t workspace_bytes, float* stats_out,
                                    const float* ln_stats, int ln_gn, const float* ln_colsum, float ln_eps, void* stream) {
    if (M == 0) return 0;
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    MUMPY_REQUIRE((act & ~0xff) == 0, MUMPY_EINVAL, "linear_lnx: fp32 matrix math only");
    MUMPY_REQUIRE(aligned16(workspace) && workspace, MUMPY_EALIGN, "linear_lnx: needs the kept (zeroed) workspace of mumpy_linear_wsz_fwd");
    MUMPY_REQUIRE((stats_out != nullptr) != (ln_stats != nullptr), MUMPY_EINVAL, "linear_lnx: exactly one of stats_out (producer) / ln_stats (consumer)");
    MUMPY_REQUIRE(!ln_stats || (ln_colsum && bias && !residual && ln_gn == (K + 127) / 128 && ln_gn <= 16 && ln_eps > 0.f), MUMPY_EINVAL,
                  "linear_lnx: the consumer takes W gamma, colsum, bias = W beta + b, no residual, and ln_gn = ceil(K / 128) tiles (got %d)", ln_gn);
    MUMPY_REQUIRE(aligned16(stats_out) && aligned16(ln_stats) && aligned16(ln_colsum), MUMPY_EALIGN, "linear_lnx: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(M * (int64_t)((N + 127) / 128) * 8 < (1ll << 31) && M * (int64_t)(ln_gn > 0 ? ln_gn : 1) * 8 < (1ll << 31), MUMPY_ERANGE,
                  "linear_lnx: statistics buffer beyond 2 GiB");
    const gemm_ws::LnArgs ln{stats_out, ln_stats, ln_colsum, ln_gn, K, ln_eps};
    return launch_linear(x, W, bias, residual, y, M, N, K, act, static_cast<float*>(workspace), workspace_bytes, as_stream(stream), 0, 0,
                         nullptr, true, &ln);
}

extern "C" int mumpy_linear_rows_fwd(const float* x, int64_t rows_per_block, int64_t block_stride, const float* W,
                                     const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    if (M == 0) return 0;
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    MUMPY_REQUIRE(rows_per_block > 0 && M % rows_per_block == 0 && block_stride % 4 == 0, MUMPY_EINVAL,
                  "linear_rows: M=%lld must be a multiple of rows_per_block=%lld and block_stride %% 4 == 0",
                  (long long)M, (long long)rows_per_block);
    return launch_linear(x, W, bias, residual, y, M, N, K, act, static_cast<float*>(workspace),
                         workspace ? workspace_bytes : 0, as_stream(stream), rows_per_block, block_stride);
}

// Rows mode with a segmented contraction index: row (blk, r) of the A operand is the concatenation of K / kseg segments of kseg
// floats, segment j at  x + blk * block_stride + j * kstride + r * kseg  -- e.g. the (B, T, n, C) token tensor read as the
// (B n) x (T C) operand of a Conv3d(k = s = (T,1,1)) head (decoder.py:62-66): rows_per_block = n, block_stride = T n C, kseg = C,
// kstride = n C.  One launch instead of T chained GEMMs (each re-reading and re-writing y).
extern "C" int mumpy_linear_rows_kseg_fwd(const float* x, int64_t rows_per_block, int64_t block_stride, int kseg, int kstride,
                                          const float* W, const float* bias, const float* residual, float* y, int64_t M, int N, int K,
                                          int act, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M == 0) return 0;
    if (int rc = check_linear_args(x, W, residual, y, M, N, K, act)) return rc;
    MUMPY_REQUIRE(rows_per_block > 0 && M % rows_per_block == 0 && block_stride % 4 == 0, MUMPY_EINVAL,
                  "linear_rows_kseg: M=%lld must be a multiple of rows_per_block=%lld and block_stride %% 4 == 0",
                  (long long)M, (long long)rows_per_block);
    MUMPY_REQUIRE(kseg > 0 && kseg % 32 == 0 && K % kseg == 0 && kstride % 4 == 0 && kstride >= 0, MUMPY_EINVAL,
                  "linear_rows_kseg: need kseg %% 32 == 0, K %% kseg == 0, kstride %% 4 == 0 (got kseg=%d K=%d kstride=%d)", kseg, K, kstride);
    return launch_linear(x, W, bias, residual, y, M, N, K, act, static_cast<float*>(workspace), workspace ? workspace_bytes : 0,
                         as_stream(stream), rows_per_block, block_stride, nullptr, false, nullptr, kseg, kstride);
}

extern "C" int64_t mumpy_conv2d_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    return mumpy_linear_workspace_bytes((int64_t)B * H * W, Cout, kh * kw * Cin);
}

extern "C" int mumpy_conv2d_nhwc_fwd(const float* x, const float* w_krsc, const float* bias, const float* residual,
                                     float* y, int B, int H, int W, int Cin, int Cout, int kh, int kw, int act,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    MUMPY_REQUIRE(B > 0 && H > 0 && W > 0 && H < 32768 && W < 32768, MUMPY_EINVAL, "conv2d: bad image size %dx%d", H, W);
    MUMPY_REQUIRE(kh > 0 && kw > 0 && (kh & 1) && (kw & 1), MUMPY_EINVAL, "conv2d: kernel %dx%d must be odd (same padding)", kh, kw);
    MUMPY_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, MUMPY_EINVAL, "conv2d: Cin=%d and Cout=%d must be multiples of 32", Cin, Cout);
    const int64_t M = (int64_t)B * H * W;
    const int K = kh * kw * Cin;
    if (int rc = check_linear_args(x, w_krsc, residual, y, M, Cout, K, act)) return rc;
    MUMPY_REQUIRE(aligned16(workspace), MUMPY_EALIGN, "conv2d: workspace must be 16-byte aligned");
    const ConvGeom cg{H, W, Cin, kh, kw, kh / 2, kw / 2, 0, 0};
    return launch_linear(x, w_krsc, bias, residual, y, M, Cout, K, act, static_cast<float*>(workspace),
                         workspace ? workspace_bytes : 0, as_stream(stream), 0, 0, &cg);
}
