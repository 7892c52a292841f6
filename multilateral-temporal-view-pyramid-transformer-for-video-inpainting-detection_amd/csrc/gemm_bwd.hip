// gemm_bwd.hip — the backward of y = x W^T + b (every nn.Linear of the path: swin:46-49,142,164; blocks:27-33,57-71;
// train.py:117-120 runs loss.backward() over them) from the row-major tensors AS THEY ARE:
//     dX[M,K] = dY[M,N] W[N,K]        contraction over N:  A = dY rows (contraction contiguous), B = W   ("NN")
//     dW[N,K] = dY[M,N]^T X[M,K]      contraction over M:  A = dY, B = X, both with the contraction as the SLOW index ("TN")
//     db[N]   = column sums of dY
// The first version formed W^T, dY^T and X^T with a transpose kernel (and a zero-padded copy when M % 32 != 0) and ran
// the forward GEMM on them: at config 5's micro-batch (B = 2) a training step was ~9,900 launches of mostly 3-8 us
// kernels, 1,300 of them transposes (profiles/r02_train_b2_before.md).  Here an operand whose contraction index is the
// slow one is staged [k][column] in LDS exactly as it lies in memory (16-B coalesced loads along the column) and the MFMA
// fragments are read from that image directly: lane (c, h) of v_mfma_f32_32x32x2_f32 supplies ONE value per operand, for
// row c and contraction slot h, so a ds_read_b64 at [k][2c] feeds TWO accumulator blocks whose rows are the even / odd
// rows of the wave's 64 -- the output row (column) permutation is undone by the store addresses, nothing is transposed.
// dW / db can be accumulated into (the caller's flat gradient buffer: no separate add kernels), deep contractions are
// split over workgroups with per-split slabs and a fixed-order reduce (bitwise reproducible).  fp32 MFMA, exact products.
// db costs no launch: the workgroups of the first column tile of the dW product sum the dY tiles they stage anyway.
// The same kernel with the tap as a grid dimension is the weight gradient of the decoder's convolutions
// (mumpy_conv2d_wgrad_nhwc: contraction over output pixels, the B row of pixel p for tap (r, s) is input pixel
// p + (r - ph) W + (s - pw) or zeros -- decided per staged 16-byte piece by two magic-number divisions; decoder.py:9,24-31).
#include "common.h"
using namespace mumpy;

namespace {

constexpr int BK = 32;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct XArgs {
    const float* A;      // A_T: [KK][lda >= R]   else: [R][lda >= KK]
    const float* B;      // always contraction-slow: [KK][ldb >= C]
    float* out;          // [R][ldo]
    float* slabs;        // ks > 1: [ks][R][C] partial products
    int64_t lda, ldb, ldo;
    int R, C, KK;
    int ks, chunks_per_split, nchunks;
    int accum;           // ks == 1: out += product
    int gc;              // tiles along C
    // A_T only: row sums of A over the contraction (= column sums of dY = the bias gradient), computed by the workgroups of
    // the first column tile from the tiles they stage anyway; null: not wanted
    float* rowsum;       // [R] (ks == 1) -- or, ks > 1, the partials go to rowsum_slabs [ks][R]
    float* rowsum_slabs;
    int rowsum_accum;
    // CONV (weight gradient of a stride-1 "same" convolution, NHWC): contraction index = output pixel p = (img, y, x); the B
    // row of pixel p for tap (r, s) = blockIdx.z is input pixel p + (r - ph) W + (s - pw) when that is inside the image,
    // zeros otherwise; the tap's product lands at column offset tap * C of the (Cout, kh, kw, Cin) gradient
    int cvH, cvW, cv_kw, cv_ph, cv_pw;
    unsigned cv_mhw, cv_shw, cv_mw, cv_sw;      // p / (H W) and rem / W as mulhi + shift
};

// WT: wave tile (32 or 64); the workgroup tile is 2 WT x 2 WT (4 waves).  A_T: A's contraction index is the slow one.
template <int WT, bool A_T, bool CONV = false>
__global__ __launch_bounds__(256) void xgemm_kernel(XArgs a) {
    constexpr int BT = 2 * WT, NB = WT / 32;
    constexpr int LDN = BK + 4;                  // [row][k] image: 36-dword rows (conflict-free ds_read_b128 down the rows)
    constexpr int LDT = BT + 4;                  // [k][col] image
    constexpr int A_DW = A_T ? BK * LDT : BT * LDN;
    constexpr int PIECES = BT / 32;              // 16-B pieces per thread and operand per chunk
    __shared__ __attribute__((aligned(16))) float As[A_DW];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x, z = blockIdx.y;
    const int tr = tile / a.gc, tc = tile - tr * a.gc;
    const int row0 = tr * BT, col0 = tc * BT;
    const int ch0 = z * a.chunks_per_split;
    int ch1 = ch0 + a.chunks_per_split;
    if (ch1 > a.nchunks) ch1 = a.nchunks;
    const int tap = CONV ? (int)blockIdx.z : 0;
    const int tap_r = CONV ? tap / a.cv_kw : 0, tap_s = CONV ? tap - tap_r * a.cv_kw : 0;
    const int dr = tap_r - a.cv_ph, dc = tap_s - a.cv_pw;
    const int64_t tap_shift = CONV ? (int64_t)dr * a.cvW + dc : 0;        // pixel displacement of the tap

    // global -> register staging runs TWO chunks ahead of the MFMAs (two register sets, 4-8 f32x4 each): with 16-64 MFMAs per
    // chunk and wave, one chunk of lead does not cover a memory round trip on the short launches
    f32x4 pa0[PIECES], pb0[PIECES], pa1[PIECES], pb1[PIECES];
    auto fetch = [&](int ch, f32x4 (&pa)[PIECES], f32x4 (&pb)[PIECES]) {
        const int k0 = ch * BK;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = tid + 256 * i;
            if (A_T) {
                const int kr = p / (BT / 4), cq = p - kr * (BT / 4);
                const int k = k0 + kr, r = row0 + 4 * cq;
                pa[i] = (k < a.KK && r < a.R) ? *reinterpret_cast<const f32x4*>(a.A + (int64_t)k * a.lda + r) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                int r = row0 + (p >> 3);
                if (r > a.R - 1) r = a.R - 1;            // rows past the edge: clamped, their products are never stored
                pa[i] = *reinterpret_cast<const f32x4*>(a.A + (int64_t)r * a.lda + k0 + 4 * (p & 7));
            }
            const int kr = p / (BT / 4), cq = p - kr * (BT / 4);
            const int k = k0 + kr, cc = col0 + 4 * cq;
            bool ok = k < a.KK && cc < a.C;
            if (CONV) {
                const unsigned img = __umulhi((unsigned)k, a.cv_mhw) >> a.cv_shw, rem = (unsigned)k - img * (unsigned)(a.cvH * a.cvW);
                const unsigned yy = __umulhi(rem, a.cv_mw) >> a.cv_sw, xx = rem - yy * (unsigned)a.cvW;
                ok = ok && (unsigned)((int)yy + dr) < (unsigned)a.cvH && (unsigned)((int)xx + dc) < (unsigned)a.cvW;
            }
            pb[i] = ok ? *reinterpret_cast<const f32x4*>(a.B + ((int64_t)k + tap_shift) * a.ldb + cc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](const f32x4 (&pa)[PIECES], const f32x4 (&pb)[PIECES]) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = tid + 256 * i;
            const int kr = p / (BT / 4), cq = p - kr * (BT / 4);
            if (A_T) *reinterpret_cast<f32x4*>(&As[kr * LDT + 4 * cq]) = pa[i];
            else *reinterpret_cast<f32x4*>(&As[(p >> 3) * LDN + 4 * (p & 7)]) = pa[i];
            *reinterpret_cast<f32x4*>(&Bs[kr * LDT + 4 * cq]) = pb[i];
        }
    };

    f32x16 acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this lane's contraction slots of a chunk: k = 16 h + t, t = 0..15 (the same pairing for A and B)
    const float* const a_n = As + (wm * WT + c) * LDN + 16 * h;                       // + 32 i LDN + 4 q
    const float* const a_t = As + 16 * h * LDT + wm * WT + (NB == 2 ? 2 * c : c);     // + t LDT
    const float* const b_t = Bs + 16 * h * LDT + wn * WT + (NB == 2 ? 2 * c : c);

    const bool want_rowsum = A_T && a.rowsum != nullptr && tc == 0 && tap == 0;
    float rowsum = 0.f;
    auto compute = [&]() {
        if (A_T && want_rowsum && tid < BT) {
#pragma unroll 8
            for (int k = 0; k < BK; ++k) rowsum += As[k * LDT + tid];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 fa[NB];
            if (!A_T) {
#pragma unroll
                for (int i = 0; i < NB; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a_n + 32 * i * LDN + 4 * q);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = 4 * q + e;
                float av[NB], bv[NB];
                if (A_T) {
                    if (NB == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(a_t + t * LDT); av[0] = v.x; av[NB - 1] = v.y; }
                    else av[0] = a_t[t * LDT];
                } else {
#pragma unroll
                    for (int i = 0; i < NB; ++i) av[i] = fa[i][e];
                }
                if (NB == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(b_t + t * LDT); bv[0] = v.x; bv[NB - 1] = v.y; }
                else bv[0] = b_t[t * LDT];
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    if (ch0 < ch1) fetch(ch0, pa0, pb0);
    if (ch0 + 1 < ch1) fetch(ch0 + 1, pa1, pb1);
    for (int ch = ch0; ch < ch1; ch += 2) {
        __syncthreads();                         // the previous chunk's fragment reads are done
        stage(pa0, pb0);
        __syncthreads();
        if (ch + 2 < ch1) fetch(ch + 2, pa0, pb0);
        compute();
        if (ch + 1 < ch1) {
            __syncthreads();
            stage(pa1, pb1);
            __syncthreads();
            if (ch + 3 < ch1) fetch(ch + 3, pa1, pb1);
            compute();
        }
    }

    if (A_T && want_rowsum && tid < BT && row0 + tid < a.R) {
        if (a.ks > 1) a.rowsum_slabs[(int64_t)z * a.R + row0 + tid] = rowsum;
        else a.rowsum[row0 + tid] = a.rowsum_accum ? a.rowsum[row0 + tid] + rowsum : rowsum;
    }
    // accumulator (i, j)[r] -> row rho = (r&3) + 8 (r>>2) + 4 h, column gamma = c of the 32x32 block; block (i, j) holds
    // wave rows 32 i + rho (A row-major) or 2 rho + i (A contraction-slow), wave columns 2 gamma + j (NB == 2) / gamma
    float* dst;
    int64_t ldd;
    if (a.ks > 1) { dst = a.slabs + ((int64_t)z * gridDim.z + tap) * a.R * a.C; ldd = a.C; }      // [split][tap][R][C]
    else { dst = a.out + (int64_t)tap * a.C; ldd = a.ldo; }
    const bool accum = a.ks == 1 && a.accum;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int row = row0 + wm * WT + (A_T ? (NB == 2 ? 2 * rho + i : rho) : 32 * i + rho);
            const int col = col0 + wn * WT + (NB == 2 ? 2 * c : c);
            if (row < a.R && col < a.C) {
                float* o = dst + (int64_t)row * ldd + col;
                if (NB == 2) {
                    f32x2 v{acc[i][0][r], acc[i][NB - 1][r]};
                    if (accum) v += *reinterpret_cast<const f32x2*>(o);
                    *reinterpret_cast<f32x2*>(o) = v;
                } else {
                    float v = acc[i][0][r];
                    if (accum) v += *o;
                    *o = v;
                }
            }
        }
}

// ---- bf16 OPERAND mode (MUMPY_MATH_BF16 OR-ed into `accumulate`; config 5's arithmetic, train.py:94-138 under bf16) --------
// Same products, same tensors in memory (fp32), same split / reduce / accumulate contract; the operands are rounded to bf16
// (RNE, v_cvt_pk_bf16_f32) while they are staged and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  A lane
// of that MFMA supplies EIGHT consecutive contraction values of one row, so both LDS images are [row][k] (32 bf16 = 16 dwords
// per row, padded to 20: conflict-free ds_read_b128 fragments).  An operand whose contraction index is the slow one in memory
// ([k][col]: dY and x of the dW product, W of the dX product) is transposed IN REGISTERS on its way to LDS: a thread loads the
// PIECES (k) x 4 (col) block k = kb PIECES .. +PIECES-1, cols 4 cq .. +3 as PIECES 16-byte loads (lanes: kb fastest, so the
// 64-128 contiguous bytes of a k row are read by neighbouring lanes and the PIECES bf16 of a column land in consecutive LDS
// dwords of neighbouring lanes -- at most the 2-way conflict that 64 lanes x 8 B have anyway) and writes one 4/8-byte
// k-run per column.  No transposed copies in HBM, no extra launches: the first version of the bf16 mode ran three transpose
// kernels + two forward GEMMs + a column sum per Linear and ~45 launches per convolution weight gradient.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int WT, bool A_T, bool CONV = false>
__global__ __launch_bounds__(256) void xgemm16_kernel(XArgs a) {
    constexpr int BT = 2 * WT, NB = WT / 32;
    constexpr int LDH = 20;                      // dwords per LDS row: 32 bf16 + 4 dwords of padding
    constexpr int PIECES = BT / 32;              // 16-B pieces per thread and operand per chunk (4 or 2)
    constexpr int KB = BK / PIECES;              // k blocks of a chunk (8 or 16)
    constexpr int CQW = 64 / KB;                 // column quads per wave (8 or 4)
    __shared__ __attribute__((aligned(16))) uint32_t As[BT * LDH];
    __shared__ __attribute__((aligned(16))) uint32_t Bs[BT * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x, z = blockIdx.y;
    const int tr = tile / a.gc, tc = tile - tr * a.gc;
    const int row0 = tr * BT, col0 = tc * BT;
    const int ch0 = z * a.chunks_per_split;
    int ch1 = ch0 + a.chunks_per_split;
    if (ch1 > a.nchunks) ch1 = a.nchunks;
    const int tap = CONV ? (int)blockIdx.z : 0;
    const int tap_r = CONV ? tap / a.cv_kw : 0, tap_s = CONV ? tap - tap_r * a.cv_kw : 0;
    const int dr = tap_r - a.cv_ph, dc = tap_s - a.cv_pw;
    const int64_t tap_shift = CONV ? (int64_t)dr * a.cvW + dc : 0;
    // transposing loader: this thread's k block and column quad
    const int kb = lane & (KB - 1), cq = (lane / KB) + CQW * wave;

    f32x4 pa0[PIECES], pb0[PIECES], pa1[PIECES], pb1[PIECES];
    auto fetch = [&](int ch, f32x4 (&pa)[PIECES], f32x4 (&pb)[PIECES]) {
        const int k0 = ch * BK;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int k = k0 + kb * PIECES + i;
            if (A_T) {
                const int r = row0 + 4 * cq;
                pa[i] = (k < a.KK && r < a.R) ? *reinterpret_cast<const f32x4*>(a.A + (int64_t)k * a.lda + r) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const int p = tid + 256 * i;
                int r = row0 + (p >> 3);
                if (r > a.R - 1) r = a.R - 1;            // rows past the edge: clamped, their products are never stored
                pa[i] = *reinterpret_cast<const f32x4*>(a.A + (int64_t)r * a.lda + k0 + 4 * (p & 7));
            }
            const int cc = col0 + 4 * cq;
            bool ok = k < a.KK && cc < a.C;
            if (CONV) {
                const unsigned img = __umulhi((unsigned)k, a.cv_mhw) >> a.cv_shw, rem = (unsigned)k - img * (unsigned)(a.cvH * a.cvW);
                const unsigned yy = __umulhi(rem, a.cv_mw) >> a.cv_sw, xx = rem - yy * (unsigned)a.cvW;
                ok = ok && (unsigned)((int)yy + dr) < (unsigned)a.cvH && (unsigned)((int)xx + dc) < (unsigned)a.cvW;
            }
            pb[i] = ok ? *reinterpret_cast<const f32x4*>(a.B + ((int64_t)k + tap_shift) * a.ldb + cc) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // k-run of column j of a transposed block: PIECES bf16 at [4 cq + j][kb PIECES ..]
    auto put_t = [&](uint32_t* img, const f32x4 (&pv)[PIECES]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t* dst = img + (4 * cq + j) * LDH + (kb * PIECES) / 2;
            if (PIECES == 4) {
                const bf16x2 lo = __builtin_convertvector(f32x2{pv[0][j], pv[1][j]}, bf16x2);
                const bf16x2 hi = __builtin_convertvector(f32x2{pv[PIECES - 2][j], pv[PIECES - 1][j]}, bf16x2);
                uint2 w;
                w.x = __builtin_bit_cast(uint32_t, lo); w.y = __builtin_bit_cast(uint32_t, hi);
                *reinterpret_cast<uint2*>(dst) = w;
            } else {
                const bf16x2 lo = __builtin_convertvector(f32x2{pv[0][j], pv[PIECES - 1][j]}, bf16x2);
                *dst = __builtin_bit_cast(uint32_t, lo);
            }
        }
    };
    auto stage = [&](const f32x4 (&pa)[PIECES], const f32x4 (&pb)[PIECES]) {
        if (A_T) put_t(As, pa);
        else {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int p = tid + 256 * i;
                const bf16x2 lo = __builtin_convertvector(f32x2{pa[i].x, pa[i].y}, bf16x2), hi = __builtin_convertvector(f32x2{pa[i].z, pa[i].w}, bf16x2);
                uint2 w;
                w.x = __builtin_bit_cast(uint32_t, lo); w.y = __builtin_bit_cast(uint32_t, hi);
                *reinterpret_cast<uint2*>(&As[(p >> 3) * LDH + 2 * (p & 7)]) = w;
            }
        }
        put_t(Bs, pb);
    };

    f32x16 acc[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // MFMA step st of a chunk: lane (c, h) supplies k = 16 st + 8 h .. + 7 of row c
    const uint32_t* const a_f = As + (wm * WT + c) * LDH + 4 * h;
    const uint32_t* const b_f = Bs + (wn * WT + c) * LDH + 4 * h;
    const bool want_rowsum = A_T && a.rowsum != nullptr && tc == 0 && tap == 0;
    float rowsum = 0.f;
    auto compute = [&]() {
        if (A_T && want_rowsum && tid < BT) {        // bias gradient from the bf16-rounded dY tile (as a bf16 autocast backward sums it)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(&As[tid * LDH + 4 * q]);
#pragma unroll
                for (int e = 0; e < 8; ++e) rowsum += (float)v[e];
            }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            bf16x8 fa[NB], fb[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a_f + 32 * i * LDH + 8 * st);
#pragma unroll
            for (int j = 0; j < NB; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(b_f + 32 * j * LDH + 8 * st);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    };
    if (ch0 < ch1) fetch(ch0, pa0, pb0);
    if (ch0 + 1 < ch1) fetch(ch0 + 1, pa1, pb1);
    for (int ch = ch0; ch < ch1; ch += 2) {
        __syncthreads();
        stage(pa0, pb0);
        __syncthreads();
        if (ch + 2 < ch1) fetch(ch + 2, pa0, pb0);
        compute();
        if (ch + 1 < ch1) {
            __syncthreads();
            stage(pa1, pb1);
            __syncthreads();
            if (ch + 3 < ch1) fetch(ch + 3, pa1, pb1);
            compute();
        }
    }

    if (A_T && want_rowsum && tid < BT && row0 + tid < a.R) {
        if (a.ks > 1) a.rowsum_slabs[(int64_t)z * a.R + row0 + tid] = rowsum;
        else a.rowsum[row0 + tid] = a.rowsum_accum ? a.rowsum[row0 + tid] + rowsum : rowsum;
    }
    // accumulator (i, j)[r]: wave row 32 i + (r&3) + 8 (r>>2) + 4 h, wave column 32 j + c (both images are [row][k]: no permutation)
    float* dst;
    int64_t ldd;
    if (a.ks > 1) { dst = a.slabs + ((int64_t)z * gridDim.z + tap) * a.R * a.C; ldd = a.C; }
    else { dst = a.out + (int64_t)tap * a.C; ldd = a.ldo; }
    const bool accum = a.ks == 1 && a.accum;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wm * WT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = col0 + wn * WT + 32 * j + c;
                if (row < a.R && col < a.C) {
                    float* o = dst + (int64_t)row * ldd + col;
                    float v = acc[i][j][r];
                    if (accum) v += *o;
                    *o = v;
                }
            }
}

// out = (accum ? out : 0) + slab[0] + slab[1] + ... in split order (fixed: bitwise reproducible); n4 = R C / 4, dense
struct RowSumArgs { const float* slabs; float* out; int R, accum; };     // the bias-gradient partials ride in the same launch

__device__ __forceinline__ void reduce_rowsum(const RowSumArgs& rs, int ks) {
    if (!rs.out) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rs.R; i += gridDim.x * 256) {
        const float s = ordered_sum(rs.slabs[i], rs.slabs + rs.R + i, (int64_t)rs.R, ks - 1);
        rs.out[i] = rs.accum ? rs.out[i] + s : s;
    }
}

__global__ __launch_bounds__(256) void xgemm_reduce_kernel(const f32x4* __restrict__ slabs, f32x4* __restrict__ out, int64_t n4,
                                                           int ks, int accum, RowSumArgs rs) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 s = ordered_sum(slabs[i], slabs + n4 + i, n4, ks - 1);
        if (accum) s += out[i];
        out[i] = s;
    }
    reduce_rowsum(rs, ks);
}

// two reductions in ONE launch (the dX and the dW product of a Linear backward, both split): blocks [0, g0) take job 0, the rest
// job 1; each job is the fixed-order sum above.  Saves a launch per nn.Linear of the training step (~250 of them).
struct ReduceJob { const f32x4* slabs; f32x4* out; int64_t n4; int ks, accum; unsigned blocks; };
__global__ __launch_bounds__(256) void xgemm_reduce2_kernel(ReduceJob j0, ReduceJob j1, RowSumArgs rs) {
    const bool second = blockIdx.x >= j0.blocks;
    const ReduceJob& j = second ? j1 : j0;
    const unsigned bid = second ? blockIdx.x - j0.blocks : blockIdx.x;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < j.n4; i += (int64_t)j.blocks * 256) {
        f32x4 s = ordered_sum(j.slabs[i], j.slabs + j.n4 + i, j.n4, j.ks - 1);
        if (j.accum) s += j.out[i];
        j.out[i] = s;
    }
    if (second) {                                   // the bias-gradient partials ride with the dW job
        if (!rs.out) return;
        for (int i = bid * 256 + threadIdx.x; i < rs.R; i += j1.blocks * 256) {
            const float s = ordered_sum(rs.slabs[i], rs.slabs + rs.R + i, (int64_t)rs.R, j1.ks - 1);
            rs.out[i] = rs.accum ? rs.out[i] + s : s;
        }
    }
}

// the same for the convolution gradient: slabs [split][tap][R][C] -> out[r][tap * C + c] (row pitch ldo = taps * C)
__global__ __launch_bounds__(256) void xgemm_reduce_taps_kernel(const f32x4* __restrict__ slabs, float* __restrict__ out, int R, int C,
                                                                int taps, int ks, int accum, RowSumArgs rs) {
    reduce_rowsum(rs, ks);
    const int c4n = C / 4;
    const int64_t per = (int64_t)taps * R * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
        f32x4 s = ordered_sum(slabs[i], slabs + per + i, per, ks - 1);
        const int64_t t = i / ((int64_t)R * c4n), rc = i - t * R * c4n;
        const int64_t r = rc / c4n, c4 = rc - r * c4n;
        f32x4* o = reinterpret_cast<f32x4*>(out + (r * taps + t) * C) + c4;
        if (accum) s += *o;
        *o = s;
    }
}

// column sums of dY (M, N): one partial row per block of rows, then a fixed-order reduce over the partial rows
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t R,
                                                             int C, int rows_per_block) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int cc = blockIdx.x * 64 + col;
    float s = 0.f;
    if (cc < C)
        for (int64_t r = r0 + rl; r < r0 + rows_per_block && r < R; r += 4) s += x[r * C + cc];
    red[rl][col] = s;
    __syncthreads();
    if (rl == 0 && cc < C) partial[(int64_t)blockIdx.y * C + cc] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int nparts,
                                                           int C, int accum) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int cc = blockIdx.x * 64 + col;
    float s = 0.f;
    if (cc < C)
        for (int p = g; p < nparts; p += 4) s += partial[(int64_t)p * C + cc];
    red[g][col] = s;
    __syncthreads();
    if (g == 0 && cc < C) {
        const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
        out[cc] = accum ? out[cc] + t : t;
    }
}

// short matrices (M <= COLSUM_DIRECT_ROWS): one launch -- 64 columns per block, 16 row lanes each summing every 16th row,
// combined through LDS in lane order (a fixed tree: bitwise reproducible)
constexpr int64_t COLSUM_DIRECT_ROWS = 4096;
__global__ __launch_bounds__(1024) void colsum_direct_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t R, int C,
                                                             int accum) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int cc = blockIdx.x * 64 + col;
    float s = 0.f;
    if (cc < C && g < R)
        s = ordered_sum(x[(int64_t)g * C + cc], x + (int64_t)(g + 16) * C + cc, (int64_t)16 * C, (int)((R - g + 15) / 16) - 1);
    red[g][col] = s;
    __syncthreads();
    if (g == 0 && cc < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][col];
        out[cc] = accum ? out[cc] + t : t;
    }
}

struct XPlan { int wt, ks, cps, nchunks, gr, gc; };

XPlan plan(int R, int C, int KK, int taps = 1) {
    XPlan p;
    const int64_t t128 = (int64_t)((R + 127) / 128) * ((C + 127) / 128) * taps;
    static const int wide_at = tune_int("MUMPY_XG_WIDE_AT", 96);
    p.wt = t128 >= wide_at ? 64 : 32;                  // wide tiles only when they fill a good part of the chip
    const int bt = 2 * p.wt;
    p.gr = (R + bt - 1) / bt; p.gc = (C + bt - 1) / bt;
    p.nchunks = (KK + BK - 1) / BK;
    const int64_t tiles = (int64_t)p.gr * p.gc * taps;
    static const int t64 = tune_int("MUMPY_XG_TARGET64", 512), t32 = tune_int("MUMPY_XG_TARGET32", 768), minc = tune_int("MUMPY_XG_MINCHUNKS", 4);
    const int64_t target = p.wt == 64 ? t64 : t32;     // workgroups wanted (2 / 3+ resident per CU)
    int ks = (int)(target / tiles);
    if (ks > p.nchunks / minc) ks = p.nchunks / minc;
    if (ks > 64) ks = 64;
    if (ks < 1) ks = 1;
    p.cps = (p.nchunks + ks - 1) / ks;
    p.ks = (p.nchunks + p.cps - 1) / p.cps;            // no empty splits
    return p;
}

int64_t colsum_blocks(int64_t R) {
    int64_t b = (R + 127) / 128;
    if (b > 1024) b = 1024;
    return b < 1 ? 1 : b;
}

// rowsum (A_T only): also write / accumulate the row sums of A over the contraction (the bias gradient) -- the slab layout
// is [ks][R][C] followed by [ks][R] row-sum partials
// defer: when non-null and the product is split, the reduce is NOT launched; its description is returned for xgemm_reduce2_kernel
struct Deferred { ReduceJob job; RowSumArgs rs; bool pending; };
int launch_xgemm(bool a_t, const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t ldo, int R, int C, int KK,
                 int accum, float* ws, int64_t ws_bytes, hipStream_t s, float* rowsum = nullptr, int rowsum_accum = 0, bool bf16 = false,
                 Deferred* defer = nullptr) {
    XPlan p = plan(R, C, KK);
    if (p.ks > 1 && (ldo != C || (int64_t)p.ks * R * (C + (rowsum ? 1 : 0)) * 4 > ws_bytes)) { p.ks = 1; p.cps = p.nchunks; }
    XArgs a;
    a.rowsum = rowsum; a.rowsum_slabs = ws ? ws + (int64_t)p.ks * R * C : nullptr; a.rowsum_accum = rowsum_accum;
    a.A = A; a.B = B; a.out = out; a.slabs = ws; a.lda = lda; a.ldb = ldb; a.ldo = ldo;
    a.R = R; a.C = C; a.KK = KK; a.ks = p.ks; a.chunks_per_split = p.cps; a.nchunks = p.nchunks; a.accum = accum; a.gc = p.gc;
    a.cvH = a.cvW = a.cv_kw = a.cv_ph = a.cv_pw = 0; a.cv_mhw = a.cv_shw = a.cv_mw = a.cv_sw = 0;
    const dim3 grid((unsigned)(p.gr * p.gc), (unsigned)p.ks);
    if (bf16) {
        if (p.wt == 64) {
            if (a_t) hipLaunchKernelGGL((xgemm16_kernel<64, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((xgemm16_kernel<64, false>), grid, dim3(256), 0, s, a);
        } else {
            if (a_t) hipLaunchKernelGGL((xgemm16_kernel<32, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((xgemm16_kernel<32, false>), grid, dim3(256), 0, s, a);
        }
    } else if (p.wt == 64) {
        if (a_t) hipLaunchKernelGGL((xgemm_kernel<64, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((xgemm_kernel<64, false>), grid, dim3(256), 0, s, a);
    } else {
        if (a_t) hipLaunchKernelGGL((xgemm_kernel<32, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((xgemm_kernel<32, false>), grid, dim3(256), 0, s, a);
    }
    MUMPY_CHECK_LAUNCH("linear_bwd(product)");
    if (defer) defer->pending = false;
    if (p.ks > 1) {
        const int64_t n4 = (int64_t)R * C / 4;
        int64_t g = (n4 + 255) / 256;
        if (g > 2048) g = 2048;
        const RowSumArgs rs{a.rowsum_slabs, rowsum, R, rowsum_accum};
        if (defer) {
            defer->job = ReduceJob{reinterpret_cast<const f32x4*>(ws), reinterpret_cast<f32x4*>(out), n4, p.ks, accum, (unsigned)g};
            defer->rs = rs;
            defer->pending = true;
            return 0;
        }
        hipLaunchKernelGGL(xgemm_reduce_kernel, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const f32x4*>(ws),
                           reinterpret_cast<f32x4*>(out), n4, p.ks, accum, rs);
        MUMPY_CHECK_LAUNCH("linear_bwd(reduce)");
    }
    return 0;
}

int launch_reduce(const Deferred& d, hipStream_t s) {
    hipLaunchKernelGGL(xgemm_reduce_kernel, dim3(d.job.blocks), dim3(256), 0, s, d.job.slabs, d.job.out, d.job.n4, d.job.ks, d.job.accum, d.rs);
    MUMPY_CHECK_LAUNCH("linear_bwd(reduce)");
    return 0;
}

}  // namespace

extern "C" int64_t mumpy_linear_bwd_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const XPlan p = plan(N, K, (int)M);
    int64_t need = p.ks > 1 ? (int64_t)p.ks * N * (K + 1) * 4 : 0;          // dW slabs + the bias-gradient partials
    const XPlan q = plan((int)M, K, N);
    const int64_t dx_slabs = q.ks > 1 ? (int64_t)q.ks * M * K * 4 : 0;      // dX slabs (deep N, few tiles); skipped when huge
    if (dx_slabs <= (256ll << 20)) need += dx_slabs;                         // BEHIND the dW slabs: both products' slabs are alive
                                                                             // until the one reduce launch that sums them
    const int64_t cs = colsum_blocks(M) * N * 4;
    return need > cs ? need : cs;
}

extern "C" int mumpy_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW, float* db, int64_t M,
                                int N, int K, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    if (M == 0) return 0;
    MUMPY_REQUIRE(dy && (dx || dW || db), MUMPY_ENULL, "linear_bwd: null pointer");
    MUMPY_REQUIRE((!dx || W) && (!dW || x), MUMPY_ENULL, "linear_bwd: dx needs W, dW needs x");
    MUMPY_REQUIRE(M > 0 && M < (1ll << 31) - 256 && N > 0 && K > 0 && N % 32 == 0 && K % 32 == 0, MUMPY_EINVAL,
                  "linear_bwd: need N %% 32 == 0 and K %% 32 == 0 (got M=%lld N=%d K=%d)", (long long)M, N, K);
    MUMPY_REQUIRE(aligned16(x) && aligned16(W) && aligned16(dy) && aligned16(dx) && aligned16(dW) && aligned16(db) && aligned16(workspace),
                  MUMPY_EALIGN, "linear_bwd: pointers must be 16-byte aligned");
    MUMPY_REQUIRE((accumulate & ~(3 | MUMPY_MATH_BF16)) == 0, MUMPY_EINVAL, "linear_bwd: unknown accumulate bits 0x%x", accumulate);
    const bool bf16 = (accumulate & MUMPY_MATH_BF16) != 0;
    MUMPY_REQUIRE(!workspace || workspace_bytes >= mumpy_linear_bwd_workspace_bytes(M, N, K), MUMPY_EINVAL,
                  "linear_bwd: workspace too small");
    MUMPY_REQUIRE(!db || dW || workspace, MUMPY_ENULL, "linear_bwd: db without dW needs the workspace");
    hipStream_t s = as_stream(stream);
    float* ws = static_cast<float*>(workspace);
    // db rides in the dW launch (row sums of dY^T over the tokens, taken from the tiles staged anyway); on its own only
    // when no dW is wanted
    if (db && dW) {
    } else if (db && M <= COLSUM_DIRECT_ROWS) {
        hipLaunchKernelGGL(colsum_direct_kernel, dim3((unsigned)((N + 63) / 64)), dim3(1024), 0, s, dy, db, M, N, (accumulate >> 1) & 1);
        MUMPY_CHECK_LAUNCH("linear_bwd(bias)");
    } else if (db) {                             // first: its partial rows share the workspace with dW's slabs
        const int64_t nb = colsum_blocks(M);
        const int rpb = (int)((M + nb - 1) / nb);
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)nb), dim3(256), 0, s, dy, ws, M, N, rpb);
        MUMPY_CHECK_LAUNCH("linear_bwd(bias partial)");
        hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, ws, db, (int)nb, N, (accumulate >> 1) & 1);
        MUMPY_CHECK_LAUNCH("linear_bwd(bias)");
    }
    // workspace layout: [dW slabs + bias-gradient partials | dX slabs]; when both products are split their reduces are ONE launch
    const XPlan pw = plan(N, K, (int)M);
    const int64_t w_bytes = (dW && pw.ks > 1) ? (int64_t)pw.ks * N * (K + 1) * 4 : 0;
    float* ws_x = ws ? ws + w_bytes / 4 : nullptr;
    const int64_t ws_x_bytes = workspace ? workspace_bytes - w_bytes : 0;
    Deferred dfx{}, dfw{};
    if (dx)          // dX[M,K] = dY[M,N] W[N,K]
        if (int rc = launch_xgemm(false, dy, N, W, K, dx, K, (int)M, K, N, 0, ws_x, ws_x_bytes > 0 ? ws_x_bytes : 0, s, nullptr, 0, bf16,
                                  dW ? &dfx : nullptr)) return rc;
    if (dW)          // dW[N,K] (+)= dY^T X
        if (int rc = launch_xgemm(true, dy, N, x, K, dW, K, N, K, (int)M, accumulate & 1, ws, workspace ? workspace_bytes : 0, s, db,
                                  (accumulate >> 1) & 1, bf16, dx ? &dfw : nullptr)) return rc;
    if (dfx.pending && dfw.pending) {
        hipLaunchKernelGGL(xgemm_reduce2_kernel, dim3(dfx.job.blocks + dfw.job.blocks), dim3(256), 0, s, dfx.job, dfw.job, dfw.rs);
        MUMPY_CHECK_LAUNCH("linear_bwd(reduce x2)");
    } else {
        if (dfx.pending) if (int rc = launch_reduce(dfx, s)) return rc;
        if (dfw.pending) if (int rc = launch_reduce(dfw, s)) return rc;
    }
    return 0;
}

// n / d for every n < 2^31 as mulhi(n, magic) >> shift (round-up method, 31-bit dividends; d >= 2)
static void magic_div31(unsigned d, unsigned& magic, unsigned& shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    magic = (unsigned)(((1ull << (31 + s)) + d - 1) / d);
    shift = s - 1;
}

extern "C" int64_t mumpy_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int kh, int kw) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0) return 0;
    const XPlan p = plan(Cout, Cin, B * H * W, kh * kw);
    return p.ks > 1 ? (int64_t)p.ks * kh * kw * Cout * Cin * 4 : 0;
}

extern "C" int mumpy_conv2d_wgrad_nhwc(const float* x, const float* dy, float* dW, int B, int H, int W, int Cin, int Cout, int kh,
                                       int kw, int accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
    MUMPY_REQUIRE(x && dy && dW, MUMPY_ENULL, "conv2d_wgrad: null pointer");
    MUMPY_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dW) && aligned16(workspace), MUMPY_EALIGN,
                  "conv2d_wgrad: pointers must be 16-byte aligned");
    MUMPY_REQUIRE(B > 0 && H > 0 && W >= 2 && (int64_t)B * H * W < (1ll << 31) - 256, MUMPY_EINVAL, "conv2d_wgrad: bad image batch %dx%dx%d", B, H, W);
    MUMPY_REQUIRE(kh > 0 && kw > 0 && (kh & 1) && (kw & 1), MUMPY_EINVAL, "conv2d_wgrad: kernel %dx%d must be odd (same padding)", kh, kw);
    MUMPY_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0, MUMPY_EINVAL, "conv2d_wgrad: Cin=%d and Cout=%d must be multiples of 32", Cin, Cout);
    MUMPY_REQUIRE((accumulate & ~(1 | MUMPY_MATH_BF16)) == 0, MUMPY_EINVAL, "conv2d_wgrad: accumulate is 0 / 1, optionally | MUMPY_MATH_BF16");
    const bool bf16 = (accumulate & MUMPY_MATH_BF16) != 0;
    accumulate &= 1;
    const int P = B * H * W, taps = kh * kw;
    XPlan p = plan(Cout, Cin, P, taps);
    if (p.ks > 1 && (!workspace || (int64_t)p.ks * taps * Cout * Cin * 4 > workspace_bytes)) { p.ks = 1; p.cps = p.nchunks; }
    XArgs a;
    a.A = dy; a.B = x; a.out = dW; a.slabs = static_cast<float*>(workspace);
    a.lda = Cout; a.ldb = Cin; a.ldo = (int64_t)taps * Cin;
    a.R = Cout; a.C = Cin; a.KK = P; a.ks = p.ks; a.chunks_per_split = p.cps; a.nchunks = p.nchunks; a.accum = accumulate; a.gc = p.gc;
    a.rowsum = nullptr; a.rowsum_slabs = nullptr; a.rowsum_accum = 0;
    a.cvH = H; a.cvW = W; a.cv_kw = kw; a.cv_ph = kh / 2; a.cv_pw = kw / 2;
    magic_div31((unsigned)(H * W), a.cv_mhw, a.cv_shw);
    magic_div31((unsigned)W, a.cv_mw, a.cv_sw);
    hipStream_t s = as_stream(stream);
    const dim3 grid((unsigned)(p.gr * p.gc), (unsigned)p.ks, (unsigned)taps);
    if (bf16 && p.wt == 64) hipLaunchKernelGGL((xgemm16_kernel<64, true, true>), grid, dim3(256), 0, s, a);
    else if (bf16) hipLaunchKernelGGL((xgemm16_kernel<32, true, true>), grid, dim3(256), 0, s, a);
    else if (p.wt == 64) hipLaunchKernelGGL((xgemm_kernel<64, true, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((xgemm_kernel<32, true, true>), grid, dim3(256), 0, s, a);
    MUMPY_CHECK_LAUNCH("conv2d_wgrad(product)");
    if (p.ks > 1) {
        const int64_t per = (int64_t)taps * Cout * (Cin / 4);
        int64_t g = (per + 255) / 256;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(xgemm_reduce_taps_kernel, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const f32x4*>(workspace), dW,
                           Cout, Cin, taps, p.ks, accumulate, RowSumArgs{nullptr, nullptr, 0, 0});
        MUMPY_CHECK_LAUNCH("conv2d_wgrad(reduce)");
    }
    return 0;
}
