// whiten.hip — the matrix-core kernels of the PCA-whitening step
// (pycleora/__init__.py:130-164 `whiten_embeddings`), for gfx950.
//
//   centred Gram    G = sum_r (x_r - mu)(x_r - mu)^T  in f64     (:138-143, without the 1/(n-1))
//     gram_kernel: v_mfma_f64_16x16x4_f64; operands are centred and widened to f64 when the X tile is staged into LDS, so the
//     accumulation is f64 end to end like the reference's `block.T @ block` on an f64 block.  Only block tiles on or above the
//     diagonal, and of a diagonal block tile only its 36 upper 16x16 MFMA tiles (9 per wave); row slices are combined in a fixed
//     order (deterministic).  Mean and Gram in ONE pass around a sampled shift, corrected exactly.
//     gram16_kernel: the same statistics from the bf16 matrix cores with split f32 operands, for the whitened loop's INTERMEDIATE
//     iterations only (v_mfma_f32_32x32x16_bf16; three or six exact bf16 products per f32 product).
//   projection      out = (X - mu_f32) @ T  in f32                 (:157-163)
//     project_split_kernel: every f32 product from six bf16 MFMAs of three-way split operands — not a reduced-precision path: its
//     error against an f64 product is that of an f32 GEMM (tests/test_gpu_whiten.py); the A split is made once per row group and
//     shared through LDS; optional row normalisation in the epilogue (the reorganised loop of abi.hip / sharded.hip).
//     project_kernel: v_mfma_f32_32x32x2_f32, LDS-tiled, any shape (d not a multiple of 32).
//
// All are matrix-core work (2 n d^2 flops against 1-3 passes over X), unlike the SpMM; docs/history.md §3.5 / 3.6 have what bounds each.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "project_common.h"

namespace cleora {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

// v from the lane a DPP control selects inside this lane's 16-lane row (0xB1 / 0x4E: quad_perm [1,0,3,2] / [2,3,0,1]; 0x141 / 0x140:
// row_half_mirror / row_mirror)
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
typedef float f16v __attribute__((ext_vector_type(16)));

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------
// centred Gram, f64 matrix cores
// ---------------------------------------------------------------------------------------------
constexpr int GT = 128;     // block tile edge (columns of X on each side)
constexpr int GKC = 16;     // rows of X per LDS chunk
constexpr int GLD = 144;    // LDS row stride in doubles: 288 dwords = 32 (mod 64) -> the two
                            // 16-lane row groups of a ds_read_b64 half hit disjoint banks

struct GramArgs {
    const float *x;
    uint64_t ldx;
    uint64_t n;
    uint32_t d;
    const double *mean;     // centring vector: the exact mean (two-pass form) or a nearby shift (one-pass form)
    double *colsum;         // [s_diag][tiles * GT]: per-slice column sums of (x - mean), written by the DIAG blocks
    double *partial;        // [slices][pairs][GT][GT]
    uint32_t tiles;         // ceil(d / GT)
    uint32_t pairs;         // tiles (tiles + 1) / 2
    uint64_t rows_per_slice;
    int w4;
};

__device__ __forceinline__ void pair_to_tiles(uint32_t p, uint32_t tiles, uint32_t &bi, uint32_t &bj) {
    bi = 0;
    uint32_t rowlen = tiles;
    while (p >= rowlen) { p -= rowlen; ++bi; --rowlen; }
    bj = bi + p;
}

// 4 consecutive floats of row `rp` starting at column c (zero beyond d / invalid row).
__device__ __forceinline__ float4 load4(const float *rp, uint32_t c, uint32_t d, bool row_ok, int w4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    if (w4) {
        if (c < d) v = *reinterpret_cast<const float4 *>(rp + c);
    } else {
        if (c + 0 < d) v.x = rp[c + 0];
        if (c + 1 < d) v.y = rp[c + 1];
        if (c + 2 < d) v.z = rp[c + 2];
        if (c + 3 < d) v.w = rp[c + 3];
    }
    return v;
}

// Tiles of a DIAGONAL block tile: only the 36 MFMA tiles (ti <= tj) of its 8 x 8 grid are needed (the rest is the
// mirror image).  Wave WV takes tile rows WV (columns WV..7) and 7-WV (columns 7-WV..7): 9 tiles each, and — the A
// fragment of tile row r being the same LDS column block as the B fragment of tile column r — only the 8-WV column
// fragments WV..7 are read per k-step.  (Round 1 dealt the tiles out in sequence and read two fragments per MFMA: 18
// reads x 8 resident waves kept the CU's LDS port busy for as long as the 9 MFMAs take; r02_whiten_pmc.json, 70 % MFMA
// busy against 81 % for the off-diagonal blocks.)  The wave index is a template parameter so that accumulators and
// fragments are fixed registers; the kernel branches on it once.
__host__ __device__ constexpr int diag_tile_row(int wv, int i) { return i < 8 - wv ? wv : 7 - wv; }
__host__ __device__ constexpr int diag_tile_col(int wv, int i) { return i < 8 - wv ? wv + i : (7 - wv) + (i - (8 - wv)); }

// FAST: d is a multiple of the 128-column tile and rows are float4-aligned — the loads carry no column checks and no
// branches (rows past the slice are read from a clamped address and zeroed when staged), so the compiler keeps the
// prefetch in flight behind counted waits instead of `s_waitcnt vmcnt(0)` after every guarded load.
template <bool DIAG, bool FAST, int WV>
__device__ __forceinline__ void gram_body(const GramArgs &a, double (&lds)[2][DIAG ? 1 : 2][GKC][GLD],
                                          uint32_t bi, uint32_t bj, uint32_t pair) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const uint64_t r_begin = (uint64_t)blockIdx.y * a.rows_per_slice;
    const uint64_t r_end = r_begin + a.rows_per_slice < a.n ? r_begin + a.rows_per_slice : a.n;

    // loader role: 4 columns (c4*4 ..) of rows lr and lr+8 of each chunk, for both panels
    const int c4 = t & 31, lr = t >> 5;
    const uint32_t colA = bi * GT + c4 * 4, colB = bj * GT + c4 * 4;
    double mA[4], mB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mA[q] = (colA + q < a.d) ? a.mean[colA + q] : 0.0;
        mB[q] = (colB + q < a.d) ? a.mean[colB + q] : 0.0;
    }

    constexpr int NACC = DIAG ? 9 : 16;
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    // LDS column of the A / B fragment of accumulator i (lane & 15 added at the read)
    int fa_col[DIAG ? 9 : 4], fb_col[DIAG ? 9 : 4];
    if constexpr (DIAG) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            fa_col[i] = diag_tile_row(WV, i) * 16;
            fb_col[i] = diag_tile_col(WV, i) * 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa_col[i] = wr * 64 + i * 16;
            fb_col[i] = wc * 64 + i * 16;
        }
    }

    float4 pa[2], pb[2];
    bool ok[2];
    pb[0] = pb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    double cs[4] = {0.0, 0.0, 0.0, 0.0};   // DIAG: this thread's share of sum_r (x[r, colA + q] - mean): every row once
    auto prefetch = [&](uint64_t row0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint64_t r = row0 + lr + 8 * h;
            ok[h] = r < r_end;
            if constexpr (FAST) {
                const float *rp = a.x + (ok[h] ? r : r_begin) * a.ldx;      // always a valid row: no branch
                pa[h] = *reinterpret_cast<const float4 *>(rp + colA);
                if (!DIAG) pb[h] = *reinterpret_cast<const float4 *>(rp + colB);
            } else {
                const float *rp = a.x + r * a.ldx;
                pa[h] = load4(rp, colA, a.d, ok[h], a.w4);
                if (!DIAG) pb[h] = load4(rp, colB, a.d, ok[h], a.w4);
            }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float va[4] = {pa[h].x, pa[h].y, pa[h].z, pa[h].w};
            const float vb[4] = {pb[h].x, pb[h].y, pb[h].z, pb[h].w};
            double *da = &lds[buf][0][lr + 8 * h][c4 * 4];
            double *db = &lds[buf][DIAG ? 0 : 1][lr + 8 * h][c4 * 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // centre in f64: block.astype(float64) - mean          (pycleora/__init__.py:141)
                da[q] = (ok[h] && (FAST || colA + q < a.d)) ? (double)va[q] - mA[q] : 0.0;
                if constexpr (DIAG) cs[q] += da[q];
                if constexpr (!DIAG) db[q] = (ok[h] && (FAST || colB + q < a.d)) ? (double)vb[q] - mB[q] : 0.0;
            }
        }
    };

    // Double-buffered LDS, one barrier per chunk: while the MFMAs of chunk c run from buffer c&1,
    // chunk c+1 (already in registers) is staged into the other buffer and chunk c+2 is fetched.
    if (r_begin < r_end) {
        prefetch(r_begin);
        stage(0);
        if (r_begin + GKC < r_end) prefetch(r_begin + GKC);
    }
    __syncthreads();
    int buf = 0;
    for (uint64_t row0 = r_begin; row0 < r_end; row0 += GKC, buf ^= 1) {
        if (row0 + GKC < r_end) {
            stage(buf ^ 1);
            if (row0 + 2 * GKC < r_end) prefetch(row0 + 2 * GKC);
        }
        constexpr int PB = DIAG ? 0 : 1;
#pragma unroll
        for (int kk = 0; kk < GKC / 4; ++kk) {
            const int krow = kk * 4 + (lane >> 4);
            if constexpr (DIAG) {
                double f[8];                                   // column fragments WV..7; f[r] is also the A fragment of tile row r
#pragma unroll
                for (int c = WV; c < 8; ++c) f[c] = lds[buf][0][krow][c * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[diag_tile_row(WV, i)], f[diag_tile_col(WV, i)], acc[i], 0, 0, 0);
            } else {
                double fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = lds[buf][0][krow][fa_col[i] + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = lds[buf][PB][krow][fb_col[j] + (lane & 15)];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i * 4 + j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if constexpr (DIAG) {
        // column sums of the centred panel: the 8 row groups (lr) of a column are added in order through LDS
        // (the main loop ended on a barrier, so the staging buffers are free)
        double *red = &lds[0][0][0][0];
#pragma unroll
        for (int q = 0; q < 4; ++q) red[lr * GT + c4 * 4 + q] = cs[q];
        __syncthreads();
        if (t < GT) {
            double sum = red[t];
#pragma unroll
            for (int g = 1; g < 8; ++g) sum += red[g * GT + t];
            a.colsum[(uint64_t)blockIdx.y * a.tiles * GT + bi * GT + t] = sum;
        }
    }

    double *out = a.partial + ((uint64_t)blockIdx.y * a.pairs + pair) * (uint64_t)(GT * GT);
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        const int trow = DIAG ? fa_col[i] : fa_col[i / 4];
        const int tcol = DIAG ? fb_col[i] : fb_col[i % 4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            // f64 16x16x4 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
            out[(trow + (lane >> 4) + 4 * reg) * GT + tcol + (lane & 15)] = acc[i][reg];
        }
    }
}

__device__ __forceinline__ uint32_t pair_index(uint32_t bi, uint32_t bj, uint32_t tiles) {
    return bi * tiles - bi * (bi - 1) / 2 + (bj - bi);
}

// Two launches so each body gets its own register budget (2 waves/SIMD each):
// DIAG: blockIdx.x = diagonal tile; off-diagonal: blockIdx.x enumerates the pairs bi < bj.
template <bool DIAG, bool FAST>
__global__ __launch_bounds__(256) void gram_kernel(const GramArgs a) {
    __shared__ __attribute__((aligned(16))) double lds[2][DIAG ? 1 : 2][GKC][GLD];
    uint32_t bi, bj;
    if constexpr (DIAG) {
        bi = bj = blockIdx.x;
    } else {
        uint32_t q = blockIdx.x, rowlen = a.tiles - 1;
        bi = 0;
        while (q >= rowlen) { q -= rowlen; ++bi; --rowlen; }
        bj = bi + 1 + q;
    }
    if constexpr (DIAG) {
        switch (threadIdx.x >> 6) {                            // whole waves take each arm; every arm meets the same barriers
            case 0: gram_body<true, FAST, 0>(a, lds, bi, bj, pair_index(bi, bj, a.tiles)); break;
            case 1: gram_body<true, FAST, 1>(a, lds, bi, bj, pair_index(bi, bj, a.tiles)); break;
            case 2: gram_body<true, FAST, 2>(a, lds, bi, bj, pair_index(bi, bj, a.tiles)); break;
            default: gram_body<true, FAST, 3>(a, lds, bi, bj, pair_index(bi, bj, a.tiles)); break;
        }
    } else {
        gram_body<false, FAST, 0>(a, lds, bi, bj, pair_index(bi, bj, a.tiles));
    }
}

// One-pass form: the Gram was centred with a shift c near the mean.  delta = sum_r (x_r - c) / n (slices added in
// order), mean = c + delta (pycleora/__init__.py:136), mean32 = (float)mean (:159).
__global__ __launch_bounds__(256) void gram_mean_kernel(const double *__restrict__ colsum, uint32_t s_diag,
                                                        uint32_t tiles, uint32_t d, uint64_t n,
                                                        const double *__restrict__ shift, double *__restrict__ delta,
                                                        double *__restrict__ mean64, float *__restrict__ mean32) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    double s = 0.0;
    for (uint32_t sl = 0; sl < s_diag; ++sl) s += colsum[(uint64_t)sl * tiles * GT + c];
    const double dl = s / (double)n;
    delta[c] = dl;
    const double m = shift[c] + dl;
    mean64[c] = m;
    mean32[c] = (float)m;
}

// gram[gi][gj] = sum over slices (fixed order) of the partial tiles; mirrors the upper triangle.
// delta != nullptr: sum_r (x-c)(x-c)^T - n delta delta^T = sum_r (x-mu)(x-mu)^T  (exact identity, mu = c + delta).
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ partial,
                                                          uint32_t s_diag, uint32_t s_off,
                                                          uint32_t pairs, uint32_t tiles, uint32_t d,
                                                          const double *__restrict__ delta, double n_rows,
                                                          double *__restrict__ gram) {
    const uint32_t p = blockIdx.y;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;  // element of the GT x GT tile
    const uint32_t r = e / GT, c = e % GT;
    uint32_t bi, bj;
    pair_to_tiles(p, tiles, bi, bj);
    const bool diag = bi == bj;
    const uint32_t slices = diag ? s_diag : s_off;
    if (diag && (r / 16) > (c / 16)) return;  // not computed: mirror image of an upper MFMA tile
    const uint32_t gi = bi * GT + r, gj = bj * GT + c;
    if (gi >= d || gj >= d) return;
    double s = 0.0;
    for (uint32_t sl = 0; sl < slices; ++sl)
        s += partial[((uint64_t)sl * pairs + p) * (uint64_t)(GT * GT) + e];
    if (delta) s -= n_rows * (delta[gi] * delta[gj]);     // the product first: the same rounding for (i, j) and (j, i)
    gram[(uint64_t)gi * d + gj] = s;
    if (!diag || (r / 16) < (c / 16)) gram[(uint64_t)gj * d + gi] = s;
}

// ---------------------------------------------------------------------------------------------
// centred Gram for the INTERMEDIATE iterations of the whitened loop, d = 256 S: shared pieces
// ---------------------------------------------------------------------------------------------
// Inside E <- whiten(l2_normalise(A E)) the intermediate whitenings only have to be whitenings (eigh.hip: the Cholesky form);
// only the last iteration, and every caller that looks at a whitened iterate, needs the f64 Gram above
// (pycleora/__init__.py:138-143).  For those iterations the Gram comes from the bf16 matrix cores with split f32 operands
// (gram16_kernel below).  Common to it (and to the f32-matrix-core form it replaced, scripts/rejected/):
//   * the columns are cut into S super-tiles of 256.  A DIAGONAL block owns one super-tile: the 36 upper 32x32 tiles of its
//     8 x 8 grid; at d = 256 that is the whole matrix and X is read exactly once.  An OFF-DIAGONAL pair (I < J) of
//     super-tiles is two blocks, each 128 columns of I against all 256 of J.  The blocks of one row slice are neighbours in
//     the grid and read the same rows at about the same time (L2 / Infinity Cache);
//   * f32 accumulators run over at most `sub_rows` (2048) rows, then fold into the block's private f64 partial in global
//     memory (each element belongs to one lane: no atomics, fixed order), and the slices are combined in f64 in a fixed
//     order by gram32_reduce_kernel: deterministic;
//   * operands are centred in f32 with an f32 shift, y = x - c32 (one rounding, 3e-8 of |y|); column sums of y in f64 beside
//     the diagonal blocks (the exact mean comes out of the same pass, as above).
constexpr int G32_D = 256, G32_TILES = 36, G32_OFF_TILES = 64;

struct Gram32Args {
    const float *x;
    uint64_t ldx, n;
    const float *shift32;    // centring vector c32 (f32), d entries
    double *colsum;          // [slices][d]
    double *diagfix;         // [slices][d]: sum_r r1^2 per column, the diagonal correction of the three-product form (nullptr: none)
    double *partial;         // [slices][tiles_per_slice][32][32]: S x 36 diagonal tiles, then S (S - 1) / 2 x 64 off-diagonal ones
    uint64_t rows_per_slice;
    uint32_t sub_rows;       // fold period, a multiple of the 32-row stage
    uint32_t d, S, tiles_per_slice;
};

__host__ __device__ constexpr int upper_tile_index(int tr, int tc) { return tr * 8 - tr * (tr - 1) / 2 + (tc - tr); }

// The block's partial starts as zeros (written by the lanes that own the elements) so that every fold is the same
// unconditional read-modify-write: a "first fold stores, later folds add" flag makes the compiler branch around each of the
// loads.  The lane offset is made opaque inside the fold, or the element addresses are computed ahead of the main loop and
// live across it in scratch; tile by tile behind sched_barrier, or all loads are batched ahead of the adds and spill.
template <int NT, class TileIndex>
__device__ __forceinline__ void gram32_fold(double *out, f16v (&acc)[NT], int i, int h, bool zero, TileIndex tile_index) {
    uint32_t lo = (uint32_t)((4 * h) * 32 + i) * 8u;
    asm volatile("" : "+v"(lo));
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        // 32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        char *p = reinterpret_cast<char *>(out) + tile_index(q) * 8192 + lo;
        if (zero) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) *reinterpret_cast<double *>(p + ((reg & 3) + 8 * (reg >> 2)) * 256) = 0.0;
        } else {
            double old[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) old[reg] = *reinterpret_cast<const double *>(p + ((reg & 3) + 8 * (reg >> 2)) * 256);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                *reinterpret_cast<double *>(p + ((reg & 3) + 8 * (reg >> 2)) * 256) = old[reg] + (double)acc[q][reg];
                acc[q][reg] = 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------
// the statistics from the bf16 matrix cores: f32-accurate Gram sums from split operands
// ---------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x16_bf16 runs at sixteen times the rate of the f32 MFMA.  Write y = y1 + r1 (y1 = bf16(y), r1 = y - y1 exact in
// f32, |r1| <= 2^-9 |y|) and y2 = bf16(r1), r2 = r1 - y2 (|r2| <= 2^-18 |y|).  Then for two columns a, b
//     y_a y_b = y1_a y1_b + y1_a y2_b + y2_a y1_b   +   r1_a r1_b   +   (y1_a r2_b + r2_a y1_b)   + O(2^-27)
// The first three products are exact in f32 on the bf16 matrix cores.  What the fourth and fifth terms do to a SUM over n rows:
//   * r1_a r1_b: 2^-20 of |y_a y_b| per row.  For a != b the rounding residuals of two columns are uncorrelated and the sum is a
//     random walk, 2^-20 / sqrt(n) of the entry; for a == b every term is positive — a systematic -6e-7 of each VARIANCE — so
//     sum_r r1_a^2 is accumulated beside the matrix cores (one FMA per element in the staging thread that owns the column, f32 over
//     the octet, f64 across) and added to the diagonal by the reducer (`diagfix`);
//   * the y1 r2 cross terms: 2^-18 of the entry per row with random sign, 2^-18 / sqrt(n) of the sum.
// So THREE bf16 MFMAs per product (LEAN) give the Gram of a LONG matrix to a few 1e-8 of the f64 one — at half the matrix-core
// work, two thirds of the split arithmetic and two thirds of the LDS traffic of the six-product form (1,1) (1,2) (2,1) (2,2) (1,3)
// (3,1) of the projection, which LEAN = false keeps for matrices of fewer than 2^20 rows (launch_gram32: the random-walk terms
// fall as 1 / sqrt(n)).
// For G = Y^T Y both MFMA operands are "column i, eight consecutive ROWS": lane (i, h) of a fragment holds rows 8h .. 8h+7 of
// column i of its 32-column block — the A fragment of column block a and the B fragment of column block b are the same kind of
// thing, and the one of a diagonal tile is one register set used twice.
//   * staging: a thread owns (column, row octet): eight dword loads down a column (a wave reads 256 contiguous bytes per row),
//     centred with the f32 shift, summed into the column sums (f32 over the octet, f64 across), split, and written as NS
//     16-byte fragment pieces — [k-step][split][column block][lane] x 16 B, lane-linear: ds_write_b128 / ds_read_b128 without
//     bank conflicts, no transposition anywhere;
//   * a block is 8 waves (2 per SIMD), one per CU, 32 rows per barrier, two stages in LDS;
//   * DIAGONAL block: the 36 upper tiles of a 256-column super-tile dealt to the 8 waves as 4 4 4 4 5 5 5 5 (G16_TR / G16_TC:
//     waves w and w + 4 share a SIMD: 9 tiles per SIMD, the matrix pipes are evenly loaded); OFF-DIAGONAL block (I < J, `half`):
//     128 columns of I against 256 of J, wave w owns tile row w >> 1 and four tile columns;
//   * accumulation: the matrix cores sum one 32-row stage from zero, the vector unit adds the stage sums (the bf16 MFMA's f32
//     accumulation is not round-to-nearest: see gram16_diag_body); f32 over <= 2048 rows, f64 across; deterministic.
constexpr int G16_KR = 32;                    // rows per stage
constexpr int G16_THREADS = 512;
constexpr int G16_TR[8][5] = {{0, 0, 0, 0, -1}, {0, 0, 0, 0, -1}, {1, 1, 1, 1, -1}, {1, 1, 1, 7, -1},
                              {2, 2, 2, 2, 2},  {2, 3, 3, 3, 3},  {3, 4, 4, 4, 4},  {5, 5, 5, 6, 6}};
constexpr int G16_TC[8][5] = {{0, 1, 2, 3, -1}, {4, 5, 6, 7, -1}, {1, 2, 3, 4, -1}, {5, 6, 7, 7, -1},
                              {2, 3, 4, 5, 6},  {7, 3, 4, 5, 6},  {7, 4, 5, 6, 7},  {5, 6, 7, 6, 7}};
__host__ __device__ constexpr int g16_ntiles(int w) { return w < 4 ? 4 : 5; }
__host__ __device__ constexpr bool g16_uses(int w, int blk) {
    for (int q = 0; q < g16_ntiles(w); ++q)
        if (G16_TR[w][q] == blk || G16_TC[w][q] == blk) return true;
    return false;
}
// byte offset of a fragment piece inside a stage: NS split pieces x NCB column blocks per k-step
template <int NS, int NCB>
__device__ __forceinline__ uint32_t g16_slot(int ks, int s, int cbk, int lane) { return (uint32_t)((((ks * NS + s) * NCB + cbk) * 64 + lane) * 16); }

// a wave-uniform address as an SGPR pair the compiler cannot fold back into per-lane 64-bit arithmetic: loads through it take
// the "scalar base + 32-bit lane offset" form of global_load
typedef const char __attribute__((address_space(1))) *gbytes;
typedef const float __attribute__((address_space(1))) *gfloat;
__device__ __forceinline__ float load_scalar_base(const float *row, uint32_t byte_offset) {
    const uint64_t v = reinterpret_cast<uint64_t>(row);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return *reinterpret_cast<gfloat>(reinterpret_cast<gbytes>(((uint64_t)hi << 32) | lo) + byte_offset);
}

// one staging task: eight rows of one column -> centred, summed, split, NS 16-byte pieces into the stage.
// Returns the octet's sum; LEAN: `resid2` += sum of r1^2 (the diagonal correction above).
template <bool LEAN, int NCB>
__device__ __forceinline__ float g16_stage_task(const float (&pre)[8], int nv, float sh, char *stage, int o, int cbk, int lane, float &resid2) {
    constexpr int NS = LEAN ? 2 : 3;
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = j < nv ? __fsub_rn(pre[j], sh) : 0.f;
    u32x4 q1, q2, q3;
    float rr = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        uint32_t p1, p2, p3;
        if constexpr (LEAN) {
            float ra, rb;
            split2_pair(y[2 * jj], y[2 * jj + 1], p1, p2, ra, rb);
            rr = __builtin_fmaf(ra, ra, rr);
            rr = __builtin_fmaf(rb, rb, rr);
            p3 = 0;
        } else {
            split3_pair(y[2 * jj], y[2 * jj + 1], p1, p2, p3);
        }
        q1[jj] = p1; q2[jj] = p2; q3[jj] = p3;
    }
    const int ks = o >> 1, ln = (o & 1) * 32 + (lane & 31);
    *reinterpret_cast<u32x4 *>(stage + g16_slot<NS, NCB>(ks, 0, cbk, ln)) = q1;
    *reinterpret_cast<u32x4 *>(stage + g16_slot<NS, NCB>(ks, 1, cbk, ln)) = q2;
    if constexpr (!LEAN) *reinterpret_cast<u32x4 *>(stage + g16_slot<NS, NCB>(ks, 2, cbk, ln)) = q3;
    resid2 += rr;
    return ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
}

#define G16_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
constexpr f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
// the products of one (tile, k-step): operand pieces (A, B); the first starts from zero at ks == 0
constexpr int G16_PA[6] = {0, 0, 1, 1, 0, 2}, G16_PB[6] = {0, 1, 0, 1, 2, 0};
constexpr int G16_LA[3] = {0, 0, 1}, G16_LB[3] = {0, 1, 0};

// acc += the sums of one staged 32-row stage over this wave's tiles of a diagonal block.
// The bf16 MFMA does not round its f32 accumulation to nearest: a long chain of same-sign terms (the variances on the
// diagonal) comes out LOW — -1.3e-6 relative after 2048 rows, uniformly (profiles/r03u_gram_error.txt), against -6e-9
// for 32-row chains.  So the matrix cores only ever sum ONE stage (32 rows) from zero, and the
// stage sums are added to the long accumulators by the vector unit (v_pk_add_f32: round to nearest even).
template <int W, bool LEAN>
__device__ __forceinline__ void g16_diag_stage_sums(const char *sb, f16v (&acc)[g16_ntiles(W)], int lane) {
    constexpr int NT = g16_ntiles(W), NCB = 8, NS = LEAN ? 2 : 3, NP = LEAN ? 3 : 6;
    f16v tmp[NT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 f[8][NS];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
            if (g16_uses(W, blk)) {
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) f[blk][sp] = *reinterpret_cast<const bf16x8 *>(sb + g16_slot<NS, NCB>(ks, sp, blk, lane));
            }
        // product-major: consecutive MFMAs go to different accumulators
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) {
            const int pa = LEAN ? G16_LA[pr] : G16_PA[pr], pb = LEAN ? G16_LB[pr] : G16_PB[pr];
#pragma unroll
            for (int q = 0; q < NT; ++q)
                tmp[q] = G16_MFMA(f[G16_TR[W][q]][pa], f[G16_TC[W][q]][pb], (ks == 0 && pr == 0) ? zero16 : tmp[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] += tmp[q];
}

template <int W, bool LEAN>
__device__ __forceinline__ void gram16_diag_body(const Gram32Args &a, char *lds, uint32_t sup) {
    constexpr int NT = g16_ntiles(W), NCB = 8, NS = LEAN ? 2 : 3, STAGE = 2 * NS * NCB * 1024;
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, h = lane >> 5;
    const uint64_t r_begin = (uint64_t)blockIdx.y * a.rows_per_slice;
    const uint64_t r_end = r_begin + a.rows_per_slice < a.n ? r_begin + a.rows_per_slice : a.n;
    const uint32_t cb = sup * G32_D;
    // loader role: tasks p = 2 W + u: column group p & 3 (64 columns), row octet p >> 2
    constexpr int CG0 = (2 * W) & 3, CG1 = (2 * W + 1) & 3, O0 = (2 * W) >> 2, O1 = (2 * W + 1) >> 2;
    const uint32_t col0 = cb + 64 * CG0 + lane, col1 = cb + 64 * CG1 + lane;
    const float sh0 = a.shift32[col0], sh1 = a.shift32[col1];
    f16v acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    double cs0 = 0.0, cs1 = 0.0, df0 = 0.0, df1 = 0.0;
    float pre0[8], pre1[8];
    int nv0 = 0, nv1 = 0;
    auto prefetch = [&](uint64_t row0) {
        const uint64_t ra = row0 + 8 * O0, rb = row0 + 8 * O1;
        if (row0 + G16_KR <= r_end) {
            // a whole stage (all but the last of a slice): one scalar row base per octet, the eight rows by additions — the
            // guarded form below spends a select and a 64-bit multiply on each of its sixteen loads (~185 scalar instructions
            // per wave and stage, on the one scalar unit the CU's eight waves share)
            // (scalar base + 32-bit byte offset of the lane's column: global_load with an SGPR base, no 64-bit vector arithmetic)
            static_assert(O0 == O1, "the two tasks of a wave read the same rows");
            nv0 = nv1 = 8;
            const float *rowp = a.x + ra * a.ldx;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pre0[j] = load_scalar_base(rowp, 4u * col0);
                pre1[j] = load_scalar_base(rowp, 4u * col1);
                rowp += a.ldx;
            }
            return;
        }
        nv0 = ra >= r_end ? 0 : (r_end - ra >= 8 ? 8 : (int)(r_end - ra));
        nv1 = rb >= r_end ? 0 : (r_end - rb >= 8 ? 8 : (int)(r_end - rb));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pre0[j] = a.x[(j < nv0 ? ra + j : r_begin) * a.ldx + col0];          // always a valid row
            pre1[j] = a.x[(j < nv1 ? rb + j : r_begin) * a.ldx + col1];
        }
    };
    auto stage = [&](int buf) {
        float r0 = 0.f, r1 = 0.f;
        cs0 += (double)g16_stage_task<LEAN, NCB>(pre0, nv0, sh0, lds + buf * STAGE, O0, 2 * CG0 + h, lane, r0);
        cs1 += (double)g16_stage_task<LEAN, NCB>(pre1, nv1, sh1, lds + buf * STAGE, O1, 2 * CG1 + h, lane, r1);
        if constexpr (LEAN) { df0 += (double)r0; df1 += (double)r1; }
    };
    double *const out = a.partial + ((uint64_t)blockIdx.y * a.tiles_per_slice + (uint64_t)sup * G32_TILES) * 1024;
    auto tile_index = [](int q) { return upper_tile_index(G16_TR[W][q], G16_TC[W][q]); };
    gram32_fold<NT>(out, acc, i, h, true, tile_index);

    if (r_begin < r_end) {
        prefetch(r_begin);
        stage(0);
        if (r_begin + G16_KR < r_end) prefetch(r_begin + G16_KR);
    }
    __syncthreads();
    int buf = 0;
    uint32_t in_sub = 0;
    // Waves w and w + 4 share a SIMD and leave every barrier together.  If both staged first and multiplied second, the SIMD's
    // vector unit would serve two staging phases (~110 instructions each) while its matrix pipe idles, then the matrix pipe two
    // MFMA phases while the vector unit idles.  So the lower four waves stage the next rows BEFORE their MFMAs and the upper four
    // AFTER: between two barriers every wave still does both, but a SIMD's two waves are in opposite phases.
    constexpr bool STAGE_FIRST = W < 4;
    auto stage_next = [&](uint64_t row0, int b) {
        if (row0 + G16_KR < r_end) {
            stage(b ^ 1);
            if (row0 + 2 * G16_KR < r_end) prefetch(row0 + 2 * G16_KR);
        }
    };
    for (uint64_t row0 = r_begin; row0 < r_end; row0 += G16_KR, buf ^= 1) {
        if constexpr (STAGE_FIRST) stage_next(row0, buf);
        g16_diag_stage_sums<W, LEAN>(lds + buf * STAGE, acc, lane);
        in_sub += G16_KR;
        if (in_sub >= a.sub_rows || row0 + G16_KR >= r_end) {
            gram32_fold<NT>(out, acc, i, h, false, tile_index);
            in_sub = 0;
        }
        if constexpr (!STAGE_FIRST) stage_next(row0, buf);
        __syncthreads();
    }

    // column sums: thread (wave, u) summed octet O_u of every stage for its column; the four octets of a column meet in LDS
    double *red = reinterpret_cast<double *>(lds);          // the loop ended on a barrier: the stages are free
    red[O0 * G32_D + 64 * CG0 + lane] = cs0;
    red[O1 * G32_D + 64 * CG1 + lane] = cs1;
    if constexpr (LEAN) {
        red[(4 + O0) * G32_D + 64 * CG0 + lane] = df0;
        red[(4 + O1) * G32_D + 64 * CG1 + lane] = df1;
    }
    __syncthreads();
    if (t < G32_D) {
        a.colsum[(uint64_t)blockIdx.y * a.d + cb + t] = ((red[t] + red[G32_D + t]) + red[2 * G32_D + t]) + red[3 * G32_D + t];
        if constexpr (LEAN)
            a.diagfix[(uint64_t)blockIdx.y * a.d + cb + t] = ((red[4 * G32_D + t] + red[5 * G32_D + t]) + red[6 * G32_D + t]) + red[7 * G32_D + t];
    }
}

// off-diagonal block: columns [256 I + 128 half, +128) (A: column blocks 0..3 of the stage) against [256 J, +256) (B: blocks 4..11);
// wave W owns tile row W >> 1 of the half and tile columns 4 (W & 1) .. +3
template <int W, bool LEAN>
__device__ __forceinline__ void gram16_off_body(const Gram32Args &a, char *lds, uint32_t I, uint32_t J, uint32_t half, uint32_t pair) {
    constexpr int NT = 4, NCB = 12, NS = LEAN ? 2 : 3, NP = LEAN ? 3 : 6, STAGE = 2 * NS * NCB * 1024, AR = W >> 1, B0 = 4 * (W & 1);
    const int t = threadIdx.x, lane = t & 63, i = lane & 31, h = lane >> 5;
    const uint64_t r_begin = (uint64_t)blockIdx.y * a.rows_per_slice;
    const uint64_t r_end = r_begin + a.rows_per_slice < a.n ? r_begin + a.rows_per_slice : a.n;
    // loader role: tasks p = 3 W + u (24 per stage): p < 8: A panel, column group p & 1, octet p >> 1; else B panel, (p - 8) & 3, (p - 8) >> 2
    uint32_t col[3];
    float sh[3];
    int oct[3], cbk[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int p = 3 * W + u;
        if (p < 8) {
            col[u] = I * G32_D + half * 128 + 64 * (p & 1) + lane;
            oct[u] = p >> 1;
            cbk[u] = 2 * (p & 1) + h;
        } else {
            col[u] = J * G32_D + 64 * ((p - 8) & 3) + lane;
            oct[u] = (p - 8) >> 2;
            cbk[u] = 4 + 2 * ((p - 8) & 3) + h;
        }
        sh[u] = a.shift32[col[u]];
    }
    f16v acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float pre[3][8];
    int nv[3] = {0, 0, 0};
    auto prefetch = [&](uint64_t row0) {
        if (row0 + G16_KR <= r_end) {                        // a whole stage: see gram16_diag_body
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                nv[u] = 8;
                const float *rowp = a.x + (row0 + 8 * oct[u]) * a.ldx;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pre[u][j] = load_scalar_base(rowp, 4u * col[u]);
                    rowp += a.ldx;
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const uint64_t r0 = row0 + 8 * oct[u];
            nv[u] = r0 >= r_end ? 0 : (r_end - r0 >= 8 ? 8 : (int)(r_end - r0));
#pragma unroll
            for (int j = 0; j < 8; ++j) pre[u][j] = a.x[(j < nv[u] ? r0 + j : r_begin) * a.ldx + col[u]];
        }
    };
    auto stage = [&](int buf) {
        float unused = 0.f;                                  // no diagonal entries in an off-diagonal block
#pragma unroll
        for (int u = 0; u < 3; ++u) (void)g16_stage_task<LEAN, NCB>(pre[u], nv[u], sh[u], lds + buf * STAGE, oct[u], cbk[u], lane, unused);
    };
    const uint32_t tile_base = a.S * G32_TILES + pair * G32_OFF_TILES + (4 * half + AR) * 8 + B0;
    double *const out = a.partial + ((uint64_t)blockIdx.y * a.tiles_per_slice + tile_base) * 1024;
    auto tile_index = [](int q) { return q; };
    gram32_fold<NT>(out, acc, i, h, true, tile_index);

    if (r_begin < r_end) {
        prefetch(r_begin);
        stage(0);
        if (r_begin + G16_KR < r_end) prefetch(r_begin + G16_KR);
    }
    __syncthreads();
    int buf = 0;
    uint32_t in_sub = 0;
    constexpr bool STAGE_FIRST = W < 4;                      // opposite phases for the two waves of a SIMD (see gram16_diag_body)
    auto stage_next = [&](uint64_t row0, int b) {
        if (row0 + G16_KR < r_end) {
            stage(b ^ 1);
            if (row0 + 2 * G16_KR < r_end) prefetch(row0 + 2 * G16_KR);
        }
    };
    for (uint64_t row0 = r_begin; row0 < r_end; row0 += G16_KR, buf ^= 1) {
        if constexpr (STAGE_FIRST) stage_next(row0, buf);
        const char *sb = lds + buf * STAGE;
        f16v tmp[NT];                                        // one stage's sums (see gram16_diag_body)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[NS], fb[4][NS];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) fa[sp] = *reinterpret_cast<const bf16x8 *>(sb + g16_slot<NS, NCB>(ks, sp, AR, lane));
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int sp = 0; sp < NS; ++sp) fb[q][sp] = *reinterpret_cast<const bf16x8 *>(sb + g16_slot<NS, NCB>(ks, sp, 4 + B0 + q, lane));
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const int pa = LEAN ? G16_LA[pr] : G16_PA[pr], pb = LEAN ? G16_LB[pr] : G16_PB[pr];
#pragma unroll
                for (int q = 0; q < NT; ++q) tmp[q] = G16_MFMA(fa[pa], fb[q][pb], (ks == 0 && pr == 0) ? zero16 : tmp[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] += tmp[q];
        in_sub += G16_KR;
        if (in_sub >= a.sub_rows || row0 + G16_KR >= r_end) {
            gram32_fold<NT>(out, acc, i, h, false, tile_index);
            in_sub = 0;
        }
        if constexpr (!STAGE_FIRST) stage_next(row0, buf);
        __syncthreads();
    }
}

// grid = (S + S (S - 1), slices): block x < S is the diagonal block of super-tile x; the others come in pairs per (I < J).
// 512 threads; dynamic LDS: two stages of NS x NCB KiB x 2 k-steps (LEAN: 64 KiB when S == 1, else 96 KiB)
template <bool LEAN>
__global__ __launch_bounds__(G16_THREADS, 1) void gram16_kernel(const Gram32Args a) {
    extern __shared__ __attribute__((aligned(16))) char g16_lds[];
    const uint32_t b = blockIdx.x;
    if (b < a.S) {
        switch (threadIdx.x >> 6) {                          // whole waves take each arm; every arm meets the same barriers
            case 0: gram16_diag_body<0, LEAN>(a, g16_lds, b); break;
            case 1: gram16_diag_body<1, LEAN>(a, g16_lds, b); break;
            case 2: gram16_diag_body<2, LEAN>(a, g16_lds, b); break;
            case 3: gram16_diag_body<3, LEAN>(a, g16_lds, b); break;
            case 4: gram16_diag_body<4, LEAN>(a, g16_lds, b); break;
            case 5: gram16_diag_body<5, LEAN>(a, g16_lds, b); break;
            case 6: gram16_diag_body<6, LEAN>(a, g16_lds, b); break;
            default: gram16_diag_body<7, LEAN>(a, g16_lds, b); break;
        }
        return;
    }
    const uint32_t pair = (b - a.S) >> 1, half = (b - a.S) & 1;
    uint32_t q = pair, rowlen = a.S - 1, I = 0;
    while (q >= rowlen) { q -= rowlen; ++I; --rowlen; }
    const uint32_t J = I + 1 + q;
    switch (threadIdx.x >> 6) {
        case 0: gram16_off_body<0, LEAN>(a, g16_lds, I, J, half, pair); break;
        case 1: gram16_off_body<1, LEAN>(a, g16_lds, I, J, half, pair); break;
        case 2: gram16_off_body<2, LEAN>(a, g16_lds, I, J, half, pair); break;
        case 3: gram16_off_body<3, LEAN>(a, g16_lds, I, J, half, pair); break;
        case 4: gram16_off_body<4, LEAN>(a, g16_lds, I, J, half, pair); break;
        case 5: gram16_off_body<5, LEAN>(a, g16_lds, I, J, half, pair); break;
        case 6: gram16_off_body<6, LEAN>(a, g16_lds, I, J, half, pair); break;
        default: gram16_off_body<7, LEAN>(a, g16_lds, I, J, half, pair); break;
    }
}

// the sampled shift as the f32 value the kernel centres with, and that same value in f64 for the exact correction
__global__ __launch_bounds__(256) void shift_round_kernel(double *__restrict__ shift64, float *__restrict__ shift32, uint32_t d) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const float s = (float)shift64[c];
    shift32[c] = s;
    shift64[c] = (double)s;
}

// gram = sum over slices (fixed order) - n delta delta^T, mirrored to the lower triangle.  grid = (4, tiles_per_slice)
__global__ __launch_bounds__(256) void gram32_reduce_kernel(const double *__restrict__ partial, uint32_t slices, uint32_t S,
                                                            uint32_t tiles_per_slice, uint32_t d,
                                                            const double *__restrict__ delta, double n_rows,
                                                            const double *__restrict__ diagfix, double *__restrict__ gram) {
    const uint32_t p = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;       // tile, element of the tile
    uint32_t gi, gj;
    bool mirror;
    if (p < S * G32_TILES) {
        const uint32_t sup = p / G32_TILES;
        uint32_t tr, tc;
        pair_to_tiles(p % G32_TILES, 8, tr, tc);
        gi = sup * G32_D + tr * 32 + e / 32;
        gj = sup * G32_D + tc * 32 + e % 32;
        // a diagonal tile is computed in full; its upper triangle is THE value of both (i, j) and (j, i): the split form sums the
        // products y1 y2 and y2 y1 of the two in different passes, and the result must be symmetric to the bit
        if (tr == tc && e / 32 > e % 32) return;
        mirror = gi != gj;
    } else {
        const uint32_t po = p - S * G32_TILES, pair = po / G32_OFF_TILES, tq = po % G32_OFF_TILES;
        uint32_t q = pair, rowlen = S - 1, I = 0;
        while (q >= rowlen) { q -= rowlen; ++I; --rowlen; }
        const uint32_t J = I + 1 + q;
        gi = I * G32_D + (tq / 8) * 32 + e / 32;
        gj = J * G32_D + (tq % 8) * 32 + e % 32;
        mirror = true;
    }
    double s = 0.0;
    for (uint32_t sl = 0; sl < slices; ++sl) s += partial[((uint64_t)sl * tiles_per_slice + p) * 1024 + e];
    if (diagfix && gi == gj)                                // the r1^2 term of the three-product form (see gram16_kernel)
        for (uint32_t sl = 0; sl < slices; ++sl) s += diagfix[(uint64_t)sl * d + gi];
    s -= n_rows * (delta[gi] * delta[gj]);                  // the product first: the same rounding for (i, j) and (j, i)
    gram[(uint64_t)gi * d + gj] = s;
    if (mirror) gram[(uint64_t)gj * d + gi] = s;
}

// Row slices per launch: the grid is sized to ONE resident round (2 blocks per CU) so there is no
// partial last round — with 1-3 block tiles per launch at d = 256 a generic "many blocks" grid left
// a third of the chip idle in its tail.  `group` = block tiles in the launch.
inline uint32_t gram_slices(uint64_t n, uint32_t group, int per_cu = 2) {
    static int cus = 0;
    if (!cus) {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        cus = c > 0 ? c : 256;
    }
    const int resident = (per_cu == 1 ? 1 : 2) * cus;
    uint64_t s = group ? (uint64_t)resident / group : 1;
    const uint64_t cap = (n + GKC - 1) / GKC;   // at least one chunk of rows per slice
    if (s > cap) s = cap;
    return (uint32_t)(s < 1 ? 1 : s);
}

struct GramPlan { uint32_t tiles, pairs, s_diag, s_off, s_max; };

inline GramPlan gram_plan(uint64_t n, uint32_t d, int per_cu = 2) {
    GramPlan p;
    p.tiles = (d + GT - 1) / GT;
    p.pairs = p.tiles * (p.tiles + 1) / 2;
    p.s_diag = gram_slices(n, p.tiles, per_cu);
    p.s_off = p.tiles > 1 ? gram_slices(n, p.pairs - p.tiles, per_cu) : 0;
    p.s_max = p.s_diag > p.s_off ? p.s_diag : p.s_off;
    return p;
}

// ---------------------------------------------------------------------------------------------
// projection, f32 matrix cores
// ---------------------------------------------------------------------------------------------
constexpr int PM = 128, PN = 128, PK = 32;
constexpr int PLA = PK + 1;  // A tile row stride (floats): odd -> conflict-free column reads

__global__ __launch_bounds__(256) void project_kernel(const ProjArgs a) {
    __shared__ __attribute__((aligned(16))) float As[PM][PLA];
    __shared__ __attribute__((aligned(16))) float Bs[PK][PN];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const uint64_t bid = CLEORA_LINEAR_BLOCK();
    if (bid >= a.n_blocks) return;
    const uint32_t bn = (uint32_t)(bid % a.nb_n);
    const uint64_t bm = bid / a.nb_n;
    const uint64_t m0 = bm * PM;
    const uint32_t n0 = bn * PN;

    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // loader roles
    const int ac4 = t & 7, ar = t >> 3;    // A: cols ac4*4.., rows ar + 32 i
    const int bc4 = t & 31, br = t >> 5;   // B: cols bc4*4.., rows br + 8 i
    float4 pa[4], pb[4], pa2[4];
    bool aok[4];
    float rs[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.rowscale) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t r = m0 + ar + 32 * i;
            rs[i] = r < a.n ? a.rowscale[r] : 1.f;
        }
    }
    auto prefetch = [&](uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t r = m0 + ar + 32 * i;
            aok[i] = r < a.n;
            pa[i] = load4(a.x + r * a.ldx, k0 + ac4 * 4, a.d, aok[i], a.w4x);
            if (a.x2) pa2[i] = load4(a.x2 + r * a.ldx2, k0 + ac4 * 4, a.d, aok[i], a.w4x);
            const uint32_t kr = k0 + br + 8 * i;
            pb[i] = load4(a.t + (uint64_t)kr * a.k, n0 + bc4 * 4, a.k, kr < a.d, a.w4t);
        }
    };
    auto stage = [&](uint32_t k0) {
        float mu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t c = k0 + ac4 * 4 + q;
            mu[q] = c < a.d ? a.mean[c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v[4] = {pa[i].x, pa[i].y, pa[i].z, pa[i].w};
            const float v2[4] = {pa2[i].x, pa2[i].y, pa2[i].z, pa2[i].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c = k0 + ac4 * 4 + q;
                // block = embeddings[i:end] - mean_f32   (f32)          (pycleora/__init__.py:161)
                float o = centre(v[q], mu[q], rs[i], a.rowscale != nullptr);
                if (a.x2) o = __fadd_rn(__fmul_rn(a.alpha, o), __fmul_rn(a.beta, __fsub_rn(v2[q], mu[q])));
                As[ar + 32 * i][ac4 * 4 + q] = (aok[i] && c < a.d) ? o : 0.f;
            }
            *reinterpret_cast<float4 *>(&Bs[br + 8 * i][bc4 * 4]) = pb[i];
        }
    };

    prefetch(0);
    for (uint32_t k0 = 0; k0 < a.d; k0 += PK) {
        stage(k0);
        __syncthreads();
        if (k0 + PK < a.d) prefetch(k0 + PK);
#pragma unroll
        for (int kk = 0; kk < PK; kk += 2) {
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = As[wr * 64 + i * 32 + (lane & 31)][kk + (lane >> 5)];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = Bs[kk + (lane >> 5)][wc * 64 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                // 32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                const uint64_t row = m0 + wr * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const uint32_t col = n0 + wc * 64 + j * 32 + (lane & 31);
                if (row < a.n && col < a.k) a.out[row * a.ldo + col] = acc[i][j][reg];
            }
}


// ---------------------------------------------------------------------------------------------
// projection, second form: f32-accurate products from the bf16 matrix cores
// ---------------------------------------------------------------------------------------------
// The f32 MFMA runs at the f32 VECTOR rate (157 TF), one sixteenth of the bf16 MFMA rate of the same chip.  An f32 value
// is exactly the sum of three bf16 values up to 2^-27 of its magnitude (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 -
// x2): 8 + 8 + 8 significand bits, every subtraction exact), and a product of two bf16 values is exact in f32.  So
//     x * t = x1 t1 + x1 t2 + x2 t1 + x2 t2 + x1 t3 + x3 t1   + O(2^-25 |x t|)
// — six bf16 MFMAs with f32 accumulation give the product to better than one f32 rounding (the three dropped terms are
// <= 2^-26 + 2^-26 + 2^-36 of |x t|), i.e. the same error class as the sgemm of pycleora/__init__.py:163 (measured against
// an f64 product in tests/test_gpu_whiten.py: not larger than the f32-MFMA kernel's own error), at 6/16 of its matrix-core
// time.  The kernel is then bound by reading X and writing the result (HBM), not by the matrix cores.
//
//   * block = 4 waves = 64 rows x 256 columns (one "pass" of output columns; wider k runs blockIdx.y passes); wave (wr, wc)
//     owns rows 32 wr .. +31 and columns 128 wc .. +127: four 32x32 accumulator tiles.  Persistent: 2 blocks per CU, each
//     walking row tiles blockIdx.x, + gridDim.x, ...; the loop is FLAT over (row tile, k-step), so the operands of the next
//     row tile are already in flight while the current one finishes.
//   * A (rows of X) goes straight from HBM into the fragment registers: lane (i, h) of k-step ks loads the two 16-byte
//     pieces [16 ks + 4 h, +4) and [16 ks + 8 + 4 h, +4) of row i (64 contiguous bytes per row and k-step, every byte of X
//     read once), RING k-steps ahead; it is centred (and blended) in f32 exactly like the other forms, then split.
//   * B (the transform) is split ONCE per call by pack_transform_split_kernel into fragment order with the same k mapping
//     — Tp[((pass*KS + ks)*3 + split)*8 + tile][lane] = 8 bf16: slot e <-> k = 16 ks + 8 (e >> 2) + 4 h + (e & 3) —
//     and streamed through a double-buffered 24 KiB LDS stage per k-step (plain loads one step ahead, ds_write_b128, one
//     barrier per step; fragments are lane-linear 1 KiB blocks: conflict-free ds_read_b128).
//   * per k-step and wave: 12 B fragments, 24 MFMAs (the six products of each of the four tiles; consecutive MFMAs go to
//     different accumulators).
//   * epilogue per row tile: optional row normalisation on the accumulators, stores.
// T (d x k row-major f32) -> split into three bf16 matrices, in fragment order, zero-padded to whole k-steps and passes.
__global__ __launch_bounds__(256) void pack_transform_split_kernel(const float *__restrict__ t, uint32_t d, uint32_t k,
                                                                   uint32_t ksteps, uint32_t passes, u32x4 *__restrict__ tp) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;     // one 16-byte unit
    if (idx >= (uint64_t)passes * ksteps * SKB) return;
    const uint32_t lane = (uint32_t)(idx & 63), j = (uint32_t)((idx >> 6) & 7), s = (uint32_t)((idx >> 9) % 3);
    const uint64_t step = idx / SKB;                                   // pass * ksteps + ks
    const uint32_t ks = (uint32_t)(step % ksteps), pass = (uint32_t)(step / ksteps);
    const uint32_t col = pass * SN + j * 32 + (lane & 31), h = lane >> 5;
    uint32_t w[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t e = 2 * m + q;
            const uint32_t kk = 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3);
            v[q] = (kk < d && col < k) ? t[(uint64_t)kk * k + col] : 0.f;
        }
        uint32_t p[3];
        split3_pair(v[0], v[1], p[0], p[1], p[2]);
        w[m] = p[s];
    }
    tp[idx] = (u32x4){w[0], w[1], w[2], w[3]};
}

// ---- the bounded-operand mode of the split form (round 6): two-way f16 splits, THREE products ---------------------------------
// Inside the whitened loop the projection's operand is bounded row by row (abi.hip: Z = A Y with normalised rows Y, |z| <= sum |a|):
// rows scaled by 2^e_r and T's columns by 2^st into f16's upper range are sums of two f16 values to 2^-22, and x t = x1 t1 + x1 t2 +
// x2 t1 (project_f16.hip has the argument and the d = 256 kernel with T resident in registers).  Here the same arithmetic for every other
// width: B stages of 16 KiB instead of 24, two A fragments instead of three, half the MFMAs; the scales are undone on the accumulators.
constexpr int SKB16 = 2 * 8 * 64;      // 16-byte units of packed T per (pass, k-step) in this mode: 16 KiB

__global__ __launch_bounds__(256) void rowinfo_f16_kernel(const float *__restrict__ rowscale, const float *__restrict__ rowbound, uint64_t n,
                                                          float2 *__restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float s = rowscale ? rowscale[r] : 1.f, b = rowbound ? rowbound[r] : 1.f;
    out[r] = make_float2(s, pf_row_scale(fabsf(b) + fabsf(s)));                 // |x - s mu| <= B + |s| (|mu| <= 1)
}
// per column of T (d x k): 2^st with max |t| 2^st in [2^13, 2^14) (colmul) and its reciprocal (colscale); one thread per column
__global__ __launch_bounds__(256) void colscale_f16_kernel(const float *__restrict__ t, uint32_t d, uint32_t k, float *__restrict__ colmul,
                                                           float *__restrict__ colscale) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k) return;
    float m = 0.f;
    for (uint32_t r = 0; r < d; ++r) m = fmaxf(m, fabsf(t[(uint64_t)r * k + c]));
    const float up = pf_row_scale(m);
    colmul[c] = up;
    colscale[c] = __uint_as_float(0x7F000000u - __float_as_uint(up));           // the reciprocal of a power of two by its exponent field
}
// T -> hi / lo f16 in fragment order: tp[((pass * KS + ks) * 2 + split) * 8 + tile][lane], slot e <-> k = 16 ks + 8 (e >> 2) + 4 h + (e & 3)
__global__ __launch_bounds__(256) void pack_transform_split_f16_kernel(const float *__restrict__ t, uint32_t d, uint32_t k, uint32_t ksteps,
                                                                       uint32_t passes, const float *__restrict__ colmul, u32x4 *__restrict__ tp) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;     // one 16-byte unit
    if (idx >= (uint64_t)passes * ksteps * SKB16) return;
    const uint32_t lane = (uint32_t)(idx & 63), j = (uint32_t)((idx >> 6) & 7), sp = (uint32_t)((idx >> 9) & 1);
    const uint64_t step = idx / SKB16;                                 // pass * ksteps + ks
    const uint32_t ks = (uint32_t)(step % ksteps), pass = (uint32_t)(step / ksteps);
    const uint32_t col = pass * SN + j * 32 + (lane & 31), h = lane >> 5;
    const float up = col < k ? colmul[col] : 0.f;
    uint32_t w[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        float v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t e = 2 * m + q;
            const uint32_t kk = 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3);
            v[q] = (kk < d && col < k) ? __fmul_rn(t[(uint64_t)kk * k + col], up) : 0.f;
        }
        uint32_t p1, p2;
        split2h_pair(v[0], v[1], p1, p2);
        w[m] = sp ? p2 : p1;
    }
    tp[idx] = (u32x4){w[0], w[1], w[2], w[3]};
}

// U = k-steps per trip of the flat loop (a divisor of ksteps, a multiple of 2 RING): the compiler copies the live part of the
// operand ring at the loop's back edge — behind a vmcnt(0) that drains the prefetch — so the back edge is taken as rarely
// as the shape allows (once per row tile at d = 256).
// RG = row groups of 32 rows per block: 2 (a 64-row tile, 4 waves, two blocks per CU) or 4 (a 128-row tile, 8 waves, one
// block per CU — for large n: the same two waves per SIMD, but one B stage feeds twice the MFMAs, so the L2 traffic of
// the B stream (60 GB per call at the C3 shape, 2.6 of 8.75 ms by the profiling builds) halves).
// Round 4: the two waves of a row group (column halves wc = 0 / 1) need the SAME A fragments.  Until round 3 each loaded,
// centred and split them for itself — the split is 4.5 vector instructions per element, ~60 of the ~90 non-MFMA instructions of a
// k-step, in a kernel that is bound by instruction issue (MFMA pipe busy 0.51: profiles/r03_whiten_pmc.json).  Now the waves take
// turns: the wave whose parity PAR matches k-step g + 1 loads, centres and splits it under the MFMAs of k-step g and leaves the
// three fragments in LDS (af[(g + 1) & 1][row group]: 3 KiB), both read them after the step's barrier.  Same values, same
// order of operations: results are bit-identical to the round-3 kernel.  RING counts a wave's OWN k-steps in flight (every second).
// PAR = wc ^ (row group >> 1), so that the two waves of a SIMD (w and w + 4) produce in opposite steps.
template <bool SCALED, bool BLEND, int RING, int U, int RG, int PAR, bool F16>
__device__ __forceinline__ void project_split_body(const ProjArgs &a, const u32x4 *__restrict__ tp, uint32_t ksteps, uint64_t tiles,
                                                   unsigned char *smem) {
    static_assert(U % 2 == 0 && (U / 2) % RING == 0, "ring slots must be compile-time functions of the unrolled step");
    static_assert(!F16 || (SCALED && !BLEND), "the bounded mode centres with the row's scale and takes no blend");
    constexpr int T = RG * 128;            // threads
    constexpr int NS = F16 ? 2 : 3;        // splits of an operand
    constexpr int SKBv = NS * 512;         // 16-byte units of a B stage (SKB / SKB16)
    constexpr int NB = SKBv / T;           // ... per thread (6 or 3; bounded mode 4 or 2)
    constexpr int SRT = RG * 32;           // rows per block tile
    u32x4 *const bs = reinterpret_cast<u32x4 *>(smem);                           // [2][SKB]
    u32x4 *const af = bs + 2 * SKBv;                                             // [2][RG][NS][64]: the A fragments of a k-step
    float *const mean_s = reinterpret_cast<float *>(af + 2 * RG * NS * 64);      // [16 ksteps]
    float *const red = mean_s + 16 * ksteps;                                     // [2 RG waves][32 rows]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1, i = lane & 31, h = lane >> 5;
    const uint32_t pass = blockIdx.y;
    const u32x4 *const tpp = tp + (uint64_t)pass * ksteps * SKBv;
    const uint64_t my_tiles = tiles > blockIdx.x ? (tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint64_t total = my_tiles * ksteps;                                    // a multiple of U (ksteps is)
    if (total == 0) return;

    for (uint32_t c = t; c < 16 * ksteps; c += T) mean_s[c] = c < a.d ? a.mean[c] : 0.f;
    // B of step 0 straight into buffer 0
#pragma unroll
    for (int u = 0; u < NB; ++u) bs[t + T * u] = tpp[t + T * u];

    // ---- operand ring: A of this wave's next RING own k-steps (k-steps PAR, PAR + 2, ...) -------------------------------------
    float4 ra[RING][2], rb[BLEND ? RING : 1][2];
    float rs[SCALED ? RING : 1];          // the row's scale travels with its operand slot (an unconditional 4-byte load per
                                          // k-step: a load behind a "first k-step of a tile" branch would cost the counted waits)
    float2 ri[F16 ? RING : 1];            // bounded mode: {s_r, 2^e_r} instead
    uint64_t ltile = blockIdx.x;          // row tile / k-step of the NEXT load
    uint32_t lks = PAR;
    auto row_of = [&](uint64_t tile) {
        const uint64_t r = tile * SRT + (uint64_t)(wr * 32 + i);
        return r < a.n ? r : a.n - 1;                                            // clamped: always a valid address
    };
    auto issue_a = [&](int slot) {
        const uint64_t r = row_of(ltile);
        const float *p = a.x + r * a.ldx + 16 * lks + 4 * h;
        ra[slot][0] = *reinterpret_cast<const float4 *>(p);
        ra[slot][1] = *reinterpret_cast<const float4 *>(p + 8);
        if constexpr (BLEND) {
            const float *p2 = a.x2 + r * a.ldx2 + 16 * lks + 4 * h;
            rb[slot][0] = *reinterpret_cast<const float4 *>(p2);
            rb[slot][1] = *reinterpret_cast<const float4 *>(p2 + 8);
        }
        if constexpr (F16) ri[slot] = a.rowinfo[r];
        else if constexpr (SCALED) rs[slot] = a.rowscale[r];
        lks += 2;                                                                // ksteps is even: the parity stays
        if (lks >= ksteps) { lks -= ksteps; ltile += gridDim.x; }
    };
    // centre (block = embeddings - mean_f32, pycleora/__init__.py:161), blend, split: slot -> three bf16x8 fragments
    auto split_a = [&](int slot, uint32_t ks, u32x4 (&as)[NS]) {
        const float4 m0 = *reinterpret_cast<const float4 *>(mean_s + 16 * ks + 4 * h);
        const float4 m1 = *reinterpret_cast<const float4 *>(mean_s + 16 * ks + 8 + 4 * h);
        const float xv[8] = {ra[slot][0].x, ra[slot][0].y, ra[slot][0].z, ra[slot][0].w, ra[slot][1].x, ra[slot][1].y, ra[slot][1].z, ra[slot][1].w};
        const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (F16) o[e] = __fmul_rn(centre(xv[e], mu[e], ri[slot].x, true), ri[slot].y);   // (x - s mu) 2^e: the scaling is exact
            else o[e] = centre(xv[e], mu[e], SCALED ? rs[slot] : 1.f, SCALED);
        }
        if constexpr (BLEND) {
            const int sb = BLEND ? slot : 0;
            const float x2v[8] = {rb[sb][0].x, rb[sb][0].y, rb[sb][0].z, rb[sb][0].w, rb[sb][1].x, rb[sb][1].y, rb[sb][1].z, rb[sb][1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __fadd_rn(__fmul_rn(a.alpha, o[e]), __fmul_rn(a.beta, __fsub_rn(x2v[e], mu[e])));
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if constexpr (F16) {
                uint32_t p1, p2;
                split2h_pair(o[2 * m], o[2 * m + 1], p1, p2);
                as[0][m] = p1; as[1][m] = p2;
            } else {
                uint32_t p1, p2, p3;
                split3_pair(o[2 * m], o[2 * m + 1], p1, p2, p3);
                as[0][m] = p1; as[1][m] = p2; as[2][m] = p3;
            }
        }
    };
    auto publish = [&](int buf, const u32x4 (&as)[NS]) {
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) af[((buf * RG + wr) * NS + sp) * 64 + lane] = as[sp];
    };
#pragma unroll
    for (int slot = 0; slot < RING; ++slot) issue_a(slot);                       // own k-steps PAR, PAR + 2, ...

    f16v acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;

    __syncthreads();                                                             // mean_s is in place
    if constexpr (PAR == 0) {                                                    // the fragments of k-step 0
        u32x4 first[NS];
        split_a(0, 0, first);
        publish(0, first);
    }
    __syncthreads();
    uint64_t tile = blockIdx.x;
    uint32_t ks0 = 0;
    for (uint64_t g0 = 0; g0 < total; g0 += U) {
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            constexpr int kRing = RING;
            const uint32_t ks = ks0 + uu;
            const int buf = uu & 1;                                              // U is even
            const bool produce = ((uu + 1) & 1) == PAR;                          // this wave makes the fragments of k-step g + 1
            // B of the next k-step: six coalesced 16-byte loads per thread, written to the other buffer at the end
            const uint32_t ksn = ks + 1 == ksteps ? 0 : ks + 1;
            u32x4 bst[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) bst[u] = tpp[(uint64_t)ksn * SKBv + t + T * u];
            // a step in which this wave does not produce refills the slot it emptied in the previous step (or the prologue), right
            // behind the B loads: the wait for B at the end of this step leaves exactly these loads in flight (vmcnt retires in
            // order); they are split three steps from now
            if (!produce) issue_a(((uu - PAR) / 2) % kRing);

            // A fragments of this k-step (made by this wave or its partner during the previous step), B fragments of this wave's
            // four tiles, all three splits
            u32x4 as_cur[NS];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp) as_cur[sp] = af[((buf * RG + wr) * NS + sp) * 64 + lane];
            u32x4 bf[NS][4];
#pragma unroll
            for (int sp = 0; sp < NS; ++sp)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bf[sp][jj] = bs[buf * SKBv + (sp * 8 + wc * 4 + jj) * 64 + lane];
            constexpr int NP = F16 ? 3 : 6;
            constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};      // (the first three: the bounded mode's products)
            u32x4 as_next[NS];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if constexpr (F16)
                        acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, as_cur[PA[q]]),
                                                                         __builtin_bit_cast(h8v, bf[PB[q]][jj]), acc[jj], 0, 0, 0);
                    else
                        acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, as_cur[PA[q]]),
                                                                          __builtin_bit_cast(bf16x8, bf[PB[q]][jj]), acc[jj], 0, 0, 0);
                }
                // the fragments of the NEXT k-step are made here, under this step's MFMAs (the matrix pipe runs them for
                // 32 cycles each; the ~50 VALU instructions of centre + split issue in their shadow)
                if (q == 0 && produce) split_a(((uu + 1 - PAR) / 2) % kRing, ksn, as_next);
            }
            if (produce) publish(buf ^ 1, as_next);

            // B of the next k-step into the other buffer (nobody reads it during this step).  Before the tile epilogue, not
            // after: behind that branch the compiler cannot count the epilogue's stores and drains everything (vmcnt(0)),
            // the operand ring included, at the end of every RING-th step.
#pragma unroll
            for (int u = 0; u < NB; ++u) bs[(buf ^ 1) * SKBv + t + T * u] = bst[u];

            if (uu == U - 1 && ks0 + U == ksteps) {
                // ---- end of a row tile: normalise (whole rows live in this block when there is one pass), store --------
                if constexpr (F16) {                                             // undo the column and row scales (powers of two: exact)
                    float csl[4], un[16];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const uint32_t col = pass * SN + wc * 128 + jj * 32 + i;
                        csl[jj] = col < a.k ? a.colscale[col] : 0.f;
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        uint64_t row = tile * SRT + (uint64_t)(wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
                        row = row < a.n ? row : a.n - 1;
                        un[reg] = __uint_as_float(0x7F000000u - __float_as_uint(a.rowinfo[row].y));
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) acc[jj][reg] = (acc[jj][reg] * csl[jj]) * un[reg];
                }
                if (a.norm) {
                    float pr[16];
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        float v = 0.f;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v += a.norm == 1 ? acc[jj][reg] * acc[jj][reg] : fabsf(acc[jj][reg]);
                        pr[reg] = v;
                    }
                    // Sum over the 32 lanes of a half-wave (the tile's 32 columns), 16 values per lane, on the vector unit alone (round 4:
                    // the butterfly of 80 ds_bpermute per wave kept the LDS pipe busy for ~2 500 cycles per tile while no MFMA ran):
                    // v_permlane16_swap exchanges the odd 16-lane row of one register with the even row of another — one swap + one
                    // add folds the two rows of values r and r + 8 together (even rows end up with r, odd rows with r + 8) —, then four
                    // DPP adds (lane ^ 1, lane ^ 2, mirror inside 8, mirror inside 16) complete the sum inside each 16-lane row.
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(pr[r]), "+v"(pr[r + 8]));
                        pr[r] += pr[r + 8];
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        pr[r] += dpp_move<0xB1>(pr[r]);                          // quad_perm [1, 0, 3, 2]
                        pr[r] += dpp_move<0x4E>(pr[r]);                          // quad_perm [2, 3, 0, 1]
                        pr[r] += dpp_move<0x141>(pr[r]);                         // row_half_mirror
                        pr[r] += dpp_move<0x140>(pr[r]);                         // row_mirror
                    }
                    float mine = 0.f;                                            // lane i with bit 3 clear publishes value (i & 7) + 8 (i >> 4)
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if ((i & 7) == r) mine = pr[r];
                    if (!(i & 8)) {
                        const int reg = (i & 7) + 8 * (i >> 4);
                        red[w * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h] = mine;
                    }
                    __syncthreads();
                    // one factor per lane — lane (i & 15) of half h takes value i & 15 of that half — instead of all sixteen in every
                    // lane (the IEEE sqrt and division are ~20 instructions each); the sixteen reach the lanes as scalars (v_readlane)
                    const int mreg = i & 15, mrl = (mreg & 3) + 8 * (mreg >> 2) + 4 * h;
                    const float msum = red[(wr * 2) * 32 + mrl] + red[(wr * 2 + 1) * 32 + mrl];       // column halves in order
                    // L2: v * (1 / max(sqrt(s), 1e-10)) like src/embedding.rs:98-102; L1: v / max(s, 1e-10) (pycleora/__init__.py:947-950)
                    const float mf = a.norm == 1 ? 1.0f / fmaxf(sqrtf(msum), 1e-10f) : fmaxf(msum, 1e-10f);
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float flo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mf), reg));
                        const float fhi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mf), 32 + reg));
                        const float f = h ? fhi : flo;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc[jj][reg] = a.norm == 1 ? acc[jj][reg] * f : acc[jj][reg] / f;
                    }
                }
                // 32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                if (tile * SRT + SRT <= a.n && (uint64_t)pass * SN + SN <= a.k) {
                    // a tile that lies inside the matrix (all but the last row tile / a ragged last pass): one 64-bit row base per
                    // lane, the sixteen rows by additions, the four column tiles at immediate offsets — the guarded form below
                    // spends a 64-bit multiply, a compare and a branch on each of its 64 stores (round 4: ~770 -> ~130 instructions
                    // per wave and tile, in the one phase of the kernel in which no wave issues MFMAs)
                    float *p0 = a.out + (tile * SRT + (uint64_t)(wr * 32 + 4 * h)) * a.ldo + (pass * SN + wc * 128 + i);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float *pg = p0 + (uint64_t)(8 * g) * a.ldo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) {
                                pg[jj * 32] = acc[jj][4 * g + r];
                                acc[jj][4 * g + r] = 0.f;
                            }
                            pg += a.ldo;
                        }
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const uint64_t row = tile * SRT + (uint64_t)(wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h);
                            const uint32_t col = pass * SN + wc * 128 + jj * 32 + i;
                            if (row < a.n && col < a.k) a.out[row * a.ldo + col] = acc[jj][reg];
                            acc[jj][reg] = 0.f;
                        }
                }
                tile += gridDim.x;
            }
            __syncthreads();
        }
        ks0 = ks0 + U == ksteps ? 0 : ks0 + U;
    }
}

template <bool SCALED, bool BLEND, int RING, int U, int RG = 2, bool F16 = false>
__global__ __launch_bounds__(RG * 128, RG == 2 ? 2 : 1) void project_split_kernel(const ProjArgs a, const u32x4 *__restrict__ tp,
                                                                                   uint32_t ksteps, uint64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int w = threadIdx.x >> 6;
    // whole waves take each arm; every arm meets the same barriers
    if ((((w >> 1) >> 1) ^ w) & 1) project_split_body<SCALED, BLEND, RING, U, RG, 1, F16>(a, tp, ksteps, tiles, smem);
    else project_split_body<SCALED, BLEND, RING, U, RG, 0, F16>(a, tp, ksteps, tiles, smem);
}

}  // namespace

constexpr uint64_t kLeanGramMinRows = 1ull << 20;
struct Gram32Plan { uint32_t S, blocks, tiles_per_slice, slices; };
inline Gram32Plan gram32_plan(uint64_t n, uint32_t d) {
    Gram32Plan p;
    p.S = d / G32_D;
    p.blocks = p.S + p.S * (p.S - 1);
    p.tiles_per_slice = p.S * G32_TILES + p.S * (p.S - 1) / 2 * G32_OFF_TILES;
    p.slices = gram_slices(n, p.blocks, 1);             // one 8-wave block per CU
    return p;
}

uint64_t gram_workspace(uint64_t n, uint32_t d) {
    const GramPlan p = gram_plan(n, d);
    // tile partials, per-slice column sums of the diagonal blocks, delta
    uint64_t need = (uint64_t)p.s_max * p.pairs * GT * GT + (uint64_t)p.s_diag * p.tiles * GT + (uint64_t)p.tiles * GT;
    if (d % G32_D == 0 && d <= 2048) {   // the split form: [slices][tiles][32][32] partials, [slices][d] column sums and diagonal corrections, delta
        const Gram32Plan q = gram32_plan(n, d);
        const uint64_t need32 = (uint64_t)q.slices * ((uint64_t)q.tiles_per_slice * 1024 + 2 * (uint64_t)d) + d;
        if (need32 > need) need = need32;
    }
    return need;
}

bool gram32_applies(const float *x, uint64_t ldx, uint64_t n, uint32_t d) {
    return d % G32_D == 0 && d <= 2048 && ldx % 4 == 0 && aligned16(x) && n >= 4096;
}

// One-pass centred Gram from the bf16 matrix cores (d a multiple of 256; see gram16_kernel).  shift64 / shift32: the sampled shift,
// rounded to f32 here (both are updated: the kernel centres with the f32 value, the exact correction uses that same value).
int launch_gram32(const float *x, uint64_t ldx, uint64_t n, uint32_t d, double *shift64, float *shift32, double *ws, double *gram,
                  hipStream_t stream, double *mean_out64, float *mean_out32) {
    CL_REQUIRE(x != nullptr && shift64 != nullptr && shift32 != nullptr && ws != nullptr && gram != nullptr && mean_out64 != nullptr &&
               mean_out32 != nullptr, "x / shift / workspace / gram / mean is NULL");
    CL_REQUIRE(gram32_applies(x, ldx, n, d), "internal: the split Gram does not apply to this shape");
    const Gram32Plan q = gram32_plan(n, d);             // the same plan gram_workspace() sized the workspace for
    CL_REQUIRE(q.slices <= 65535, "internal: too many Gram slices");
    // Three products from kLeanGramMinRows rows on: the terms they drop are random-signed per row, so their sum falls as
    // 1 / sqrt(n) relative to the Gram — ||G3 - G64||_F / ||G64||_F = sqrt(d) 2^-18 / sqrt(n), measured 8.0e-7 / 2.2e-7 / 1.05e-7 /
    // 1.9e-8 at n = 5 003 / 70 001 / 300 007 / 10 M (d = 256) against 8e-9 for the six-product form at 10 M (profiles/r04_gram_forms.txt).
    // From 2^20 rows that is <= 6e-8 at d = 256 (the class of the f32 matrix cores' 4.4e-8, round 3) for 3.5 instead of 4.7 ms at
    // the C3 shape; shorter matrices keep the six products — their Gram is a fraction of a millisecond either way.
    const bool six = n < kLeanGramMinRows;
    Gram32Args a{};
    a.x = x;
    a.ldx = ldx;
    a.n = n;
    a.d = d;
    a.S = q.S;
    a.tiles_per_slice = q.tiles_per_slice;
    a.partial = ws;
    a.colsum = ws + (uint64_t)q.slices * q.tiles_per_slice * 1024;
    a.diagfix = a.colsum + (uint64_t)q.slices * d;
    double *delta = a.diagfix + (uint64_t)q.slices * d;
    uint64_t rps = (n + q.slices - 1) / q.slices;
    a.rows_per_slice = (rps + G16_KR - 1) / G16_KR * G16_KR;           // whole stages
    a.sub_rows = 2048;
    hipLaunchKernelGGL(shift_round_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, shift64, shift32, d);
    // mean_out32 may be the buffer that holds shift32: the kernel reads it before gram_mean_kernel (same stream) rewrites it
    a.shift32 = shift32;
    auto go = [&](auto kernel, size_t lds) {
        // per DEVICE and cheap: set on every call (a process-wide "done" flag broke the second GPU of a multi-device host)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) hipLaunchKernelGGL(kernel, dim3(q.blocks, q.slices), dim3(G16_THREADS), lds, stream, a);
        return e;
    };
    const size_t ncb = q.S == 1 ? 8 : 12;
    if (six) CL_HIP(go(gram16_kernel<false>, 2 * 2 * 3 * ncb * 1024));
    else CL_HIP(go(gram16_kernel<true>, 2 * 2 * 2 * ncb * 1024));
    hipLaunchKernelGGL(gram_mean_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, a.colsum, q.slices, d / GT, d, n, shift64, delta,
                       mean_out64, mean_out32);
    hipLaunchKernelGGL(gram32_reduce_kernel, dim3(4, q.tiles_per_slice), dim3(256), 0, stream, ws, q.slices, q.S, q.tiles_per_slice, d,
                       delta, (double)n, six ? nullptr : a.diagfix, gram);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

// mean_out64 / mean_out32 == nullptr: `mean` is the exact mean (two-pass form, pycleora/__init__.py:136-143 literally).
// Otherwise `mean` is a shift near the mean; the exact mean comes out of the same pass over X (gram_mean_kernel).
// blocks_per_cu: 2 fills the chip (2 waves per SIMD, ~206 registers each); 1 leaves half of every register file to a
// kernel running beside it (the SpMM of the overlapped whitened loop needs its occupancy to keep HBM busy).
int launch_gram(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const double *mean,
                double *ws, double *gram, hipStream_t stream, double *mean_out64, float *mean_out32, int blocks_per_cu) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && mean != nullptr && ws != nullptr && gram != nullptr,
               "x / mean / workspace / gram is NULL");
    CL_REQUIRE((mean_out64 == nullptr) == (mean_out32 == nullptr), "mean outputs come in pairs");
    const GramPlan p = gram_plan(n, d, blocks_per_cu == 1 ? 1 : 2);     // (the workspace is sized for 2: enough for fewer)
    GramArgs a{};
    a.colsum = ws + (uint64_t)p.s_max * p.pairs * GT * GT;
    double *delta = a.colsum + (uint64_t)p.s_diag * p.tiles * GT;
    a.x = x;
    a.ldx = ldx;
    a.n = n;
    a.d = d;
    a.mean = mean;
    a.partial = ws;
    a.tiles = p.tiles;
    a.pairs = p.pairs;
    a.w4 = (d % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
    auto rows_per_slice = [&](uint32_t slices) {
        uint64_t rps = (n + slices - 1) / slices;
        rps = (rps + GKC - 1) / GKC * GKC;
        return rps ? rps : (uint64_t)GKC;
    };
    CL_REQUIRE(p.s_max <= 65535, "internal: too many Gram slices");
    // quadrants that are never computed are skipped by the reducer, so no memset is needed
    const bool fast = a.w4 && d % GT == 0 && n > 0;
    a.rows_per_slice = rows_per_slice(p.s_diag);
    if (fast) hipLaunchKernelGGL((gram_kernel<true, true>), dim3(p.tiles, p.s_diag), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gram_kernel<true, false>), dim3(p.tiles, p.s_diag), dim3(256), 0, stream, a);
    if (p.tiles > 1) {
        a.rows_per_slice = rows_per_slice(p.s_off);
        if (fast) hipLaunchKernelGGL((gram_kernel<false, true>), dim3(p.pairs - p.tiles, p.s_off), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((gram_kernel<false, false>), dim3(p.pairs - p.tiles, p.s_off), dim3(256), 0, stream, a);
    }
    if (mean_out64)
        hipLaunchKernelGGL(gram_mean_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, a.colsum, p.s_diag, p.tiles, d, n,
                           mean, delta, mean_out64, mean_out32);
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(GT * GT / 256, p.pairs), dim3(256), 0, stream, ws,
                       p.s_diag, p.s_off, p.pairs, p.tiles, d, mean_out64 ? delta : nullptr, (double)n, gram);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

int launch_project(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *mean,
                   const float *t, uint32_t k, float *out, uint64_t ldo, hipStream_t stream,
                   const float *rowscale, const float *x2, uint64_t ldx2, float alpha, float beta, int norm,
                   bool *norm_done, const float *rowbound, bool *bounded_form) {
    if (norm_done) *norm_done = false;
    if (bounded_form) *bounded_form = false;
    CL_REQUIRE(d > 0 && k > 0 && ldx >= d && ldo >= k, "bad d / k / leading dimension");
    CL_REQUIRE(x != nullptr && mean != nullptr && t != nullptr && out != nullptr,
               "x / mean / transform / out is NULL");
    CL_REQUIRE((const void *)x != (const void *)out && (const void *)x2 != (const void *)out, "x and out must not alias");
    CL_REQUIRE(x2 == nullptr || ldx2 >= d, "bad leading dimension of the second operand");
    if (n == 0) return CLEORA_OK;
    ProjArgs a{};
    a.rowscale = rowscale;
    a.x2 = x2;
    a.ldx2 = ldx2;
    a.alpha = x2 ? alpha : 1.0f;
    a.beta = beta;
    a.x = x;
    a.ldx = ldx;
    a.n = n;
    a.d = d;
    a.mean = mean;
    a.t = t;
    a.k = k;
    a.out = out;
    a.ldo = ldo;
    a.nb_n = (k + PN - 1) / PN;
    a.w4x = (d % 4 == 0) && (ldx % 4 == 0) && aligned16(x) && (!x2 || (ldx2 % 4 == 0 && aligned16(x2)));
    a.w4t = (k % 4 == 0) && aligned16(t);
    // split-bf16 form: any d that is a multiple of 32 whose mean fits the block's LDS beside the two B stages, any k.
    // Everything else (d % 32 != 0, unaligned rows, d beyond ~28k) takes the tiled f32-MFMA kernel below.
    const uint32_t ksteps = d / 16, passes = (k + SN - 1) / SN;
    // two B stages, the A fragments of two k-steps for up to four row groups, the mean, the row-norm partials
    const size_t lds_bytes = (size_t)2 * SKB * 16 + (size_t)2 * 4 * 3 * 1024 + (size_t)16 * ksteps * sizeof(float) + 8 * 32 * sizeof(float);
    if (a.w4x && d % 32 == 0 && lds_bytes <= 160 * 1024) {
        static int cus = 0;
        if (!cus) {
            int dev = 0, c = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
            cus = c > 0 ? c : 256;
        }
        a.norm = (norm && passes == 1) ? norm : 0;                            // whole rows inside one block only
        if (norm_done) *norm_done = a.norm != 0;
        const bool scaled = rowscale != nullptr, blend = x2 != nullptr;
        if (rowbound != nullptr && !blend) {
            // ---- the bounded-operand mode: |x[r][j]| <= rowbound[r], |mean[j]| <= 1 (the caller's promise): three f16 products ----
            const uint64_t units16 = (uint64_t)passes * ksteps * SKB16;
            const size_t b_tp = units16 * sizeof(u32x4), b_ri = ((n * sizeof(float2) + 15) / 16) * 16, b_k = (((size_t)k * sizeof(float) + 15) / 16) * 16;
            char *blk = nullptr;
            CL_HIP(hipMallocAsync(reinterpret_cast<void **>(&blk), b_tp + b_ri + 2 * b_k, stream));
            u32x4 *tp16 = reinterpret_cast<u32x4 *>(blk);
            float2 *rinfo = reinterpret_cast<float2 *>(blk + b_tp);
            float *colmul = reinterpret_cast<float *>(blk + b_tp + b_ri), *colscale = reinterpret_cast<float *>(blk + b_tp + b_ri + b_k);
            hipLaunchKernelGGL(rowinfo_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, rowscale, rowbound, n, rinfo);
            hipLaunchKernelGGL(colscale_f16_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, t, d, k, colmul, colscale);
            hipLaunchKernelGGL(pack_transform_split_f16_kernel, dim3((unsigned)((units16 + 255) / 256)), dim3(256), 0, stream, t, d, k, ksteps,
                               passes, colmul, tp16);
            a.rowinfo = rinfo;
            a.colscale = colscale;
            const bool wide16 = (n + 127) / 128 >= (uint64_t)cus;
            const uint64_t tiles16 = wide16 ? (n + 127) / 128 : (n + SR - 1) / SR;
            const uint64_t resident16 = wide16 ? (uint64_t)cus : 2ull * (uint64_t)cus;
            const dim3 grid16((unsigned)(tiles16 < resident16 ? tiles16 : resident16), passes);
            hipError_t e16 = hipSuccess;
            auto go16 = [&](auto RGt) {
                constexpr int RG = decltype(RGt)::value;
#define CLEORA_SPLIT16_LAUNCH(RING, UU)                                                                                               \
                do {                                                                                                                   \
                    e16 = hipFuncSetAttribute(reinterpret_cast<const void *>(project_split_kernel<true, false, RING, UU, RG, true>),   \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                            \
                    if (e16 == hipSuccess)                                                                                             \
                        hipLaunchKernelGGL((project_split_kernel<true, false, RING, UU, RG, true>), grid16, dim3(RG * 128), lds_bytes, \
                                           stream, a, tp16, ksteps, tiles16);                                                          \
                } while (0)
                if (ksteps % 16 == 0) CLEORA_SPLIT16_LAUNCH(2, 16);
                else if (ksteps % 4 == 0) CLEORA_SPLIT16_LAUNCH(2, 4);
                else CLEORA_SPLIT16_LAUNCH(1, 2);
#undef CLEORA_SPLIT16_LAUNCH
            };
            if (wide16) go16(std::integral_constant<int, 4>{});
            else go16(std::integral_constant<int, 2>{});
            const hipError_t le16 = e16 != hipSuccess ? e16 : hipGetLastError();
            CL_HIP(hipFreeAsync(blk, stream));
            CL_HIP(le16);
            if (bounded_form) *bounded_form = true;
            return CLEORA_OK;
        }
        const uint64_t units = (uint64_t)passes * ksteps * SKB;
        u32x4 *tp = nullptr;
        CL_HIP(hipMallocAsync(reinterpret_cast<void **>(&tp), units * sizeof(u32x4), stream));
        hipLaunchKernelGGL(pack_transform_split_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, stream, t, d, k,
                           ksteps, passes, tp);
        // large n: 128-row tiles, one 8-wave block per CU (RG = 4: half the B-stage traffic per MFMA); otherwise 64-row tiles,
        // two 4-wave blocks per CU
        const bool wide = (n + 127) / 128 >= (uint64_t)cus;
        const uint64_t tiles = wide ? (n + 127) / 128 : (n + SR - 1) / SR;
        const uint64_t resident = wide ? (uint64_t)cus : 2ull * (uint64_t)cus;
        const unsigned gx = (unsigned)(tiles < resident ? tiles : resident);
        const dim3 grid(gx, passes);
        hipError_t attr_err = hipSuccess;
        auto launch_shape = [&](auto SCt, auto BLt, auto RGt) {
            constexpr bool SC = decltype(SCt)::value, BL = decltype(BLt)::value;
            constexpr int RG = decltype(RGt)::value;
            // the mean of a wide row pushes the block past the 64 KiB a kernel gets without asking (d >= 3872): raise the limit for
            // the instantiation being launched — per DEVICE and cheap, so on every call
#define CLEORA_SPLIT_LAUNCH(RING, UU)                                                                                                  \
            do {                                                                                                                       \
                attr_err = hipFuncSetAttribute(reinterpret_cast<const void *>(project_split_kernel<SC, BL, RING, UU, RG>),            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                           \
                if (attr_err == hipSuccess)                                                                                            \
                    hipLaunchKernelGGL((project_split_kernel<SC, BL, RING, UU, RG>), grid, dim3(RG * 128), lds_bytes, stream, a, tp,   \
                                       ksteps, tiles);                                                                                 \
            } while (0)
            // RING = a wave's own k-steps in flight (every second k-step is its own): 2 = four k-steps ahead; the blended operand
            // doubles the registers of a slot: 1 there.  U / 2 must be a multiple of RING.
            if constexpr (!BL) {
                if (ksteps % 16 == 0) { CLEORA_SPLIT_LAUNCH(2, 16); return; }
                if (ksteps % 4 == 0) { CLEORA_SPLIT_LAUNCH(2, 4); return; }
            }
            if (ksteps % 8 == 0) CLEORA_SPLIT_LAUNCH(1, 8);
            else CLEORA_SPLIT_LAUNCH(1, 2);
#undef CLEORA_SPLIT_LAUNCH
        };
        auto launch_rg = [&](auto RGt) {
            if (blend) {
                if (scaled) launch_shape(std::true_type{}, std::true_type{}, RGt); else launch_shape(std::false_type{}, std::true_type{}, RGt);
            } else {
                if (scaled) launch_shape(std::true_type{}, std::false_type{}, RGt); else launch_shape(std::false_type{}, std::false_type{}, RGt);
            }
        };
        if (wide) launch_rg(std::integral_constant<int, 4>{});
        else launch_rg(std::integral_constant<int, 2>{});
        const hipError_t le = attr_err != hipSuccess ? attr_err : hipGetLastError();
        CL_HIP(hipFreeAsync(tp, stream));
        CL_HIP(le);
        return CLEORA_OK;
    }
    const uint64_t blocks = ((n + PM - 1) / PM) * a.nb_n;
    a.n_blocks = blocks;
    hipLaunchKernelGGL(project_kernel, grid_1d_as_2d(blocks), dim3(256), 0, stream, a);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
