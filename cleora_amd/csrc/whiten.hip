// whiten.hip — the two matrix-core kernels of the PCA-whitening step
// (pycleora/__init__.py:130-164 `whiten_embeddings`), for gfx950.
//
//   centered Gram   G = sum_r (x_r - mu)(x_r - mu)^T  in f64     (:138-143, without the 1/(n-1))
//     v_mfma_f64_16x16x4_f64; operands are centred and widened to f64 when the X tile is staged
//     into LDS, so the accumulation is f64 end to end like the reference's `block.T @ block`
//     on an f64 block.  Only block tiles on or above the diagonal are computed, and of a diagonal
//     block tile only its 36 upper 16x16 MFMA tiles (9 per wave); the row range is cut into slices
//     that are combined in a fixed order (deterministic).
//   projection      out = (X - mu_f32) @ T  in f32                 (:157-163)
//     v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate — the same arithmetic class
//     as the reference's sgemm; summation order differs, tolerance documented in the tests).
//
// Both are MFMA-bound (2 n d^2 flops against 1-3 passes over X), unlike the SpMM.
#include "common.h"

namespace cleora {
namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
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
    const double *mean;
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

// Tiles of a DIAGONAL block tile: only the 36 MFMA tiles (ti <= tj) of its 8 x 8 grid are needed
// (the rest is the mirror image).  They are dealt 9 per wave; (ti, tj) = kDiagTiles[wave][t].
__constant__ unsigned char kDiagTiles[4][9][2] = {
    {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {0, 7}, {1, 1}},
    {{1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6}, {1, 7}, {2, 2}, {2, 3}, {2, 4}},
    {{2, 5}, {2, 6}, {2, 7}, {3, 3}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {4, 4}},
    {{4, 5}, {4, 6}, {4, 7}, {5, 5}, {5, 6}, {5, 7}, {6, 6}, {6, 7}, {7, 7}}};

template <bool DIAG>
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
            fa_col[i] = kDiagTiles[w][i][0] * 16;
            fb_col[i] = kDiagTiles[w][i][1] * 16;
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
    auto prefetch = [&](uint64_t row0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint64_t r = row0 + lr + 8 * h;
            ok[h] = r < r_end;
            const float *rp = a.x + r * a.ldx;
            pa[h] = load4(rp, colA, a.d, ok[h], a.w4);
            if (!DIAG) pb[h] = load4(rp, colB, a.d, ok[h], a.w4);
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
                da[q] = (ok[h] && colA + q < a.d) ? (double)va[q] - mA[q] : 0.0;
                if constexpr (!DIAG) db[q] = (ok[h] && colB + q < a.d) ? (double)vb[q] - mB[q] : 0.0;
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
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const double fa = lds[buf][0][krow][fa_col[i] + (lane & 15)];
                    const double fb = lds[buf][0][krow][fb_col[i] + (lane & 15)];
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa, fb, acc[i], 0, 0, 0);
                }
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
template <bool DIAG>
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
    gram_body<DIAG>(a, lds, bi, bj, pair_index(bi, bj, a.tiles));
}

// gram[gi][gj] = sum over slices (fixed order) of the partial tiles; mirrors the upper triangle.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ partial,
                                                          uint32_t s_diag, uint32_t s_off,
                                                          uint32_t pairs, uint32_t tiles, uint32_t d,
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
    gram[(uint64_t)gi * d + gj] = s;
    if (!diag || (r / 16) < (c / 16)) gram[(uint64_t)gj * d + gi] = s;
}

// Row slices per launch: the grid is sized to ONE resident round (2 blocks per CU) so there is no
// partial last round — with 1-3 block tiles per launch at d = 256 a generic "many blocks" grid left
// a third of the chip idle in its tail.  `group` = block tiles in the launch.
inline uint32_t gram_slices(uint64_t n, uint32_t group) {
    static int resident = 0;
    if (!resident) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        resident = 2 * (cus > 0 ? cus : 256);
    }
    uint64_t s = group ? (uint64_t)resident / group : 1;
    const uint64_t cap = (n + GKC - 1) / GKC;   // at least one chunk of rows per slice
    if (s > cap) s = cap;
    return (uint32_t)(s < 1 ? 1 : s);
}

struct GramPlan { uint32_t tiles, pairs, s_diag, s_off, s_max; };

inline GramPlan gram_plan(uint64_t n, uint32_t d) {
    GramPlan p;
    p.tiles = (d + GT - 1) / GT;
    p.pairs = p.tiles * (p.tiles + 1) / 2;
    p.s_diag = gram_slices(n, p.tiles);
    p.s_off = p.tiles > 1 ? gram_slices(n, p.pairs - p.tiles) : 0;
    p.s_max = p.s_diag > p.s_off ? p.s_diag : p.s_off;
    return p;
}

// ---------------------------------------------------------------------------------------------
// projection, f32 matrix cores
// ---------------------------------------------------------------------------------------------
constexpr int PM = 128, PN = 128, PK = 32;
constexpr int PLA = PK + 1;  // A tile row stride (floats): odd -> conflict-free column reads

struct ProjArgs {
    const float *x;
    uint64_t ldx;
    uint64_t n;
    uint32_t d;
    const float *mean;
    const float *t;   // d x k row-major
    uint32_t k;
    float *out;
    uint64_t ldo;
    uint32_t nb_n;    // column blocks
    uint64_t n_blocks;
    int w4x, w4t;
};

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
    float4 pa[4], pb[4];
    bool aok[4];
    auto prefetch = [&](uint32_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t r = m0 + ar + 32 * i;
            aok[i] = r < a.n;
            pa[i] = load4(a.x + r * a.ldx, k0 + ac4 * 4, a.d, aok[i], a.w4x);
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
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c = k0 + ac4 * 4 + q;
                // block = embeddings[i:end] - mean_f32   (f32)          (pycleora/__init__.py:161)
                As[ar + 32 * i][ac4 * 4 + q] = (aok[i] && c < a.d) ? __fsub_rn(v[q], mu[q]) : 0.f;
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

}  // namespace

uint64_t gram_workspace(uint64_t n, uint32_t d) {
    const GramPlan p = gram_plan(n, d);
    return (uint64_t)p.s_max * p.pairs * GT * GT;
}

int launch_gram(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const double *mean,
                double *ws, double *gram, hipStream_t stream) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && mean != nullptr && ws != nullptr && gram != nullptr,
               "x / mean / workspace / gram is NULL");
    const GramPlan p = gram_plan(n, d);
    GramArgs a{};
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
    a.rows_per_slice = rows_per_slice(p.s_diag);
    hipLaunchKernelGGL(gram_kernel<true>, dim3(p.tiles, p.s_diag), dim3(256), 0, stream, a);
    if (p.tiles > 1) {
        a.rows_per_slice = rows_per_slice(p.s_off);
        hipLaunchKernelGGL(gram_kernel<false>, dim3(p.pairs - p.tiles, p.s_off), dim3(256), 0, stream, a);
    }
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(GT * GT / 256, p.pairs), dim3(256), 0, stream, ws,
                       p.s_diag, p.s_off, p.pairs, p.tiles, d, gram);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

int launch_project(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *mean,
                   const float *t, uint32_t k, float *out, uint64_t ldo, hipStream_t stream) {
    CL_REQUIRE(d > 0 && k > 0 && ldx >= d && ldo >= k, "bad d / k / leading dimension");
    CL_REQUIRE(x != nullptr && mean != nullptr && t != nullptr && out != nullptr,
               "x / mean / transform / out is NULL");
    CL_REQUIRE((const void *)x != (const void *)out, "x and out must not alias");
    if (n == 0) return CLEORA_OK;
    ProjArgs a{};
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
    a.w4x = (d % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
    a.w4t = (k % 4 == 0) && aligned16(t);
    const uint64_t blocks = ((n + PM - 1) / PM) * a.nb_n;
    a.n_blocks = blocks;
    hipLaunchKernelGGL(project_kernel, grid_1d_as_2d(blocks), dim3(256), 0, stream, a);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
