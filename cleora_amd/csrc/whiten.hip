// whiten.hip — the two matrix-core kernels of the PCA-whitening step
// (pycleora/__init__.py:130-164 `whiten_embeddings`), for gfx950.
//
//   centered Gram   G = sum_r (x_r - mu)(x_r - mu)^T  in f64     (:138-143, without the 1/(n-1))
//     v_mfma_f64_16x16x4_f64; operands are centred and widened to f64 when the X tile is staged
//     into LDS, so the accumulation is f64 end to end like the reference's `block.T @ block`
//     on an f64 block.  Only block tiles on or above the diagonal are computed; the row range is
//     cut into slices that are combined in a fixed order (deterministic).
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

__global__ __launch_bounds__(256) void gram_kernel(const GramArgs a) {
    __shared__ __attribute__((aligned(16))) double lds[2][GKC][GLD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    uint32_t bi, bj;
    pair_to_tiles(blockIdx.x, a.tiles, bi, bj);
    const bool diag = bi == bj;
    const bool compute = !(diag && wr > wc);
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

    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

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
            if (!diag) pb[h] = load4(rp, colB, a.d, ok[h], a.w4);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float va[4] = {pa[h].x, pa[h].y, pa[h].z, pa[h].w};
            const float vb[4] = {pb[h].x, pb[h].y, pb[h].z, pb[h].w};
            double *da = &lds[0][lr + 8 * h][c4 * 4];
            double *db = &lds[1][lr + 8 * h][c4 * 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // centre in f64: block.astype(float64) - mean          (pycleora/__init__.py:141)
                da[q] = (ok[h] && colA + q < a.d) ? (double)va[q] - mA[q] : 0.0;
                if (!diag) db[q] = (ok[h] && colB + q < a.d) ? (double)vb[q] - mB[q] : 0.0;
            }
        }
    };

    if (r_begin < r_end) prefetch(r_begin);
    for (uint64_t row0 = r_begin; row0 < r_end; row0 += GKC) {
        stage();
        __syncthreads();
        if (row0 + GKC < r_end) prefetch(row0 + GKC);
        if (compute) {
            const int pb_sel = diag ? 0 : 1;
#pragma unroll
            for (int kk = 0; kk < GKC / 4; ++kk) {
                const int krow = kk * 4 + (lane >> 4);
                double fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = lds[0][krow][wr * 64 + i * 16 + (lane & 15)];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = lds[pb_sel][krow][wc * 64 + j * 16 + (lane & 15)];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    if (compute) {
        double *out = a.partial + ((uint64_t)blockIdx.y * a.pairs + blockIdx.x) * (uint64_t)(GT * GT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    // f64 16x16x4 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
                    const int row = wr * 64 + i * 16 + (lane >> 4) + 4 * reg;
                    const int col = wc * 64 + j * 16 + (lane & 15);
                    out[row * GT + col] = acc[i][j][reg];
                }
    }
}

// gram[gi][gj] = sum over slices (fixed order) of the partial tiles; mirrors the upper triangle.
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double *__restrict__ partial,
                                                          uint32_t slices, uint32_t pairs,
                                                          uint32_t tiles, uint32_t d,
                                                          double *__restrict__ gram) {
    const uint32_t p = blockIdx.y;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;  // element of the GT x GT tile
    const uint32_t r = e / GT, c = e % GT;
    uint32_t bi, bj;
    pair_to_tiles(p, tiles, bi, bj);
    const bool diag = bi == bj;
    if (diag && (r / 64) > (c / 64)) return;  // not computed: mirrored from the (0,1) quadrant
    const uint32_t gi = bi * GT + r, gj = bj * GT + c;
    if (gi >= d || gj >= d) return;
    double s = 0.0;
    for (uint32_t sl = 0; sl < slices; ++sl)
        s += partial[((uint64_t)sl * pairs + p) * (uint64_t)(GT * GT) + e];
    gram[(uint64_t)gi * d + gj] = s;
    if (!diag || (r / 64) < (c / 64)) gram[(uint64_t)gj * d + gi] = s;
}

inline uint32_t gram_slices(uint64_t n, uint32_t pairs) {
    uint64_t s = (2048 + pairs - 1) / pairs;
    const uint64_t cap = (n + 255) / 256;
    if (s > cap) s = cap;
    return (uint32_t)(s < 1 ? 1 : s);
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
    int w4x, w4t;
};

__global__ __launch_bounds__(256) void project_kernel(const ProjArgs a) {
    __shared__ __attribute__((aligned(16))) float As[PM][PLA];
    __shared__ __attribute__((aligned(16))) float Bs[PK][PN];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    const uint32_t bn = blockIdx.x % a.nb_n;
    const uint64_t bm = blockIdx.x / a.nb_n;
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
    const uint32_t tiles = (d + GT - 1) / GT;
    const uint32_t pairs = tiles * (tiles + 1) / 2;
    return (uint64_t)gram_slices(n, pairs) * pairs * GT * GT;
}

int launch_gram(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const double *mean,
                double *ws, double *gram, hipStream_t stream) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && mean != nullptr && ws != nullptr && gram != nullptr,
               "x / mean / workspace / gram is NULL");
    GramArgs a{};
    a.x = x;
    a.ldx = ldx;
    a.n = n;
    a.d = d;
    a.mean = mean;
    a.partial = ws;
    a.tiles = (d + GT - 1) / GT;
    a.pairs = a.tiles * (a.tiles + 1) / 2;
    const uint32_t slices = gram_slices(n, a.pairs);
    uint64_t rps = (n + slices - 1) / slices;
    rps = (rps + GKC - 1) / GKC * GKC;
    a.rows_per_slice = rps ? rps : GKC;
    a.w4 = (d % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
    CL_REQUIRE(slices <= 65535, "internal: too many Gram slices");
    // quadrants that are never computed are skipped by the reducer, so no memset is needed
    hipLaunchKernelGGL(gram_kernel, dim3(a.pairs, slices), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(GT * GT / 256, a.pairs), dim3(256), 0, stream, ws,
                       slices, a.pairs, a.tiles, d, gram);
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
    CL_REQUIRE(blocks < (1ull << 31), "too many row blocks");
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
