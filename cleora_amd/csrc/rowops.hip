// rowops.hip — the small streaming kernels around the SpMM: deterministic init,
// fixed-order f64 reductions (RMSE, column sums for the whitening mean).
#include "common.h"

namespace cleora {
namespace {

// init_value (src/lib.rs:478-488): FxHasher::write_i64 from state 0 is x * K (mod 2^64);
// Rust `%` on i64 is the truncated remainder, as C++'s; |r| < 2^23 so the cast and the
// division by 2^23 are exact: the result is bit-identical to the CPU's.
constexpr uint64_t kFxK = 0x517cc1b727220a95ULL;

__global__ __launch_bounds__(256) void init_kernel(const uint64_t *__restrict__ hash, uint64_t n,
                                                   uint32_t d, int64_t seed, float *__restrict__ x,
                                                   uint64_t ldx) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const uint64_t h = hash[row];
    float *xr = x + row * ldx;
    for (uint32_t c = lane; c < d; c += 64) {
        const uint64_t s = h + (uint64_t)c + (uint64_t)seed;  // wrapping i64 add
        const int64_t hv = (int64_t)(s * kFxK);
        const int64_t r = hv % (int64_t)(8 * 1024 * 1024);
        xr[c] = (float)r / 8388608.0f;
    }
}

// ---- deterministic f64 sum --------------------------------------------------------------------
constexpr int kReduceBlocks = 1024;

__device__ __forceinline__ double block_sum_256(double v, double *sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) sm[t] += sm[t + s];
        __syncthreads();
    }
    return sm[0];
}

__global__ __launch_bounds__(256) void reduce_stage1(const double *__restrict__ v, uint64_t n,
                                                     uint64_t per_block, double *__restrict__ part) {
    __shared__ double sm[256];
    const uint64_t b0 = (uint64_t)blockIdx.x * per_block;
    const uint64_t b1 = b0 + per_block < n ? b0 + per_block : n;
    double s = 0.0;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += 256) s += v[i];
    const double t = block_sum_256(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void reduce_stage2(const double *__restrict__ part, int nb,
                                                     double *__restrict__ out) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
    const double t = block_sum_256(s, sm);
    if (threadIdx.x == 0) out[0] = t;
}

// ---- column sums in f64 (whitening mean, pycleora/__init__.py:136) ----------------------------------
constexpr int kColsumMaxBlocks = 2048;

inline int colsum_blocks(uint64_t n) {
    const uint64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > kColsumMaxBlocks ? kColsumMaxBlocks : b));
}

// Thread t owns column chunk (t % tc) of every (256/tc)-th row of the block's strip.
template <int W>
__global__ __launch_bounds__(256) void colsum_stage1(const float *__restrict__ x, uint64_t ldx,
                                                     uint64_t n, uint32_t d, int tc,
                                                     uint64_t rows_per_block,
                                                     double *__restrict__ part) {
    __shared__ double sm[256 * W];
    const int t = threadIdx.x;
    const int c0 = t % tc, rsub = t / tc, rstep = 256 / tc;
    const uint64_t r0 = (uint64_t)blockIdx.x * rows_per_block;
    const uint64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    const uint32_t chunks = (d + W - 1) / W;
    for (uint32_t cb = 0; cb < chunks; cb += tc) {
        const uint32_t ch = cb + c0;
        double acc[W];
#pragma unroll
        for (int q = 0; q < W; ++q) acc[q] = 0.0;
        if (ch < chunks) {
            for (uint64_t r = r0 + rsub; r < r1; r += rstep) {
                const float *p = x + r * ldx + (uint64_t)ch * W;
                if constexpr (W == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(p);
                    acc[0] += (double)v.x; acc[1] += (double)v.y; acc[2] += (double)v.z; acc[3] += (double)v.w;
                } else {
                    acc[0] += (double)p[0];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < W; ++q) sm[q * 256 + t] = acc[q];
        __syncthreads();
        if (rsub == 0 && ch < chunks) {
#pragma unroll
            for (int q = 0; q < W; ++q) {
                double s = 0.0;
                for (int k = 0; k < rstep; ++k) s += sm[q * 256 + k * tc + c0];  // fixed order
                const uint32_t c = ch * W + q;
                if (c < d) part[(uint64_t)blockIdx.x * d + c] = s;
            }
        }
        __syncthreads();
    }
}

// 32 columns per block, 8 threads per column: each adds a contiguous eighth of the block partials in
// order, the eight sums are then added in order (fixed order => deterministic).
__global__ __launch_bounds__(256) void colsum_stage2(const double *__restrict__ part, int nb,
                                                     uint32_t d, double *__restrict__ out) {
    __shared__ double sm[8][32];
    const int cl = threadIdx.x & 31, p = threadIdx.x >> 5;
    const uint32_t c = blockIdx.x * 32 + cl;
    const int per = (nb + 7) / 8;
    const int b0 = p * per, b1 = (b0 + per) < nb ? (b0 + per) : nb;
    double s = 0.0;
    if (c < d)
        for (int b = b0; b < b1; ++b) s += part[(uint64_t)b * d + c];
    sm[p][cl] = s;
    __syncthreads();
    if (p == 0 && c < d) {
        double t = sm[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sm[k][cl];
        out[c] = t;
    }
}

// ---- cosine scores against one query (find_most_similar, pycleora/__init__.py:753-781) --------------
// scores[r] = (x[r] . q) / max(||x[r]||, 1e-10); one wavefront per row, one pass over X (HBM-bound).
template <bool W4>
__global__ __launch_bounds__(256) void cosine_kernel(const float *__restrict__ x, uint64_t ldx, uint64_t n,
                                                     uint32_t d, const float *__restrict__ q,
                                                     float *__restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + row * ldx;
    float dot = 0.f, sq = 0.f;
    if constexpr (W4) {                       // 16 B per lane: one wavefront load covers a whole 1 KiB row
        for (uint32_t c = lane * 4; c < d; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + c);
            const float4 w = *reinterpret_cast<const float4 *>(q + c);
            dot += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
            sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    } else {
        for (uint32_t c = lane; c < d; c += 64) {
            const float v = xr[c];
            dot += v * q[c];
            sq += v * v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dot += __shfl_xor(dot, o, 64);
        sq += __shfl_xor(sq, o, 64);
    }
    if (lane == 0) scores[row] = dot / fmaxf(sqrtf(sq), 1e-10f);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// s[r] = sum of the stored values of row r = (A 1)[r]: what the propagate-before-project form of the whitened loop
// needs to move the mean through the SpMM, A (Y - 1 mu^T) = A Y - s mu^T.  One wavefront per row, f64 partials.
__global__ __launch_bounds__(256) void csr_rowsum_kernel(const uint64_t *__restrict__ rowptr, const float *__restrict__ val,
                                                         uint64_t n_rows, float *__restrict__ out, float *__restrict__ abs_out) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    double s = 0.0, sa = 0.0;
    for (uint64_t e = rowptr[row] + lane; e < rowptr[row + 1]; e += 64) {
        const double v = (double)val[e];
        s += v;
        sa += fabs(v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        sa += __shfl_xor(sa, o, 64);
    }
    if (lane == 0) {
        out[row] = (float)s;
        // rounded UP: this is a bound on |(A y)[r][j]| for |y| <= 1 (project_f16.hip scales a row by it)
        if (abs_out) abs_out[row] = __double2float_ru(sa);
    }
}

}  // namespace

int launch_csr_rowsum(const cleora_graph *g, int kind, float *out, hipStream_t stream, float *abs_out) {
    CL_REQUIRE(g != nullptr && out != nullptr, "graph / out is NULL");
    CL_REQUIRE((kind == CLEORA_LEFT || kind == CLEORA_SYMMETRIC) && g->val[kind] != nullptr, "no values for this markov_type");
    if (g->n_rows == 0) return CLEORA_OK;
    hipLaunchKernelGGL(csr_rowsum_kernel, grid_1d_as_2d((g->n_rows + 3) / 4), dim3(256), 0, stream, g->rowptr, g->val[kind],
                       g->n_rows, out, abs_out);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

int launch_cosine(const float *x, uint64_t ldx, uint64_t n, uint32_t d, const float *q, float *scores,
                  hipStream_t stream) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && q != nullptr && scores != nullptr, "x / q / scores is NULL");
    if (n == 0) return CLEORA_OK;
    if ((d % 4 == 0) && (ldx % 4 == 0) && aligned16(x) && aligned16(q))
        hipLaunchKernelGGL(cosine_kernel<true>, grid_1d_as_2d((n + 3) / 4), dim3(256), 0, stream, x, ldx, n, d, q, scores);
    else
        hipLaunchKernelGGL(cosine_kernel<false>, grid_1d_as_2d((n + 3) / 4), dim3(256), 0, stream, x, ldx, n, d, q, scores);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

int launch_init(const uint64_t *hash, uint64_t n, uint32_t d, int64_t seed, float *x, uint64_t ldx,
                hipStream_t stream) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(hash != nullptr && x != nullptr, "hash / x is NULL");
    if (n == 0) return CLEORA_OK;
    hipLaunchKernelGGL(init_kernel, grid_1d_as_2d((n + 3) / 4), dim3(256), 0, stream, hash, n, d,
                       seed, x, ldx);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

uint64_t reduce_workspace(uint64_t) { return kReduceBlocks; }

int launch_reduce_sum(const double *v, uint64_t n, double *ws, double *out, hipStream_t stream) {
    CL_REQUIRE(ws != nullptr && out != nullptr, "workspace / out is NULL");
    CL_REQUIRE(n == 0 || v != nullptr, "v is NULL");
    uint64_t per_block = (n + kReduceBlocks - 1) / kReduceBlocks;
    per_block = (per_block + 255) / 256 * 256;
    if (per_block == 0) per_block = 256;
    int nb = (int)((n + per_block - 1) / per_block);
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(reduce_stage1, dim3(nb), dim3(256), 0, stream, v, n, per_block, ws);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(256), 0, stream, ws, nb, out);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

uint64_t colsum_workspace(uint64_t n, uint32_t d) { return (uint64_t)colsum_blocks(n) * d; }

int launch_colsum(const float *x, uint64_t ldx, uint64_t n, uint32_t d, double *ws, double *out,
                  hipStream_t stream) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && ws != nullptr && out != nullptr, "x / workspace / out is NULL");
    const int nb = colsum_blocks(n);
    const uint64_t rows_per_block = (n + nb - 1) / nb;
    const bool w4 = (d % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
    const uint32_t chunks = w4 ? d / 4 : d;
    int tc = 1;
    while (tc < 256 && (uint32_t)tc < chunks) tc <<= 1;
    if (w4)
        hipLaunchKernelGGL(colsum_stage1<4>, dim3(nb), dim3(256), 0, stream, x, ldx, n, d, tc,
                           rows_per_block ? rows_per_block : 1, ws);
    else
        hipLaunchKernelGGL(colsum_stage1<1>, dim3(nb), dim3(256), 0, stream, x, ldx, n, d, tc,
                           rows_per_block ? rows_per_block : 1, ws);
    hipLaunchKernelGGL(colsum_stage2, dim3((d + 31) / 32), dim3(256), 0, stream, ws, nb, d, out);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
