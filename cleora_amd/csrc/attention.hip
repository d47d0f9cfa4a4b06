// attention.hip — per-edge attention weights of embed_with_attention (pycleora/__init__.py:241-268).
//
// For every stored edge (r, c) of the graph:
//   s_e   = cos(x_r, x_c) / temperature                       (:243-248, rows normalised with max(||.||, 1e-10))
//   a_e   = exp(s_e - max_row s) / max(sum_row exp, 1e-10)    (:250-262, softmax over the row's edges)
//   w_e   = a_e * adj_e / max(sum_row a * adj, 1e-10)         (:264-267, re-weighted and row-normalised)
// The values are written in the graph's edge order, ready for cleora_propagate_vals_dev.
//
// One wavefront per row: the row's edges are visited once for the scores (one gathered X row per edge —
// the same traffic as an SpMM), then three lane-parallel passes over the row's scores do the softmax
// arithmetic in f64 like the reference.  Not a headline kernel: rows are not split, so a hub row is
// served by a single wavefront.
#include "common.h"

namespace cleora {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void edge_attention_kernel(const uint64_t *__restrict__ rowptr,
                                                             const uint32_t *__restrict__ col,
                                                             const float *__restrict__ adj,
                                                             const float *__restrict__ x, uint64_t ldx, uint32_t d,
                                                             uint64_t n_rows, float temperature,
                                                             float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint64_t beg = rowptr[row], end = rowptr[row + 1];
    if (beg == end) return;
    const float *xr = x + row * ldx;
    float sq = 0.f;
    for (uint32_t j = lane; j < d; j += 64) sq += xr[j] * xr[j];
    const float nr = fmaxf(sqrtf(wave_sum(sq)), 1e-10f);
    // scores (lane 0 writes them), running row maximum
    double mx = -INFINITY;
    for (uint64_t e = beg; e < end; ++e) {
        const float *xc = x + (uint64_t)col[e] * ldx;
        float dot = 0.f, sc = 0.f;
        for (uint32_t j = lane; j < d; j += 64) {
            const float v = xc[j];
            dot += xr[j] * v;
            sc += v * v;
        }
        dot = wave_sum(dot);
        const float nc = fmaxf(sqrtf(wave_sum(sc)), 1e-10f);
        const float s = dot / (nr * nc) / temperature;
        if (lane == 0) out[e] = s;
        mx = fmax(mx, (double)s);
    }
    __threadfence();                               // lane 0's scores are read by every lane below
    // ... through device-scope loads: the per-CU vector L1 is not coherent, and a cache line of `out` also holds
    // the entries of neighbouring rows, so another wavefront of this CU may have pulled a stale copy of it
    auto score = [&](uint64_t e) { return (double)__hip_atomic_load(out + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    double se = 0.0;
    for (uint64_t e = beg + lane; e < end; e += 64) se += exp(score(e) - mx);
    se = fmax(wave_sum(se), 1e-10);
    double sw = 0.0;
    for (uint64_t e = beg + lane; e < end; e += 64) sw += exp(score(e) - mx) / se * (double)adj[e];
    sw = fmax(wave_sum(sw), 1e-10);
    for (uint64_t e = beg + lane; e < end; e += 64)
        out[e] = (float)(exp(score(e) - mx) / se * (double)adj[e] / sw);
}

}  // namespace

int launch_edge_attention(const cleora_graph *g, int kind, const float *x, uint64_t ldx, uint32_t d,
                          float temperature, float *vals_out, hipStream_t stream) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    CL_REQUIRE(kind == CLEORA_LEFT || kind == CLEORA_SYMMETRIC, "unknown markov_type");
    CL_REQUIRE(g->val[kind] != nullptr, "graph has no values for this markov_type");
    CL_REQUIRE(g->n_rows == g->n_cols, "edge attention needs the whole (square) graph");
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && vals_out != nullptr, "x / vals_out is NULL");
    CL_REQUIRE(temperature > 0.0f, "attention_temperature must be positive");
    if (g->n_rows == 0 || g->nnz == 0) return CLEORA_OK;
    CL_HIP(hipSetDevice(g->device));
    hipLaunchKernelGGL(edge_attention_kernel, grid_1d_as_2d((g->n_rows + 3) / 4), dim3(256), 0, stream, g->rowptr,
                       g->col, g->val[kind], x, ldx, d, g->n_rows, temperature, vals_out);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
