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
// arithmetic in f64 like the reference.  Two forms: edge_attention_vec_kernel (aligned rows of up to 2048 floats:
// 16-byte gathers, 8 rows in flight, scores in registers — below) and the scalar edge_attention_kernel for
// every other shape.  Rows are not split: a hub row is served by a single wavefront.
#include "common.h"
#include "row_epilogue.h"

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

// Eight per-lane partial sums -> their eight wave totals in 10 lane exchanges instead of 8 butterflies of 6: the first three
// steps are a reduce-scatter (the lanes of a pair keep half of the values each and add the partner's copy of that half:
// 4 + 2 + 1 exchanges), the last three a butterfly on the one value left.  The total of v[u] ends up in every lane whose
// bits 5, 4, 3 spell u — read it with wave8_total(r, u) (lane 8 u).
__device__ __forceinline__ float wave8_reduce(const float (&v)[8], int lane) {
    float w4[4], w2[2], w1;
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = b5 ? v[4 + j] : v[j], send = b5 ? v[j] : v[4 + j];
        w4[j] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = b4 ? w4[2 + j] : w4[j], send = b4 ? w4[j] : w4[2 + j];
        w2[j] = keep + __shfl_xor(send, 16, 64);
    }
    {
        const float keep = b3 ? w2[1] : w2[0], send = b3 ? w2[0] : w2[1];
        w1 = keep + __shfl_xor(send, 8, 64);
    }
    w1 += __shfl_xor(w1, 4, 64);
    w1 += __shfl_xor(w1, 2, 64);
    w1 += __shfl_xor(w1, 1, 64);
    return w1;
}
__device__ __forceinline__ float wave8_total(float r, int u) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(r), 8 * u));
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

// ---- the form for aligned rows of up to 2048 floats: whole 16-byte gathers, 8 in flight, scores in registers ------------
// One wavefront per row, lane l holding elements [4(l + 64v), +4) of every embedding row it touches (V float4 per lane).
// Row norms come from one streaming pass (row_norm_kernel) instead of a second wave reduction per edge.  The row's own
// vector stays in registers; its edges are visited 64 at a time (one coalesced load of col / adj per chunk, column
// indices broadcast with v_readlane), 8 neighbour rows in flight; lane k of the chunk keeps edge k's score in a
// register, RC chunks deep — rows of up to 64*RC edges never touch memory for the softmax.  Longer rows keep their
// scores in `out` as before (the wave re-reads its own writes after a device-scope fence).
constexpr int RC = 8;

__global__ __launch_bounds__(256) void row_norm_kernel(const float *__restrict__ x, uint64_t ldx, uint64_t n, uint32_t d,
                                                       float *__restrict__ norm) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + row * ldx;
    float sq = 0.f;
    for (uint32_t c = lane * 4; c < d; c += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(xr + c);
        sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    sq = wave_sum(sq);
    if (lane == 0) norm[row] = fmaxf(sqrtf(sq), 1e-10f);
}

template <int V>
__global__ __launch_bounds__(256) void edge_attention_vec_kernel(const uint64_t *__restrict__ rowptr,
                                                                 const uint32_t *__restrict__ col,
                                                                 const float *__restrict__ adj,
                                                                 const float *__restrict__ x, uint64_t ldx, uint32_t d,
                                                                 const float *__restrict__ norm, uint64_t n_rows,
                                                                 float temperature, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint64_t beg = rowptr[row], end = rowptr[row + 1];
    if (beg == end) return;
    float4 xr[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const uint32_t c = (uint32_t)(v * 64 + lane) * 4;
        xr[v] = c < d ? *reinterpret_cast<const float4 *>(x + row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float nr = norm[row];
    const bool in_regs = end - beg <= (uint64_t)64 * RC;
    float keep[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) keep[c] = -INFINITY;
    double mx = -INFINITY;
    int chunk = 0;
    for (uint64_t e0 = beg; e0 < end; e0 += 64, ++chunk) {
        const uint32_t cnt = end - e0 < 64 ? (uint32_t)(end - e0) : 64u;
        const uint32_t cv = (uint32_t)lane < cnt ? col[e0 + lane] : 0u;
        const float ncv = (uint32_t)lane < cnt ? norm[cv] : 1.f;   // the chunk's 64 neighbour norms in one gather (not one load per edge)
        float mine = -INFINITY;                                   // score of edge e0 + lane
        for (uint32_t k0 = 0; k0 < cnt; k0 += 8) {
            float4 g[8][V];
            float dot[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cv, (int)(k0 + u < cnt ? k0 + u : k0));
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const uint32_t cc = (uint32_t)(v * 64 + lane) * 4;
                    g[u][v] = cc < d ? *reinterpret_cast<const float4 *>(x + (uint64_t)c * ldx + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float s = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) s += xr[v].x * g[u][v].x + xr[v].y * g[u][v].y + xr[v].z * g[u][v].z + xr[v].w * g[u][v].w;
                dot[u] = s;
            }
            const float dots = wave8_reduce(dot, lane);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u < cnt) {
                    const float nc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ncv), (int)(k0 + u)));
                    const float s = wave8_total(dots, u) / (nr * nc) / temperature;   // (:243-248)
                    if ((uint32_t)lane == k0 + u) mine = s;
                }
        }
        mx = fmax(mx, (double)mine);
        if (in_regs) {
#pragma unroll
            for (int c = 0; c < RC; ++c)
                if (c == chunk) keep[c] = mine;
        } else if ((uint32_t)lane < cnt) {
            out[e0 + lane] = mine;
        }
    }
    mx = wave_max(mx);
    if (!in_regs) __threadfence();          // this wave re-reads its scores from `out` through device-scope loads below
    auto score = [&](int c, uint64_t e) -> double {
        if (in_regs) {
            float v = -INFINITY;
#pragma unroll
            for (int k = 0; k < RC; ++k)
                if (k == c) v = keep[k];
            return (double)v;
        }
        return (double)__hip_atomic_load(out + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    double se = 0.0;
    chunk = 0;
    for (uint64_t e = beg + lane; e < end; e += 64, ++chunk) se += exp(score(chunk, e) - mx);
    se = fmax(wave_sum(se), 1e-10);                                   // softmax over the row's edges (:250-262)
    double sw = 0.0;
    chunk = 0;
    for (uint64_t e = beg + lane; e < end; e += 64, ++chunk) sw += exp(score(chunk, e) - mx) / se * (double)adj[e];
    sw = fmax(wave_sum(sw), 1e-10);                                   // re-weighted and row-normalised (:264-267)
    chunk = 0;
    for (uint64_t e = beg + lane; e < end; e += 64, ++chunk)
        out[e] = (float)(exp(score(chunk, e) - mx) / se * (double)adj[e] / sw);
}

// ---- attention weights AND the weighted SpMM in one pass over the edges ---------------------------------------------------------
// embed_with_attention computes the weights from the current iterate and multiplies with them at once (:241-269):
//     y_r = sum_e w_e x_{c_e},   w_e = exp(s_e - max_r) adj_e / (se_r * max(sum_e' exp(s_e' - max_r) adj_e' / se_r, 1e-10)),   se_r = max(sum exp, 1e-10)
// The two-kernel route (edge_attention_vec_kernel, then the SpMM with those values) gathers every neighbour row twice.  Here the
// softmax is accumulated ONLINE: the running maximum m, l = sum exp(s - m) adj, se = sum exp(s - m) and the weighted sum
// acc = sum exp(s - m) adj x_c are rescaled by exp(m_old - m_new) whenever an edge raises the maximum, so each neighbour row is
// gathered once, used for its score and for the sum, and dropped.  Same function of the inputs as the reference's (f32 here where
// the reference holds the weights in f64: tests carry the tolerance of the two-kernel route, 2e-5 on unit rows); the row
// epilogue (residual, normalisation, squared difference) is the SpMM's (row_epilogue.h).  One wavefront per row, rows not split.
template <int V>
__global__ __launch_bounds__(256) void attention_spmm_kernel(const uint64_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                             const float *__restrict__ adj, const float *__restrict__ x, uint64_t ldx,
                                                             uint32_t d, const float *__restrict__ norm, uint64_t n_rows,
                                                             float temperature, const RowArgs ra) {
    const int lane = threadIdx.x & 63;
    uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    row = uniform_u64(row);
    const uint64_t beg = uniform_u64(rowptr[row]), end = uniform_u64(rowptr[row + 1]);
    float acc[V][4];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v][0] = acc[v][1] = acc[v][2] = acc[v][3] = 0.f;
    if (beg < end) {
        float4 xr[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const uint32_t c = (uint32_t)(v * 64 + lane) * 4;
            xr[v] = c < d ? *reinterpret_cast<const float4 *>(x + row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float nr = norm[row];
        float m = -INFINITY, l = 0.f, se = 0.f;
        for (uint64_t e0 = beg; e0 < end; e0 += 64) {
            const uint32_t cnt = end - e0 < 64 ? (uint32_t)(end - e0) : 64u;
            const uint32_t cv = (uint32_t)lane < cnt ? col[e0 + lane] : 0u;
            const float av = (uint32_t)lane < cnt ? adj[e0 + lane] : 0.f;
            // 1 / (|x_r| |x_c| temperature) for the chunk's 64 edges in one gather: a norm[c] load per edge inside the serial
            // part below cost a memory round trip per edge (round 3: 9.8 ms per iteration at the C2 shape, 3.3x the plain SpMM)
            const float qv = (uint32_t)lane < cnt ? 1.0f / (nr * norm[cv] * temperature) : 0.f;
            for (uint32_t k0 = 0; k0 < cnt; k0 += 8) {
                float4 g[8][V];
                float dot[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cv, (int)(k0 + u < cnt ? k0 + u : k0));
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const uint32_t cc = (uint32_t)(v * 64 + lane) * 4;
                        g[u][v] = cc < d ? *reinterpret_cast<const float4 *>(x + (uint64_t)c * ldx + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    float s = 0.f;
#pragma unroll
                    for (int v = 0; v < V; ++v) s += xr[v].x * g[u][v].x + xr[v].y * g[u][v].y + xr[v].z * g[u][v].z + xr[v].w * g[u][v].w;
                    dot[u] = s;
                }
                const float dots = wave8_reduce(dot, lane);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + u < cnt) {                                             // wave-uniform
                        const float a_e = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(av), (int)(k0 + u)));
                        const float q_e = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(qv), (int)(k0 + u)));
                        // (v_readlane puts the dot product into an SGPR: the score is wave-uniform by construction and the branch
                        // below a scalar one instead of an exec-masked region per edge)
                        const float s = wave8_total(dots, u) * q_e;                 // cos(x_r, x_c) / temperature (:243-248)
                        if (s > m) {                                                // a new row maximum: rescale what was summed so far
                            const float sc = expf(m - s);                          // exp(-inf) = 0 for the first edge
                            l *= sc;
                            se *= sc;
#pragma unroll
                            for (int v = 0; v < V; ++v) { acc[v][0] *= sc; acc[v][1] *= sc; acc[v][2] *= sc; acc[v][3] *= sc; }
                            m = s;
                        }
                        const float ex = expf(s - m);
                        const float p = ex * a_e;
                        se += ex;
                        l += p;
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            acc[v][0] += p * g[u][v].x; acc[v][1] += p * g[u][v].y; acc[v][2] += p * g[u][v].z; acc[v][3] += p * g[u][v].w;
                        }
                    }
            }
        }
        const float se_c = fmaxf(se, 1e-10f);                                       // (:257-258)
        const float inv = 1.0f / (se_c * fmaxf(l / se_c, 1e-10f));                 // (:264-267)
#pragma unroll
        for (int v = 0; v < V; ++v) { acc[v][0] *= inv; acc[v][1] *= inv; acc[v][2] *= inv; acc[v][3] *= inv; }
    }
    finish_row<64, V, 4, false>(ra, row, lane, 0, acc);
}

}  // namespace

// y = epilogue(attention-weighted A x) in one pass over the edges (see attention_spmm_kernel); CLEORA_E_INVALID for shapes the
// vector form does not take (d % 4 != 0, unaligned, d > 2048): the caller then runs the two-kernel route.
int launch_propagate_attention(const cleora_graph *g, int kind, const float *x, uint64_t ldx, uint32_t d, float temperature,
                               float *y, uint64_t ldy, uint32_t flags, float rw, const float *x_self, double *row_sqdiff,
                               hipStream_t stream) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    CL_REQUIRE(kind == CLEORA_LEFT || kind == CLEORA_SYMMETRIC, "unknown markov_type");
    CL_REQUIRE(g->val[kind] != nullptr, "graph has no values for this markov_type");
    CL_REQUIRE(g->n_rows == g->n_cols, "edge attention needs the whole (square) graph");
    CL_REQUIRE(d > 0 && ldx >= d && ldy >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && y != nullptr && x != y, "x / y is NULL or aliased");
    CL_REQUIRE(temperature > 0.0f, "attention_temperature must be positive");
    CL_REQUIRE(!(flags & (CLEORA_F_ROWSQ | CLEORA_F_SCALE)), "ROWSQ / SCALE are not available on the attention path");
    CL_REQUIRE(!((flags & CLEORA_F_L1NORM) && (flags & CLEORA_F_L2NORM)), "L1NORM is exclusive with L2NORM");
    const bool vec = d % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 &&
                     (reinterpret_cast<uintptr_t>(y) & 15u) == 0 && d <= 2048;
    CL_REQUIRE(vec, "the fused attention SpMM takes 16-byte aligned rows of at most 2048 floats, d % 4 == 0");
    if (flags & CLEORA_F_RESIDUAL) {                     // the gate of launch_propagate (spmm.hip gate_residual)
        const bool on = rw > 0.0f && (rw < 1.0f || (flags & CLEORA_F_BLEND_ANY));
        if (!on) flags &= ~CLEORA_F_RESIDUAL;
    }
    if (flags & (CLEORA_F_RESIDUAL | CLEORA_F_SQDIFF)) {
        if (!x_self) x_self = x;
        CL_REQUIRE((reinterpret_cast<uintptr_t>(x_self) & 15u) == 0, "x_self must be 16-byte aligned");
    }
    if (flags & CLEORA_F_SQDIFF) CL_REQUIRE(row_sqdiff != nullptr, "row_sqdiff is NULL");
    if (g->n_rows == 0) return CLEORA_OK;
    CL_HIP(hipSetDevice(g->device));
    RowArgs ra{};
    ra.y = y;
    ra.ldy = ldy;
    ra.x_self = x_self;
    ra.ldxs = ldx;
    ra.row_sqdiff = row_sqdiff;
    ra.row_sumsq = nullptr;
    ra.rw = rw;
    ra.alpha = 1.0f - rw;
    ra.flags = flags;
    ra.d = d;
    float *norm = nullptr;                      // n floats of scratch, stream-ordered
    CL_HIP(hipMallocAsync(reinterpret_cast<void **>(&norm), g->n_rows * sizeof(float), stream));
    const dim3 grid = grid_1d_as_2d((g->n_rows + 3) / 4);
    hipLaunchKernelGGL(row_norm_kernel, grid, dim3(256), 0, stream, x, ldx, g->n_rows, d, norm);
#define CLEORA_ATT(VV) hipLaunchKernelGGL(attention_spmm_kernel<VV>, grid, dim3(256), 0, stream, g->rowptr, g->col, g->val[kind], x, ldx, d, norm, g->n_rows, temperature, ra)
    switch ((d + 255) / 256) {
        case 1: CLEORA_ATT(1); break;
        case 2: CLEORA_ATT(2); break;
        case 3: case 4: CLEORA_ATT(4); break;
        default: CLEORA_ATT(8); break;
    }
#undef CLEORA_ATT
    const hipError_t le = hipGetLastError();
    CL_HIP(hipFreeAsync(norm, stream));
    CL_HIP(le);
    return CLEORA_OK;
}

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
    const bool vec = d % 4 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && d <= 2048;
    if (vec) {
        float *norm = nullptr;                  // n floats of scratch, stream-ordered
        CL_HIP(hipMallocAsync(reinterpret_cast<void **>(&norm), g->n_rows * sizeof(float), stream));
        const dim3 grid = grid_1d_as_2d((g->n_rows + 3) / 4);
        hipLaunchKernelGGL(row_norm_kernel, grid, dim3(256), 0, stream, x, ldx, g->n_rows, d, norm);
        switch ((d + 255) / 256) {
            case 1: hipLaunchKernelGGL(edge_attention_vec_kernel<1>, grid, dim3(256), 0, stream, g->rowptr, g->col, g->val[kind], x, ldx, d, norm, g->n_rows, temperature, vals_out); break;
            case 2: hipLaunchKernelGGL(edge_attention_vec_kernel<2>, grid, dim3(256), 0, stream, g->rowptr, g->col, g->val[kind], x, ldx, d, norm, g->n_rows, temperature, vals_out); break;
            case 3: case 4: hipLaunchKernelGGL(edge_attention_vec_kernel<4>, grid, dim3(256), 0, stream, g->rowptr, g->col, g->val[kind], x, ldx, d, norm, g->n_rows, temperature, vals_out); break;
            default: hipLaunchKernelGGL(edge_attention_vec_kernel<8>, grid, dim3(256), 0, stream, g->rowptr, g->col, g->val[kind], x, ldx, d, norm, g->n_rows, temperature, vals_out); break;
        }
        const hipError_t le = hipGetLastError();
        CL_HIP(hipFreeAsync(norm, stream));
        CL_HIP(le);
        return CLEORA_OK;
    }
    hipLaunchKernelGGL(edge_attention_kernel, grid_1d_as_2d((g->n_rows + 3) / 4), dim3(256), 0, stream, g->rowptr,
                       g->col, g->val[kind], x, ldx, d, g->n_rows, temperature, vals_out);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
