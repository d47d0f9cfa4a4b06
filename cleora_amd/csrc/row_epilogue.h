// row_epilogue.h — the per-row pieces shared by the SpMM kernels (spmm.hip) and the fused attention SpMM (attention.hip):
// row loads / stores for a group of G lanes, the lane broadcasts, and finish_row — residual blend, L1 / L2 normalisation,
// squared difference and the store, on the accumulator registers (src/embedding.rs:88-136, pycleora/__init__.py:942-950).
#pragma once
#include "common.h"

namespace cleora {
namespace {

// Internal flag bit (never part of the ABI's flag set): the epilogue's row stores are non-temporal.  An iterate far larger than the
// caches is written once per launch and read by nobody before the launch ends: left to the default policy its 10 GB stream through
// L2 and Infinity Cache and take the place of the rows the gather cache policy keeps there (hot.hip).  Measured at C3 on the same
// (X, Y) pairs, one process, builds side by side (scripts/r06/store_policy_probe.py): 30.7-31.7 ms against 31.6-33.1 with the default
// policy in the fast placement class, 35.1-36.9 against 36.3 in the slow one; sc0 / sc1 / sc0 sc1 stores: no different from the default.
// (Round 1 had measured +0.1 % — before there was a hot set to protect.)  Set by the launchers for outputs of 256 MiB and more.
constexpr uint32_t kStreamStores = 1u << 30;
constexpr uint64_t kStreamStoreMinBytes = 256ull << 20;

struct RowArgs {
    float *y;
    uint64_t ldy;
    const float *x_self;
    uint64_t ldxs;
    double *row_sqdiff;
    float *row_sumsq;
    float rw, alpha;
    uint32_t flags;  // CLEORA_F_* with RESIDUAL already gated on 0 < rw < 1
    uint32_t d;
};

__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }

template <int G>
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, uint32_t src, int gbase) {
    if constexpr (G == 64) {
        return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
    } else {
        return (uint32_t)__shfl((int)v, gbase + (int)src, 64);
    }
}
template <int G>
__device__ __forceinline__ float bcast_f32(float v, uint32_t src, int gbase) {
    return __uint_as_float(bcast_u32<G>(__float_as_uint(v), src, gbase));
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// element index of chunk v of lane gl:  j = (v*G + gl) * W
template <int G, int V, int W, bool FULL>
__device__ __forceinline__ void load_row(const float *__restrict__ p, int gl, uint32_t d,
                                         float (&r)[V][W]) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const uint32_t j = (uint32_t)(v * G + gl) * W;
        if constexpr (W == 4) {
            if (FULL || j < d) {
                const float4 t = *reinterpret_cast<const float4 *>(p + j);
                r[v][0] = t.x; r[v][1] = t.y; r[v][2] = t.z; r[v][3] = t.w;
            } else {
                r[v][0] = r[v][1] = r[v][2] = r[v][3] = 0.f;
            }
        } else {
            r[v][0] = (FULL || j < d) ? p[j] : 0.f;
        }
    }
}

template <int G, int V, int W, bool FULL>
__device__ __forceinline__ void store_row(float *__restrict__ p, int gl, uint32_t d,
                                          const float (&r)[V][W], bool nt = false) {
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const uint32_t j = (uint32_t)(v * G + gl) * W;
        if constexpr (W == 4) {
            if (FULL || j < d) {
                if (nt) {                                                         // streaming store: see kStreamStores
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    const nt_f4 q = {r[v][0], r[v][1], r[v][2], r[v][3]};
                    __builtin_nontemporal_store(q, reinterpret_cast<nt_f4 *>(p + j));
                } else {
                    *reinterpret_cast<float4 *>(p + j) = make_float4(r[v][0], r[v][1], r[v][2], r[v][3]);
                }
            }
        } else {
            if (FULL || j < d) p[j] = r[v][0];
        }
    }
}

// Residual blend, L2 normalise, squared difference, store — on the accumulator registers.
template <int G, int V, int W, bool FULL>
__device__ __forceinline__ void finish_row(const RowArgs &ra, uint64_t row, int gl, int gbase,
                                           float (&acc)[V][W]) {
    const uint32_t d = ra.d;
    float xs[V][W];
    if (ra.flags & (CLEORA_F_RESIDUAL | CLEORA_F_SQDIFF))
        load_row<G, V, W, FULL>(ra.x_self + row * ra.ldxs, gl, d, xs);

    if (ra.flags & CLEORA_F_RESIDUAL) {
        // dst[j] = alpha * dst[j] + rw * src[j]                      (src/embedding.rs:121-129)
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int q = 0; q < W; ++q)
                acc[v][q] = fadd(fmul(ra.alpha, acc[v][q]), fmul(ra.rw, xs[v][q]));
    }

    if (ra.flags & (CLEORA_F_L2NORM | CLEORA_F_ROWSQ | CLEORA_F_SCALE | CLEORA_F_L1NORM)) {
        const bool l1 = (ra.flags & CLEORA_F_L1NORM) != 0;   // sum |v| instead of sum v*v
        float s = 0.f;
        if (ra.flags & CLEORA_F_SCALE) {
            s = ra.row_sumsq[row];  // complete (all-reduced) sum of squares of the whole row
        } else if (ra.flags & CLEORA_F_FASTNORM) {
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int q = 0; q < W; ++q) s = fadd(s, l1 ? fabsf(acc[v][q]) : fmul(acc[v][q], acc[v][q]));
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) s = fadd(s, __shfl_xor(s, o, 64));
        } else {
            // the reference's order: sum_sq += v*v for j = 0..d-1   (src/embedding.rs:94-97); ROWSQ_CONT: these d columns are a slice of
            // a wider row and the sum continues where the columns to the left ended
            if ((ra.flags & CLEORA_F_ROWSQ_CONT) && (ra.flags & CLEORA_F_ROWSQ)) s = ra.row_sumsq[row];
            const uint32_t nchunks = (d + W - 1) / W;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float sq[W];
#pragma unroll
                for (int q = 0; q < W; ++q) sq[q] = l1 ? fabsf(acc[v][q]) : fmul(acc[v][q], acc[v][q]);
                const uint32_t base = (uint32_t)v * G;
                const uint32_t lim = nchunks > base ? ((nchunks - base) < (uint32_t)G ? (nchunks - base) : (uint32_t)G) : 0u;
                for (uint32_t g = 0; g < lim; ++g) {
#pragma unroll
                    for (int q = 0; q < W; ++q) s = fadd(s, bcast_f32<G>(sq[q], g, gbase));
                }
            }
        }
        if ((ra.flags & CLEORA_F_ROWSQ) && gl == 0) ra.row_sumsq[row] = s;
        if (l1) {
            // norms = max(sum |v|, 1e-10); emb / norms: a true division    (pycleora/__init__.py:947-950)
            const float norm = fmaxf(s, 1e-10f);
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int q = 0; q < W; ++q) acc[v][q] = acc[v][q] / norm;
        } else if (ra.flags & (CLEORA_F_L2NORM | CLEORA_F_SCALE)) {
            // norm = sum_sq.sqrt().max(1e-10); inv = 1/norm; v *= inv    (src/embedding.rs:98-102)
            const float norm = fmaxf(sqrtf(s), 1e-10f);
            const float inv = (1.0f / norm);
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int q = 0; q < W; ++q) acc[v][q] = fmul(acc[v][q], inv);
        }
    }

    if (ra.flags & CLEORA_F_SQDIFF) {
        // delta = dst - src (f32, as the reference), accumulated in f64 instead of the
        // reference's whole-matrix f32 accumulator                   (src/embedding.rs:169-176)
        double ds = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const uint32_t j = (uint32_t)(v * G + gl) * W + q;
                if (FULL || j < d) {
                    // f32 difference like src/embedding.rs:172, or the f64 difference of _compute_rmse (pycleora/__init__.py:975)
                    const double delta = (ra.flags & CLEORA_F_SQDIFF64) ? (double)acc[v][q] - (double)xs[v][q]
                                                                        : (double)fsub(acc[v][q], xs[v][q]);
                    ds += delta * delta;
                }
            }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
        if (gl == 0) ra.row_sqdiff[row] = ds;
    }

    store_row<G, V, W, FULL>(ra.y + row * ra.ldy, gl, d, acc, (ra.flags & kStreamStores) != 0);
}

}  // namespace
}  // namespace cleora
