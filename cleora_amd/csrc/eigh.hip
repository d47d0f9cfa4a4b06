// eigh.hip — from the Gram matrix to the whitening transform, without leaving the device.
//
//   cov = G / (n-1); lambda, V = eigh(cov); descending; T = V * 1/sqrt(max(lambda, 1e-10)) as f32
//   (pycleora/__init__.py:143-156)
//
// The d x d symmetric eigenproblem is a library call, not a kernel of ours: rocSOLVER's dsyevd (the
// device counterpart of the LAPACK routine behind the reference's np.linalg.eigh).  rocSOLVER is
// bound lazily with dlopen on the first whitening call, so the propagation path has no dependency
// on it and hosts that never whiten never load it.  CLEORA_ROCSOLVER=<path> overrides the name.
#include <dlfcn.h>
#include <link.h>
#include <rocsolver/rocsolver.h>

#include <cstdlib>
#include <cstring>
#include <map>

#include "common.h"

namespace cleora {
// dxd_host.cpp: the d x d step on the host (0 = transform written, 1 = not safely positive definite)
int cholesky_whiten_host(const double *gram, double scale, uint32_t d, float *transform, double min_pivot2, double max_trace_inverse,
                         double *trace_inverse_out, double *min_pivot2_out);
namespace {

struct Solver {
    void *lib = nullptr;
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_destroy_handle) destroy = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocsolver_dsyevd) dsyevd = nullptr;
    decltype(&rocsolver_dpotrf) dpotrf = nullptr;
    decltype(&rocsolver_dtrtri) dtrtri = nullptr;
    std::string error;
    bool rocblas_was_resident = false;           // rocBLAS already mapped by the host before we loaded anything (see below)
    std::mutex mu;                               // one eigenproblem at a time per process
    std::map<int, rocblas_handle> handles;       // one rocBLAS handle per device, created on demand
};

Solver &solver() {
    static Solver s;
    static std::once_flag once;
    std::call_once(once, [] {
        dl_iterate_phdr(
            [](dl_phdr_info *info, size_t, void *found) {
                if (info->dlpi_name && std::strstr(info->dlpi_name, "librocblas")) *static_cast<bool *>(found) = true;
                return 0;
            },
            &s.rocblas_was_resident);
        const char *env = std::getenv("CLEORA_ROCSOLVER");
        const char *names[] = {env, "librocsolver.so.0", "librocsolver.so"};
        for (const char *name : names) {
            if (!name || !*name) continue;
            s.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (s.lib) break;
            s.error = dlerror();
        }
        if (!s.lib) return;
        // dlsym on a handle also searches the library's dependencies (rocBLAS)
        s.create = reinterpret_cast<decltype(s.create)>(dlsym(s.lib, "rocblas_create_handle"));
        s.destroy = reinterpret_cast<decltype(s.destroy)>(dlsym(s.lib, "rocblas_destroy_handle"));
        s.set_stream = reinterpret_cast<decltype(s.set_stream)>(dlsym(s.lib, "rocblas_set_stream"));
        s.dsyevd = reinterpret_cast<decltype(s.dsyevd)>(dlsym(s.lib, "rocsolver_dsyevd"));
        s.dpotrf = reinterpret_cast<decltype(s.dpotrf)>(dlsym(s.lib, "rocsolver_dpotrf"));
        s.dtrtri = reinterpret_cast<decltype(s.dtrtri)>(dlsym(s.lib, "rocsolver_dtrtri"));
        if (!s.create || !s.destroy || !s.set_stream || !s.dsyevd || !s.dpotrf || !s.dtrtri) {
            s.error = "rocSOLVER / rocBLAS entry points not found in the loaded library";
            s.lib = nullptr;
        }
    });
    return s;
}

__global__ __launch_bounds__(256) void mean_kernel(const double *__restrict__ colsum, uint64_t n, uint32_t d,
                                                   double *__restrict__ mean64, float *__restrict__ mean32) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const double m = colsum[c] / (double)n;      // np.mean(axis=0, dtype=float64)   (:136)
    mean64[c] = m;
    mean32[c] = (float)m;                        // mean.astype(np.float32)          (:159)
}

__global__ __launch_bounds__(256) void cov_kernel(const double *__restrict__ gram, uint64_t elems, double inv,
                                                  double *__restrict__ cov) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < elems) cov[i] = gram[i] * inv;       // cov /= (n - 1)                   (:143)
}

// dsyevd leaves ascending eigenvalues in w and eigenvector j in column j (column-major: v[i + j*d]).
// Output column j of the transform is eigenvector d-1-j (descending order, :147-149) scaled by
// 1/sqrt(max(lambda, 1e-10)) in f64, then cast to f32 (:155-156).
__global__ __launch_bounds__(256) void transform_kernel(const double *__restrict__ v, const double *__restrict__ w,
                                                        uint32_t d, uint32_t k, float *__restrict__ t,
                                                        double *__restrict__ w_desc) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < d && w_desc) w_desc[idx] = w[d - 1 - idx];
    if (idx >= (uint64_t)d * k) return;
    const uint32_t i = (uint32_t)(idx / k), j = (uint32_t)(idx % k);
    const uint32_t src = d - 1 - j;
    const double lam = w[src];
    const double scale = 1.0 / sqrt(lam > 1e-10 ? lam : 1e-10);
    t[idx] = (float)(v[(uint64_t)i + (uint64_t)src * d] * scale);
}

// Cholesky whitening (the rotation-equivalent transform of the intermediate iterations, launch_whiten_transform below):
// smallest squared pivot of the factor (a lower bound test for near-singularity) ...
__global__ __launch_bounds__(256) void min_pivot_kernel(const double *__restrict__ l, uint32_t d, double *__restrict__ out) {
    __shared__ double sm[256];
    double m = INFINITY;
    for (uint32_t i = threadIdx.x; i < d; i += 256) {
        const double p = l[(uint64_t)i * d + i];
        m = fmin(m, p * p);
    }
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// ... and the transform T = L^-T as f32: rocSOLVER worked on the column-major LOWER triangle, whose memory read row-major
// is the upper triangle of L^-T; the other triangle still holds the covariance and is zeroed here.
__global__ __launch_bounds__(256) void tri_transform_kernel(const double *__restrict__ linv, uint32_t d, float *__restrict__ t) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (uint64_t)d * d) return;
    const uint32_t i = (uint32_t)(idx / d), j = (uint32_t)(idx % d);
    t[idx] = j >= i ? (float)linv[idx] : 0.0f;
}

// ... and the guard that makes the Cholesky form safe to use in place of the reference's clamped PCA form: T = L^-T, so
// ||T||_F^2 = trace(L^-T L^-1) = trace(cov^-1) = sum_i 1/lambda_i >= 1/lambda_min.  If that sum is <= 1e10 then EVERY
// eigenvalue of the covariance is >= 1e-10, the reference's clamp max(lambda, 1e-10) (pycleora/__init__.py:155) is inactive
// and its transform is a whitening like ours (equal up to a rotation); otherwise the caller takes the PCA form, which
// reproduces the clamp.  (The smallest pivot alone only bounds lambda_min from ABOVE: ADVICE round 2.)  One block, fixed
// order: deterministic.
__global__ __launch_bounds__(256) void frob2_kernel(const float *__restrict__ t, uint64_t elems, double *__restrict__ out) {
    __shared__ double sm[256];
    double s = 0.0;
    for (uint64_t i = threadIdx.x; i < elems; i += 256) {
        const double v = (double)t[i];
        s += v * v;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sm[threadIdx.x] += sm[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}
}  // namespace
constexpr double kMaxTraceInverseForHost = 0.999e10;   // = kMaxTraceInverse below
namespace {
constexpr uint32_t kHostDxdMax = 256;            // the d x d step of an intermediate iteration runs on the host up to here
constexpr double kMaxTraceInverse = 0.999e10;   // sum 1/lambda_i <= this  =>  lambda_min >= 1e-10 (with a margin for the f32 T)

// ---- Cholesky whitening transform for d <= 256 in ONE launch, no library, no host synchronisation inside -------------
// cov = gram / (n-1) = L L^T ;  transform = L^-T as f32 (row-major d x d, upper triangular).
// One workgroup of 512 threads keeps a triangle of the matrix in REGISTERS (d (d+1)/2 <= 32 896 values; 512 KiB of f64
// would not fit the LDS) and runs two right-looking eliminations with one barrier per step:
//   phase 1, the factorisation.  Ownership by COLUMN pairs: columns q and d-1-q hold d+1 entries together; the four threads
//     of pair q own positions sub, sub+4, ... of the two columns laid end to end — at most 65 values per thread, and a
//     slot's coordinates cost one compare.  Step j: everyone reads the still-unscaled column j from LDS (published at
//     the end of step j-1 by its owners, one wave), a(i,k) -= c_i c_k / pivot with c_k / pivot hoisted (two per thread);
//     the owners of column j scale it (it is final: L[:, j]); the owners of column j+1 publish theirs.  Threads whose
//     columns are both final skip the step (whole waves retire as j advances).
//   phase 2, M = L^-1 by the same scheme on rows: M starts as I; step j: row j of M is final after / L[j][j], and
//     m(i,k) -= (L[i][j] / L[j][j]) m_j(k) for i > j.  Ownership by ROW pairs (rows q and d-1-q), the multiplier is per
//     row (two per thread), row j of M travels through LDS, column j of L is prefetched a step ahead from the
//     transposed copy phase 1 left in global scratch.  T[k][i] = M[i][k] leaves as f32.
// meta[0] = 0 ok / 1 a pivot was not positive, meta[1] = smallest pivot (= squared diagonal of L).  The ownership and
// update order were checked against numpy in an index-exact emulation before this was written (d = 1 ... 256).
constexpr int kCholSlots = 65, kCholThreads = 512, kCholPad = 8;

__global__ __launch_bounds__(kCholThreads) void cholesky_whiten_kernel(const double *__restrict__ gram, double inv_nm1,
                                                                       uint32_t d, double *__restrict__ lt,
                                                                       float *__restrict__ transform,
                                                                       double *__restrict__ meta) {
    // vb: phase 1, column j of the trailing matrix, unscaled; phase 2, row j of M, unscaled.  Padded on both sides so that a
    // slot's read address is one of two per-thread bases plus a constant (32 s) with no clamp: indices -8 .. 519 exist.
    __shared__ double vb_store[2][kCholPad + 520];
    __shared__ double lb[2][256];      // phase 2: column j of L
    double *const vb0 = vb_store[0] + kCholPad, *const vb1 = vb_store[1] + kCholPad;
    __shared__ int failed;
    const uint32_t t = threadIdx.x, q = t >> 2, sub = t & 3;
    __builtin_amdgcn_s_setprio(3);                                 // one latency-bound workgroup beside a chip full of SpMM waves
    const bool active = q < (d + 1) / 2;
    const uint32_t l0 = q, l1 = d - 1 - q;                         // the two lines (columns, then rows) of this thread
    double a[kCholSlots];

    // ------------------------------------------------ phase 1: L ------------------------------------------------
    uint32_t len0 = d - l0;                                        // column l0 holds rows l0..d-1; then column l1, rows l1..
    uint32_t len_all = active ? (l0 == l1 ? len0 : d + 1) : 0u;   // an odd d leaves the middle line unpaired
#pragma unroll
    for (int s = 0; s < kCholSlots; ++s) {
        const uint32_t p = (uint32_t)s * 4 + sub;
        const bool second = p >= len0;
        const uint32_t i = second ? p - 1 : l0 + p, k = second ? l1 : l0;
        a[s] = p < len_all ? gram[(uint64_t)i * d + k] * inv_nm1 : 0.0;
    }
    for (uint32_t e = t; e < d * d; e += kCholThreads) transform[e] = 0.0f;
    if (t == 0) failed = 0;
    for (uint32_t e = t; e < kCholPad + 520; e += kCholThreads) vb_store[0][e] = vb_store[1][e] = 0.0;   // finite everywhere:
    if (t < 256) lb[0][t] = lb[1][t] = 0.0;                                                              // retired slots use 0 x it
    __syncthreads();
    if (active && l0 == 0) {                                       // column 0
#pragma unroll
        for (int s = 0; s < kCholSlots; ++s) {
            const uint32_t p = (uint32_t)s * 4 + sub;
            if (p < len_all && p < len0) vb0[p] = a[s];
        }
    }
    double min_pivot = INFINITY;
    for (uint32_t j = 0; j < d; ++j) {
        __syncthreads();
        const double *c = (j & 1) ? vb1 : vb0;
        double *cn = (j & 1) ? vb0 : vb1;
        const double pivot = c[j];
        if (!(pivot > 0.0) && t == 0) failed = 1;                  // no early exit (a second loop exit made the compiler copy
                                                                   // all 65 values every step): the NaNs that follow are discarded
        min_pivot = fmin(min_pivot, pivot);
        const double inv = 1.0 / pivot, rs = 1.0 / sqrt(pivot);
        // opaque copies keep the compiler from hoisting 65 sets of slot coordinates (and their predicate masks) out of
        // the step loop, which spilled 350 registers
        uint32_t sub_j = sub, len0_j = len0, l0_j = l0, len_j = len_all;
        asm volatile("" : "+v"(sub_j), "+v"(len0_j), "+v"(l0_j), "+v"(len_j));
        // a wave-uniform branch (lanes whose columns are final run with multipliers 0): a divergent one made the compiler
        // keep two copies of the 65 values across it
        const bool go = active && l1 > j;
        if (__builtin_amdgcn_ballot_w64(go) != 0) {
            const double m0 = (go && l0 > j) ? c[l0] * inv : 0.0, m1 = go ? c[l1] * inv : 0.0;
            // LDS reads in batches of 13 (5 x 13 = 65): issued back to back, then consumed — left alone the compiler
            // waits for every single read before its multiply, ~100 cycles x 65 per step.  Address = one of two bases
            // (first column: row l0 + p, second: row p - 1) + the constant 4 s: compare, select, read.
            const int c0 = (int)(l0_j + sub_j), c1 = (int)sub_j - 1;
#pragma unroll
            for (int s0 = 0; s0 < kCholSlots; s0 += 13) {
                double cv[13];
#pragma unroll
                for (int u = 0; u < 13; ++u) cv[u] = c[((uint32_t)(4 * (s0 + u)) + sub_j >= len0_j ? c1 : c0) + 4 * (s0 + u)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 13; ++u)
                    a[s0 + u] = __builtin_fma(-((uint32_t)(4 * (s0 + u)) + sub_j >= len0_j ? m1 : m0), cv[u], a[s0 + u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (active && (l0 == j || l1 == j)) {                      // column j is final: scale it
            const bool which = l0 != j;
#pragma unroll
            for (int s = 0; s < kCholSlots; ++s) {
                const uint32_t p = (uint32_t)s * 4 + sub_j;
                if (p < len_j && (p >= len0_j) == which) a[s] *= rs;
            }
        }
        if (active && (l0 == j + 1 || l1 == j + 1)) {              // publish column j+1, updated and unscaled
            const bool which = l0 != j + 1;
#pragma unroll
            for (int s = 0; s < kCholSlots; ++s) {
                const uint32_t p = (uint32_t)s * 4 + sub_j;
                const bool second = p >= len0_j;
                if (p < len_j && second == which) cn[second ? p - 1 : l0_j + p] = a[s];
            }
        }
    }
    __syncthreads();
    const bool bad = failed != 0;
#pragma unroll
    for (int s = 0; s < kCholSlots; ++s) {                         // lt[k][i] = L[i][k]: column k of L is contiguous
        const uint32_t p = (uint32_t)s * 4 + sub;
        const bool second = p >= len0;
        if (p < len_all) lt[(uint64_t)(second ? l1 : l0) * d + (second ? p - 1 : l0 + p)] = a[s];
    }
    if (t == 0) {
        meta[0] = bad ? 1.0 : 0.0;
        meta[1] = min_pivot;
    }
    __threadfence_block();
    __syncthreads();
    if (bad) return;

    // ---------------------------------------------- phase 2: M = L^-1 ----------------------------------------------
    len0 = l0 + 1;                                                 // row l0 holds columns 0..l0; then row l1, columns 0..l1
    len_all = active ? (l0 == l1 ? len0 : d + 1) : 0u;
#pragma unroll
    for (int s = 0; s < kCholSlots; ++s) {
        const uint32_t p = (uint32_t)s * 4 + sub;
        const bool second = p >= len0;
        a[s] = (p < len_all && (second ? p - len0 : p) == (second ? l1 : l0)) ? 1.0 : 0.0;
    }
    for (uint32_t e = t; e < kCholPad + 520; e += kCholThreads) vb_store[0][e] = vb_store[1][e] = 0.0;
    if (t < 256) lb[0][t] = t < d ? lt[t] : 0.0;                   // column 0 of L
    __syncthreads();
    if (t == 0) vb0[0] = 1.0;                                      // row 0 of M (unscaled) = e_0
    for (uint32_t j = 0; j < d; ++j) {
        __syncthreads();
        const double *r = (j & 1) ? vb1 : vb0, *lc = lb[j & 1];
        double *rn = (j & 1) ? vb0 : vb1, *lcn = lb[(j + 1) & 1];
        const double l_next = (t < d && j + 1 < d) ? lt[(uint64_t)(j + 1) * d + t] : 0.0;   // consumed at the end of the step
        const double rinv = 1.0 / lc[j];
        uint32_t sub_j = sub, len0_j = len0, len_j = len_all;
        asm volatile("" : "+v"(sub_j), "+v"(len0_j), "+v"(len_j));
        const bool go = active && l1 > j;
        if (__builtin_amdgcn_ballot_w64(go) != 0) {
            const double m0 = (go && l0 > j) ? lc[l0] * rinv : 0.0, m1 = go ? lc[l1] * rinv : 0.0;
            const int r0 = (int)sub_j, r1 = (int)sub_j - (int)len0_j;  // first row: column p, second: column p - len0
#pragma unroll
            for (int s0 = 0; s0 < kCholSlots; s0 += 13) {
                double rv[13];
#pragma unroll
                for (int u = 0; u < 13; ++u) rv[u] = r[((uint32_t)(4 * (s0 + u)) + sub_j >= len0_j ? r1 : r0) + 4 * (s0 + u)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 13; ++u)
                    a[s0 + u] = __builtin_fma(-((uint32_t)(4 * (s0 + u)) + sub_j >= len0_j ? m1 : m0), rv[u], a[s0 + u]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (active && (l0 == j || l1 == j)) {                      // row j is final
            const bool which = l0 != j;
#pragma unroll
            for (int s = 0; s < kCholSlots; ++s) {
                const uint32_t p = (uint32_t)s * 4 + sub_j;
                if (p < len_j && (p >= len0_j) == which) a[s] *= rinv;
            }
        }
        if (active && (l0 == j + 1 || l1 == j + 1)) {              // publish row j+1
            const bool which = l0 != j + 1;
#pragma unroll
            for (int s = 0; s < kCholSlots; ++s) {
                const uint32_t p = (uint32_t)s * 4 + sub_j;
                const bool second = p >= len0_j;
                if (p < len_j && second == which) rn[second ? p - len0_j : p] = a[s];
            }
        }
        if (t < 256) lcn[t] = l_next;
    }
    uint32_t sub_e = sub, len0_e = len0, len_e = len_all;          // opaque again: 65 store addresses computed ahead of the
    asm volatile("" : "+v"(sub_e), "+v"(len0_e), "+v"(len_e));     // loop would live across it in scratch
#pragma unroll
    for (int s = 0; s < kCholSlots; ++s) {                         // T[k][i] = M[i][k]
        const uint32_t p = (uint32_t)s * 4 + sub_e;
        const bool second = p >= len0_e;
        if (p < len_e) transform[(uint64_t)(second ? p - len0_e : p) * d + (second ? l1 : l0)] = (float)a[s];
    }
}

inline uint64_t align256(uint64_t b) { return (b + 255) / 256 * 256; }

struct TransformWs {      // carved out of the caller's workspace
    double *cov, *w, *e;
    int *info;
};
inline uint64_t transform_ws_bytes(uint32_t d) {
    return align256((uint64_t)d * d * 8) + 2 * align256((uint64_t)d * 8) + 256;
}
inline TransformWs carve_transform(void *ws, uint32_t d) {
    char *p = static_cast<char *>(ws);
    TransformWs t;
    t.cov = reinterpret_cast<double *>(p); p += align256((uint64_t)d * d * 8);
    t.w = reinterpret_cast<double *>(p); p += align256((uint64_t)d * 8);
    t.e = reinterpret_cast<double *>(p); p += align256((uint64_t)d * 8);
    t.info = reinterpret_cast<int *>(p);
    return t;
}

}  // namespace

uint64_t eigh_workspace(uint32_t d) { return transform_ws_bytes(d); }

int launch_mean(const double *colsum, uint64_t n, uint32_t d, double *mean64, float *mean32, hipStream_t stream) {
    CL_REQUIRE(colsum != nullptr && mean64 != nullptr && mean32 != nullptr, "colsum / mean is NULL");
    CL_REQUIRE(n > 0 && d > 0, "n and d must be positive");
    hipLaunchKernelGGL(mean_kernel, dim3((d + 255) / 256), dim3(256), 0, stream, colsum, n, d, mean64, mean32);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

// Any W with W^T cov W = I whitens; the PCA form (eigenvectors, :145-156) is what the reference RETURNS, but inside the
// loop E <- whiten(l2_normalise(A E)) every such W leads to the same final result: two whitenings differ by an orthogonal
// factor R on the right, the SpMM and the row-wise L2 normalisation commute with R, and the PCA whitening of the last
// iteration removes it.  The Cholesky form W = L^-T (cov = L L^T) costs potrf + trtri — a handful of launches — instead
// of dsyevd's ~d dependent steps (6 ms at d = 256, two thirds of a whitening at |V| = 1M).  Returns 1 (not an error) when
// the covariance is too close to singular for that (pivot^2 < 1e-8, near the reference's 1e-10 eigenvalue clamp, or potrf
// reports a non-positive pivot) or when trace(cov^-1) = ||L^-T||_F^2 > 1e10, i.e. whenever lambda_min >= 1e-10 is not
// PROVEN: the caller then takes the eigenvector form, which reproduces the clamp.
int launch_whiten_transform_cholesky(const double *gram, uint64_t n, uint32_t d, float *transform, void *workspace,
                                     hipStream_t stream) {
    CL_REQUIRE(gram != nullptr && transform != nullptr && workspace != nullptr, "gram / transform / workspace is NULL");
    CL_REQUIRE(n >= 2 && d > 0 && d <= (1u << 15), "bad shape");
    int device = 0;
    CL_HIP(hipGetDevice(&device));
    const TransformWs w = carve_transform(workspace, d);
    const uint64_t elems = (uint64_t)d * d;
    // Two routes to the same transform for d <= 256 (larger d: rocSOLVER only).
    //   "library": rocSOLVER's potrf + trtri — many small launches that slip in beside the SpMM of the overlapped loop
    //     (7.96 ms / 53.3 ms per whitened iteration at BASELINE configs 2 / 3), but they sit on rocBLAS, whose first use in
    //     a process loads its kernel library: seconds when warm, 6 minutes measured in a torch-free C host on a cold box.
    //   "kernel": the single launch above — no rocBLAS, but it needs a whole CU's registers and therefore starts only
    //     when the SpMM beside it drains (9.9 ms / 54.7 ms).
    // Default: the library where the host process had rocBLAS mapped already (a PyTorch host), the kernel elsewhere.
    // CLEORA_CHOLESKY=library|kernel overrides (read per call).
    //   "host" (round 3, the default for d <= 256): Gram to the host, factorisation on one host core (dxd_host.cpp, ~1 ms at
    //     d = 256), transform back — nothing on the GPU at all: the ~215 small launches of the library route cost 3.5 ms of
    //     launch latency per iteration (more than config 2's whole SpMM) and slow the SpMM beside them by ~2 ms at config 3.
    bool library_route = solver().rocblas_was_resident;
    bool host_route = d <= kHostDxdMax;
    if (const char *env = std::getenv("CLEORA_CHOLESKY")) {
        host_route = d <= kHostDxdMax && std::strcmp(env, "host") == 0;
        library_route = std::strcmp(env, "kernel") != 0;
    }
    if (host_route) {
        std::vector<double> g(elems);
        std::vector<float> t(elems);
        CL_HIP(hipMemcpyAsync(g.data(), gram, elems * sizeof(double), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipStreamSynchronize(stream));
        const int bad = cholesky_whiten_host(g.data(), 1.0 / (double)(n - 1), d, t.data(), 1e-8, kMaxTraceInverse, nullptr, nullptr);
        CL_HIP(hipMemsetAsync(w.info, 0, sizeof(int), stream));            // whiten_info(): nothing failed to converge
        if (bad) return 1;
        CL_HIP(hipMemcpyAsync(transform, t.data(), elems * sizeof(float), hipMemcpyHostToDevice, stream));
        CL_HIP(hipStreamSynchronize(stream));                              // `t` dies with this call
        return CLEORA_OK;
    }
    if (d <= 256 && !library_route) {
        hipLaunchKernelGGL(cholesky_whiten_kernel, dim3(1), dim3(kCholThreads), 0, stream, gram, 1.0 / (double)(n - 1), d, w.cov,
                           transform, w.w);
        hipLaunchKernelGGL(frob2_kernel, dim3(1), dim3(256), 0, stream, transform, elems, w.w + 2);
        CL_HIP(hipGetLastError());
        double meta[3] = {1.0, 0.0, INFINITY};
        CL_HIP(hipMemcpyAsync(meta, w.w, sizeof(meta), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipStreamSynchronize(stream));
        CL_HIP(hipMemsetAsync(w.info, 0, sizeof(int), stream));            // whiten_info(): nothing failed to converge
        return (meta[0] != 0.0 || !(meta[1] >= 1e-8) || !(meta[2] <= kMaxTraceInverse)) ? 1 : CLEORA_OK;
    }
    Solver &s = solver();
    if (!s.lib) {
        set_error("whitening needs rocSOLVER (dlopen failed: " + s.error + "); set CLEORA_ROCSOLVER to its path");
        return CLEORA_E_HIP;
    }
    hipLaunchKernelGGL(cov_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, gram, elems,
                       1.0 / (double)(n - 1), w.cov);
    CL_HIP(hipGetLastError());
    int info_potrf = 0;
    double min_pivot2 = 0.0;
    {
        std::lock_guard<std::mutex> lock(s.mu);
        rocblas_handle &h = s.handles[device];
        if (!h && s.create(&h) != rocblas_status_success) {
            h = nullptr;
            set_error("rocblas_create_handle failed");
            return CLEORA_E_HIP;
        }
        if (s.set_stream(h, stream) != rocblas_status_success) {
            set_error("rocblas_set_stream failed");
            return CLEORA_E_HIP;
        }
        if (s.dpotrf(h, rocblas_fill_lower, (rocblas_int)d, w.cov, (rocblas_int)d, w.info) != rocblas_status_success) {
            set_error("rocsolver_dpotrf failed");
            return CLEORA_E_HIP;
        }
        hipLaunchKernelGGL(min_pivot_kernel, dim3(1), dim3(256), 0, stream, w.cov, d, w.w);
        CL_HIP(hipMemcpyAsync(&info_potrf, w.info, sizeof(int), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipMemcpyAsync(&min_pivot2, w.w, sizeof(double), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipStreamSynchronize(stream));
        if (info_potrf != 0 || !(min_pivot2 >= 1e-8)) return 1;              // not safely positive definite
        if (s.dtrtri(h, rocblas_fill_lower, rocblas_diagonal_non_unit, (rocblas_int)d, w.cov, (rocblas_int)d, w.info) !=
            rocblas_status_success) {
            set_error("rocsolver_dtrtri failed");
            return CLEORA_E_HIP;
        }
    }
    hipLaunchKernelGGL(tri_transform_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, w.cov, d, transform);
    // (w.info was overwritten by trtri with 0: whiten_info() keeps reporting success for this workspace)
    // the clamp guard: trace(cov^-1) = ||T||_F^2 <= 1e10 proves lambda_min >= 1e-10 (frob2_kernel)
    hipLaunchKernelGGL(frob2_kernel, dim3(1), dim3(256), 0, stream, transform, elems, w.w + 1);
    CL_HIP(hipGetLastError());
    double trace_inv = INFINITY;
    CL_HIP(hipMemcpyAsync(&trace_inv, w.w + 1, sizeof(double), hipMemcpyDeviceToHost, stream));
    CL_HIP(hipStreamSynchronize(stream));
    return trace_inv <= kMaxTraceInverse ? CLEORA_OK : 1;
}

int launch_whiten_transform(const double *gram, uint64_t n, uint32_t d, uint32_t k, float *transform,
                            double *eigenvalues, void *workspace, hipStream_t stream) {
    CL_REQUIRE(gram != nullptr && transform != nullptr && workspace != nullptr, "gram / transform / workspace is NULL");
    CL_REQUIRE(n >= 2, "whitening needs at least two rows");
    CL_REQUIRE(d > 0 && k > 0 && k <= d, "need 0 < k <= d");
    CL_REQUIRE(d <= (1u << 15), "d too large for the eigensolver");
    Solver &s = solver();
    if (!s.lib) {
        set_error("whitening needs rocSOLVER (dlopen failed: " + s.error + "); set CLEORA_ROCSOLVER to its path");
        return CLEORA_E_HIP;
    }
    int device = 0;
    CL_HIP(hipGetDevice(&device));
    const TransformWs w = carve_transform(workspace, d);
    const uint64_t elems = (uint64_t)d * d;
    hipLaunchKernelGGL(cov_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, gram, elems,
                       1.0 / (double)(n - 1), w.cov);
    CL_HIP(hipGetLastError());
    {
        std::lock_guard<std::mutex> lock(s.mu);
        rocblas_handle &h = s.handles[device];
        if (!h && s.create(&h) != rocblas_status_success) {
            h = nullptr;
            set_error("rocblas_create_handle failed");
            return CLEORA_E_HIP;
        }
        if (s.set_stream(h, stream) != rocblas_status_success) {
            set_error("rocblas_set_stream failed");
            return CLEORA_E_HIP;
        }
        // symmetric input: row-major and column-major coincide; eigenvectors come back column-major
        const rocblas_status st = s.dsyevd(h, rocblas_evect_original, rocblas_fill_upper, (rocblas_int)d, w.cov,
                                           (rocblas_int)d, w.w, w.e, w.info);
        if (st != rocblas_status_success) {
            set_error("rocsolver_dsyevd failed with status " + std::to_string((int)st));
            return CLEORA_E_HIP;
        }
    }
    const uint64_t cells = (uint64_t)d * k > d ? (uint64_t)d * k : d;
    hipLaunchKernelGGL(transform_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, w.cov, w.w, d, k,
                       transform, eigenvalues);
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

// ---- the whole of whiten_embeddings on device buffers ------------------------------------------
namespace {
struct WhitenWs {
    double *colsum_ws, *colsum, *mean64, *shift64, *gram_ws, *gram;
    float *mean32, *transform;
    void *eigh;
};
inline uint64_t whiten_ws_layout(uint64_t n, uint32_t d, void *base, WhitenWs *out) {
    uint64_t off = 0;
    auto take = [&](uint64_t bytes) {
        char *p = base ? static_cast<char *>(base) + off : nullptr;
        off += align256(bytes);
        return p;
    };
    char *a = take(colsum_workspace(n, d) * 8);
    char *b = take((uint64_t)d * 8);
    char *c = take((uint64_t)d * 8);
    char *c2 = take((uint64_t)d * 8);
    char *e = take(gram_workspace(n, d) * 8);
    char *f = take((uint64_t)d * d * 8);
    char *g = take((uint64_t)d * 4);
    char *h = take((uint64_t)d * d * 4);
    char *i = take(transform_ws_bytes(d));
    if (out) {
        out->colsum_ws = reinterpret_cast<double *>(a);
        out->colsum = reinterpret_cast<double *>(b);
        out->mean64 = reinterpret_cast<double *>(c);
        out->shift64 = reinterpret_cast<double *>(c2);
        out->gram_ws = reinterpret_cast<double *>(e);
        out->gram = reinterpret_cast<double *>(f);
        out->mean32 = reinterpret_cast<float *>(g);
        out->transform = reinterpret_cast<float *>(h);
        out->eigh = i;
    }
    return off;
}
}  // namespace

uint64_t whiten_workspace(uint64_t n, uint32_t d) { return whiten_ws_layout(n, d, nullptr, nullptr); }

// ---- optional per-stage timing of launch_whiten (cleora_whiten_set_timing) ---------------------------------
// 5 events per call on the launch stream: [statistics | Gram | transform (eigh) | projection].
namespace {
struct WhitenTiming {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> pool, used;
};
WhitenTiming &wt() {
    static WhitenTiming t;
    return t;
}
void wt_mark(hipStream_t stream) {
    WhitenTiming &t = wt();
    std::lock_guard<std::mutex> lock(t.mu);
    if (!t.on) return;
    hipEvent_t e = nullptr;
    if (!t.pool.empty()) {
        e = t.pool.back();
        t.pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        return;
    }
    t.used.push_back(e);
    (void)hipEventRecord(e, stream);
}
}  // namespace

int whiten_set_timing(bool enable) {
    WhitenTiming &t = wt();
    std::lock_guard<std::mutex> lock(t.mu);
    t.on = enable;
    return CLEORA_OK;
}

int whiten_get_timing(double ms[4], uint64_t *calls) {
    WhitenTiming &t = wt();
    std::lock_guard<std::mutex> lock(t.mu);
    for (int k = 0; k < 4; ++k) ms[k] = 0.0;
    *calls = t.used.size() / 5;
    if (!t.used.empty()) CL_HIP(hipEventSynchronize(t.used.back()));
    for (size_t i = 0; i + 4 < t.used.size(); i += 5)
        for (int k = 0; k < 4; ++k) {
            float v = 0.f;
            CL_HIP(hipEventElapsedTime(&v, t.used[i + k], t.used[i + k + 1]));
            ms[k] += (double)v;
        }
    t.pool.insert(t.pool.end(), t.used.begin(), t.used.end());
    t.used.clear();
    return CLEORA_OK;
}

// dsyevd's convergence flag of the last launch_whiten on this workspace (0 = converged), on the device
const int *whiten_info(void *workspace, uint64_t n, uint32_t d) {
    WhitenWs w;
    whiten_ws_layout(n, d, workspace, &w);
    return carve_transform(w.eigh, d).info;
}

// The fit in two halves, so that a caller can slip other launches between the MFMA-bound statistics and the
// eigensolver (whose library call synchronises with the host): launch_whiten_fit = stats + solve.
int launch_whiten_fit_stats(const float *x, uint64_t ldx, uint64_t n, uint32_t d, void *workspace, hipStream_t stream,
                            int gram_blocks_per_cu, bool intermediate) {
    CL_REQUIRE(d > 0 && ldx >= d && n >= 2, "bad shape");
    CL_REQUIRE(x != nullptr && workspace != nullptr, "x / workspace is NULL");
    WhitenWs w;
    whiten_ws_layout(n, d, workspace, &w);
    int rc;
    // One pass over X for mean AND covariance: centre with a shift c = the mean of <= 4096 rows sampled with a
    // constant stride (a 4 MB read), accumulate sum (x - c) beside the Gram of (x - c), and correct exactly:
    // mu = c + sum / n,  sum (x-mu)(x-mu)^T = sum (x-c)(x-c)^T - n (mu-c)(mu-c)^T.  With |mu - c| ~ sigma / 64 the
    // correction is 2e-4 of the diagonal, so nothing cancels: the result is the two-pass result to f64 rounding.
    const uint64_t m = n < 4096 ? n : 4096, stride = n / m;
    wt_mark(stream);
    if ((rc = launch_colsum(x, ldx * stride, m, d, w.colsum_ws, w.colsum, stream)) != CLEORA_OK) return rc;
    if ((rc = launch_mean(w.colsum, m, d, w.shift64, w.mean32, stream)) != CLEORA_OK) return rc;
    wt_mark(stream);
    // intermediate iterations of the whitened loop (the caller vouches that nobody looks at this whitening): the f32-matrix-core
    // Gram where it applies (whiten.hip, d = 256); everything else — the last iteration, cleora_whiten_dev — is f64 end to end
    if (intermediate && gram32_applies(x, ldx, n, d))
        rc = launch_gram32(x, ldx, n, d, w.shift64, w.mean32, w.gram_ws, w.gram, stream, w.mean64, w.mean32, gram_blocks_per_cu);
    else
        rc = launch_gram(x, ldx, n, d, w.shift64, w.gram_ws, w.gram, stream, w.mean64, w.mean32, gram_blocks_per_cu);
    if (rc != CLEORA_OK) return rc;
    wt_mark(stream);
    return CLEORA_OK;
}

int launch_whiten_fit_solve(uint64_t n, uint32_t d, uint32_t k, void *workspace, double *eigenvalues, hipStream_t stream,
                            bool any_whitening) {
    CL_REQUIRE(k >= 1 && k <= d && workspace != nullptr, "bad shape");
    WhitenWs w;
    whiten_ws_layout(n, d, workspace, &w);
    int rc = 1;
    if (any_whitening && k == d) {            // the cheap transform where the result does not depend on which one
        rc = launch_whiten_transform_cholesky(w.gram, n, d, w.transform, w.eigh, stream);
        if (rc < 0) return rc;
    }
    if (rc == 1) rc = launch_whiten_transform(w.gram, n, d, k, w.transform, eigenvalues, w.eigh, stream);
    if (rc != CLEORA_OK) return rc;
    wt_mark(stream);
    return CLEORA_OK;
}

int launch_whiten_fit(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t k, void *workspace,
                      double *eigenvalues, hipStream_t stream, int gram_blocks_per_cu) {
    const int rc = launch_whiten_fit_stats(x, ldx, n, d, workspace, stream, gram_blocks_per_cu);
    if (rc != CLEORA_OK) return rc;
    return launch_whiten_fit_solve(n, d, k, workspace, eigenvalues, stream);
}

int whiten_fit_copy_stats(void *workspace, uint64_t n, uint32_t d, double *mean64, double *gram, hipStream_t stream) {
    WhitenWs w;
    whiten_ws_layout(n, d, workspace, &w);
    CL_HIP(hipMemcpyAsync(mean64, w.mean64, (size_t)d * sizeof(double), hipMemcpyDeviceToDevice, stream));
    CL_HIP(hipMemcpyAsync(gram, w.gram, (size_t)d * d * sizeof(double), hipMemcpyDeviceToDevice, stream));
    return CLEORA_OK;
}

void whiten_fit_result(void *workspace, uint64_t n, uint32_t d, const float **mean32, const float **transform) {
    WhitenWs w;
    whiten_ws_layout(n, d, workspace, &w);
    *mean32 = w.mean32;
    *transform = w.transform;
}

int launch_whiten(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t k, float *y, uint64_t ldy,
                  void *workspace, double *eigenvalues, hipStream_t stream) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    if (n == 0) return CLEORA_OK;
    CL_REQUIRE(x != nullptr && y != nullptr, "x / y is NULL");
    CL_REQUIRE((const void *)x != (const void *)y, "x and y must not alias");
    if (n == 1) {                                 // `if n <= 1: return embeddings.copy()`   (:132-133)
        CL_REQUIRE(ldy >= d, "one row is returned unchanged: y needs d columns");
        CL_HIP(hipMemcpyAsync(y, x, (uint64_t)d * sizeof(float), hipMemcpyDeviceToDevice, stream));
        return CLEORA_OK;
    }
    if (k == 0 || k > d) k = d;                   // n_components=None, or >= d             (:151)
    CL_REQUIRE(ldy >= k, "bad output leading dimension");
    CL_REQUIRE(workspace != nullptr, "workspace is NULL");
    int rc = launch_whiten_fit(x, ldx, n, d, k, workspace, eigenvalues, stream);
    if (rc != CLEORA_OK) return rc;
    const float *mean32, *transform;
    whiten_fit_result(workspace, n, d, &mean32, &transform);
    rc = launch_project(x, ldx, n, d, mean32, transform, k, y, ldy, stream);
    wt_mark(stream);
    return rc;
}

}  // namespace cleora

extern "C" int cleora_cholesky_whiten_host(const double *gram_host, uint64_t n, uint32_t d, float *transform_host, double *trace_inverse_out) {
    if (!gram_host || !transform_host || n < 2 || d == 0) return -1;
    return cleora::cholesky_whiten_host(gram_host, 1.0 / (double)(n - 1), d, transform_host, 1e-8, cleora::kMaxTraceInverseForHost,
                                        trace_inverse_out, nullptr);
}
