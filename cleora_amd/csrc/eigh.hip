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
#include <rocsolver/rocsolver.h>

#include <cstdlib>
#include <map>

#include "common.h"

namespace cleora {
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
    std::mutex mu;                               // one eigenproblem at a time per process
    std::map<int, rocblas_handle> handles;       // one rocBLAS handle per device, created on demand
};

Solver &solver() {
    static Solver s;
    static std::once_flag once;
    std::call_once(once, [] {
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

// ---- Cholesky whitening transform for d <= 256 in ONE launch, no library, no host synchronisation inside -------------
// cov = gram / (n-1) = L L^T ;  transform = L^-T as f32 (row-major d x d, upper triangular).
// One workgroup of 512 threads keeps the lower triangle of the matrix in REGISTERS (d (d+1)/2 <= 32 896 values; 512 KiB
// of f64 would not fit the LDS).  Ownership is by ROW PAIRS so that a slot's (i, k) costs one compare instead of a stored
// index: rows r and d-1-r together hold d+1 elements, four threads share a pair, thread `sub` of the four owns positions
// sub, sub+4, ... of the pair's concatenated rows — at most 65 values per thread.  The factorisation is right-looking
// with one barrier per column: the still-unscaled column j+1 is published to LDS by its owners at the end of step j,
// every thread reads the pivot from it and folds the scaling into its update, a -= c_i c_k / pivot.  L then goes to
// global scratch and thread j < d solves L m = e_j forward (column j of L^-1), writing it as row j of the transform.
// meta[0] = 0 ok / 1 a pivot was not positive, meta[1] = smallest pivot (= squared diagonal of L).
constexpr int kCholSlots = 65, kCholThreads = 512;

__global__ __launch_bounds__(kCholThreads) void cholesky_whiten_kernel(const double *__restrict__ gram, double inv_nm1,
                                                                       uint32_t d, double *__restrict__ lfull,
                                                                       float *__restrict__ transform,
                                                                       double *__restrict__ meta) {
    __shared__ double col[2][256];
    __shared__ int failed;
    const uint32_t t = threadIdx.x, pair = t >> 2, sub = t & 3;
    const uint32_t r0 = pair, r1 = d - 1 - pair;                   // r0 <= r1 for the pairs that exist
    const uint32_t len0 = r0 + 1;                                  // row r0 holds k = 0..r0
    const uint32_t len = pair < (d + 1) / 2 ? (r0 == r1 ? len0 : d + 1) : 0;
    double a[kCholSlots];
#pragma unroll
    for (int s = 0; s < kCholSlots; ++s) {
        const uint32_t p = (uint32_t)s * 4 + sub;
        const bool second = p >= len0;
        const uint32_t i = second ? r1 : r0, k = second ? p - len0 : p;
        a[s] = p < len ? gram[(uint64_t)i * d + k] * inv_nm1 : 0.0;
    }
    if (t == 0) failed = 0;
    if (len != 0 && sub == 0) {                                    // column 0 = position 0 of each row
        col[0][r0] = a[0];
        if (r1 != r0) col[0][r1] = gram[(uint64_t)r1 * d] * inv_nm1;
    }
    double min_pivot = INFINITY;
    for (uint32_t j = 0; j < d; ++j) {
        __syncthreads();
        const double *c = col[j & 1];
        double *cn = col[(j + 1) & 1];
        const double pivot = c[j];
        if (!(pivot > 0.0)) {                                      // uniform: every thread reads the same value
            if (t == 0) failed = 1;
            break;
        }
        min_pivot = fmin(min_pivot, pivot);
        const double inv = 1.0 / pivot, rs = 1.0 / sqrt(pivot);
        const double ci0 = len != 0 ? c[r0] * inv : 0.0, ci1 = len != 0 ? c[r1] * inv : 0.0;
        // the slot coordinates are two instructions each; opaque copies keep the compiler from hoisting 65 sets of them
        // (and their predicate masks) out of the column loop, which spilled 350 registers
        uint32_t sub_j = sub, len_j = len, len0_j = len0;
        asm volatile("" : "+v"(sub_j), "+v"(len_j), "+v"(len0_j));
#pragma unroll
        for (int s = 0; s < kCholSlots; ++s) {
            const uint32_t p = (uint32_t)s * 4 + sub_j;
            if (p >= len_j) continue;
            const bool second = p >= len0_j;
            const uint32_t k = second ? p - len0_j : p;
            if (k == j) {
                a[s] = a[s] * rs;                                  // final: L[i][j] (the diagonal becomes sqrt(pivot))
            } else if (k > j) {
                a[s] -= (second ? ci1 : ci0) * c[k];
                if (k == j + 1) cn[second ? r1 : r0] = a[s];       // the next column, still unscaled
            }
        }
    }
    __syncthreads();
    const bool bad = failed != 0;
#pragma unroll
    for (int s = 0; s < kCholSlots; ++s) {
        const uint32_t p = (uint32_t)s * 4 + sub;
        const bool second = p >= len0;
        if (p < len) lfull[(uint64_t)(second ? r1 : r0) * d + (second ? p - len0 : p)] = a[s];
    }
    for (uint32_t e = t; e < d * d; e += kCholThreads) transform[e] = 0.0f;
    if (t == 0) {
        meta[0] = bad ? 1.0 : 0.0;
        meta[1] = min_pivot;
    }
    __threadfence_block();
    __syncthreads();
    if (bad || t >= d) return;
    // Column t of M = L^-1 by forward substitution, M[i] = -(sum_{t <= k < i} L[i][k] M[k]) / L[i][i] for i > t.  The
    // column is parked in the unused upper triangle, row t of lfull (this thread is its only reader and writer, program
    // order suffices), and leaves as row t of the transform: T[t][i] = (L^-T)[t][i] = M[i][t].
    double *m = lfull + (uint64_t)t * d;
    const double m_tt = 1.0 / lfull[(uint64_t)t * d + t];
    transform[(uint64_t)t * d + t] = (float)m_tt;
    for (uint32_t i = t + 1; i < d; ++i) {
        const double *li = lfull + (uint64_t)i * d;
        double sum = li[t] * m_tt;
        for (uint32_t k = t + 1; k < i; ++k) sum += li[k] * m[k];
        const double v = -sum / li[i];
        m[i] = v;
        transform[(uint64_t)t * d + i] = (float)v;
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
// reports a non-positive pivot): the caller then takes the eigenvector form, which reproduces the clamp.
int launch_whiten_transform_cholesky(const double *gram, uint64_t n, uint32_t d, float *transform, void *workspace,
                                     hipStream_t stream) {
    CL_REQUIRE(gram != nullptr && transform != nullptr && workspace != nullptr, "gram / transform / workspace is NULL");
    CL_REQUIRE(n >= 2 && d > 0 && d <= (1u << 15), "bad shape");
    int device = 0;
    CL_HIP(hipGetDevice(&device));
    const TransformWs w = carve_transform(workspace, d);
    const uint64_t elems = (uint64_t)d * d;
    if (d <= 256) {
        // our own single-launch kernel: no rocBLAS underneath (whose first use in a process loads its whole kernel
        // library — minutes on a cold box), nothing that synchronises with the host except the 16-byte verdict
        hipLaunchKernelGGL(cholesky_whiten_kernel, dim3(1), dim3(kCholThreads), 0, stream, gram, 1.0 / (double)(n - 1), d, w.cov,
                           transform, w.w);
        CL_HIP(hipGetLastError());
        double meta[2] = {1.0, 0.0};
        CL_HIP(hipMemcpyAsync(meta, w.w, sizeof(meta), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipStreamSynchronize(stream));
        CL_HIP(hipMemsetAsync(w.info, 0, sizeof(int), stream));            // whiten_info(): nothing failed to converge
        return (meta[0] != 0.0 || !(meta[1] >= 1e-8)) ? 1 : CLEORA_OK;
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
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
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
                            int gram_blocks_per_cu) {
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
    if ((rc = launch_gram(x, ldx, n, d, w.shift64, w.gram_ws, w.gram, stream, w.mean64, w.mean32, gram_blocks_per_cu)) != CLEORA_OK) return rc;
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
