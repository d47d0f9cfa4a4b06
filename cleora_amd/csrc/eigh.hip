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
    std::mutex mu;                               // one eigenproblem at a time per process
    std::map<int, rocblas_handle> handles;       // one rocBLAS handle per device, created on demand
    // The handle owns the solvers' device workspace, so two problems on ONE handle must not overlap on the GPU either (the
    // mutex only serialises the enqueues): every call first makes its stream wait for the previous call's end.  Matters when
    // several shards of one process share a device (multi.hip's one-GPU form) or a host drives one device from two streams.
    std::map<int, hipEvent_t> last_use;
    int begin_use(int device, hipStream_t stream) {
        auto it = last_use.find(device);
        if (it != last_use.end()) CL_HIP(hipStreamWaitEvent(stream, it->second, 0));
        return CLEORA_OK;
    }
    int end_use(int device, hipStream_t stream) {
        hipEvent_t &e = last_use[device];
        if (!e) CL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CL_HIP(hipEventRecord(e, stream));
        return CLEORA_OK;
    }
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
// iteration removes it.  The Cholesky form W = L^-T (cov = L L^T) costs potrf + trtri instead of dsyevd's ~d dependent steps
// (6 ms at d = 256, two thirds of a whitening at |V| = 1M).  Returns 1 (not an error) whenever the Cholesky form is not
// PROVEN equivalent — the caller then takes the eigenvector form, which reproduces the reference's clamp:
//   * potrf reports a non-positive pivot, or the smallest squared pivot is < 1e-8;
//   * trace(cov^-1) = ||L^-T||_F^2 > 1e10: sum 1/lambda_i >= 1/lambda_min, so passing proves lambda_min >= 1e-10, i.e. that
//     np.maximum(eigenvalues, 1e-10) (pycleora/__init__.py:155) is inactive;
//   * approximate_gram (the Gram came from the split-bf16 form: ||dG||_2 <~ 1e-7 n trace(cov) / d — whiten.hip launch_gram32) and
//     sum_i (trace(cov) / d) / lambda_i > kMaxRelativeSpread = 1e4: in a direction of variance lambda the
//     transform built on that Gram is off by ~1e-7 (trace / d) / lambda, so a spread below 1e4 keeps W^T cov_exact W within
//     1e-3 of I in the worst direction (typically two orders less: the sum runs over d terms) — and with it the clamp check
//     holds for the exact covariance too (lambda_min >= 1e-4 trace / d >> 1e-10 + the Gram's error).
// Two routes: d <= 256 on one host core (dxd_host.cpp: Gram down, transform up, ~1 ms — the ~215 small launches of the library
// route cost 3.5 ms of launch latency per iteration and slow the SpMM beside them), wider matrices on rocSOLVER's potrf + trtri.
// (A third route, a single-launch in-house kernel, measured slower than both: scripts/rejected/cholesky_whiten_kernel.hip.txt.)
constexpr double kMaxRelativeSpread = 1e4;

__global__ __launch_bounds__(256) void diag_sum_kernel(const double *__restrict__ m, uint32_t d, double scale, double *__restrict__ out) {
    __shared__ double sm[256];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < d; i += 256) s += m[(uint64_t)i * d + i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0] * scale;
}

int launch_whiten_transform_cholesky(const double *gram, uint64_t n, uint32_t d, float *transform, void *workspace,
                                     hipStream_t stream, bool approximate_gram) {
    CL_REQUIRE(gram != nullptr && transform != nullptr && workspace != nullptr, "gram / transform / workspace is NULL");
    CL_REQUIRE(n >= 2 && d > 0 && d <= (1u << 15), "bad shape");
    int device = 0;
    CL_HIP(hipGetDevice(&device));
    const TransformWs w = carve_transform(workspace, d);
    const uint64_t elems = (uint64_t)d * d;
    const double scale = 1.0 / (double)(n - 1);
    auto trace_bound = [&](double trace_cov) {        // the largest trace(cov^-1) that passes both guards
        double bound = kMaxTraceInverse;
        if (approximate_gram && trace_cov > 0.0 && kMaxRelativeSpread * d / trace_cov < bound) bound = kMaxRelativeSpread * d / trace_cov;
        return bound;
    };
    if (d <= kHostDxdMax) {
        std::vector<double> g(elems);
        std::vector<float> t(elems);
        CL_HIP(hipMemcpyAsync(g.data(), gram, elems * sizeof(double), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipStreamSynchronize(stream));
        double trace_cov = 0.0;
        for (uint32_t i = 0; i < d; ++i) trace_cov += g[(uint64_t)i * d + i] * scale;
        const int bad = cholesky_whiten_host(g.data(), scale, d, t.data(), 1e-8, trace_bound(trace_cov), nullptr, nullptr);
        CL_HIP(hipMemsetAsync(w.info, 0, sizeof(int), stream));            // whiten_info(): nothing failed to converge
        if (bad) return 1;
        CL_HIP(hipMemcpyAsync(transform, t.data(), elems * sizeof(float), hipMemcpyHostToDevice, stream));
        CL_HIP(hipStreamSynchronize(stream));                              // `t` dies with this call
        return CLEORA_OK;
    }
    Solver &s = solver();
    if (!s.lib) {
        set_error("whitening needs rocSOLVER (dlopen failed: " + s.error + "); set CLEORA_ROCSOLVER to its path");
        return CLEORA_E_HIP;
    }
    hipLaunchKernelGGL(cov_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, gram, elems, scale, w.cov);
    hipLaunchKernelGGL(diag_sum_kernel, dim3(1), dim3(256), 0, stream, gram, d, scale, w.w + 2);      // trace(cov)
    CL_HIP(hipGetLastError());
    int info_potrf = 0;
    double min_pivot2 = 0.0, trace_cov = 0.0;
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
        if (int ru = s.begin_use(device, stream)) return ru;
        if (s.dpotrf(h, rocblas_fill_lower, (rocblas_int)d, w.cov, (rocblas_int)d, w.info) != rocblas_status_success) {
            set_error("rocsolver_dpotrf failed");
            return CLEORA_E_HIP;
        }
        hipLaunchKernelGGL(min_pivot_kernel, dim3(1), dim3(256), 0, stream, w.cov, d, w.w);
        CL_HIP(hipMemcpyAsync(&info_potrf, w.info, sizeof(int), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipMemcpyAsync(&min_pivot2, w.w, sizeof(double), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipMemcpyAsync(&trace_cov, w.w + 2, sizeof(double), hipMemcpyDeviceToHost, stream));
        CL_HIP(hipStreamSynchronize(stream));
        if (info_potrf != 0 || !(min_pivot2 >= 1e-8)) return 1;              // not safely positive definite
        if (s.dtrtri(h, rocblas_fill_lower, rocblas_diagonal_non_unit, (rocblas_int)d, w.cov, (rocblas_int)d, w.info) !=
            rocblas_status_success) {
            set_error("rocsolver_dtrtri failed");
            return CLEORA_E_HIP;
        }
        if (int ru = s.end_use(device, stream)) return ru;
    }
    hipLaunchKernelGGL(tri_transform_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream, w.cov, d, transform);
    // (w.info was overwritten by trtri with 0: whiten_info() keeps reporting success for this workspace)
    hipLaunchKernelGGL(frob2_kernel, dim3(1), dim3(256), 0, stream, transform, elems, w.w + 1);     // trace(cov^-1) = ||T||_F^2
    CL_HIP(hipGetLastError());
    double trace_inv = INFINITY;
    CL_HIP(hipMemcpyAsync(&trace_inv, w.w + 1, sizeof(double), hipMemcpyDeviceToHost, stream));
    CL_HIP(hipStreamSynchronize(stream));
    return trace_inv <= trace_bound(trace_cov) ? CLEORA_OK : 1;
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
        if (int ru = s.begin_use(device, stream)) return ru;
        // symmetric input: row-major and column-major coincide; eigenvectors come back column-major
        const rocblas_status st = s.dsyevd(h, rocblas_evect_original, rocblas_fill_upper, (rocblas_int)d, w.cov,
                                           (rocblas_int)d, w.w, w.e, w.info);
        if (st != rocblas_status_success) {
            set_error("rocsolver_dsyevd failed with status " + std::to_string((int)st));
            return CLEORA_E_HIP;
        }
        if (int ru = s.end_use(device, stream)) return ru;
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

// the same flag for a caller that owns the d x d step's workspace alone (eigh_workspace(d): sharded.hip)
const int *transform_info(void *eigh_ws, uint32_t d) { return carve_transform(eigh_ws, d).info; }

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
    // intermediate iterations of the whitened loop (the caller vouches that nobody looks at this whitening): the split-bf16
    // Gram where it applies (whiten.hip, d = 256 S); everything else — the last iteration, cleora_whiten_dev — is f64 end to end
    if (intermediate && gram32_applies(x, ldx, n, d))
        rc = launch_gram32(x, ldx, n, d, w.shift64, w.mean32, w.gram_ws, w.gram, stream, w.mean64, w.mean32);
    else
        rc = launch_gram(x, ldx, n, d, w.shift64, w.gram_ws, w.gram, stream, w.mean64, w.mean32, gram_blocks_per_cu);
    if (rc != CLEORA_OK) return rc;
    wt_mark(stream);
    return CLEORA_OK;
}

int launch_whiten_fit_solve(uint64_t n, uint32_t d, uint32_t k, void *workspace, double *eigenvalues, hipStream_t stream,
                            bool any_whitening, bool approximate_gram, bool *need_exact_gram) {
    CL_REQUIRE(k >= 1 && k <= d && workspace != nullptr, "bad shape");
    WhitenWs w;
    whiten_ws_layout(n, d, workspace, &w);
    if (need_exact_gram) *need_exact_gram = false;
    int rc = 1;
    if (any_whitening && k == d) {            // the cheap transform where the result does not depend on which one
        rc = launch_whiten_transform_cholesky(w.gram, n, d, w.transform, w.eigh, stream, approximate_gram);
        if (rc < 0) return rc;
    }
    if (rc == 1 && approximate_gram) {
        // the PCA form reproduces the reference's eigenvalue clamp, and an eigenvalue near the clamp is exactly where a Gram that is
        // only ~1e-8 accurate must not be trusted: the caller recomputes the statistics in f64 and comes back
        CL_REQUIRE(need_exact_gram != nullptr, "internal: an approximate Gram needs a caller that can recompute it");
        *need_exact_gram = true;
        return CLEORA_OK;
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
