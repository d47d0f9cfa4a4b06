// dxd_host.cpp — the d x d step of the INTERMEDIATE whitened iterations on the host, for small d (plain C++, no HIP).
//
// Inside E <- whiten(l2_normalise(A E)) any W with W^T C W = I serves (eigh.hip: the Cholesky form, W = L^-T for C = L L^T;
// reference op: the eigendecomposition + clamp of pycleora/__init__.py:145-156, of which this is a rotation when the clamp is
// inactive).  On the device that is rocSOLVER's potrf + trtri: ~215 launches of a few microseconds each at d = 256 — 3.5 ms of
// launch latency per iteration even on an idle chip, longer than the whole SpMM of BASELINE config 2, and every one of them
// interrupts the SpMM running beside it (profiles/r03r_loop_timeline.jsonl: the SpMM of config 3 takes 38.6 ms with the chain
// beside it, 36.7 without).  11 MFLOP of f64 at d = 256 are a millisecond of ONE host core, the host is idle while the
// SpMM runs, and the Gram it needs is 512 KiB: so for d <= 256 the factorisation runs here, between a D2H copy of the Gram and
// an H2D copy of the transform.  It also removes rocBLAS (and its cold start) from every iteration but the last.
//
// With U = L^T (upper, row-major) both phases are rows of axpys over contiguous memory — no dot-product reductions, so the
// compiler vectorises them without reassociation flags:
//   factor   C = U^T U, right-looking:  U[k][k] = sqrt(c_kk);  U[k][k+1..] /= U[k][k];  row i (i > k): C[i][i..] -= U[k][i] U[k][i..]
//   invert   T = U^-1 (upper), rows from the bottom:  T[i][i] = 1/U[i][i];  T[i][j>i] = -(sum_{k>i} U[i][k] T[k][j]) / U[i][i]
// (four source rows per pass over a destination row: the matrix is 512 KiB at d = 256, larger than the L1)
// and T = L^-T is the transform (d x d row-major, upper triangular), rounded to f32 on the way out.
#include <cmath>
#include <cstdint>
#include <vector>

namespace cleora {

// gram: d x d row-major f64 (symmetric; the upper triangle is read), scale = 1 / (n - 1).  transform: d x d row-major f32.
// Returns 0 and fills `transform`, or 1 if C is not SAFELY positive definite — a pivot that is not finite and positive, a
// squared pivot below min_pivot2, or trace(C^-1) = ||T||_F^2 above max_trace_inverse (which proves lambda_min >= 1e-10, i.e.
// that the reference's clamp np.maximum(eigenvalues, 1e-10), pycleora/__init__.py:155, is inactive): the caller then takes
// the PCA form for this iteration.  *trace_inverse_out / *min_pivot2_out (may be NULL): what the verdict was taken on.
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
#endif
int cholesky_whiten_host(const double *gram, double scale, uint32_t d, float *transform, double min_pivot2, double max_trace_inverse,
                         double *trace_inverse_out, double *min_pivot2_out) {
    const size_t n = d;
    std::vector<double> ubuf(n * n), tbuf(n * n, 0.0);
    double *U = ubuf.data(), *T = tbuf.data();
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i; j < n; ++j) U[i * n + j] = gram[i * n + j] * scale;
    double smallest = INFINITY;
    // factor, four pivot rows at a time: the rows of the block are finished one by one (their updates stay inside the block's
    // four rows), then every row below takes the four updates in ONE pass over its elements
    for (size_t kb = 0; kb < n; kb += 4) {
        const size_t ke = kb + 4 < n ? kb + 4 : n;
        for (size_t k = kb; k < ke; ++k) {
            const double p = U[k * n + k];
            if (!(p > 0.0) || !std::isfinite(p)) return 1;
            smallest = p < smallest ? p : smallest;       // p IS the squared pivot
            const double r = std::sqrt(p), inv = 1.0 / r;
            double *uk = U + k * n;
            uk[k] = r;
            for (size_t j = k + 1; j < n; ++j) uk[j] *= inv;
            for (size_t i = k + 1; i < ke; ++i) {
                const double f = uk[i];
                double *ui = U + i * n;
                for (size_t j = i; j < n; ++j) ui[j] -= f * uk[j];
            }
        }
        if (ke - kb == 4) {
            const double *u0 = U + kb * n, *u1 = u0 + n, *u2 = u1 + n, *u3 = u2 + n;
            for (size_t i = ke; i < n; ++i) {
                const double f0 = u0[i], f1 = u1[i], f2 = u2[i], f3 = u3[i];
                double *ui = U + i * n;
                for (size_t j = i; j < n; ++j) ui[j] -= (f0 * u0[j] + f1 * u1[j]) + (f2 * u2[j] + f3 * u3[j]);
            }
        }                                                 // (a short last block has no rows below it)
    }
    if (min_pivot2_out) *min_pivot2_out = smallest;
    if (!(smallest >= min_pivot2)) return 1;
    // invert, rows from the bottom; row i gathers the rows below it four at a time (row k of T starts at column k)
    double frob2 = 0.0;
    for (size_t ii = n; ii-- > 0;) {
        double *ti = T + ii * n;
        const double *ui = U + ii * n;
        size_t k = ii + 1;
        for (; k + 4 <= n; k += 4) {
            const double f0 = ui[k], f1 = ui[k + 1], f2 = ui[k + 2], f3 = ui[k + 3];
            const double *t0 = T + k * n, *t1 = t0 + n, *t2 = t1 + n, *t3 = t2 + n;
            ti[k] += f0 * t0[k];
            ti[k + 1] += f0 * t0[k + 1] + f1 * t1[k + 1];
            ti[k + 2] += (f0 * t0[k + 2] + f1 * t1[k + 2]) + f2 * t2[k + 2];
            for (size_t j = k + 3; j < n; ++j) ti[j] += (f0 * t0[j] + f1 * t1[j]) + (f2 * t2[j] + f3 * t3[j]);
        }
        for (; k < n; ++k) {
            const double f = ui[k];
            const double *tk = T + k * n;
            for (size_t j = k; j < n; ++j) ti[j] += f * tk[j];
        }
        const double inv = 1.0 / ui[ii];
        for (size_t j = ii + 1; j < n; ++j) ti[j] *= -inv;
        ti[ii] = inv;
        for (size_t j = ii; j < n; ++j) frob2 += ti[j] * ti[j];
    }
    if (trace_inverse_out) *trace_inverse_out = frob2;
    if (!(frob2 <= max_trace_inverse)) return 1;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) transform[i * n + j] = j >= i ? (float)T[i * n + j] : 0.0f;
    return 0;
}

}  // namespace cleora
