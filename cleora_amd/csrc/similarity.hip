// similarity.hip — top-k cosine neighbours on the device: the `normed @ normed[src]`, masking and
// `argsort()[::-1][:top_k]` of predict_links / find_most_similar (pycleora/__init__.py:636-681, 753-781) for a BATCH of
// query rows per pass over X, with the selection on the device: no per-query launch + sync + n-float download.
//
//   scores   up to 8 queries: one wavefront per row of X, the query rows normalised into LDS, X read once per batch
//            (HBM-bound for one query, VALU-bound at 8);  more than 8 queries: the batch IS a GEMM — the normalised
//            query rows packed as a d x q matrix and X . Q on the f32 matrix cores through the projection kernel of
//            whiten.hip (up to 64 queries per pass over X), then one row-scale pass by 1 / ||x_r||;
//   mask     the query itself and, optionally, every r with a stored edge (q, r) or (r, q) get -2 like the reference
//            (one pass over the CSR per batch);
//   top-k    per 2048-element chunk k rounds of a block-wide arg-max in LDS (ties: the LARGER row index first, which
//            is what numpy's argsort()[::-1] yields), then the chunk winners are merged the same way.  For n >= 256 Ki
//            rows that selection runs on a SHORT LIST instead of on all n scores: a threshold from a stratified row
//            sample (the r-th largest of S sampled scores, r chosen so that fewer than k elements pass it with
//            probability < 1e-9), ONE pass over the scores that compacts everything at or above it (a few thousand
//            elements per query), the rounds on that list.  The pass count is checked on the host; a list shorter
//            than k or longer than its buffer sends that batch through the full selection: same result either way.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "common.h"

namespace cleora {
namespace {

constexpr int QB = 8;           // queries per pass, VALU form (scores laid out [query][row])
constexpr int QM = 256;         // queries per pass at most, MFMA form (scores laid out [row][query]); 64 for batches <= 64
constexpr float kMasked = -3.0e38f;   // what mask_kernel writes; the selection turns it into the reference's -2
constexpr int CHUNK = 2048;     // elements per selection block
constexpr uint32_t SHORT_MIN_N = 256 * 1024;   // the short-list selection from this many rows on
constexpr uint32_t SHORT_CAP = 32768;          // candidates per query the short list holds
constexpr uint32_t SHORT_RMAX = 64;            // largest sample rank the threshold may take
constexpr uint32_t SHORT_SMAX = 131072;        // largest sample per query

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// scores[q * n + r] = (x[r] . qn_q) / max(||x[r]||, 1e-10), qn_q = x[query_q] / max(||x[query_q]||, 1e-10)
template <bool W4>
__global__ __launch_bounds__(256) void cosine_batch_kernel(const float *__restrict__ x, uint64_t ldx, uint64_t n, uint32_t d,
                                                           const uint32_t *__restrict__ queries, uint32_t nq,
                                                           float *__restrict__ scores) {
    extern __shared__ float qs[];                       // [nq][d]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t q = w; q < nq; q += 4) {              // one wave normalises one query row into LDS
        const float *xq = x + (uint64_t)queries[q] * ldx;
        float sq = 0.f;
        for (uint32_t c = lane; c < d; c += 64) sq += xq[c] * xq[c];
        const float inv = 1.0f / fmaxf(sqrtf(wsum(sq)), 1e-10f);
        for (uint32_t c = lane; c < d; c += 64) qs[q * d + c] = xq[c] * inv;
    }
    __syncthreads();
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    for (uint64_t row = (uint64_t)blockIdx.x * 4 + w; row < n; row += waves) {
        const float *xr = x + row * ldx;
        float dot[QB], sq = 0.f;
#pragma unroll
        for (int q = 0; q < QB; ++q) dot[q] = 0.f;
        if constexpr (W4) {
            for (uint32_t c = lane * 4; c < d; c += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(xr + c);
                sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
                for (int q = 0; q < QB; ++q)
                    if ((uint32_t)q < nq) {
                        const float4 t = *reinterpret_cast<const float4 *>(qs + q * d + c);
                        dot[q] += v.x * t.x + v.y * t.y + v.z * t.z + v.w * t.w;
                    }
            }
        } else {
            for (uint32_t c = lane; c < d; c += 64) {
                const float v = xr[c];
                sq += v * v;
#pragma unroll
                for (int q = 0; q < QB; ++q)
                    if ((uint32_t)q < nq) dot[q] += v * qs[q * d + c];
            }
        }
        const float inv = 1.0f / fmaxf(sqrtf(wsum(sq)), 1e-10f);
#pragma unroll
        for (int q = 0; q < QB; ++q)
            if ((uint32_t)q < nq) {
                const float s = wsum(dot[q]);
                if (lane == 0) scores[(uint64_t)q * n + row] = s * inv;
            }
    }
}

// One bit per row of X: is it a query of this pass?  (Lets mask_kernel test an edge with one load instead of nq compares.)
__global__ void mark_queries_kernel(const uint32_t *__restrict__ queries, uint32_t nq, uint32_t *__restrict__ bits) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nq) atomicOr(&bits[queries[j] >> 5], 1u << (queries[j] & 31));
}

// Masked (the reference's -2, written as kMasked and turned into -2 by the selection): the query itself and both directions
// of every stored edge that touches it (:650-660).  Element (query q, row r) lives at scores[q * qs + r * rs].
__global__ __launch_bounds__(256) void mask_kernel(const uint64_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                   uint64_t n_rows, uint64_t qs, uint64_t rs, const uint32_t *__restrict__ queries,
                                                   uint32_t nq, const uint32_t *__restrict__ bits, int exclude_self,
                                                   int exclude_edges, float *__restrict__ scores) {
    __shared__ uint32_t qv[QM];
    for (uint32_t j = threadIdx.x; j < nq; j += 256) qv[j] = queries[j];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    if (exclude_self && blockIdx.x == 0)
        for (uint32_t j = threadIdx.x; j < nq; j += 256) scores[(uint64_t)j * qs + (uint64_t)qv[j] * rs] = kMasked;
    if (!exclude_edges) return;
    for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += waves) {
        const uint64_t b = rowptr[row], e = rowptr[row + 1];
        if ((bits[row >> 5] >> (row & 31)) & 1u) {            // (src, other): the row is a query — its whole edge list
            for (uint32_t q = 0; q < nq; ++q)
                if (qv[q] == (uint32_t)row)
                    for (uint64_t k = b + lane; k < e; k += 64) scores[(uint64_t)q * qs + (uint64_t)col[k] * rs] = kMasked;
        }
        for (uint64_t k = b + lane; k < e; k += 64) {
            const uint32_t c = col[k];
            if ((bits[c >> 5] >> (c & 31)) & 1u)              // (other, src)
                for (uint32_t q = 0; q < nq; ++q)
                    if (qv[q] == c) scores[(uint64_t)q * qs + row * rs] = kMasked;
        }
    }
}

// MFMA form, step 1: column j of the d x nq matrix = x[query_j] / max(||x[query_j]||, 1e-10)   (one wave per query)
__global__ __launch_bounds__(64) void pack_queries_kernel(const float *__restrict__ x, uint64_t ldx, uint32_t d,
                                                          const uint32_t *__restrict__ queries, uint32_t nq,
                                                          float *__restrict__ qmat) {
    const uint32_t j = blockIdx.x, lane = threadIdx.x;
    const float *xq = x + (uint64_t)queries[j] * ldx;
    float sq = 0.f;
    for (uint32_t c = lane; c < d; c += 64) sq += xq[c] * xq[c];
    const float inv = 1.0f / fmaxf(sqrtf(wsum(sq)), 1e-10f);
    for (uint32_t c = lane; c < d; c += 64) qmat[(uint64_t)c * nq + j] = xq[c] * inv;
}

// MFMA form: inv[r] = 1 / max(||x_r||, 1e-10), once per call; the selection applies it when it loads a raw score
__global__ __launch_bounds__(256) void inv_row_norm_kernel(const float *__restrict__ x, uint64_t ldx, uint64_t n, uint32_t d,
                                                           float *__restrict__ inv) {
    const int lane = threadIdx.x & 63;
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    for (uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += waves) {
        const float *xr = x + row * ldx;
        float sq = 0.f;
        for (uint32_t c = lane; c < d; c += 64) sq += xr[c] * xr[c];
        sq = wsum(sq);
        if (lane == 0) inv[row] = 1.0f / fmaxf(sqrtf(sq), 1e-10f);
    }
}

// One block: the k best of its CHUNK elements of query q, descending, ties -> larger index.  Element p of query q is
// score_in[q * stride_in + p * elem_stride], times scale[p] if given (the MFMA form's raw dot products); kMasked
// becomes -2.  idx_in == nullptr: the element's position is its index (first level).  Q_FAST: the query is the fast
// grid dimension — for the [row][query] layout, where the blocks of one chunk share their cache lines.
// Output: out_score / out_index [q][block][k].
template <bool Q_FAST>
__global__ __launch_bounds__(256) void topk_chunk_kernel(const float *__restrict__ score_in, const uint32_t *__restrict__ idx_in,
                                                         const float *__restrict__ scale, uint64_t len, uint64_t stride_in,
                                                         uint64_t elem_stride, uint32_t k,
                                                         float *__restrict__ out_score, uint32_t *__restrict__ out_index,
                                                         uint64_t stride_out, const uint32_t *__restrict__ len_of_query) {
    __shared__ float sv[CHUNK];
    __shared__ uint32_t si[CHUNK];
    __shared__ float rv[4];
    __shared__ uint32_t ri[4], rp[4];
    const uint32_t q = Q_FAST ? blockIdx.x : blockIdx.y, chunk = Q_FAST ? blockIdx.y : blockIdx.x;
    const uint64_t base = (uint64_t)chunk * CHUNK;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (len_of_query) len = len_of_query[q] < len ? len_of_query[q] : len;     // a short list: its own length per query
    for (int i = t; i < CHUNK; i += 256) {
        const uint64_t p = base + i;
        const bool ok = p < len;
        float v = -INFINITY;
        if (ok) {
            v = score_in[q * stride_in + p * elem_stride];
            if (v == kMasked) v = -2.0f;                 // (-inf stays -inf: the padding of a short candidate list)
            else if (scale) v *= scale[p];
        }
        sv[i] = v;
        si[i] = ok ? (idx_in ? idx_in[q * stride_in + p] : (uint32_t)p) : 0u;
    }
    __syncthreads();
    float *os = out_score + q * stride_out + (uint64_t)chunk * k;
    uint32_t *oi = out_index + q * stride_out + (uint64_t)chunk * k;
    for (uint32_t round = 0; round < k; ++round) {
        float bv = -INFINITY;
        uint32_t bi = 0, bp = 0xffffffffu;
        for (int i = t; i < CHUNK; i += 256) {
            const float v = sv[i];
            const uint32_t id = si[i];
            // first live element, a larger score, or an equal score at a larger row index
            if (v > -INFINITY && (bp == 0xffffffffu || v > bv || (v == bv && id > bi))) { bv = v; bi = id; bp = (uint32_t)i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const uint32_t oid = (uint32_t)__shfl_xor((int)bi, o, 64), op = (uint32_t)__shfl_xor((int)bp, o, 64);
            const bool take = op != 0xffffffffu && (bp == 0xffffffffu || ov > bv || (ov == bv && oid > bi));
            if (take) { bv = ov; bi = oid; bp = op; }
        }
        if (lane == 0) { rv[w] = bv; ri[w] = bi; rp[w] = bp; }
        __syncthreads();
        if (t == 0) {
            float fv = rv[0];
            uint32_t fi = ri[0], fp = rp[0];
            for (int j = 1; j < 4; ++j) {
                const bool take = rp[j] != 0xffffffffu && (fp == 0xffffffffu || rv[j] > fv || (rv[j] == fv && ri[j] > fi));
                if (take) { fv = rv[j]; fi = ri[j]; fp = rp[j]; }
            }
            os[round] = fp == 0xffffffffu ? -INFINITY : fv;
            oi[round] = fi;
            if (fp != 0xffffffffu) sv[fp] = -INFINITY;       // taken: out of the next rounds
        }
        __syncthreads();
    }
}

// Short list, step 1: S scores of query q at stratified row positions (one per stratum of `stratum` rows, at a hashed
// offset inside it), as the selection sees them (-2 for masked, scaled).  grid (S / 256, nq).
__global__ __launch_bounds__(256) void sample_scores_kernel(const float *__restrict__ scores, const float *__restrict__ scale,
                                                            uint64_t qs, uint64_t rs, uint32_t S, uint32_t stratum,
                                                            float *__restrict__ sample) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
    if (i >= S) return;
    const uint64_t p = (uint64_t)i * stratum + ((i * 2654435761u) >> 8) % stratum;
    float v = scores[q * qs + p * rs];
    if (v == kMasked) v = -2.0f;
    else if (scale) v *= scale[p];
    sample[(uint64_t)q * S + i] = v;
}

// Short list, step 2: every (score, row) of query q with score >= tau[q * tau_stride] appended to the query's list (order
// arbitrary: the rounds that follow order by score and row index).  count[q] = how many passed, also beyond the capacity.
// Layout [query][row] (rs == 1): blockIdx.y = query.  Layout [row][query] (qs == 1): flat over the n * nq elements.
__global__ __launch_bounds__(256) void compact_kernel(const float *__restrict__ scores, const float *__restrict__ scale, uint64_t n,
                                                      uint64_t qs, uint64_t rs, uint32_t nq, const float *__restrict__ tau,
                                                      uint32_t tau_stride, float *__restrict__ cand_score,
                                                      uint32_t *__restrict__ cand_index, uint32_t *__restrict__ count) {
    const uint64_t threads = (uint64_t)gridDim.x * 256, first = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    auto visit = [&](uint32_t q, uint64_t r, float v, float t) {
        if (v == kMasked) v = -2.0f;
        else if (scale) v *= scale[r];
        if (v >= t) {
            const uint32_t slot = atomicAdd(&count[q], 1u);
            if (slot < SHORT_CAP) {
                cand_score[(uint64_t)q * SHORT_CAP + slot] = v;
                cand_index[(uint64_t)q * SHORT_CAP + slot] = (uint32_t)r;
            }
        }
    };
    if (rs == 1) {
        const uint32_t q = blockIdx.y;
        const float t = tau[(uint64_t)q * tau_stride];
        for (uint64_t r = first; r < n; r += threads) visit(q, r, scores[q * qs + r], t);
    } else {
        // element e = r * nq + q; the stride of the loop splits into whole rows and a remainder of queries
        const uint64_t dr = threads / nq;
        const uint32_t dq = (uint32_t)(threads - dr * nq);
        uint64_t r = first / nq;
        uint32_t q = (uint32_t)(first - r * nq);
        while (r < n) {
            visit(q, r, scores[r * nq + q], tau[(uint64_t)q * tau_stride]);
            r += dr;
            q += dq;
            if (q >= nq) { q -= nq; ++r; }
        }
    }
}

}  // namespace

static inline uint64_t chunks_of(uint64_t len) { return (len + CHUNK - 1) / CHUNK; }

constexpr uint32_t kMaxD = 16384;       // widest row the query staging takes

static inline uint32_t per_pass(uint32_t n_queries) {
    return n_queries <= (uint32_t)QB ? (uint32_t)QB : n_queries <= 64u ? 64u : (uint32_t)QM;
}

// The short list's plan for n rows and k results: sample size S (one score per stratum of n / S rows), the rank r of the
// threshold in the sample.  The number of the true top-k that land in a sample of fraction S / n is about Poisson(k S / n);
// the list holds all of them unless at least r do, so r is the smallest rank whose tail is below 1e-9 (and the host checks
// the count in any case).  Expected list length r n / S.  cleora_topk_set_route (thread-local): 1 switches it off, 2 takes it
// from 2048 rows on and for any batch (the tests force both forms on one input; 0 = this plan).
static inline uint64_t short_sample_size(uint64_t n) {               // n / 256 in whole chunks, within [CHUNK, SHORT_SMAX]
    const uint64_t S = ((n / 256 + CHUNK - 1) / CHUNK) * CHUNK;
    return S < (uint64_t)CHUNK ? (uint64_t)CHUNK : S > SHORT_SMAX ? (uint64_t)SHORT_SMAX : S;
}
// per calling thread (a process-global raced between threads / devices)
static thread_local int t_route = 0, t_last_route = 0;
struct ShortPlan { bool use; uint32_t S, stratum, r; };
static ShortPlan short_plan(uint64_t n, uint32_t k, uint32_t queries_per_pass) {
    ShortPlan p{false, 0, 0, 0};
    const bool forced = t_route == 2;
    if (t_route == 1 || n < (forced ? (uint64_t)CHUNK : (uint64_t)SHORT_MIN_N)) return p;
    // a handful of results for a handful of queries: the rounds over all chunks run side by side on the chip and cost less
    // than the short list's five launches and its host check (measured at n = 1M: break-even near 500 results per pass)
    if (!forced && (uint64_t)k * queries_per_pass < 512) return p;
    const uint64_t S = short_sample_size(n);
    p.S = (uint32_t)S;
    p.stratum = (uint32_t)(n / S);
    const double lambda = (double)k * (double)S / (double)n;
    double term = exp(-lambda), cdf = term;
    uint32_t r = 1;
    while (1.0 - cdf > 1e-9 && r <= SHORT_RMAX) { term *= lambda / r; cdf += term; ++r; }
    p.r = r;
    p.use = r <= SHORT_RMAX && 3.0 * r * (double)p.stratum <= (double)SHORT_CAP;
    return p;
}

int topk_last_route() { return t_last_route; }
int topk_set_route(int route) {
    if (route < 0 || route > 2) return -1;
    t_route = route;
    return 0;
}

// bytes: scores [per][n] + two candidate levels (score + index each) + the query bitmap + the short list (sample, list,
// thresholds, counts) + (MFMA form) 1/||x_r||, the d x per query matrix and a zero "mean" of d floats for the projection kernel
static inline uint64_t short_floats(uint64_t n, uint64_t per) {
    if (n < (uint64_t)CHUNK) return 0;
    return per * short_sample_size(n) + 2 * per * SHORT_CAP + 2 * per * SHORT_RMAX + per;
}
uint64_t topk_workspace_bytes(uint64_t n, uint32_t k, uint32_t n_queries) {
    const uint64_t per = per_pass(n_queries);
    const uint64_t l1 = chunks_of(n) * k;
    const uint64_t l2 = chunks_of(l1) * k;
    uint64_t floats = per * n + 2 * per * l1 + 2 * per * l2 + (n + 31) / 32 + short_floats(n, per);
    if (per > (uint64_t)QB) floats += n + (uint64_t)kMaxD * per + kMaxD;
    return floats * 4 + 1024;
}

int launch_topk_cosine(const cleora_graph *g, const float *x, uint64_t ldx, uint64_t n, uint32_t d,
                       const uint32_t *queries_dev, uint32_t n_queries, uint32_t k, int exclude_self, int exclude_edges,
                       uint32_t *out_index, float *out_score, void *workspace, hipStream_t stream) {
    CL_REQUIRE(d > 0 && ldx >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && queries_dev != nullptr && out_index != nullptr && out_score != nullptr && workspace != nullptr,
               "x / queries / out / workspace is NULL");
    CL_REQUIRE(k >= 1 && k <= 1024 && (uint64_t)k <= n, "need 1 <= k <= min(n, 1024)");
    CL_REQUIRE(n < (1ull << 32), "more than 2^32 rows");
    CL_REQUIRE(!exclude_edges || (g != nullptr && g->n_rows == n && g->n_cols == n), "exclude_existing needs the square graph of X");
    CL_REQUIRE(d <= kMaxD, "row too wide for the query staging (d <= 16384)");
    if (n_queries == 0) return CLEORA_OK;
    const bool w4 = (d % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
    const uint64_t slots = per_pass(n_queries);                       // what the workspace was sized for
    const bool mfma = slots > (uint64_t)QB;
    // queries per pass: VALU form — as many as fit 64 KiB of LDS, at most QB; MFMA form — 64 or 256
    uint32_t per = mfma ? (uint32_t)slots : (uint32_t)((64 * 1024) / ((uint64_t)d * sizeof(float)));
    if (!mfma && per > (uint32_t)QB) per = QB;
    float *scores = static_cast<float *>(workspace);
    const uint64_t l1 = chunks_of(n) * k, l2 = chunks_of(l1) * k;
    float *s1 = scores + slots * n;
    uint32_t *i1 = reinterpret_cast<uint32_t *>(s1 + slots * l1);
    float *s2 = reinterpret_cast<float *>(i1 + slots * l1);
    uint32_t *i2 = reinterpret_cast<uint32_t *>(s2 + slots * l2);
    uint32_t *bits = i2 + slots * l2;
    const uint64_t bit_words = (n + 31) / 32;
    // the short list: sample [slots][S], list (score, row) [slots][SHORT_CAP], thresholds (score, unused index) [slots][SHORT_RMAX], counts
    const ShortPlan plan = short_plan(n, k, n_queries < per ? n_queries : per);
    float *sample = reinterpret_cast<float *>(bits + bit_words);
    const uint64_t sfl = short_floats(n, slots);
    float *cs = sample + (sfl ? slots * short_sample_size(n) : 0);
    uint32_t *ci = reinterpret_cast<uint32_t *>(cs + slots * SHORT_CAP);
    float *ts = reinterpret_cast<float *>(ci + slots * SHORT_CAP);
    uint32_t *ti = reinterpret_cast<uint32_t *>(ts + slots * SHORT_RMAX);
    uint32_t *count = ti + slots * SHORT_RMAX;
    float *inv = reinterpret_cast<float *>(sample + sfl);             // MFMA form only from here on: [n]
    int route = -1;                                                   // 1 every batch by the short list, 0 none, 2 some
    // ... then [d][nq] and d zeros, on a 256-byte boundary (the projection's fast form wants an aligned mean)
    float *qmat = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(inv + n) + 255) & ~(uintptr_t)255);
    float *zeros = qmat + (uint64_t)kMaxD * slots;
    const unsigned grid = (unsigned)((n + 3) / 4 < 4096 ? (n + 3) / 4 : 4096);
    if (mfma) {
        CL_HIP(hipMemsetAsync(zeros, 0, (size_t)d * sizeof(float), stream));
        hipLaunchKernelGGL(inv_row_norm_kernel, dim3(grid), dim3(256), 0, stream, x, ldx, n, d, inv);
    }
    for (uint32_t q0 = 0; q0 < n_queries; q0 += per) {
        const uint32_t nq = n_queries - q0 < per ? n_queries - q0 : per;
        uint64_t qs, rs;                                              // element (query q, row r) at scores[q * qs + r * rs]
        if (mfma) {
            hipLaunchKernelGGL(pack_queries_kernel, dim3(nq), dim3(64), 0, stream, x, ldx, d, queries_dev + q0, nq, qmat);
            CL_HIP(hipGetLastError());
            const int rc = launch_project(x, ldx, n, d, zeros, qmat, nq, scores, nq, stream);
            if (rc != CLEORA_OK) return rc;
            qs = 1;
            rs = nq;
        } else {
            const size_t lds = (size_t)nq * d * sizeof(float);
            if (w4)
                hipLaunchKernelGGL(cosine_batch_kernel<true>, dim3(grid), dim3(256), lds, stream, x, ldx, n, d, queries_dev + q0, nq, scores);
            else
                hipLaunchKernelGGL(cosine_batch_kernel<false>, dim3(grid), dim3(256), lds, stream, x, ldx, n, d, queries_dev + q0, nq, scores);
            qs = n;
            rs = 1;
        }
        if (exclude_self || exclude_edges) {
            if (exclude_edges) {
                CL_HIP(hipMemsetAsync(bits, 0, (size_t)bit_words * 4, stream));
                hipLaunchKernelGGL(mark_queries_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, queries_dev + q0, nq, bits);
            }
            hipLaunchKernelGGL(mask_kernel, dim3(grid), dim3(256), 0, stream, exclude_edges ? g->rowptr : nullptr,
                               exclude_edges ? g->col : nullptr, exclude_edges ? g->n_rows : 0, qs, rs, queries_dev + q0, nq,
                               bits, exclude_self, exclude_edges, scores);
        }
        // selection: chunks -> chunk winners -> ... -> one block per query.  Element p of query q at sin[q * stride + p * estride].
        auto select = [&](const float *sin, const uint32_t *iin, const float *scale, uint64_t len, const uint32_t *len_dev,
                          uint64_t stride, uint64_t estride, uint32_t kk, float *final_score, uint32_t *final_index, uint64_t final_stride) {
            float *so = s1;
            uint32_t *io = i1;
            uint64_t ostride = l1;
            for (;;) {
                const uint64_t nb = chunks_of(len);
                if (nb == 1) {   // final: straight into the destination
                    hipLaunchKernelGGL(topk_chunk_kernel<false>, dim3(1, nq), dim3(256), 0, stream, sin, iin, scale, len, stride, estride, kk,
                                       final_score, final_index, final_stride, len_dev);
                    break;
                }
                if (estride != 1 && nb <= 65535)     // [row][query] layout: the blocks of one chunk next to each other
                    hipLaunchKernelGGL(topk_chunk_kernel<true>, dim3(nq, (unsigned)nb), dim3(256), 0, stream, sin, iin, scale, len, stride, estride, kk, so, io, ostride, len_dev);
                else
                    hipLaunchKernelGGL(topk_chunk_kernel<false>, dim3((unsigned)nb, nq), dim3(256), 0, stream, sin, iin, scale, len, stride, estride, kk, so, io, ostride, len_dev);
                sin = so;
                iin = io;
                scale = nullptr;
                len_dev = nullptr;                   // (blocks beyond a query's own length have written -inf entries)
                len = nb * kk;
                stride = ostride;
                estride = 1;
                // the two candidate buffers alternate (level 1 fits l2 again: it is smaller than level 0's output)
                if (so == s1) { so = s2; io = i2; ostride = l2; } else { so = s1; io = i1; ostride = l1; }
            }
        };
        const float *scale = mfma ? inv : nullptr;
        bool by_short_list = false;
        if (plan.use) {
            CL_HIP(hipMemsetAsync(count, 0, (size_t)nq * 4, stream));
            hipLaunchKernelGGL(sample_scores_kernel, dim3((plan.S + 255) / 256, nq), dim3(256), 0, stream, scores, scale, qs, rs, plan.S,
                               plan.stratum, sample);
            select(sample, nullptr, nullptr, plan.S, nullptr, plan.S, 1, plan.r, ts, ti, SHORT_RMAX);
            const uint64_t elems = rs == 1 ? n : n * nq;
            const unsigned gx = (unsigned)((elems + 1023) / 1024 < 2048 ? (elems + 1023) / 1024 : 2048);
            hipLaunchKernelGGL(compact_kernel, dim3(gx, rs == 1 ? nq : 1), dim3(256), 0, stream, scores, scale, n, qs, rs, nq,
                               ts + (plan.r - 1), SHORT_RMAX, cs, ci, count);
            CL_HIP(hipGetLastError());
            uint32_t passed[QM];
            CL_HIP(hipMemcpyAsync(passed, count, (size_t)nq * 4, hipMemcpyDeviceToHost, stream));
            CL_HIP(hipStreamSynchronize(stream));
            uint32_t longest = 0;
            by_short_list = true;
            for (uint32_t j = 0; j < nq; ++j) {
                if (passed[j] < k || passed[j] > SHORT_CAP) by_short_list = false;
                longest = passed[j] > longest ? passed[j] : longest;
            }
            if (by_short_list)
                select(cs, ci, nullptr, longest, count, SHORT_CAP, 1, k, out_score + (uint64_t)q0 * k, out_index + (uint64_t)q0 * k, (uint64_t)k);
        }
        if (!by_short_list)
            select(scores, nullptr, scale, n, nullptr, qs, rs, k, out_score + (uint64_t)q0 * k, out_index + (uint64_t)q0 * k, (uint64_t)k);
        route = route < 0 ? (by_short_list ? 1 : 0) : (route == (by_short_list ? 1 : 0) ? route : 2);
    }
    t_last_route = route < 0 ? 0 : route;
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
