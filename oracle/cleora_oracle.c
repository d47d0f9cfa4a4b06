/*
 * cleora_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the arithmetic on pycleora 3.2.1's Markov-propagation
 * hot path, written from the reference's behaviour (file:line citations are
 * relative to /root/reference).  It exists so that the HIP kernels can be
 * checked against the reference's operation order on the CPU.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (cleora_amd/) never does.
 *
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 *   -ffp-contract=off because rustc never contracts `a += v * b` into an FMA
 *   (src/embedding.rs:80-82), and neither must we.
 *
 * PARITY PIN: the SpMM restatement is pinned by the reference's four insta
 * snapshots (tests/snapshots/snapshot__tests__markov_{left,sym}_{01,02}.snap)
 * through tests/test_oracle_golden.py.  xxh64 is pinned by python-xxhash and
 * the XXH64 spec vectors.  init_value / FxHasher / l2 / embed_full have NO
 * reference-side golden vectors ("parity unpinned" by the reference; they are
 * pinned here only by derived regression values).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* XXH64, seed 0 — twox-hash 1.6.3 `XxHash64::default()` as used by    */
/* hash_entity (src/entity.rs:109-114).  Restated from the public      */
/* XXH64 specification (the crate itself is not under /root/reference).*/
/* ------------------------------------------------------------------ */
#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; } /* LE host */
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t xxround(uint64_t acc, uint64_t in) {
    acc += in * P2; acc = rotl64(acc, 31); return acc * P1;
}
static inline uint64_t xxmerge(uint64_t h, uint64_t v) {
    v = xxround(0, v); h ^= v; return h * P1 + P4;
}

uint64_t oracle_xxh64(const uint8_t *p, uint64_t len, uint64_t seed) {
    const uint8_t *end = p + len;
    uint64_t h;
    if (len >= 32) {
        const uint8_t *limit = end - 32;
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = xxround(v1, rd64(p)); p += 8;
            v2 = xxround(v2, rd64(p)); p += 8;
            v3 = xxround(v3, rd64(p)); p += 8;
            v4 = xxround(v4, rd64(p)); p += 8;
        } while (p <= limit);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxmerge(h, v1); h = xxmerge(h, v2); h = xxmerge(h, v3); h = xxmerge(h, v4);
    } else {
        h = seed + P5;
    }
    h += len;
    while (p + 8 <= end) { h ^= xxround(0, rd64(p)); h = rotl64(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P1; h = rotl64(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = rotl64(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

/* ------------------------------------------------------------------ */
/* init_value (src/lib.rs:478-488).  FxHasher (rustc-hash 1.1.0)       */
/* write_i64 from the default state 0 is one step                      */
/*   h = (rotl(0,5) ^ x) * 0x517cc1b727220a95 = x * K  (mod 2^64);     */
/* the sum hsh + col + seed wraps (release build), `%` is Rust's       */
/* truncated remainder (C99 `%` has the same sign rule), and the cast  */
/* and the divide by 2^23 are exact in f32.                            */
/* ------------------------------------------------------------------ */
#define FX_K 0x517cc1b727220a95ULL

float oracle_init_value(uint64_t col, uint64_t hsh, int64_t seed) {
    uint64_t x = hsh + col + (uint64_t)seed;          /* wrapping i64 add */
    int64_t hv = (int64_t)(x * FX_K);                 /* FxHasher::finish() as i64 */
    const int64_t MAXH = 8 * 1024 * 1024;
    int64_t r = hv % MAXH;                            /* sign follows dividend */
    return ((float)r) / (float)MAXH;
}

/* initialize_deterministically_rust (src/lib.rs:69-81), with the entity
 * hashes precomputed (the reference rehashes the id string per call). */
void oracle_init(const uint64_t *hashes, uint64_t n, uint64_t d, int64_t seed, float *x) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++)
        for (uint64_t c = 0; c < d; c++)
            x[(uint64_t)i * d + c] = oracle_init_value(c, hashes[i], seed);
}

/* ------------------------------------------------------------------ */
/* spmm_kernel (src/embedding.rs:52-86).  One task per output row;     */
/* edges of the row in stored order; acc[j] += v * src[j] as separate  */
/* f32 multiply and add; rows with no edges keep their previous        */
/* contents (:66-68) — the callers zero-fill first (:21, :48).         */
/* The reference stores edges AoS {u32 col, f32 left, f32 sym}         */
/* (src/sparse_matrix.rs:73-78); the oracle takes the SoA split the    */
/* C-ABI uses (one value stream per call), which reads the same        */
/* numbers in the same order.                                          */
/* ------------------------------------------------------------------ */
void oracle_spmm(uint64_t n_rows, const uint64_t *rowptr, const uint32_t *col,
                 const float *val, const float *x, uint64_t d, float *y, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (d ? d : 1));
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < (int64_t)n_rows; r++) {
            uint64_t b = rowptr[r], e = rowptr[r + 1];
            if (b == e) continue;
            for (uint64_t j = 0; j < d; j++) acc[j] = 0.0f;
            for (uint64_t k = b; k < e; k++) {
                const float v = val[k];
                const float *src = x + (uint64_t)col[k] * d;
                for (uint64_t j = 0; j < d; j++) acc[j] += v * src[j];
            }
            memcpy(y + (uint64_t)r * d, acc, sizeof(float) * d);
        }
        free(acc);
    }
}

/* AoS variant: the reference's own 12-byte Edge layout, used by bench.py's
 * cpu_baseline leg so the CPU line pays the same bytes per edge as the
 * reference does (src/sparse_matrix.rs:73-78). */
typedef struct { uint32_t col; float left; float sym; } oracle_edge_t;

void oracle_spmm_aos(uint64_t n_rows, const uint64_t *rowptr, const oracle_edge_t *edges,
                     int symmetric, const float *x, uint64_t d, float *y, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * (d ? d : 1));
#pragma omp for schedule(dynamic, 64)
        for (int64_t r = 0; r < (int64_t)n_rows; r++) {
            uint64_t b = rowptr[r], e = rowptr[r + 1];
            if (b == e) continue;
            for (uint64_t j = 0; j < d; j++) acc[j] = 0.0f;
            for (uint64_t k = b; k < e; k++) {
                const float v = symmetric ? edges[k].sym : edges[k].left;
                const float *src = x + (uint64_t)edges[k].col * d;
                for (uint64_t j = 0; j < d; j++) acc[j] += v * src[j];
            }
            memcpy(y + (uint64_t)r * d, acc, sizeof(float) * d);
        }
        free(acc);
    }
}

/* l2_normalize_inplace (src/embedding.rs:88-104): sequential f32 sum of
 * squares, norm = max(sqrt(s), 1e-10), multiply by the reciprocal. */
void oracle_l2_normalize(float *m, uint64_t n, uint64_t d, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)n; r++) {
        float *row = m + (uint64_t)r * d;
        float s = 0.0f;
        for (uint64_t j = 0; j < d; j++) s += row[j] * row[j];
        float norm = sqrtf(s);
        if (!(norm > 1e-10f)) norm = 1e-10f;   /* f32::max(norm, 1e-10): NaN -> 1e-10 */
        const float inv = 1.0f / norm;
        for (uint64_t j = 0; j < d; j++) row[j] *= inv;
    }
}

/* embed_full / embed_full_with_convergence (src/embedding.rs:106-188).
 * `x` holds the initial matrix on entry and the result on exit; returns
 * the number of iterations run.  check_convergence = (threshold > 0),
 * tested from iter >= 1 with a sequential f32 accumulator (:169-183);
 * residual only for 0 < rw < 1 (:116), single-threaded loop order is
 * irrelevant (elementwise).  fill(0) before each SpMM (:47). */
uint64_t oracle_embed(uint64_t n, const uint64_t *rowptr, const uint32_t *col, const float *val,
                      float *x, uint64_t d, uint64_t max_iterations, float residual_weight,
                      float convergence_threshold, int threads) {
    const uint64_t total = n * d;
    float *src = x;
    float *dst = (float *)calloc(total ? total : 1, sizeof(float));
    const int use_residual = residual_weight > 0.0f && residual_weight < 1.0f;
    const int check = convergence_threshold > 0.0f;
    uint64_t actual = max_iterations;
    for (uint64_t it = 0; it < max_iterations; it++) {
        memset(dst, 0, sizeof(float) * total);
        oracle_spmm(n, rowptr, col, val, src, d, dst, threads);
        if (use_residual) {
            const float alpha = 1.0f - residual_weight, rw = residual_weight;
            for (uint64_t j = 0; j < total; j++) dst[j] = alpha * dst[j] + rw * src[j];
        }
        oracle_l2_normalize(dst, n, d, threads);
        if (check && it > 0) {
            float diff = 0.0f;
            for (uint64_t j = 0; j < total; j++) {
                const float delta = dst[j] - src[j];
                diff += delta * delta;
            }
            const float rmse = sqrtf(diff / (float)total);
            if (rmse < convergence_threshold) {
                float *t = src; src = dst; dst = t;
                actual = it + 1;
                break;
            }
        }
        float *t = src; src = dst; dst = t;
    }
    if (src != x) { memcpy(x, src, sizeof(float) * total); free(src); }
    else free(dst);
    return actual;
}

/* Threads worth using on this host.  NOT omp_get_max_threads(): that returns whatever the last omp_set_num_threads() call of
 * this process asked for — after any single-threaded oracle call (the default of oracle.spmm) it is 1, and the at-scale
 * checks that followed it in one pytest session ran on one core (round 3: 87 s for 20 of 40 iterations). */
int oracle_max_threads(void) {
#ifdef _OPENMP
    int n = omp_get_num_procs();
    return n > 0 ? n : 1;
#else
    return 1;
#endif
}
