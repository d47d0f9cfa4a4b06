// spmm.hip — CSR x dense f32 SpMM with the fused row epilogue, for gfx950.
//
// Replaces NdArrayMatrix::spmm_kernel / multiply_into / l2_normalize_inplace and the
// per-iteration body of embed_full (src/embedding.rs:41-136).
//
// Mapping to the hardware:
//   * one GROUP of G lanes owns one output row; G = 64 (a whole wavefront) for d >= 256, so
//     one gathered embedding row is ONE coalesced `global_load_dwordx4` per 1 KiB (lane l
//     reads bytes [16 l, 16 l + 16) of the row); narrower d packs 64/G rows per wavefront.
//   * the row's (col, val) slice is fetched 64 (G) entries at a time with one coalesced load
//     into registers and broadcast edge by edge with v_readlane (G = 64: the column index
//     lands in an SGPR, so the gather address is scalar-base + lane offset) or ds_bpermute.
//   * U*V >= 8 independent 16-byte loads are in flight per lane before the first use; with 8
//     waves per SIMD that is >= 256 KiB of gathers in flight per CU.
//   * edges are consumed in stored order with separate f32 multiply and add
//     (-ffp-contract=off + __fmul_rn/__fadd_rn), so every row that is not split is
//     bit-identical to the reference's sequential accumulate (src/embedding.rs:80-82).
//   * rows longer than hub_threshold (hub rows) are summed IN THE REFERENCE'S ORDER too, by
//     hub_inorder_kernel on a side stream beside the main launch: one wavefront per (hub row,
//     64-column slab), four edges per 16-byte load instruction (a lane quad per edge), 48 loads =
//     192 edges in flight per wavefront, the running sum handed from quad to quad with a DPP row
//     rotate so that every output element sees `acc += w * x` edge by edge, like
//     src/embedding.rs:76-83.  hub_epilogue_kernel then runs the row epilogue on the whole row.
//   * with CLEORA_F_HUB_SEGMENTS the hub rows are instead cut into hub_segment-edge segments that
//     are the FIRST work items of the main launch, each on its own wavefront, and combined in a
//     fixed order by hub_finish_kernel: deterministic and immune to a pathological hub (a chain
//     of 10^8 dependent adds), but not the reference's summation order.
//   * the epilogue (residual blend, L2 normalise, squared difference) runs on the
//     accumulator registers, so Y is written exactly once and never re-read.
#include "common.h"
#include "row_epilogue.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace cleora {
namespace {

struct SpmmArgs {
    const uint64_t *rowptr;
    const uint32_t *col;
    const float *val;
    const float *x;
    uint64_t ldx;
    uint64_t x_bytes;      // span of x (whole-matrix buffer descriptor of the sub-wave hot path)
    uint64_t n_items;      // hub segments + rows
    uint64_t n_segments;
    uint32_t hub_threshold;
    // hub kernels
    const uint32_t *seg_row;
    const uint64_t *seg_begin;
    uint32_t hub_segment;
    const uint32_t *hub_rows;
    const uint64_t *hub_seg_first;
    float *partial;
    const uint32_t *mid_rows;     // reference-order schedule: the first n_segments items are these rows (longest first), not segments
    // in-order hub kernel
    const uint32_t *hub_by_len;   // hub indices, longest row first
    uint64_t nnz;
    uint32_t n_slabs;             // 64-column slabs per row
    uint32_t hub_first;           // hub_inorder_kernel: its first row in hub_by_len (the rows before it take hub_chain_kernel)
    RowArgs r;
};

// One gathered row.  HOT: bit 31 of the column index marks a frequently referenced row; those are
// loaded with the default cache policy and every other (cold) row non-temporally, so the cold stream
// does not evict the hot set (hot.hip: -10 % at C3).  The
// policy is an immediate of buffer_load, so the row goes through a buffer descriptor: built from the
// (wave-uniform) row pointer when a whole wavefront owns the row — its num_records = d*4 also does
// the tail bounds check — or spanning the whole matrix (< 4 GiB) for sub-wave groups.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int G, int V, int W, bool FULL, bool HOT>
__device__ __forceinline__ void gather_row(const SpmmArgs &a, uint32_t c, int gl, uint32_t d,
                                           float (&r)[V][W]) {
    if constexpr (!HOT) {
        load_row<G, V, W, FULL>(a.x + (uint64_t)c * a.ldx, gl, d, r);
    } else {
        static_assert(W == 4, "hot policy is implemented for the float4 path");
        const uint32_t cc = c & 0x7fffffffu;
        const bool hot = (c >> 31) != 0;
        if constexpr (G == 64) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x + (uint64_t)cc * a.ldx), 0,
                                                              (int)(d * 4u), 0x00020000);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int off = (v * 64 + gl) * 16;
                u32x4 t;
                if (hot) t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                else t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);       // aux 2 = nt
                r[v][0] = __uint_as_float(t.x); r[v][1] = __uint_as_float(t.y);
                r[v][2] = __uint_as_float(t.z); r[v][3] = __uint_as_float(t.w);
            }
        } else {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, (int)(uint32_t)a.x_bytes, 0x00020000);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint32_t j = (uint32_t)(v * G + gl) * 4u;
                const int off = (int)((uint32_t)((uint64_t)cc * a.ldx + j) * 4u);
                u32x4 t = {0u, 0u, 0u, 0u};
                if (FULL || j < d) {
                    if (hot) t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                    else t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 2);
                }
                r[v][0] = __uint_as_float(t.x); r[v][1] = __uint_as_float(t.y);
                r[v][2] = __uint_as_float(t.z); r[v][3] = __uint_as_float(t.w);
            }
        }
    }
}

// acc += sum over edges [beg, end) in stored order.  For G == 64 beg/end are wave-uniform.
template <int G, int V, int W, bool FULL, bool HOT = false>
__device__ __forceinline__ void accumulate(const SpmmArgs &a, uint64_t beg, uint64_t end, int gl,
                                           int gbase, float (&acc)[V][W]) {
    constexpr int U = (8 / V) > 0 ? (8 / V) : 1;
    const uint32_t d = a.r.d;
    for (uint64_t e = beg; e < end; e += G) {
        const uint32_t cnt = (end - e) < (uint64_t)G ? (uint32_t)(end - e) : (uint32_t)G;
        uint32_t cv = 0;
        float wv = 0.f;
        if ((uint32_t)gl < cnt) {
            cv = a.col[e + gl];                       // (loaded non-temporally the edge stream measured no better in the fast placement
            wv = a.val[e + gl];                       //  class and 1-2 ms worse in the slow one: scripts/r06/store_policy_probe.py)
        }
        uint32_t k = 0;
        {
            for (; k + U <= cnt; k += U) {
                float r[U][V][W];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t c = bcast_u32<G>(cv, k + u, gbase);
                    gather_row<G, V, W, FULL, HOT>(a, c, gl, d, r[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float w = bcast_f32<G>(wv, k + u, gbase);
#pragma unroll
                    for (int v = 0; v < V; ++v)
#pragma unroll
                        for (int q = 0; q < W; ++q) acc[v][q] = fadd(acc[v][q], fmul(w, r[u][v][q]));
                }
            }
        }
        for (; k < cnt; ++k) {
            float r[V][W];
            const uint32_t c = bcast_u32<G>(cv, k, gbase);
            const float w = bcast_f32<G>(wv, k, gbase);
            gather_row<G, V, W, FULL, HOT>(a, c, gl, d, r);
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int q = 0; q < W; ++q) acc[v][q] = fadd(acc[v][q], fmul(w, r[v][q]));
        }
    }
}

template <int G, int V, int W>
__device__ __forceinline__ void zero(float (&acc)[V][W]) {
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int q = 0; q < W; ++q) acc[v][q] = 0.f;
}

// ---- main kernel ---------------------------------------------------------------------------
// Work items: first the hub SEGMENTS (so the longest work starts first), then one item per row.
// A segment item writes its partial sum to scratch; a row item runs the epilogue and writes Y.
template <int G, int V, int W, bool FULL, bool HOT = false>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const SpmmArgs a) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    uint64_t item = CLEORA_LINEAR_BLOCK() * (256 / G) + (threadIdx.x / G);
    bool active = item < a.n_items;
    if constexpr (G == 64) {
        if (!active) return;
        item = uniform_u64(item);
    }
    const bool is_mid = a.mid_rows != nullptr && item < a.n_segments;      // a long row scheduled first: a whole row like any other
    const bool is_seg = a.mid_rows == nullptr && item < a.n_segments;
    uint64_t row = 0, beg = 0, end = 0;
    if (active) {
        if (is_mid) {
            row = a.mid_rows[item];
            beg = a.rowptr[row];
            end = a.rowptr[row + 1];
        } else if (is_seg) {
            row = a.seg_row[item];
            beg = a.seg_begin[item];
            const uint64_t rend = a.rowptr[row + 1];
            end = beg + a.hub_segment < rend ? beg + a.hub_segment : rend;
        } else {
            row = item - a.n_segments;
            beg = a.rowptr[row];
            end = a.rowptr[row + 1];
            if (end - beg > a.hub_threshold) active = false;  // hub row: its segments own it
        }
    }
    if constexpr (G == 64) {
        if (!active) return;
        row = uniform_u64(row);
        beg = uniform_u64(beg);
        end = uniform_u64(end);
    } else if (!active) {
        beg = end = 0;
    }
    float acc[V][W];
    zero<G, V, W>(acc);
    accumulate<G, V, W, FULL, HOT>(a, beg, end, gl, gbase, acc);
    if (active) {
        if (is_seg) store_row<G, V, W, FULL>(a.partial + item * (uint64_t)a.r.d, gl, a.r.d, acc);
        else finish_row<G, V, W, FULL>(a.r, row, gl, gbase, acc);
    }
}

// ---- hub rows: in-order combine of the segment partials + epilogue -----------------------------
// One 256-thread block per hub row.  The row's segments are cut into 256/G contiguous runs, one
// per lane group; each group adds its run in order with 8 loads in flight, the groups' sums are
// then added in group order through LDS.  Fixed order => deterministic.
template <int G, int V, int W>
__global__ __launch_bounds__(256) void hub_finish_kernel(const SpmmArgs a) {
    constexpr int NG = 256 / G;
    __shared__ float sm[NG][G * V * W];
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    const int grp = threadIdx.x / G;
    const uint64_t h = blockIdx.x;
    const uint64_t row = a.hub_rows[h];
    const uint64_t s0 = a.hub_seg_first[h], s1 = a.hub_seg_first[h + 1];
    const uint64_t per = (s1 - s0 + NG - 1) / NG;
    const uint64_t b = s0 + grp * per < s1 ? s0 + grp * per : s1;
    const uint64_t e = b + per < s1 ? b + per : s1;
    const uint32_t d = a.r.d;
    float acc[V][W];
    zero<G, V, W>(acc);
    constexpr int U = (8 / V) > 0 ? (8 / V) : 1;
    uint64_t s = b;
    for (; s + U <= e; s += U) {
        float p[U][V][W];
#pragma unroll
        for (int u = 0; u < U; ++u) load_row<G, V, W, false>(a.partial + (s + u) * (uint64_t)d, gl, d, p[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int q = 0; q < W; ++q) acc[v][q] = fadd(acc[v][q], p[u][v][q]);
    }
    for (; s < e; ++s) {
        float p[V][W];
        load_row<G, V, W, false>(a.partial + s * (uint64_t)d, gl, d, p);
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int q = 0; q < W; ++q) acc[v][q] = fadd(acc[v][q], p[v][q]);
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int q = 0; q < W; ++q) sm[grp][(v * G + gl) * W + q] = acc[v][q];
    __syncthreads();
    if (grp != 0) return;
    for (int g2 = 1; g2 < NG; ++g2)
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int q = 0; q < W; ++q) acc[v][q] = fadd(acc[v][q], sm[g2][(v * G + gl) * W + q]);
    finish_row<G, V, W, false>(a.r, row, gl, gbase, acc);
}

// ---- hub rows in the reference's order ---------------------------------------------------------
// src/embedding.rs:76-83 adds a row's edges into ONE accumulator in stored order.  Every output element's chain is
// independent of the others, so a hub row can be cut by COLUMN without touching the order: one wavefront per (hub row,
// 64-column slab).  What a single in-order consumer lacks is memory parallelism, so the loads are decoupled from the adds:
//   * a ring of R 16-byte loads per lane stays in flight (R = 48: 48 KiB per wavefront, the vmcnt counter's reach); the
//     ring is refilled for step j + R right after step j's registers are read;
//   * QUAD form (16-byte-aligned rows): one load instruction covers FOUR edges — lane = (DPP row rho, quad q, i) reads
//     columns [64 slab + 16 rho + 4 i, +4) of edge 4 j + q — and the running sum travels through the quads:
//     round g computes acc = fadd(row_ror:4(acc), w * x) in all lanes, of which quad g's is the true prefix sum after
//     edge 4 j + g (the other quads' values are never used), so after four rounds quad 3 holds the sum through edge
//     4 j + 3 and row_ror:4 hands it to quad 0 of the next step.  20 VALU instructions per 4 edges;
//   * LANE form (any alignment / width): a lane owns one column, one edge per step, no cross-lane traffic;
//   * the row's (col, val) slice streams through LDS in chunks of R steps, fetched two chunks ahead with bounds-checked
//     buffer loads (ordinary loads would sit in the same in-order vmcnt queue as the ring and drain it), read back with
//     one ds_read per step at a compile-time offset.
// The last step of a row runs only its valid rounds (the padding lanes hold the next row's edges).
// L = lanes per edge (each lane 16 bytes = 4 columns): 4 -> 4 edges per load, a 64-column slab per wavefront (the form described
// above); 2 -> 8 edges per load, 32-column slabs (1 -> 16 edges per load compiles too, but its unrolled body spills and outgrows the
// instruction cache: not instantiated).  Fewer lanes per edge = more edges in
// flight per wavefront (the vmcnt counter caps the LOADS at 63) and more wavefronts per row: a shorter chain for the longest row
// — which, beside a main kernel that saturates HBM, advances one step per (load latency / R) — at the price of proportionally more
// vector instructions in total (every lane still adds 4 floats per edge).  The launcher picks L per launch (hub_lanes()).
// L = 0: the LANE form (one column per lane, one edge per step; any alignment).
template <int R, int L>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void hub_inorder_kernel(const SpmmArgs a) {
    constexpr bool QUAD = L != 0;
    constexpr int EPS = QUAD ? 16 / L : 1;        // edges per step
    constexpr int COLS = QUAD ? 16 * L : 64;      // columns per wavefront
    constexpr int CH = R * EPS;                   // edges per chunk
    constexpr int NF = CH > 64 ? (CH + 255) / 256 : 0;    // dwordx4 fetches per lane, stream and chunk (0: one dword)
    constexpr int SLOT = NF ? 256 * NF : 64;      // entries of an LDS slot
    static_assert(SLOT >= CH, "a chunk fits its LDS slot");
    __shared__ __attribute__((aligned(16))) uint32_t s_col[2][SLOT];
    __shared__ __attribute__((aligned(16))) uint32_t s_val[2][SLOT];
    const int lane = threadIdx.x;
    const int q = QUAD ? (lane & 15) / (QUAD ? L : 1) : 0;                        // which edge of the step this lane loads
    const uint64_t item = CLEORA_LINEAR_BLOCK();
    const uint32_t k = (uint32_t)(item / a.n_slabs), slab = (uint32_t)(item - (uint64_t)k * a.n_slabs);
    const uint32_t h = a.hub_by_len[k + a.hub_first];
    const uint64_t row = a.hub_rows[h];
    const uint64_t beg = a.rowptr[row], n = a.rowptr[row + 1] - beg;
    const uint32_t d = a.r.d;
    const uint32_t coff = QUAD ? slab * (uint32_t)COLS + (uint32_t)(lane >> 4) * (4u * (QUAD ? L : 1)) + (uint32_t)((lane & 15) % (QUAD ? L : 1)) * 4u
                               : slab * 64u + (uint32_t)lane;
    const bool in_range = coff < d;
    const float *xb = a.x + (in_range ? coff : 0u);

    const uint64_t left_bytes = (a.nnz - beg) * 4u;
    const int records = (int)(left_bytes > 0xfffffffcull ? 0xfffffffcu : (uint32_t)left_bytes);
    const auto rs_col = __builtin_amdgcn_make_buffer_rsrc((void *)(a.col + beg), 0, records, 0x00020000);
    const auto rs_val = __builtin_amdgcn_make_buffer_rsrc((void *)(a.val + beg), 0, records, 0x00020000);
    // entries past the row's end are the next rows' (valid gather addresses, never consumed) or, past the array, zero
    u32x4 pc[NF ? NF : 1], pv[NF ? NF : 1];
    auto fetch = [&](uint64_t chunk) {
        if constexpr (NF != 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int off = (int)(uint32_t)((chunk * CH + (uint64_t)(256 * f + 4 * lane)) * 4u);
                pc[f] = __builtin_amdgcn_raw_buffer_load_b128(rs_col, off, 0, 0);
                pv[f] = __builtin_amdgcn_raw_buffer_load_b128(rs_val, off, 0, 0);
            }
        } else {
            const int off = (int)(uint32_t)((chunk * CH + (uint64_t)lane) * 4u);
            pc[0].x = __builtin_amdgcn_raw_buffer_load_b32(rs_col, off, 0, 0);
            pv[0].x = __builtin_amdgcn_raw_buffer_load_b32(rs_val, off, 0, 0);
        }
    };
    auto stash = [&](int slot) {
        if constexpr (NF != 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                *reinterpret_cast<u32x4 *>(&s_col[slot][256 * f + 4 * lane]) = pc[f];
                *reinterpret_cast<u32x4 *>(&s_val[slot][256 * f + 4 * lane]) = pv[f];
            }
        } else {
            s_col[slot][lane] = pc[0].x;
            s_val[slot][lane] = pv[0].x;
        }
    };
    using Vec = std::conditional_t<QUAD, float4, float>;
    const uint32_t ldx32 = (uint32_t)a.ldx;     // < 2^32 (checked by the launcher): one v_mad_u64_u32 per address
    auto gather = [&](uint32_t c) -> Vec { return *reinterpret_cast<const Vec *>(xb + (uint64_t)c * ldx32); };

    fetch(0); stash(0);
    fetch(1); stash(1);
    fetch(2);
    Vec ring[R];
#pragma unroll
    for (int s = 0; s < R; ++s) ring[s] = gather(s_col[0][EPS * s + q]);

    float acc[QUAD ? 4 : 1];
#pragma unroll
    for (int e = 0; e < (QUAD ? 4 : 1); ++e) acc[e] = 0.f;
    // One chunk: R steps.  WHOLE: all CH edges belong to the row (no tests); else `left` < CH of them do.
    auto chunk_body = [&](auto whole, int cur, uint32_t left) {
        constexpr bool WHOLE = decltype(whole)::value;
        const int nxt = cur ^ 1;
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const float w = __uint_as_float(s_val[cur][EPS * s + q]);
            const uint32_t cn = s_col[nxt][EPS * s + q];
            if constexpr (QUAD) {
                // the products first, THEN the refill of the same ring slot (step s of the next chunk): the old and the new
                // value never live together, so the slot keeps its registers around the loop (no copies at the back edge,
                // which would wait for the loads)
                float t0 = fmul(w, ring[s].x), t1 = fmul(w, ring[s].y), t2 = fmul(w, ring[s].z), t3 = fmul(w, ring[s].w);
                if constexpr (WHOLE) ring[s] = gather(cn);
                // one round: acc = row_ror:L(acc) + t in every lane.  Written as asm so that all rounds are the 1-instruction
                // DPP add (the compiler turns the last round of a step into v_mov_dpp x4 + v_pk_add x2).  A DPP read needs 2 wait
                // states after a VALU write of its source: inside a round the four chains interleave; the s_nop covers a copy
                // or product the compiler schedules right in front of the block (it does not look for hazards inside asm —
                // without it the first chain read stale sums in the tail chunk).
#define CLEORA_HUB_ROUND_(ROR)                                                                                        \
    asm volatile("s_nop 1\n\t"                                                                                      \
                 "v_add_f32_dpp %0, %0, %4 row_ror:" ROR " row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %1, %1, %5 row_ror:" ROR " row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %2, %2, %6 row_ror:" ROR " row_mask:0xf bank_mask:0xf\n\t"                           \
                 "v_add_f32_dpp %3, %3, %7 row_ror:" ROR " row_mask:0xf bank_mask:0xf"                                \
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                            \
                 : "v"(t0), "v"(t1), "v"(t2), "v"(t3))
#define CLEORA_HUB_ROUND                                            \
    do {                                                            \
        if constexpr (L == 4) { CLEORA_HUB_ROUND_("4"); }           \
        else if constexpr (L == 2) { CLEORA_HUB_ROUND_("2"); }      \
        else { CLEORA_HUB_ROUND_("1"); }                            \
    } while (0)
                if (WHOLE || left >= (uint32_t)(EPS * s + EPS)) {
#pragma unroll
                    for (int g = 0; g < EPS; ++g) CLEORA_HUB_ROUND;
                } else if (left > (uint32_t)(EPS * s)) {
                    const uint32_t m = left - EPS * s;   // 1 .. EPS - 1 edges in the row's last step
                    CLEORA_HUB_ROUND;
#pragma unroll
                    for (int g = 1; g < EPS - 1; ++g)
                        if (m > (uint32_t)g) CLEORA_HUB_ROUND;
                }
#undef CLEORA_HUB_ROUND
#undef CLEORA_HUB_ROUND_
            } else {
                const float t = fmul(w, ring[s]);
                if constexpr (WHOLE) ring[s] = gather(cn);
                if (WHOLE || left > (uint32_t)s) acc[0] = fadd(acc[0], t);
            }
        }
    };
    const uint64_t whole_chunks = n / CH;
    for (uint64_t c = 0; c < whole_chunks; ++c) {
        const int cur = (int)(c & 1);
        chunk_body(std::true_type{}, cur, 0u);
        stash(cur);        // chunk c + 2, in flight since the start of chunk c
        fetch(c + 3);
    }
    const uint32_t tail = (uint32_t)(n - whole_chunks * CH);
    if (tail) chunk_body(std::false_type{}, (int)(whole_chunks & 1), tail);
    float *p = a.partial + (uint64_t)h * d + coff;
    if constexpr (QUAD) {
        if (in_range && (uint32_t)q == (uint32_t)((n - 1) % EPS)) *reinterpret_cast<float4 *>(p) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        if (in_range) *p = acc[0];
    }
}

// ---- the LONGEST rows: products by producer waves, the in-order chain by a wave that only adds -------------------------------
// hub_inorder_kernel's wavefront does everything for its (row, slab): (col, val) staging, address arithmetic, gathers, products and
// the chain of adds — ~30 instructions per 4 edges in ONE in-order instruction stream.  Alone beside a 33 ms main kernel that is
// hidden; for one rank of an 8-way partition it is the critical path (profiles/r06_plan_c3.json: the rank that owns C3's longest
// row, 431 465 edges, needs 11.3 ms for its blocks against 4.2 ms with the segmented sum: 26 ns per edge), and a 10^7-edge hub
// costs 0.3 s per SpMM.  Here the chain wave does NOTHING but add: a block is 8 waves for one (row, 64-column slab) —
//   * seven PRODUCER waves gather (one 16-byte load per lane covers four edges x 64 columns, like the QUAD form), multiply by the
//     edge values (fmul: the reference's `v * src`, src/embedding.rs:80-82) and write the PRODUCTS to an LDS ring, [edge][column];
//     each keeps D chunks of loads in flight.  Vector-memory operations retire in order, so every load of the loop has the same
//     lead: the gathers and values of chunk c + D and the columns of chunk c + 2 D are issued during chunk c (6 operations per
//     chunk, 48 in flight at D = 8; the counter reaches 63): 7 x 8 x 16 = 896 edges in flight per block;
//   * the CONSUMER wave owns one column per lane and adds the products in stored order, two edges per LDS instruction
//     (ds_read2st64_b32) + two dependent adds (fadd: `acc += ...`), the reads of the next group in flight under the adds;
//   * chunks of 112 edges alternate between two ring halves, one barrier per chunk: producers fill chunk c while the consumer adds
//     chunk c - 1.
// Same products, same order of additions: the bits of hub_inorder_kernel and of the reference.  Measured (rocprofv3, one rank's
// blocks of C3 / 8 alone on the GPU): 3.0 ms for the 431 465-edge row = 6.9 ns per edge — the consumer's ~2 instructions per edge
// in one wave's issue slots; a form that computed the products in a separate launch on every CU and streamed them in ran the same
// 3.0 ms (the gathers are not what bounds it) plus the products' 1.6 ms, and is not kept.  Taken for the few longest rows only
// (hub_chain_rows()): a block holds a whole CU's registers.
constexpr int HC_NP = 7, HC_U = 4, HC_CH = HC_NP * HC_U * 4;     // producers, loads per producer and chunk, edges per chunk (112)
// the block's barrier as an LDS-only hand-off (the producers' global loads are consumed by the producers themselves)
__device__ __forceinline__ void hc_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int D>
__global__ __launch_bounds__(512) void hub_chain_kernel(const SpmmArgs a) {
    static_assert(D % 2 == 0, "ring halves are compile-time functions of the unrolled interval");
    __shared__ __attribute__((aligned(16))) float ring[2][HC_CH][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint64_t item = CLEORA_LINEAR_BLOCK();
    const uint32_t k = (uint32_t)(item / a.n_slabs), slab = (uint32_t)(item - (uint64_t)k * a.n_slabs);
    const uint32_t h = a.hub_by_len[k];
    const uint64_t row = a.hub_rows[h];
    const uint64_t beg = a.rowptr[row], n = a.rowptr[row + 1] - beg;
    const uint32_t d = a.r.d;
    const uint64_t nchunks = (n + HC_CH - 1) / HC_CH;
    // interval j: producers make chunk j, the consumer adds chunk j - 1; whole trips of D intervals (the surplus ones are idle)
    const uint64_t trips = (nchunks + 1 + D - 1) / D;
    if (w == 0) {
        // ---- consumer: lane = column ----------------------------------------------------------------------------------
        float acc = 0.f;
        for (uint64_t j = 0; j < trips * D; ++j) {
            if (j >= 1 && j <= nchunks) {
                const float *r = &ring[(j - 1) & 1][0][lane];
                const uint64_t done = (j - 1) * HC_CH;
                if (n - done >= (uint64_t)HC_CH) {
                    // eight groups of 14 products through two register sets: the reads of group g + 1 are in flight under the adds of
                    // group g (7 ds_read2st64 per group: the LDS counter tracks 15 operations, two groups fit), ONE wait per group
                    // (LDS reads return in order: "at most 7 outstanding" = the current group has landed).  Left to itself the compiler
                    // read eight products, waited for all of them, added, and paid the LDS latency fourteen times per chunk
                    constexpr int NG = 8, GR = HC_CH / NG;
                    float v[2][GR];
#pragma unroll
                    for (int e = 0; e < GR; ++e) v[0][e] = r[e * 64];
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (grp + 1 < NG) {
#pragma unroll
                            for (int e = 0; e < GR; ++e) v[(grp + 1) & 1][e] = r[((grp + 1) * GR + e) * 64];
                            __builtin_amdgcn_s_waitcnt(0xC77F);                   // lgkmcnt(7)
                        } else {
                            __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0)
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int e = 0; e < GR; ++e) acc = fadd(acc, v[grp & 1][e]);
                    }
                } else {
                    const uint32_t left = (uint32_t)(n - done);
                    for (uint32_t e = 0; e < left; ++e) acc = fadd(acc, r[e * 64]);
                }
            }
            hc_barrier();
        }
        const uint32_t c = slab * 64u + (uint32_t)lane;
        if (c < d) a.partial[(uint64_t)h * d + c] = acc;
        return;
    }
    // ---- producers: lane = (edge quad qr | four columns) -------------------------------------------------------------
    const int p = w - 1, qr = lane >> 4, i4 = lane & 15;
    const uint32_t coff = slab * 64u + 4u * (uint32_t)i4;
    const float *xb = a.x + (coff < d ? coff : 0u);                               // (columns past d: a valid address; the consumer never stores them)
    const uint64_t left_bytes = (a.nnz - beg) * 4u;
    const int records = (int)(left_bytes > 0xfffffffcull ? 0xfffffffcu : (uint32_t)left_bytes);
    const auto rs_col = __builtin_amdgcn_make_buffer_rsrc((void *)(a.col + beg), 0, records, 0x00020000);
    const auto rs_val = __builtin_amdgcn_make_buffer_rsrc((void *)(a.val + beg), 0, records, 0x00020000);
    // this lane's four edges of chunk c: c * 112 + 16 p + 4 qr + u; entries past the row's end are the next rows' (valid gather
    // addresses, products never added) or, past the array, zero
    const uint32_t lane_edge = (uint32_t)(16 * p + 4 * qr);
    auto stream_off = [&](uint64_t chunk) { return (int)(uint32_t)((chunk * HC_CH + lane_edge) * 4u); };
    const uint32_t ldx32 = (uint32_t)a.ldx;                                       // < 2^32 (checked by the launcher)
    auto gather = [&](uint32_t c) { return *reinterpret_cast<const float4 *>(xb + (uint64_t)c * ldx32); };
    float4 g[D][HC_U];
    u32x4 pc[D], pv[D];
#pragma unroll
    for (int s = 0; s < D; ++s) pc[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_col, stream_off(s), 0, 0);
#pragma unroll
    for (int s = 0; s < D; ++s) {
#pragma unroll
        for (int u = 0; u < HC_U; ++u) g[s][u] = gather(pc[s][u]);
        pv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_val, stream_off(s), 0, 0);
        pc[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_col, stream_off(D + s), 0, 0);
        // (in this order: a prologue the scheduler rearranges — it issued chunk 0's values LAST — makes the loop's first wait a full
        // drain, on the entry path and therefore, merged, on every trip)
        __builtin_amdgcn_sched_barrier(0);
    }
    for (uint64_t c0 = 0; c0 < trips * D; c0 += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const uint64_t c = c0 + s;
            float4 *const dst = reinterpret_cast<float4 *>(&ring[s & 1][lane_edge][4 * i4]);      // (c & 1 == s & 1: D is even)
            // the products of chunk c, THEN the refill of the same registers (chunk c + D): no copies at the loop's back edge
#pragma unroll
            for (int u = 0; u < HC_U; ++u) {
                const float wv = __uint_as_float(pv[s][u]);
                dst[16 * u] = make_float4(fmul(wv, g[s][u].x), fmul(wv, g[s][u].y), fmul(wv, g[s][u].z), fmul(wv, g[s][u].w));
            }
            // (the scheduler must not lift a refill above a product that still reads the old value: it would keep the old value in a
            // copy made at the loop's back edge, behind a wait for nearly everything in flight)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < HC_U; ++u) g[s][u] = gather(pc[s][u]);
            pv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_val, stream_off(c + D), 0, 0);
            pc[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_col, stream_off(c + 2 * D), 0, 0);
            hc_barrier();
        }
    }
}

// The row epilogue of the in-order hub rows: one lane group per hub row reads the complete sums back.
template <int G, int V, int W>
__global__ __launch_bounds__(256) void hub_epilogue_kernel(const SpmmArgs a, uint64_t n_hub_rows) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    const uint64_t h = (uint64_t)blockIdx.x * (256 / G) + (threadIdx.x / G);
    if (h >= n_hub_rows) return;
    float acc[V][W];
    load_row<G, V, W, false>(a.partial + h * (uint64_t)a.r.d, gl, a.r.d, acc);
    finish_row<G, V, W, false>(a.r, a.hub_rows[h], gl, gbase, acc);
}

// ---- stand-alone row epilogue (l2_normalize_inplace & friends) --------------------------
template <int G, int V, int W>
__global__ __launch_bounds__(256) void rowops_kernel(const float *x, uint64_t ldx, uint64_t n,
                                                     const RowArgs ra) {
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);
    const int gbase = lane & ~(G - 1);
    const uint64_t row = CLEORA_LINEAR_BLOCK() * (256 / G) + (threadIdx.x / G);
    if (row >= n) return;
    float acc[V][W];
    load_row<G, V, W, false>(x + row * ldx, gl, ra.d, acc);
    finish_row<G, V, W, false>(ra, row, gl, gbase, acc);
}

// ---- stand-alone exact-order L2 normalise: 16 lanes per row, E4 float4 per lane --------------------
// NdArrayMatrix::l2_normalize_inplace (src/embedding.rs:88-104) sums the squares of a row strictly in index
// order.  With one wavefront per row that chain is 256 dependent full-wave adds (the fused SpMM hides them
// behind its gathers; alone they made this pass ALU-bound at 1.5 TB/s).  Here a row lives in ONE 16-lane DPP
// row, lane l holding the 4*E4 consecutive elements [l*4*E4, (l+1)*4*E4): the running sum walks the lane's
// elements, then moves one lane to the right with `row_shr:1` (lane 0 receives 0).  Every round recomputes all
// lanes from their left neighbour, so after round r lanes 0..r hold the exact prefix sums; 16 rounds of
// (1 + 4*E4) instructions serve FOUR rows: 17 instead of 320 wave instructions per row at d = 256, same bits.
template <int E4>
__global__ __launch_bounds__(256) void l2_exact16_kernel(const float *__restrict__ x, uint64_t ldx, uint64_t n,
                                                         float *__restrict__ y, uint64_t ldy) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const uint64_t row = (CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < n;
    float4 v[E4];
    const float *xr = x + (live ? row : 0) * ldx + (uint32_t)sub * (4 * E4);
#pragma unroll
    for (int i = 0; i < E4; ++i) v[i] = live ? *reinterpret_cast<const float4 *>(xr + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    float sq[4 * E4];
#pragma unroll
    for (int i = 0; i < E4; ++i) {
        sq[4 * i + 0] = fmul(v[i].x, v[i].x);
        sq[4 * i + 1] = fmul(v[i].y, v[i].y);
        sq[4 * i + 2] = fmul(v[i].z, v[i].z);
        sq[4 * i + 3] = fmul(v[i].w, v[i].w);
    }
    float s = 0.f;
#pragma unroll 1
    for (int r = 0; r < 16; ++r) {
        // previous lane's running sum; row_shr:1 with bound_ctrl: lane 0 of each 16-lane row reads 0
        s = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x111, 0xF, 0xF, true));
#pragma unroll
        for (int e = 0; e < 4 * E4; ++e) s = fadd(s, sq[e]);
    }
    s = __shfl(s, lane | 15, 64);                       // the row's total sits in its last lane
    const float norm = fmaxf(sqrtf(s), 1e-10f);        // src/embedding.rs:98-102
    const float inv = (1.0f / norm);
    if (!live) return;
    float *yr = y + row * ldy + (uint32_t)sub * (4 * E4);
#pragma unroll
    for (int i = 0; i < E4; ++i)
        *reinterpret_cast<float4 *>(yr + 4 * i) = make_float4(fmul(v[i].x, inv), fmul(v[i].y, inv), fmul(v[i].z, inv), fmul(v[i].w, inv));
}

// ---- rows wider than the register-resident shapes: wave per row, two passes -----------------
__global__ __launch_bounds__(256) void rowops_wide_kernel(const float *x, uint64_t ldx, uint64_t n,
                                                          const RowArgs ra) {
    const int lane = threadIdx.x & 63;
    const uint64_t row = CLEORA_LINEAR_BLOCK() * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *xr = x + row * ldx;
    const float *xs = (ra.flags & (CLEORA_F_RESIDUAL | CLEORA_F_SQDIFF)) ? ra.x_self + row * ra.ldxs : nullptr;
    float *yr = ra.y + row * ra.ldy;
    const uint32_t d = ra.d;
    const bool l1 = (ra.flags & CLEORA_F_L1NORM) != 0;
    float s = ((ra.flags & CLEORA_F_ROWSQ_CONT) && (ra.flags & CLEORA_F_ROWSQ)) ? ra.row_sumsq[row] : 0.f;
    for (uint32_t j0 = 0; j0 < d; j0 += 64) {
        const uint32_t j = j0 + lane;
        float v = j < d ? xr[j] : 0.f;
        if ((ra.flags & CLEORA_F_RESIDUAL) && j < d) v = fadd(fmul(ra.alpha, v), fmul(ra.rw, xs[j]));
        if (j < d) yr[j] = v;
        if ((ra.flags & (CLEORA_F_L2NORM | CLEORA_F_ROWSQ | CLEORA_F_L1NORM)) && !(ra.flags & CLEORA_F_SCALE)) {
            const float sq = l1 ? fabsf(v) : fmul(v, v);
            if (ra.flags & CLEORA_F_FASTNORM) {
                float t = sq;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) t = fadd(t, __shfl_xor(t, o, 64));
                s = fadd(s, t);
            } else {
                const uint32_t lim = (d - j0) < 64u ? (d - j0) : 64u;
                for (uint32_t g = 0; g < lim; ++g) s = fadd(s, bcast_f32<64>(sq, g, 0));
            }
        }
    }
    float inv = 1.0f, div = 1.0f;
    if (ra.flags & CLEORA_F_SCALE) s = ra.row_sumsq[row];
    if ((ra.flags & CLEORA_F_ROWSQ) && lane == 0) ra.row_sumsq[row] = s;
    const bool scale = ra.flags & (CLEORA_F_L2NORM | CLEORA_F_SCALE);
    if (scale) inv = (1.0f / fmaxf(sqrtf(s), 1e-10f));
    if (l1) div = fmaxf(s, 1e-10f);
    double ds = 0.0;
    for (uint32_t j0 = 0; j0 < d; j0 += 64) {
        const uint32_t j = j0 + lane;
        if (j < d) {
            float v = yr[j];
            if (l1) v = v / div;
            else if (scale) v = fmul(v, inv);
            if (ra.flags & CLEORA_F_SQDIFF) {
                const double delta = (ra.flags & CLEORA_F_SQDIFF64) ? (double)v - (double)xs[j] : (double)fsub(v, xs[j]);
                ds += delta * delta;
            }
            yr[j] = v;
        }
    }
    if (ra.flags & CLEORA_F_SQDIFF) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ds += __shfl_xor(ds, o, 64);
        if (lane == 0) ra.row_sqdiff[row] = ds;
    }
}

// ---- shape dispatch -----------------------------------------------------------------------------
template <int N> using I = std::integral_constant<int, N>;
constexpr uint32_t kMaxD4 = 64 * 8 * 4;  // widest register-resident row, float4 path
constexpr uint32_t kMaxD1 = 64 * 16;     // scalar path

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// The residual blend runs for 0 < rw < 1 (src/embedding.rs:116), or for any rw > 0 when the caller asks for the
// semantics of the Python loop (pycleora/__init__.py:111-115, CLEORA_F_BLEND_ANY).
inline uint32_t gate_residual(uint32_t flags, float rw) {
    if (!(flags & CLEORA_F_RESIDUAL)) return flags;
    const bool on = rw > 0.0f && (rw < 1.0f || (flags & CLEORA_F_BLEND_ANY));
    return on ? flags : (flags & ~CLEORA_F_RESIDUAL);
}

int check_norm_flags(uint32_t flags) {
    CL_REQUIRE(!((flags & CLEORA_F_ROWSQ) && (flags & CLEORA_F_SCALE)), "ROWSQ and SCALE are exclusive");
    CL_REQUIRE(!((flags & CLEORA_F_L1NORM) && (flags & (CLEORA_F_L2NORM | CLEORA_F_ROWSQ | CLEORA_F_SCALE))),
               "L1NORM is exclusive with L2NORM / ROWSQ / SCALE");
    CL_REQUIRE(!(flags & CLEORA_F_ROWSQ_CONT) || ((flags & CLEORA_F_ROWSQ) && !(flags & (CLEORA_F_FASTNORM | CLEORA_F_L2NORM))),
               "ROWSQ_CONT continues an exact-order ROWSQ (no L2NORM / FASTNORM in the same call)");
    return CLEORA_OK;
}

// Calls f(G, V, W, FULL) with the compile-time shape for a row of d floats.
template <class F>
bool dispatch_shape(uint32_t d, bool w4, F &&f) {
    if (w4) {
        const uint32_t chunks = d / 4;
        if (chunks <= 8) { chunks == 8 ? f(I<8>{}, I<1>{}, I<4>{}, I<1>{}) : f(I<8>{}, I<1>{}, I<4>{}, I<0>{}); return true; }
        if (chunks <= 16) { chunks == 16 ? f(I<16>{}, I<1>{}, I<4>{}, I<1>{}) : f(I<16>{}, I<1>{}, I<4>{}, I<0>{}); return true; }
        if (chunks <= 32) { chunks == 32 ? f(I<32>{}, I<1>{}, I<4>{}, I<1>{}) : f(I<32>{}, I<1>{}, I<4>{}, I<0>{}); return true; }
        if (chunks <= 64) { chunks == 64 ? f(I<64>{}, I<1>{}, I<4>{}, I<1>{}) : f(I<64>{}, I<1>{}, I<4>{}, I<0>{}); return true; }
        if (chunks <= 128) { chunks == 128 ? f(I<64>{}, I<2>{}, I<4>{}, I<1>{}) : f(I<64>{}, I<2>{}, I<4>{}, I<0>{}); return true; }
        if (chunks <= 256) { chunks == 256 ? f(I<64>{}, I<4>{}, I<4>{}, I<1>{}) : f(I<64>{}, I<4>{}, I<4>{}, I<0>{}); return true; }
        if (chunks <= 512) { chunks == 512 ? f(I<64>{}, I<8>{}, I<4>{}, I<1>{}) : f(I<64>{}, I<8>{}, I<4>{}, I<0>{}); return true; }
        return false;
    }
    if (d <= 64) { f(I<64>{}, I<1>{}, I<1>{}, I<0>{}); return true; }
    if (d <= 128) { f(I<64>{}, I<2>{}, I<1>{}, I<0>{}); return true; }
    if (d <= 256) { f(I<64>{}, I<4>{}, I<1>{}, I<0>{}); return true; }
    if (d <= 512) { f(I<64>{}, I<8>{}, I<1>{}, I<0>{}); return true; }
    if (d <= 1024) { f(I<64>{}, I<16>{}, I<1>{}, I<0>{}); return true; }
    return false;
}

inline dim3 grid_for(uint64_t items, int per_block) {
    return grid_1d_as_2d((items + per_block - 1) / per_block);
}

// Scratch for the hub-segment partial sums, grown when a wider d arrives: stream-ordered (the old block is released
// behind the launches that still use it), so a `*_dev` call stays enqueue-only.
int ensure_partial(const cleora_graph *g, uint32_t d, bool segmented, hipStream_t stream) {
    const uint64_t need = (segmented ? g->n_hub_segments : g->n_io_rows) * (uint64_t)d;
    if (need <= g->hub_partial_elems) return CLEORA_OK;
    if (g->hub_partial) CL_HIP(hipFreeAsync(g->hub_partial, stream));
    g->hub_partial = nullptr;
    g->hub_partial_elems = 0;
    CL_HIP(hipMallocAsync(reinterpret_cast<void **>(&g->hub_partial), need * sizeof(float), stream));
    g->hub_partial_elems = need;
    return CLEORA_OK;
}

hipEvent_t take_event(const cleora_graph *g) {
    hipEvent_t e = nullptr;
    if (!g->ev_pool.empty()) {
        e = g->ev_pool.back();
        g->ev_pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        return nullptr;
    }
    g->ev_used.push_back(e);
    return e;
}

inline void mark(const cleora_graph *g, hipStream_t stream) {
    if (!g->timing) return;
    if (hipEvent_t e = take_event(g)) (void)hipEventRecord(e, stream);
}

// Lanes per edge of the in-order hub launch (hub_inorder_kernel<R, L>).  The longest row is one in-order chain; beside a main kernel
// that saturates HBM a wavefront advances one step per (load latency / 48) — measured 10 us per request at C3, i.e. ~52 ns per edge at
// 4 edges per step.  The launch should end well before the main kernel does (~nnz * d * 4 bytes at ~6.4 TB/s): the narrowest form whose
// estimated chain stays under HALF the main kernel's estimated time, else the 2-lane form.  Narrower forms cost proportionally more
// vector instructions in total, so the wide form stays wherever the chain is hidden anyway (C5: 1.13 M edges beside 190 ms).
// g->hub_lanes != 0 forces a form (cleora_graph_set_hub_lanes: tests, A/B runs).
// Rows of the in-order hub launch (longest first) that take hub_chain_kernel.  Automatic: the rows whose chain in hub_inorder_kernel
// (26 ns per edge measured with an idle memory system, 52 under a saturating main kernel) would outlast a quarter of the main kernel's
// estimated time — config 3 on one GPU: the one row of 431 465 edges; a 6 M-edge block of an 8-way partition: every row beyond ~10 k
// edges — at least 4 096 edges, at most 128 blocks (a block holds a CU's whole register file).  g->hub_chain_min forces a threshold
// (cleora_graph_set_hub_chain_min: 1 = every row of the hub launch, UINT64_MAX = none).
uint64_t hub_chain_rows(const cleora_graph *g, uint32_t d) {
    if (g->io_len_desc.empty()) return 0;
    uint64_t min_edges = g->hub_chain_min, cap_blocks = ~0ull;
    if (min_edges == 0) {
        const double main_ns = (double)g->nnz * (double)d * 4.0 / 6.4e3;      // bytes / (6.4e12 B/s) in ns
        min_edges = (uint64_t)(0.25 * main_ns / 26.0);
        if (min_edges < 4096) min_edges = 4096;
        cap_blocks = 128;
    }
    const uint64_t slabs = (d + 63) / 64;
    uint64_t rows = 0;
    while (rows < g->io_len_desc.size() && g->io_len_desc[rows] >= min_edges && (rows + 1) * slabs <= cap_blocks) ++rows;
    return rows;
}

int hub_lanes(const cleora_graph *g, uint32_t d) {
    if (g->hub_lanes == 4 || g->hub_lanes == 2) return g->hub_lanes;
    const double main_ns = (double)g->nnz * (double)d * 4.0 / 6.4e3;          // bytes / (6.4e12 B/s) in ns
    const uint64_t n_chain = hub_chain_rows(g, d);
    const uint64_t longest = n_chain < g->io_len_desc.size() ? g->io_len_desc[n_chain] : 0;     // the longest row hub_inorder_kernel keeps
    const double chain4_ns = (double)longest * 52.0;
    return chain4_ns <= 0.5 * main_ns ? 4 : 2;
}

// Side stream and the fork / join events of the in-order hub launch, created on first use.  DEFAULT priority on purpose: with a
// highest- (or lowest-) priority stream, the hub launch of the third and fifth handle of a process did not overlap the main kernel
// at all — 37.4 ms per SpMM instead of 33.1 at C3, the hub launch's 4 ms in front of the main kernel (the runtime keeps few hardware
// queues per non-default priority) — while default-priority streams overlapped for every handle (32.7-32.9 ms;
// profiles/r05_hub_schedule_ab.jsonl, scripts/r05/threshold_probe.py).  hipExtAnyOrderLaunch, which would let the two kernels share
// ONE stream, is not honoured on gfx9 parts (scripts/r05/anyorder_probe.hip: 10.0 ms for two 5 ms kernels).
int ensure_hub_stream(const cleora_graph *g) {
    if (g->hub_stream) return CLEORA_OK;
    CL_HIP(hipStreamCreateWithFlags(&g->hub_stream, hipStreamNonBlocking));
    CL_HIP(hipEventCreateWithFlags(&g->hub_fork, hipEventDisableTiming));
    CL_HIP(hipEventCreateWithFlags(&g->hub_join, hipEventDisableTiming));
    return CLEORA_OK;
}

// One SpMM over a column panel that fits the register-resident shapes.
int propagate_panel(const cleora_graph *g, SpmmArgs a, bool w4, bool segmented, hipStream_t stream, hipEvent_t *hub_join_out = nullptr) {
    const uint32_t d = a.r.d;
    bool ok = true;
    const bool inorder = g->n_hub_rows && !segmented;      // the reference-order schedule of the long rows
    const bool hub_launch = inorder && g->n_io_rows;       // ... of which some run on the in-order hub launch
    mark(g, stream);
    if (hub_launch) {
        // the hub rows, longest first, on the side stream beside the main launch (src/embedding.rs:76-83's order)
        int rc = ensure_hub_stream(g);
        if (rc != CLEORA_OK) return rc;
        CL_HIP(hipEventRecord(g->hub_fork, stream));
        CL_HIP(hipStreamWaitEvent(g->hub_stream, g->hub_fork, 0));
        a.hub_by_len = g->hub_by_len;
        a.hub_rows = g->io_rows;
        a.nnz = g->nnz;
        // the few longest rows: one 8-wave block per (row, 64-column slab) — products by producer waves, the chain by a wave that only adds
        const uint64_t n_chain = w4 ? hub_chain_rows(g, d) : 0;
        if (n_chain) {
            a.n_slabs = (d + 63) / 64;
            a.hub_first = 0;
            hipLaunchKernelGGL((hub_chain_kernel<8>), grid_1d_as_2d(n_chain * (uint64_t)a.n_slabs), dim3(512), 0, g->hub_stream, a);
        }
        a.hub_first = (uint32_t)n_chain;
        const int lanes = w4 ? hub_lanes(g, d) : 0;
        a.n_slabs = lanes ? (d + 16 * lanes - 1) / (16 * lanes) : (d + 63) / 64;
        if (g->n_io_rows > n_chain) {
            const dim3 grid = grid_1d_as_2d((g->n_io_rows - n_chain) * (uint64_t)a.n_slabs);
            if (lanes == 4) hipLaunchKernelGGL((hub_inorder_kernel<48, 4>), grid, dim3(64), 0, g->hub_stream, a);
            else if (lanes == 2) hipLaunchKernelGGL((hub_inorder_kernel<48, 2>), grid, dim3(64), 0, g->hub_stream, a);
            else hipLaunchKernelGGL((hub_inorder_kernel<64, 0>), grid, dim3(64), 0, g->hub_stream, a);
        }
        a.hub_first = 0;
        ok = dispatch_shape(d, w4, [&](auto G, auto V, auto W, auto) {
            constexpr int kG = decltype(G)::value;
            hipLaunchKernelGGL((hub_epilogue_kernel<kG, decltype(V)::value, decltype(W)::value>),
                               grid_for(g->n_io_rows, 256 / kG), dim3(256), 0, g->hub_stream, a, g->n_io_rows);
        });
        CL_HIP(hipEventRecord(g->hub_join, g->hub_stream));
    }
    mark(g, stream);
    a.hub_rows = g->hub_rows;
    a.mid_rows = inorder ? g->mid_rows : nullptr;
    a.n_segments = inorder ? g->n_mid_rows : g->n_hub_segments;
    if (inorder && !g->n_mid_rows) a.mid_rows = nullptr;
    a.n_items = a.n_segments + g->n_rows;
    if (ok && a.n_items) {
        ok = dispatch_shape(d, w4, [&](auto G, auto V, auto W, auto FULL) {
            constexpr int kG = decltype(G)::value, kV = decltype(V)::value, kW = decltype(W)::value;
            constexpr bool kFull = decltype(FULL)::value != 0;
            if constexpr (kW == 4) {
                // hot-column cache policy: needs the marked column copy and, for sub-wave groups, a
                // matrix that one buffer descriptor can span
                const uint32_t *hot = (kG == 64 || a.x_bytes < (1ull << 32)) ? ensure_hot_cols(g, d, a.ldx, stream) : nullptr;
                if (hot) {
                    SpmmArgs h = a;
                    h.col = hot;
                    hipLaunchKernelGGL((spmm_rows_kernel<kG, kV, kW, kFull, true>),
                                       grid_for(h.n_items, 256 / kG), dim3(256), 0, stream, h);
                    return;
                }
            }
            hipLaunchKernelGGL((spmm_rows_kernel<kG, kV, kW, kFull, false>),
                               grid_for(a.n_items, 256 / kG), dim3(256), 0, stream, a);
        });
    }
    mark(g, stream);
    if (hub_launch) {
        // the caller may take the join itself (sharded.hip: only the gather of this block — and the next iteration — need the hub
        // rows, so the next block's launch need not wait for this block's longest chain)
        if (hub_join_out) *hub_join_out = g->hub_join;
        else CL_HIP(hipStreamWaitEvent(stream, g->hub_join, 0));
    } else if (ok && g->n_hub_rows && !inorder) {
        ok = dispatch_shape(d, w4, [&](auto G, auto V, auto W, auto) {
            hipLaunchKernelGGL((hub_finish_kernel<decltype(G)::value, decltype(V)::value, decltype(W)::value>),
                               dim3((unsigned)g->n_hub_rows), dim3(256), 0, stream, a);
        });
    }
    mark(g, stream);
    if (!ok) {
        set_error("internal: no kernel shape for d");
        return CLEORA_E_INVALID;
    }
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace

int launch_propagate(const cleora_graph *g, int kind, const float *x, uint64_t ldx, uint32_t d,
                     float *y, uint64_t ldy, uint32_t flags, float rw, const float *x_self,
                     double *row_sqdiff, float *row_sumsq, hipStream_t stream, const float *val_override, hipEvent_t *hub_join_out) {
    CL_REQUIRE(g != nullptr, "graph handle is NULL");
    if (hub_join_out) *hub_join_out = nullptr;
    if (!val_override) {
        CL_REQUIRE(kind == CLEORA_LEFT || kind == CLEORA_SYMMETRIC, "unknown markov_type");
        CL_REQUIRE(g->val[kind] != nullptr, "graph has no values for this markov_type");
    }
    CL_REQUIRE(d > 0 && ldx >= d && ldy >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && y != nullptr, "x / y is NULL");
    CL_REQUIRE(x != y, "x and y must not alias");
    flags = gate_residual(flags, rw);
    if (flags & (CLEORA_F_RESIDUAL | CLEORA_F_SQDIFF)) {
        if (!x_self && g->n_rows == g->n_cols) x_self = x;
        CL_REQUIRE(x_self != nullptr, "x_self is required for RESIDUAL / SQDIFF on a row shard");
    }
    if (flags & CLEORA_F_SQDIFF) CL_REQUIRE(row_sqdiff != nullptr, "row_sqdiff is NULL");
    if (flags & (CLEORA_F_ROWSQ | CLEORA_F_SCALE)) CL_REQUIRE(row_sumsq != nullptr, "row_sumsq is NULL");
    if (int rc = check_norm_flags(flags)) return rc;
    if (g->n_rows == 0) return CLEORA_OK;

    std::lock_guard<std::mutex> lock(g->mu);
    CL_HIP(hipSetDevice(g->device));
    // hub rows: the reference's order unless the caller asks for the segmented sum (or a row is too long for the
    // 32-bit offsets of the in-order kernel's (col, val) stream)
    const bool segmented = (flags & CLEORA_F_HUB_SEGMENTS) || !g->hub_inorder_ok || ldx >= (1ull << 32);
    flags &= ~CLEORA_F_HUB_SEGMENTS;
    if (g->n_hub_rows) {
        const int rc = ensure_partial(g, d, segmented, stream);
        if (rc != CLEORA_OK) return rc;
    }

    SpmmArgs a{};
    a.rowptr = g->rowptr;
    a.col = g->col;
    a.val = val_override ? val_override : g->val[kind];
    a.x = x;
    a.ldx = ldx;
    a.x_bytes = g->n_cols * ldx * sizeof(float);
    a.hub_threshold = g->hub_threshold;
    a.seg_row = g->seg_row;
    a.seg_begin = g->seg_begin;
    a.hub_segment = g->hub_segment;
    a.hub_rows = g->hub_rows;
    a.hub_seg_first = g->hub_seg_first;
    a.partial = g->hub_partial;
    a.r.y = y;
    a.r.ldy = ldy;
    a.r.x_self = x_self;
    a.r.ldxs = ldx;
    a.r.row_sqdiff = row_sqdiff;
    a.r.row_sumsq = row_sumsq;
    a.r.rw = rw;
    a.r.alpha = 1.0f - rw;
    a.r.flags = flags | (g->n_rows * ldy * sizeof(float) >= kStreamStoreMinBytes ? kStreamStores : 0u);
    a.r.d = d;

    const bool w4 = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) && aligned16(y) &&
                    (!x_self || aligned16(x_self));
    if (d <= (w4 ? kMaxD4 : kMaxD1)) return propagate_panel(g, a, w4, segmented, stream, hub_join_out);

    // Wider rows: SpMM column panel by column panel without the epilogue, then the wide row pass.
    const uint32_t panel = w4 ? kMaxD4 : kMaxD1;
    for (uint32_t c0 = 0; c0 < d; c0 += panel) {
        SpmmArgs p = a;
        p.x = x + c0;
        p.r.y = y + c0;
        p.r.d = (d - c0) < panel ? (d - c0) : panel;
        p.r.flags = 0;
        const int rc = propagate_panel(g, p, w4, segmented, stream);
        if (rc != CLEORA_OK) return rc;
    }
    if (flags & (CLEORA_F_L2NORM | CLEORA_F_L1NORM | CLEORA_F_RESIDUAL | CLEORA_F_SQDIFF | CLEORA_F_ROWSQ | CLEORA_F_SCALE))
        return launch_rowops(y, ldy, g->n_rows, d, y, ldy, flags, rw, x_self, row_sqdiff, row_sumsq, stream, ldx);
    return CLEORA_OK;
}

int launch_rowops(const float *x, uint64_t ldx, uint64_t n, uint32_t d, float *y, uint64_t ldy,
                  uint32_t flags, float rw, const float *x_self, double *row_sqdiff,
                  float *row_sumsq, hipStream_t stream, uint64_t ldxs) {
    CL_REQUIRE(d > 0 && ldx >= d && ldy >= d, "bad d / leading dimension");
    CL_REQUIRE(x != nullptr && y != nullptr, "x / y is NULL");
    if (ldxs == 0) ldxs = ldx;   // x_self rows are strided like x unless the caller says otherwise
    flags = gate_residual(flags & ~CLEORA_F_HUB_SEGMENTS, rw);   // (a row pass has no hub rows: the loops hand their flag set through)
    if (flags & (CLEORA_F_RESIDUAL | CLEORA_F_SQDIFF)) CL_REQUIRE(x_self != nullptr, "x_self is NULL");
    if (flags & CLEORA_F_SQDIFF) CL_REQUIRE(row_sqdiff != nullptr, "row_sqdiff is NULL");
    if (flags & (CLEORA_F_ROWSQ | CLEORA_F_SCALE)) CL_REQUIRE(row_sumsq != nullptr, "row_sumsq is NULL");
    if (int rc = check_norm_flags(flags)) return rc;
    if (n == 0) return CLEORA_OK;
    RowArgs ra{};
    ra.y = y;
    ra.ldy = ldy;
    ra.x_self = x_self;
    ra.ldxs = ldxs;
    ra.row_sqdiff = row_sqdiff;
    ra.row_sumsq = row_sumsq;
    ra.rw = rw;
    ra.alpha = 1.0f - rw;
    ra.flags = flags | (n * ldy * sizeof(float) >= kStreamStoreMinBytes ? kStreamStores : 0u);
    ra.d = d;
    const bool w4 = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(x) && aligned16(y) &&
                    (!x_self || (aligned16(x_self) && ldxs % 4 == 0));
    if (flags == CLEORA_F_L2NORM && w4 && d % 64 == 0) {   // plain exact-order normalise: 16 lanes per row
        const dim3 grid = grid_for(n, 16);
        bool done = true;
        switch (d / 64) {
            case 1: hipLaunchKernelGGL(l2_exact16_kernel<1>, grid, dim3(256), 0, stream, x, ldx, n, y, ldy); break;
            case 2: hipLaunchKernelGGL(l2_exact16_kernel<2>, grid, dim3(256), 0, stream, x, ldx, n, y, ldy); break;
            case 4: hipLaunchKernelGGL(l2_exact16_kernel<4>, grid, dim3(256), 0, stream, x, ldx, n, y, ldy); break;
            case 8: hipLaunchKernelGGL(l2_exact16_kernel<8>, grid, dim3(256), 0, stream, x, ldx, n, y, ldy); break;
            case 16: hipLaunchKernelGGL(l2_exact16_kernel<16>, grid, dim3(256), 0, stream, x, ldx, n, y, ldy); break;
            default: done = false;
        }
        if (done) {
            CL_HIP(hipGetLastError());
            return CLEORA_OK;
        }
    }
    const bool ok = dispatch_shape(d, w4, [&](auto G, auto V, auto W, auto) {
        hipLaunchKernelGGL((rowops_kernel<decltype(G)::value, decltype(V)::value, decltype(W)::value>),
                           grid_for(n, 256 / decltype(G)::value), dim3(256), 0, stream, x, ldx, n, ra);
    });
    if (!ok) {
        hipLaunchKernelGGL(rowops_wide_kernel, grid_for(n, 4), dim3(256), 0, stream, x, ldx, n, ra);
    }
    CL_HIP(hipGetLastError());
    return CLEORA_OK;
}

}  // namespace cleora
