// project_f16.hip — the projection of the whitened loop's INTERMEDIATE iterations at d = k = 256 (BASELINE configs 2 and 3):
//     out[r] = normalise( (x[r] - s_r mu) T )                 (pycleora/__init__.py:157-163 moved behind the SpMM: abi.hip)
// on the f16 matrix cores with the transform RESIDENT IN REGISTERS.
//
// Why another form beside project_split_kernel (whiten.hip).  That kernel makes every f32 product from six bf16 MFMAs of
// three-way split operands and streams the 384 KiB of split T through LDS for every 64-row tile: at the C3 shape it is bound by
// instruction issue and the LDS port (8.4-8.9 ms against an HBM floor of 3.2 ms; MFMA pipe busy 0.50).  Inside the loop the
// operand is BOUNDED — x = A Y with unit rows Y, so |x_rj| <= sum_j |a_rj| =: B_r, |mu_j| <= 1 — which makes f16 usable: an f32
// value scaled into the top of the f16 range is the sum of two f16 values to 2^-22 (11 + 11 significand bits; the residual of a
// small element falls into f16's subnormals, whose ABSOLUTE spacing, 2^-24 against row values scaled to ~2^8..2^14, is what a
// dot product cares about), and a product of two f16 values is exact in f32.  So
//     x t = x1 t1 + x1 t2 + x2 t1   + O(2^-21 |x t|)
// — THREE MFMAs per product instead of six, a two-way instead of a three-way split, and a split T of 256 x 256 x 2 x 2 B =
// 256 KiB = 128 VGPRs per lane for a wave that owns 32 output columns: no B traffic at all.  The per-element error, ~2^-22
// relative with random sign (rms over a 256-term dot product ~1e-7 of its magnitude), is the error class of the f32 GEMM this
// replaces (pycleora/__init__.py:163 is numpy's sgemm); it is used for intermediate iterations only — the last iteration's
// projection, whose output IS the result, and every projection of unbounded user data keep the six-product bf16 form.
//
//   * block = 8 waves (one block per CU, persistent over 64-row tiles); wave w owns output columns [32 w, 32 w + 32): its 32 B
//     fragments (16 k-steps x hi / lo) stay in 128 VGPRs for the whole launch.
//   * A: every wave loads 128-byte pieces of eight rows per instruction (wave w = columns [32 w, +32)), TWO tiles ahead, in registers;
//     centred, scaled by the row's power of two (2^e_r with (B_r + |s_r|) 2^e_r < 2^14), split, and written as f16 fragments —
//     [row half][k-step][hi / lo][lane] x 16 B, lane-linear: conflict-free ds_write_b64 / ds_read_b128 — into one of two 64 KiB
//     buffers, between the MFMAs of the tile before.  All eight waves read all fragments of a tile.
//   * per tile and wave: 64 ds_read_b128, 96 MFMAs with the operands SWAPPED (D^T = T^T x^T: a lane's accumulators are sixteen columns
//     of one row), two 32-row halves on alternating accumulators.
//   * epilogue straight from the accumulators: column scales (T's columns are scaled into f16's range by powers of two as well), the
//     lane's partial sum of its row, 16 partial sums per row through 8 KiB of LDS and the tile's ONE barrier, the row's factor, lane
//     swaps, 16-byte stores.  Scaling by powers of two commutes with every rounding involved, so the row scale is undone in the row's
//     final factor: same values as unscaled arithmetic.
// Where the time goes (cycle counters around the phases in a development build, 611 tiles per block at the C3 shape: scripts/rejected/
// project_f16_all_forms.hip.txt): products 6 500-6 800 cycles per tile (the matrix pipe's own time: 6 144), the epilogue 5 100-5 400 —
// ~600 for the partial sums, 500-2 400 at the barrier, 1 800-3 900 from the barrier to the last store issued (the waves dispatched second
// lose the arbitration).  Without the stores the launch takes 4.05 ms instead of 4.9, without the reloads 4.05-4.2: the vector, LDS and
// memory instructions of a tile (~390 per wave beside its 96 MFMAs) are what the matrix pipe waits for — not their count (a fused centring
// and v_fma_mix_f32 residuals, a quarter fewer vector instructions in the fragment production: 4.82-4.84 against 4.87; no SLP packing: the
// same) but their latencies in two in-order waves per SIMD — and three rearrangements of
// them — the epilogue pipelined into the next tile's MFMAs (32-row tiles), the two waves of a SIMD in opposite phases, row-major
// read-back through LDS for whole-line stores — measured 4.8 / 5.5 / no gain (same file).  The first form of this kernel (accumulators
// staged through LDS for a row-major pass, two barriers per tile) ran 5.3-5.45 ms.
#include <type_traits>

#include "common.h"
#include "project_common.h"

namespace cleora {
namespace {


constexpr int PF_D = 256;              // d = k
constexpr int PF_KS = PF_D / 16;       // k-steps
constexpr int PF_ROWS = 64;            // rows per tile
constexpr int PF_THREADS = 512;
constexpr int PF_BUF = 2 * PF_KS * 2 * 1024;        // fragment bytes per tile: [row half][k-step][split] x 1 KiB = 64 KiB

// T (256 x 256 row-major f32) -> per-column power-of-two scale, hi / lo f16 fragments in the consumer's register order:
// tp[((w * 16 + ks) * 2 + sp) * 64 + lane] = 8 x f16: lane (j, h) <-> column 32 w + pi^-1(j), k = 16 ks + 8 h + e, where the fragment
// lane of column c (of the wave's 32) is pi(c) = (c & 3) + 4 ((c >> 3) & 1) + 8 (2 (c >> 4) + ((c >> 2) & 1)) — chosen so that after the
// kernel's lane swaps a store instruction's four pieces per row are 64 contiguous bytes.   One block per column.
__global__ __launch_bounds__(256) void pack_transform_f16_kernel(const float *__restrict__ t, _Float16 *__restrict__ tp, float *__restrict__ colscale) {
    __shared__ float red[4];
    const uint32_t col = blockIdx.x, kk = threadIdx.x;
    const float v = t[(uint64_t)kk * PF_D + col];
    float m = fabsf(v);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((kk & 63) == 0) red[kk >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    // max |t| * 2^st in [2^13, 2^14); an all-zero (or non-finite) column is left alone
    int st = (m > 0.f && m < __builtin_inff()) ? PF_TOP - __builtin_amdgcn_frexp_expf(m) : 0;
    st = st > 100 ? 100 : (st < -100 ? -100 : st);
    const float tv = ldexpf(v, st);
    const _Float16 hi = (_Float16)tv;
    const _Float16 lo = (_Float16)(tv - (float)hi);
    const uint32_t w = col >> 5, cl = col & 31, ks = kk >> 4, h = (kk >> 3) & 1, e = kk & 7;
    const uint32_t j = (cl & 3) + 4 * ((cl >> 3) & 1) + 8 * (2 * (cl >> 4) + ((cl >> 2) & 1));   // fragment lane of column cl (see the store path)
    const uint64_t unit = ((uint64_t)(w * PF_KS + ks) * 2) * 64 + (h * 32 + j);
    tp[unit * 8 + e] = hi;
    tp[(unit + 64) * 8 + e] = lo;
    if (kk == 0) colscale[col] = ldexpf(1.0f, -st);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The MFMA's operands are swapped — D^T = T^T x^T — so that a lane's accumulators are SIXTEEN COLUMNS OF ONE ROW (32x32 C/D map:
// lane & 31 = tile row, reg -> fragment lane (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) of the wave's 32 columns) instead of sixteen rows
// of one column.  What that buys over staging the accumulator tile through LDS for a row-major pass:
//   * a row's sum of squares is an in-lane sum over registers: only 16 partial sums per row (8 waves x 2 lane halves) cross waves;
//   * four consecutive registers are four consecutive columns: the tile leaves as 16-byte stores straight from the accumulators;
//   * no staging buffer, so the two fragment buffers are a true double buffer and ONE barrier per tile remains: the next tile's
//     fragments are produced (centre, scale, split, ds_write) between the MFMAs of the current one instead of in a phase of their own.
// The loop is one basic block (full tiles only; the ragged last tile has a block of its own with masked stores): the loads of tile
// k + 2 stay in flight across the back edge behind counted waits.
constexpr size_t PF_RED = 2 * 16 * 64 * 4;                   // [parity][8 waves x 2 halves][row] partial sums
constexpr size_t PF_INFO = 8 * 2 * 64 * 8;                   // [wave][parity][row] {s_r, 2^e_r}: wave-private copies (no barrier)
constexpr size_t PF_LDS = 2 * (size_t)PF_BUF + PF_RED + PF_INFO + 2 * PF_D * 4;   // 146 KiB

__device__ __forceinline__ void pf_barrier() {                 // LDS-only hand-off: global loads / stores in flight are not drained
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

struct F16Args {
    ProjArgs p;
    const u32x4 *tp;
    const float *colscale;
    const float *rowscale;     // per row s_r (never null: the launcher passes ones)
    const float *rowbound;     // per row B_r (never null)
    uint64_t full_blocks;      // blocks sharing the full tiles (gridDim.x minus the ragged tile's block)
};

template <int NORM>
__global__ __launch_bounds__(PF_THREADS, 2) void project_f16_kernel(const F16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const frag = smem;                                             // [2][PF_BUF]
    float *const red = reinterpret_cast<float *>(smem + 2 * PF_BUF);              // [2][16][64]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = lane & 31, h = lane >> 5;
    float2 *const rinfo = reinterpret_cast<float2 *>(smem + 2 * PF_BUF + PF_RED) + w * 2 * 64;   // [2][64], this wave's
    float *const mean_s = reinterpret_cast<float *>(smem + 2 * PF_BUF + PF_RED + PF_INFO);
    float *const cs_s = mean_s + PF_D;
    const ProjArgs &p = a.p;
    // blocks 0 .. G-1 share the FULL tiles (no row of theirs needs a guard); block G, if launched, owns the ragged last tile
    const uint64_t tiles_full = p.n / PF_ROWS, G = a.full_blocks, b = blockIdx.x;
    const bool ragged = b >= G;
    const uint64_t cnt_full = ragged ? 0 : (tiles_full - b + G - 1) / G;

    h8v bhi[PF_KS], blo[PF_KS];
#pragma unroll
    for (int ks = 0; ks < PF_KS; ++ks) {
        bhi[ks] = __builtin_bit_cast(h8v, a.tp[((uint64_t)(w * PF_KS + ks) * 2 + 0) * 64 + lane]);
        blo[ks] = __builtin_bit_cast(h8v, a.tp[((uint64_t)(w * PF_KS + ks) * 2 + 1) * 64 + lane]);
    }
    if (t < PF_D) { mean_s[t] = p.mean[t]; cs_s[t] = a.colscale[t]; }

    // producer role (as in the staged form): rows 8 j + r8 of the tile, columns 32 w + 4 pc .. + 3
    const int r8 = (lane >> 1) & 7, pc = ((lane >> 4) << 1) | (lane & 1);
    const int ksub = pc >> 2, hh = (pc >> 1) & 1, half = pc & 1;
    const uint32_t col0 = 32u * w + 4u * pc;
    const uint32_t frag_lane_off = (uint32_t)((2 * w + ksub) * 2 * 1024 + (hh * 32 + r8) * 16 + half * 8);
    float4 P[8];
    float ri_s, ri_b;
    // full tiles: the tile index is clamped to the last full one (uniform: a prefetch past the block's share re-reads valid rows);
    // the ragged tile's block clamps rows instead
    auto issue_tile = [&](uint64_t tile) {
        tile = tile < tiles_full ? tile : tiles_full - 1;
        const uint64_t r0 = tile * PF_ROWS;
        ri_s = a.rowscale[r0 + (uint64_t)lane];                                   // the oldest loads of the batch: first to be waited for
        ri_b = a.rowbound[r0 + (uint64_t)lane];
        const float *const base = p.x + (r0 + (uint64_t)r8) * p.ldx + col0;
#pragma unroll
        for (int j = 0; j < 8; ++j) P[j] = *reinterpret_cast<const float4 *>(base + (uint64_t)(8 * j) * p.ldx);
    };
    auto issue_ragged = [&]() {
        const uint64_t r0 = tiles_full * PF_ROWS, last_row = p.n - 1;
        const uint64_t rl = r0 + (uint64_t)lane < last_row ? r0 + (uint64_t)lane : last_row;
        ri_s = a.rowscale[rl];
        ri_b = a.rowbound[rl];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t row = r0 + (uint64_t)(8 * j + r8) < last_row ? r0 + (uint64_t)(8 * j + r8) : last_row;
            P[j] = *reinterpret_cast<const float4 *>(p.x + row * p.ldx + col0);
        }
    };
    auto publish = [&](int par) {                                                 // lane L <-> row L of the tile, this wave's copy
        const float bound = fabsf(ri_b) + fabsf(ri_s);                            // |x - s mu| <= B + |s| (|mu| <= 1)
        int e = (bound > 0.f && bound < __builtin_inff()) ? PF_TOP - __builtin_amdgcn_frexp_expf(bound) : 0;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
        rinfo[par * 64 + lane] = make_float2(ri_s, ldexpf(1.0f, e));
    };
    auto produce_piece = [&](int j, unsigned char *fb, int par) {
        const float2 info = rinfo[par * 64 + 8 * j + r8];
        const float4 mu4 = *reinterpret_cast<const float4 *>(mean_s + col0);     // (re-read per piece: four registers fewer across the loop)
        const float xv[4] = {P[j].x, P[j].y, P[j].z, P[j].w};
        const float mv[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fmul_rn(centre(xv[e], mv[e], info.x, true), info.y);   // (x - s mu) 2^e: the scaling is exact
        uint32_t h0, l0, h1, l1;
        split2h_pair(o[0], o[1], h0, l0);
        split2h_pair(o[2], o[3], h1, l1);
        unsigned char *const dst = fb + (j >> 2) * (PF_BUF / 2) + frag_lane_off + (j & 3) * 128;
        *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(dst + 1024) = make_uint2(l0, l1);
    };

    // ---- prologue: fragments of the first tile, operands of the second in flight -----------------------------------------------
    if (ragged) issue_ragged(); else issue_tile(b);
    __syncthreads();                                                              // mean_s, cs_s
    publish(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) produce_piece(j, frag, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!ragged) issue_tile(b + G);
    pf_barrier();

    // The loop is ROTATED — { finish tile k ; products of tile k + 1 } — so that its back edge sits right behind the loads of tile
    // k + 3: entering from the prologue and from the back edge the same ten loads are the youngest vector-memory operations, and the
    // waits the compiler counts for them (behind the eight stores of `finish`) are the same on both paths (merged at the loop
    // header, the more conservative path wins: with the loop entered behind the stores the prologue's state did).
    f16v acc[2];
    auto products = [&](uint64_t k, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int par = (int)(k & 1);
        const uint64_t T = LAST ? tiles_full : b + k * G;
        const unsigned char *const fa = frag + par * PF_BUF + lane * 16;
        unsigned char *const fb = frag + (par ^ 1) * PF_BUF;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        // ---- the tile's products; the next tile's fragments between them ---------------------------------------------------------
        // one set of fragment registers: a k-step's hi fragments are re-read for the next k-step as soon as their four MFMAs have
        // issued (the two lo products cover the read), the lo fragments after theirs (the next k-step's four hi products cover it)
        h8v ah[2], al[2];
        auto read_hi = [&](int ks) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) ah[rt] = *reinterpret_cast<const h8v *>(fa + rt * (PF_BUF / 2) + (ks * 2 + 0) * 1024);
        };
        auto read_lo = [&](int ks) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) al[rt] = *reinterpret_cast<const h8v *>(fa + rt * (PF_BUF / 2) + (ks * 2 + 1) * 1024);
        };
        read_hi(0);
        read_lo(0);
#pragma unroll
        for (int ks = 0; ks < PF_KS; ++ks) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[ks], ah[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[ks], ah[1], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo[ks], ah[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(blo[ks], ah[1], acc[1], 0, 0, 0);
            if (ks + 1 < PF_KS) read_hi(ks + 1);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[ks], al[0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bhi[ks], al[1], acc[1], 0, 0, 0);
            if (ks + 1 < PF_KS) read_lo(ks + 1);
            if constexpr (!LAST) {
                if (ks == 5) publish(par ^ 1);
                if (ks >= 6 && ks < 14) produce_piece(ks - 6, fb, par ^ 1);
            }
        }
        if constexpr (!LAST) {
            __builtin_amdgcn_sched_barrier(0);
            issue_tile(T + 2 * G);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto finish = [&](uint64_t k, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int par = (int)(k & 1);
        const uint64_t T = LAST ? tiles_full : b + k * G;
        // ---- column scales; the lane's partial sums of its two rows -------------------------------------------------------------
        float part[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                             // four columns' scales at a time (registers)
            const float4 cv = *reinterpret_cast<const float4 *>(cs_s + 32 * w + 16 * (q >> 1) + 8 * h + 4 * (q & 1));   // register group q <-> these columns
            const float csq[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float m[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[rt][4 * q + e] * csq[e];
                    acc[rt][4 * q + e] = v;
                    m[e] = NORM == 2 ? fabsf(v) : v * v;
                }
                part[rt] += (m[0] + m[1]) + (m[2] + m[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NORM != 0) {
            red[(par * 16 + 2 * w + h) * 64 + c] = part[0];                      // [parity][slot][row]: lane-linear writes and reads
            red[(par * 16 + 2 * w + h) * 64 + 32 + c] = part[1];
        }
        pf_barrier();
        // ---- whole-row factors, 16-byte stores straight from the accumulators ---------------------------------------------------
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt) __builtin_amdgcn_sched_barrier(0);                            // one row's sixteen partial sums in registers at a time
            const int rr = 32 * rt + c;
            const float sc = rinfo[par * 64 + rr].y;                              // 2^e_r; its reciprocal by the exponent field
            const float unscale = __uint_as_float(0x7F000000u - __float_as_uint(sc));
            float f = unscale;
            if constexpr (NORM != 0) {
                const float *const rp = red + par * 16 * 64 + rr;
                float u[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) u[q] = rp[q * 64];
                const float s = (((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]))) +
                                (((u[8] + u[9]) + (u[10] + u[11])) + ((u[12] + u[13]) + (u[14] + u[15])));
                // L2: v (1 / max(sqrt(S), 1e-10)) like src/embedding.rs:98-102; L1: v / max(S, 1e-10) (pycleora/__init__.py:947-950); the
                // sums were taken on rows scaled by 2^e: S_true = S 2^-2e (L2) / S 2^-e (L1)
                f = NORM == 1 ? unscale * (1.0f / fmaxf(sqrtf(s) * unscale, 1e-10f)) : fmaxf(s * unscale, 1e-10f);
            }
            float o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = NORM == 2 ? (acc[rt][r] * unscale) / f : acc[rt][r] * f;
            // A lane holds 16-byte pieces of ONE row; a store instruction that took one piece per lane would touch 32 rows (32 cache
            // lines, 32 B each: ~60 cycles of the CU's one address unit per instruction, 64 such instructions per tile — measured as
            // the critical path).  v_permlane16_swap trades the odd 16-lane rows of register group 2 g with the even rows of group
            // 2 g + 1: afterwards group 2 g holds, in its four 16-lane rows, the four pieces (64 contiguous bytes: the fragment order
            // of T's columns was chosen for that, pack_transform_f16_kernel) of tile rows 0-15, group 2 g + 1 those of rows 16-31.
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(o[8 * g + e]), "+v"(o[8 * g + 4 + e]));
            const uint64_t row0 = T * PF_ROWS + (uint64_t)(32 * rt + (lane & 15));
            float *const orow = p.out + row0 * p.ldo + 32 * w + 4 * (lane >> 4);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) {                                     // u: tile rows 0-15 / 16-31 of this half
                    const float4 v = make_float4(o[8 * g + 4 * u], o[8 * g + 4 * u + 1], o[8 * g + 4 * u + 2], o[8 * g + 4 * u + 3]);
                    if (!LAST || row0 + 16 * u < p.n)
                        *reinterpret_cast<float4 *>(orow + (uint64_t)(16 * u) * p.ldo + 16 * g) = v;
                }
        }
    };
    if (ragged) {
        products(0, std::true_type{});
        finish(0, std::true_type{});
        return;
    }
    products(0, std::false_type{});
    for (uint64_t k = 0; k + 1 < cnt_full; ++k) {
        finish(k, std::false_type{});
        products(k + 1, std::false_type{});
    }
    finish(cnt_full - 1, std::false_type{});
}

__global__ __launch_bounds__(256) void fill_ones_kernel(float *__restrict__ v, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = 1.0f;
}

}  // namespace

bool project_f16_applies(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t k, const float *out, uint64_t ldo, const float *x2) {
    auto aligned16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return d == PF_D && k == PF_D && x2 == nullptr && n >= 1 && ldx % 4 == 0 && ldo % 4 == 0 && aligned16(x) && aligned16(out) &&
           ldx < (1u << 24) && ldo < (1u << 24);                              // 32-bit per-lane byte offsets
}

// out = normalise((x - rowscale (x) mean) T) for BOUNDED operands: |x[r][j]| <= rowbound[r] (nullptr: 1), |mean[j]| <= 1.
// norm: 0 none, 1 L2, 2 L1 (always applied in the epilogue: whole rows live in one block).
int launch_project_f16(const float *x, uint64_t ldx, uint64_t n, const float *mean, const float *t, float *out, uint64_t ldo,
                       hipStream_t stream, const float *rowscale, const float *rowbound, int norm) {
    CL_REQUIRE(project_f16_applies(x, ldx, n, PF_D, PF_D, out, ldo, nullptr), "internal: the f16 projection does not apply to this shape");
    CL_REQUIRE(mean != nullptr && t != nullptr, "mean / transform is NULL");
    _Float16 *tp = nullptr;
    float *colscale = nullptr;
    CL_HIP(hipMallocAsync(reinterpret_cast<void **>(&tp), (size_t)PF_D * PF_D * 2 * sizeof(_Float16) + PF_D * sizeof(float), stream));
    colscale = reinterpret_cast<float *>(tp + (size_t)PF_D * PF_D * 2);
    hipLaunchKernelGGL(pack_transform_f16_kernel, dim3(PF_D), dim3(256), 0, stream, t, tp, colscale);
    static int cus = 0;
    if (!cus) {
        int dev = 0, c = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev);
        cus = c > 0 ? c : 256;
    }
    hipError_t e = hipSuccess;
    // the kernel reads both per-row vectors unconditionally (its loop is one basic block): an absent one is a vector of ones
    float *ones = nullptr;
    if (!rowscale || !rowbound) {
        CL_HIP(hipMallocAsync(reinterpret_cast<void **>(&ones), n * sizeof(float), stream));
        hipLaunchKernelGGL(fill_ones_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ones, n);
    }
    F16Args a{};
    a.p.x = x;
    a.p.ldx = ldx;
    a.p.n = n;
    a.p.d = PF_D;
    a.p.mean = mean;
    a.p.k = PF_D;
    a.p.out = out;
    a.p.ldo = ldo;
    a.p.norm = norm;
    a.tp = reinterpret_cast<const u32x4 *>(tp);
    a.colscale = colscale;
    a.rowscale = rowscale ? rowscale : ones;
    a.rowbound = rowbound ? rowbound : ones;
    const uint64_t tiles_full = n / PF_ROWS;
    a.full_blocks = tiles_full < (uint64_t)cus ? tiles_full : (uint64_t)cus;
    const unsigned gt = (unsigned)a.full_blocks + (n % PF_ROWS ? 1u : 0u);          // + the ragged tile's block
    auto go = [&](auto kernel) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF_LDS);
        if (e == hipSuccess) hipLaunchKernelGGL(kernel, dim3(gt), dim3(PF_THREADS), PF_LDS, stream, a);
    };
    if (norm == 1) go(project_f16_kernel<1>);
    else if (norm == 2) go(project_f16_kernel<2>);
    else go(project_f16_kernel<0>);
    const hipError_t le = e != hipSuccess ? e : hipGetLastError();
    if (ones) CL_HIP(hipFreeAsync(ones, stream));
    CL_HIP(hipFreeAsync(tp, stream));
    CL_HIP(le);
    return CLEORA_OK;
}

}  // namespace cleora
