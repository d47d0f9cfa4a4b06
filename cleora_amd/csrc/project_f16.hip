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
//   * A: every wave loads 128-byte pieces of eight rows per instruction (wave w = columns [32 w, +32): the column means are four
//     registers), one tile ahead, in registers; centred, scaled by the row's power of two (2^e_r with (B_r + |s_r|) 2^e_r < 2^14),
//     split, and written as f16 fragments — [row half][k-step][hi / lo][lane] x 16 B, lane-linear: conflict-free ds_write_b64 /
//     ds_read_b128 — into one of two 64 KiB buffers.  All eight waves read all fragments of the tile.
//   * per tile and wave: 64 ds_read_b128, 96 MFMAs (two 32-row halves: consecutive MFMAs alternate accumulators).
//   * epilogue: the accumulators go through LDS (the buffer just consumed) so that every wave finishes whole ROWS: column scales
//     (T's columns are scaled to the f16 range by powers of two as well), sum of squares, the row's factor, one coalesced 1 KiB
//     store per row.  Scaling by powers of two commutes with every rounding involved, so the row scale is undone in the row's
//     final factor: same values as unscaled arithmetic.
#include "common.h"
#include "project_common.h"

namespace cleora {
namespace {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

constexpr int PF_D = 256;              // d = k
constexpr int PF_KS = PF_D / 16;       // k-steps
constexpr int PF_ROWS = 64;            // rows per tile
constexpr int PF_THREADS = 512;
constexpr int PF_BUF = 2 * PF_KS * 2 * 1024;        // fragment bytes per tile: [row half][k-step][split] x 1 KiB = 64 KiB
constexpr int PF_TOP = 14;             // operands are scaled below 2^14 (f16 overflows at 65504)
constexpr size_t PF_LDS = 2 * (size_t)PF_BUF + 8 * 8 * 64 * 4 + 2 * 64 * 16 + 2 * PF_D * 4;   // 148 KiB

// v from the lane a DPP control selects inside this lane's 16-lane row (0xB1 / 0x4E: quad_perm [1,0,3,2] / [2,3,0,1]; 0x141 / 0x140:
// row_half_mirror / row_mirror)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// (lo, hi) -> packed f16 pairs p1 = f16(v), p2 = f16(v - p1): v = p1 + p2 to 2^-22 |v| (plus f16's subnormal spacing, 2^-24 absolute)
__device__ __forceinline__ void split2h_pair(float lo, float hi, uint32_t &p1, uint32_t &p2) {
    const f2v v = {lo, hi};
    const h2v a = __builtin_convertvector(v, h2v);                 // round to nearest even
    p1 = __builtin_bit_cast(uint32_t, a);
    const f2v r = v - __builtin_convertvector(a, f2v);             // exact
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, h2v));
}

// T (256 x 256 row-major f32) -> per-column power-of-two scale, hi / lo f16 fragments in the consumer's register order:
// tp[((w * 16 + ks) * 2 + sp) * 64 + lane] = 8 x f16: lane (j, h) <-> column 32 w + j, k = 16 ks + 8 h + e.   One block per column.
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
    const uint32_t w = col >> 5, j = col & 31, ks = kk >> 4, h = (kk >> 3) & 1, e = kk & 7;
    const uint64_t unit = ((uint64_t)(w * PF_KS + ks) * 2) * 64 + (h * 32 + j);
    tp[unit * 8 + e] = hi;
    tp[(unit + 64) * 8 + e] = lo;
    if (kk == 0) colscale[col] = ldexpf(1.0f, -st);
}

struct F16Args {
    ProjArgs p;
    const u32x4 *tp;
    const float *colscale;
    const float *rowbound;     // per row: a bound on |x[r][j]| (nullptr: 1)
    uint64_t tiles;
};

template <bool SCALED>
__global__ __launch_bounds__(PF_THREADS, 2) void project_f16_kernel(const F16Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *const frag = smem;                                             // [2][PF_BUF]; a consumed buffer doubles as the output stage
    float *const red = reinterpret_cast<float *>(smem + 2 * PF_BUF);              // [8 waves][8 rows][64 lanes]
    float4 *const rowinfo = reinterpret_cast<float4 *>(red + 8 * 8 * 64);         // [2][64]: {s_r, 2^e_r, 2^-e_r, -}
    float *const mean_s = reinterpret_cast<float *>(rowinfo + 2 * 64);            // [256]
    float *const cs_s = mean_s + PF_D;                                            // [256]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const ProjArgs &p = a.p;
    const uint64_t G = gridDim.x;
    uint64_t T = blockIdx.x;
    if (T >= a.tiles) return;

    // B: this wave's 32 columns of the split transform, for the whole launch
    h8v bhi[PF_KS], blo[PF_KS];
#pragma unroll
    for (int ks = 0; ks < PF_KS; ++ks) {
        bhi[ks] = __builtin_bit_cast(h8v, a.tp[((uint64_t)(w * PF_KS + ks) * 2 + 0) * 64 + lane]);
        blo[ks] = __builtin_bit_cast(h8v, a.tp[((uint64_t)(w * PF_KS + ks) * 2 + 1) * 64 + lane]);
    }
    if (t < PF_D) { mean_s[t] = p.mean[t]; cs_s[t] = a.colscale[t]; }

    // producer role: lane = (piece pair (ksub, hh) | row r8 | half): rows 8 j + r8 of the tile, columns 32 w + 4 pc .. + 3
    const int r8 = (lane >> 1) & 7, pc = ((lane >> 4) << 1) | (lane & 1);
    const int ksub = pc >> 2, hh = (pc >> 1) & 1, half = pc & 1;
    const uint32_t col0 = 32u * w + 4u * pc;
    const uint32_t frag_lane_off = (uint32_t)((2 * w + ksub) * 2 * 1024 + (hh * 32 + r8) * 16 + half * 8);   // + rt * 32 KiB + (j & 3) * 128 + sp * 1 KiB
    auto row_clamped = [&](uint64_t tile, int r) {
        const uint64_t row = tile * PF_ROWS + (uint64_t)r;
        return row < p.n ? row : p.n - 1;                                         // always a valid address
    };
    float4 P[8];
    auto issue_tile = [&](uint64_t tile) {
#pragma unroll
        for (int j = 0; j < 8; ++j) P[j] = *reinterpret_cast<const float4 *>(p.x + row_clamped(tile, 8 * j + r8) * p.ldx + col0);
    };
    // per-row constants of a tile, by the first 64 threads: {s, 2^e, 2^-e}
    float ri_s = 1.f, ri_b = 1.f;
    auto load_rowinfo = [&](uint64_t tile) {
        if (t < PF_ROWS) {
            const uint64_t row = row_clamped(tile, t);
            ri_s = SCALED ? p.rowscale[row] : 1.f;
            ri_b = a.rowbound ? a.rowbound[row] : 1.f;
        }
    };
    auto publish_rowinfo = [&](int buf) {
        if (t < PF_ROWS) {
            const float bound = fabsf(ri_b) + fabsf(ri_s);                       // |x - s mu| <= B + |s| (|mu| <= 1)
            int e = (bound > 0.f && bound < __builtin_inff()) ? PF_TOP - __builtin_amdgcn_frexp_expf(bound) : 0;
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
            rowinfo[buf * 64 + t] = make_float4(ri_s, ldexpf(1.0f, e), ldexpf(1.0f, -e), 0.f);
        }
    };
    float4 mu4;                                                                   // the lane's four column means (set after the barrier below)
    auto produce = [&](int buf) {
        unsigned char *const fb = frag + buf * PF_BUF;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 info = rowinfo[buf * 64 + 8 * j + r8];
            const float xv[4] = {P[j].x, P[j].y, P[j].z, P[j].w};
            const float mv[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __fmul_rn(centre(xv[e], mv[e], info.x, SCALED), info.y);   // (x - s mu) 2^e: the scaling is exact
            uint32_t h0, l0, h1, l1;
            split2h_pair(o[0], o[1], h0, l0);
            split2h_pair(o[2], o[3], h1, l1);
            unsigned char *const dst = fb + (j >> 2) * (PF_BUF / 2) + frag_lane_off + (j & 3) * 128;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(dst + 1024) = make_uint2(l0, l1);
        }
    };

    // ---- prologue: fragments of the first tile, operands of the second in flight ----------------------------------------------
    load_rowinfo(T);
    issue_tile(T);
    publish_rowinfo(0);
    __syncthreads();                                                              // mean_s, cs_s, rowinfo[0]
    mu4 = *reinterpret_cast<const float4 *>(mean_s + col0);
    const float4 cs4 = *reinterpret_cast<const float4 *>(cs_s + 4 * lane);       // row phase: the lane's four output columns
    produce(0);
    if (T + G < a.tiles) { load_rowinfo(T + G); issue_tile(T + G); }
    __syncthreads();

    f16v acc[2];
    int it = 0;
    for (; T < a.tiles; T += G, it ^= 1) {
        const int bufA = it, bufB = it ^ 1;
        const uint64_t Tn = T + G, Tnn = T + 2 * G;
        if (Tn < a.tiles) publish_rowinfo(bufB);                                  // (read by produce() behind barrier X)
        if (Tnn < a.tiles) load_rowinfo(Tnn);
        // ---- (a) the tile's products ---------------------------------------------------------------------------------------
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        const unsigned char *const fa = frag + bufA * PF_BUF + lane * 16;
#pragma unroll
        for (int ks = 0; ks < PF_KS; ++ks) {
            h8v ah[2], al[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                ah[rt] = *reinterpret_cast<const h8v *>(fa + rt * (PF_BUF / 2) + (ks * 2 + 0) * 1024);
                al[rt] = *reinterpret_cast<const h8v *>(fa + rt * (PF_BUF / 2) + (ks * 2 + 1) * 1024);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bhi[ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bhi[ks], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], blo[ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], blo[ks], acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[0], bhi[ks], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[1], bhi[ks], acc[1], 0, 0, 0);
        }
        __syncthreads();                                                          // X: nobody reads bufA's fragments any more; bufB is free (its rows went out)
        // ---- (b) the next tile's fragments, the tile after it into flight --------------------------------------------------
        if (Tn < a.tiles) {
            produce(bufB);
            if (Tnn < a.tiles) issue_tile(Tnn);
        }
        // ---- (c) accumulators -> rows (32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) ----
        float *const stage = reinterpret_cast<float *>(frag + bufA * PF_BUF);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                stage[(32 * rt + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * PF_D + 32 * w + (lane & 31)] = acc[rt][reg];
        const float unscale = rowinfo[bufA * 64 + 8 * w + (lane >> 3)].z;        // row phase: lane L finishes row 8 w + (L >> 3)
        __syncthreads();                                                          // Y
        // ---- (d) whole rows: column scales, norm, store ---------------------------------------------------------------------
        float4 v[8];
        float part[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = *reinterpret_cast<const float4 *>(stage + (8 * w + q) * PF_D + 4 * lane);
            v[q].x *= cs4.x; v[q].y *= cs4.y; v[q].z *= cs4.z; v[q].w *= cs4.w;
            part[q] = p.norm == 2 ? (fabsf(v[q].x) + fabsf(v[q].y)) + (fabsf(v[q].z) + fabsf(v[q].w))
                                  : (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
        }
        float g = unscale;                                                        // norm == 0: only the row scale is undone
        if (p.norm) {
            float *const rw = red + w * 8 * 64;
#pragma unroll
            for (int q = 0; q < 8; ++q) rw[q * 64 + lane] = part[q];
            // (one wave, in-order LDS queue: its own writes are visible to its reads) lane L sums eight partials of row L >> 3
            const float4 s0 = *reinterpret_cast<const float4 *>(rw + (lane >> 3) * 64 + (lane & 7) * 8);
            const float4 s1 = *reinterpret_cast<const float4 *>(rw + (lane >> 3) * 64 + (lane & 7) * 8 + 4);
            float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w));
            s += dpp_row<0xB1>(s);                                                // quad_perm [1, 0, 3, 2]
            s += dpp_row<0x4E>(s);                                                // quad_perm [2, 3, 0, 1]
            s += dpp_row<0x141>(s);                                               // row_half_mirror: the eight lanes of a row's group
            // L2: v (1 / max(sqrt(S), 1e-10)) like src/embedding.rs:98-102; L1: v / max(S, 1e-10) (pycleora/__init__.py:947-950); the
            // sums were taken on rows scaled by 2^e: S_true = S 2^-2e (L2) / S 2^-e (L1), every step mirrors exactly
            g = p.norm == 1 ? unscale * (1.0f / fmaxf(sqrtf(s) * unscale, 1e-10f)) : fmaxf(s * unscale, 1e-10f);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float f = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), 8 * q));
            const float u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, unscale), 8 * q));
            float4 o;
            if (p.norm == 2) { o.x = (v[q].x * u) / f; o.y = (v[q].y * u) / f; o.z = (v[q].z * u) / f; o.w = (v[q].w * u) / f; }
            else { o.x = v[q].x * f; o.y = v[q].y * f; o.z = v[q].z * f; o.w = v[q].w * f; }
            const uint64_t row = T * PF_ROWS + (uint64_t)(8 * w + q);
            if (row < p.n) *reinterpret_cast<float4 *>(p.out + row * p.ldo + 4 * lane) = o;
        }
    }
}

}  // namespace

bool project_f16_applies(const float *x, uint64_t ldx, uint64_t n, uint32_t d, uint32_t k, const float *out, uint64_t ldo, const float *x2) {
    auto aligned16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    return d == PF_D && k == PF_D && x2 == nullptr && n >= 1 && ldx % 4 == 0 && ldo % 4 == 0 && aligned16(x) && aligned16(out);
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
    F16Args a{};
    a.p.x = x;
    a.p.ldx = ldx;
    a.p.n = n;
    a.p.d = PF_D;
    a.p.mean = mean;
    a.p.k = PF_D;
    a.p.out = out;
    a.p.ldo = ldo;
    a.p.rowscale = rowscale;
    a.p.norm = norm;
    a.tp = reinterpret_cast<const u32x4 *>(tp);
    a.colscale = colscale;
    a.rowbound = rowbound;
    a.tiles = (n + PF_ROWS - 1) / PF_ROWS;
    const unsigned gx = (unsigned)(a.tiles < (uint64_t)cus ? a.tiles : (uint64_t)cus);
    hipError_t e;
    if (rowscale) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(project_f16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF_LDS);
        if (e == hipSuccess) hipLaunchKernelGGL(project_f16_kernel<true>, dim3(gx), dim3(PF_THREADS), PF_LDS, stream, a);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(project_f16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF_LDS);
        if (e == hipSuccess) hipLaunchKernelGGL(project_f16_kernel<false>, dim3(gx), dim3(PF_THREADS), PF_LDS, stream, a);
    }
    const hipError_t le = e != hipSuccess ? e : hipGetLastError();
    CL_HIP(hipFreeAsync(tp, stream));
    CL_HIP(le);
    return CLEORA_OK;
}

}  // namespace cleora
