// project_fat.hip — the split-bf16 projection for large n (see whiten.hip "projection, third form" for the arithmetic).
// A translation unit of its own because it wants the OPPOSITE register policy from whiten.hip: that file is built with
// -amdgpu-mfma-vgpr-form (its kernels keep <= 144 accumulator registers and the AGPR shuttle cost them 30 %); this kernel
// runs one wave per SIMD with a 64 x 128 accumulator (128 registers) beside ~200 VGPRs of operands, which only fits when the
// accumulators live in the AGPR half of the unified register file.
#include <type_traits>

#include "project_common.h"

namespace cleora {
namespace {

// ---- the same arithmetic for large n: one wave per SIMD, 128 rows per block --------------------------------------------------
// Profiling the 64-row form at the C3 shape (profiles/r03b_*): 8.75 ms, of which 2.6 ms are the B stage's global loads — every
// 64-row tile streams all 384 KiB of the split transform through L2 (60 GB per call, 6.9 TB/s of L2 traffic) —, 1 ms stores,
// 0.5 ms A loads, and the rest runs the matrix pipe at 51 % with ~90 VALU instructions and a barrier per 24 MFMAs.  This form
// halves all of that per MFMA: a block of 4 waves, ONE per SIMD (accumulators in the AGPR half of the register file, ~200
// VGPRs of operands beside them), owns 128 rows; wave (wr, wc) keeps the 64 x 128 accumulator of rows 64 wr .. +63, columns
// 128 wc .. +127 (2 row groups x 4 column tiles = 128 AGPRs), so one 24 KiB B stage feeds 48 MFMAs per wave, a B fragment read
// from LDS feeds two of them, and the operand split costs ~2 VALU per MFMA.  (A 64 x 256 accumulator per wave — all 256 AGPRs —
// was tried first: hipcc spills ~290 registers per lane around it.)
//   * LDS: FOUR B stage buffers; step g reads buffer g & 3 and writes B(g + 2) into (g + 2) & 3, so the fragments of a
//     step were published TWO barriers earlier and every fragment read can be issued half a step ahead of its MFMAs —
//     also across the end-of-step barrier (two 6-fragment register sets alternate by column-tile pair): a wave alone on
//     its SIMD never waits for LDS latency behind the barrier.
constexpr int FR = kFatRows;           // rows per block tile
constexpr int FBUF = 4;                // B stage buffers

template <bool SCALED, bool BLEND, int U>
__global__ __launch_bounds__(256, 1) void project_split_fat_kernel(const ProjArgs a, const u32x4 *__restrict__ tp,
                                                                   uint32_t ksteps, uint64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *const bs = reinterpret_cast<u32x4 *>(smem);                           // [FBUF][SKB]
    float *const mean_s = reinterpret_cast<float *>(smem + FBUF * SKB * 16);     // [16 ksteps]
    float *const red = mean_s + 16 * ksteps;                                     // [4 waves][64 rows]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, i = lane & 31, h = lane >> 5;
    const int wr = w >> 1, wc = w & 1;
    const uint32_t pass = blockIdx.y;
    const u32x4 *const tpp = tp + (uint64_t)pass * ksteps * SKB;
    const uint64_t my_tiles = tiles > blockIdx.x ? (tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const uint64_t total = my_tiles * ksteps;                                    // a multiple of U
    if (total == 0) return;

    for (uint32_t c = t; c < 16 * ksteps; c += 256) mean_s[c] = c < a.d ? a.mean[c] : 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) {                                                // B of steps 0 and 1
        bs[t + 256 * u] = tpp[t + 256 * u];
        bs[SKB + t + 256 * u] = tpp[(uint64_t)(1 % ksteps) * SKB + t + 256 * u];
    }

    // operand ring: two k-steps of A for both row groups of this wave
    float4 ra[2][2][2], rb[BLEND ? 2 : 1][2][2];
    float rs[SCALED ? 2 : 1][2];
    uint64_t ltile = blockIdx.x;
    uint32_t lks = 0;
    auto issue_a = [&](int slot) {
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            uint64_t r = ltile * FR + (uint64_t)(wr * 64 + grp * 32 + i);
            r = r < a.n ? r : a.n - 1;                                           // clamped: always a valid address
            const float *p = a.x + r * a.ldx + 16 * lks + 4 * h;
            ra[slot][grp][0] = *reinterpret_cast<const float4 *>(p);
            ra[slot][grp][1] = *reinterpret_cast<const float4 *>(p + 8);
            if constexpr (BLEND) {
                const float *p2 = a.x2 + r * a.ldx2 + 16 * lks + 4 * h;
                rb[slot][grp][0] = *reinterpret_cast<const float4 *>(p2);
                rb[slot][grp][1] = *reinterpret_cast<const float4 *>(p2 + 8);
            }
            if constexpr (SCALED) rs[slot][grp] = a.rowscale[r];
        }
        if (++lks == ksteps) { lks = 0; ltile += gridDim.x; }
    };
    // centre (block = embeddings - mean_f32, pycleora/__init__.py:161), blend, split one row group's operand
    auto split_a = [&](int slot, int grp, uint32_t ks, u32x4 (&as)[3]) {
        const float4 m0 = *reinterpret_cast<const float4 *>(mean_s + 16 * ks + 4 * h);
        const float4 m1 = *reinterpret_cast<const float4 *>(mean_s + 16 * ks + 8 + 4 * h);
        const float4 x0 = ra[slot][grp][0], x1 = ra[slot][grp][1];
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = centre(xv[e], mu[e], SCALED ? rs[SCALED ? slot : 0][grp] : 1.f, SCALED);
        if constexpr (BLEND) {
            const float4 y0 = rb[BLEND ? slot : 0][grp][0], y1 = rb[BLEND ? slot : 0][grp][1];
            const float x2v[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __fadd_rn(__fmul_rn(a.alpha, o[e]), __fmul_rn(a.beta, __fsub_rn(x2v[e], mu[e])));
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint32_t p1, p2, p3;
            split3_pair(o[2 * m], o[2 * m + 1], p1, p2, p3);
            as[0][m] = p1; as[1][m] = p2; as[2][m] = p3;
        }
    };
    issue_a(0);
    issue_a(1);

    f16v acc[2][4];
#pragma unroll
    for (int grp = 0; grp < 2; ++grp)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[grp][jt][r] = 0.f;

    __syncthreads();
    u32x4 as[2][2][3];                     // [step parity][row group][split]
    split_a(0, 0, 0, as[0][0]);
    split_a(0, 1, 0, as[0][1]);
    u32x4 fs[2][3][2];                     // two fragment sets: [set][split][tile of the pair]
    auto read_frags = [&](int set, int buf, int pair) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) fs[set][sp][jj] = bs[buf * SKB + (sp * 8 + wc * 4 + pair * 2 + jj) * 64 + lane];
    };
    read_frags(0, 0, 0);                   // column-tile pair 0 of step 0

    uint64_t tile = blockIdx.x;
    uint32_t ks0 = 0;
    for (uint64_t g0 = 0; g0 < total; g0 += U) {
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            // nothing moves across a step boundary: left alone, the scheduler hoists the address arithmetic (and loads) of all U
            // unrolled steps to the top of the body and spills
            __builtin_amdgcn_sched_barrier(0);
            const int par = uu & 1, buf = uu & (FBUF - 1);                       // U is a multiple of FBUF
            const uint32_t ks = ks0 + uu;
            const uint32_t ksn = ks + 1 == ksteps ? 0 : ks + 1;
            const uint32_t ks2 = ksn + 1 == ksteps ? 0 : ksn + 1;
            u32x4 bst[6];                                                        // B of the step after the next
#pragma unroll
            for (int u = 0; u < 6; ++u) bst[u] = tpp[(uint64_t)ks2 * SKB + t + 256 * u];
            issue_a(par);                                                        // A of step g + 2 into the slot split during step g - 1
            constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
            for (int pair = 0; pair < 2; ++pair) {
                // the fragments of the NEXT column-tile pair (of the next step's first pair: published two barriers ago)
                if (pair == 1) {
#pragma unroll
                    for (int u = 0; u < 6; ++u) bs[((buf + 2) & (FBUF - 1)) * SKB + t + 256 * u] = bst[u];
                    read_frags(0, (buf + 1) & (FBUF - 1), 0);
                } else {
                    read_frags(1, buf, 1);
                }
                __builtin_amdgcn_sched_barrier(0);                               // the reads stay ahead of this pair's MFMAs
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                            acc[grp][pair * 2 + jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(bf16x8, as[par][grp][PA[q]]), __builtin_bit_cast(bf16x8, fs[pair][PB[q]][jj]),
                                acc[grp][pair * 2 + jj], 0, 0, 0);
                // the next step's operand fragments are made under these MFMAs: one row group per pair
                split_a(par ^ 1, pair, ksn, as[par ^ 1][pair]);
                __builtin_amdgcn_sched_barrier(0);
            }

            if (uu == U - 1 && ks0 + U == ksteps) {
                // ---- end of a row tile: normalise (a row's two column halves meet in LDS), store -------------------------------
                if (a.norm) {
#pragma unroll
                    for (int grp = 0; grp < 2; ++grp) {
                        float pr[16];
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            float v = 0.f;
#pragma unroll
                            for (int jt = 0; jt < 4; ++jt) v += a.norm == 1 ? acc[grp][jt][reg] * acc[grp][jt][reg] : fabsf(acc[grp][jt][reg]);
                            pr[reg] = v;
                        }
#pragma unroll
                        for (int ofs = 16; ofs > 0; ofs >>= 1)
#pragma unroll
                            for (int reg = 0; reg < 16; ++reg) pr[reg] += __shfl_xor(pr[reg], ofs, 64);
                        float mine = 0.f;                                        // lane i < 16 of half h publishes row slot i
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg)
                            if (i == reg) mine = pr[reg];
                        if (i < 16) red[w * 64 + grp * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] = mine;
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int grp = 0; grp < 2; ++grp)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int rl = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                        float f = 1.f;
                        if (a.norm) {
                            const float sum = red[(wr * 2) * 64 + grp * 32 + rl] + red[(wr * 2 + 1) * 64 + grp * 32 + rl];   // column halves in order
                            // L2: v * (1 / max(sqrt(s), 1e-10)) like src/embedding.rs:98-102; L1: v / max(s, 1e-10) (pycleora/__init__.py:947-950)
                            f = a.norm == 1 ? 1.0f / fmaxf(sqrtf(sum), 1e-10f) : fmaxf(sum, 1e-10f);
                        }
                        // 32x32 C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
                        const uint64_t row = tile * FR + (uint64_t)(wr * 64 + grp * 32 + rl);
#pragma unroll
                        for (int jt = 0; jt < 4; ++jt) {
                            const uint32_t col = pass * SN + wc * 128 + jt * 32 + i;
                            const float v = acc[grp][jt][reg];
                            if (row < a.n && col < a.k) a.out[row * a.ldo + col] = a.norm == 2 ? v / f : v * f;
                            acc[grp][jt][reg] = 0.f;
                        }
                    }
                tile += gridDim.x;
            }
            __syncthreads();
        }
        ks0 = ks0 + U == ksteps ? 0 : ks0 + U;
    }
}

}  // namespace

hipError_t launch_project_split_fat(const ProjArgs &a, const u32x4 *tp, uint32_t ksteps, uint32_t passes, int cus, hipStream_t stream) {
    const uint64_t tiles = (a.n + FR - 1) / FR;
    const dim3 fgrid((unsigned)(tiles < (uint64_t)cus ? tiles : (uint64_t)cus), passes);
    const size_t flds = (size_t)FBUF * SKB * 16 + (size_t)16 * ksteps * sizeof(float) + 4 * 64 * sizeof(float);
    // U = 4 k-steps per trip (= the B stage ring): wider unrolls make hipcc spill; the residual blend (a second operand stream)
    // does not fit beside the 128 accumulators without scratch and stays with the 64-row form (launch_project)
    auto launch_fat = [&](auto SCt) -> hipError_t {
        constexpr bool SC = decltype(SCt)::value;
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(project_split_fat_kernel<SC, false, 4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((project_split_fat_kernel<SC, false, 4>), fgrid, dim3(256), flds, stream, a, tp, ksteps, tiles);
        return hipSuccess;
    };
    if (a.x2 != nullptr || ksteps % 4 != 0) return hipErrorInvalidValue;     // the caller checks both
    return a.rowscale != nullptr ? launch_fat(std::true_type{}) : launch_fat(std::false_type{});
}

}  // namespace cleora
