// project_common.h — the projection's argument block and the three-way bf16 split (whiten.hip).
#pragma once
#include "common.h"

namespace cleora {

typedef float f16v __attribute__((ext_vector_type(16)));

struct ProjArgs {
    const float *x;
    uint64_t ldx;
    uint64_t n;
    uint32_t d;
    const float *mean;
    const float *t;   // d x k row-major
    uint32_t k;
    float *out;
    uint64_t ldo;
    uint32_t nb_n;    // column blocks
    uint64_t n_blocks;
    int w4x, w4t;
    // generalised centring (the propagate-before-project form of the embed loop, abi.hip): the operand row is
    //   alpha * (x[r] - rowscale[r] * mean) + beta * (x2[r] - mean)
    // rowscale == nullptr: scale 1; x2 == nullptr: no second term (alpha is then 1): the plain (x - mean).
    const float *rowscale;
    const float *x2;
    uint64_t ldx2;
    float alpha, beta;
    int norm;         // split form, one column pass (k <= 256): 1 = L2-normalise, 2 = L1-normalise every output row in the epilogue
    // bounded-operand (f16, three-product) mode of the split form: per row {s_r, 2^e_r} with (B_r + |s_r|) 2^e_r < 2^14, per column 2^-st
    const float2 *rowinfo;
    const float *colscale;
};

static __device__ __forceinline__ float centre(float v, float mu, float s, bool scaled) {
    return scaled ? __fsub_rn(v, __fmul_rn(s, mu)) : __fsub_rn(v, mu);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
constexpr int PF_TOP = 14;             // bounded operands are scaled below 2^14 (f16 overflows at 65504)

// (lo, hi) -> packed f16 pairs p1 = f16(v), p2 = f16(v - p1): v = p1 + p2 to 2^-22 |v| (plus f16's subnormal spacing, 2^-24 absolute)
static __device__ __forceinline__ void split2h_pair(float lo, float hi, uint32_t &p1, uint32_t &p2) {
    const float __attribute__((ext_vector_type(2))) v = {lo, hi};
    const h2v a = __builtin_convertvector(v, h2v);                 // round to nearest even
    p1 = __builtin_bit_cast(uint32_t, a);
    const float __attribute__((ext_vector_type(2))) r = v - __builtin_convertvector(a, float __attribute__((ext_vector_type(2))));   // exact
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, h2v));
}
// the power of two that scales a value bounded by `bound` just below 2^PF_TOP (1 for a zero / non-finite bound)
static __device__ __forceinline__ float pf_row_scale(float bound) {
    int e = (bound > 0.f && bound < __builtin_inff()) ? PF_TOP - __builtin_amdgcn_frexp_expf(bound) : 0;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    return ldexpf(1.0f, e);
}

constexpr int SR = 64;                 // rows per block tile
constexpr int SN = 256;                // output columns per pass
constexpr int SKB = 3 * 8 * 64;        // 16-byte units of packed T per (pass, k-step): 24 KiB

// (lo, hi) -> three packed bf16 pairs whose sum is (lo, hi) to 2^-27: v_cvt_pk_bf16_f32 (round to nearest even), the bf16
// back as f32 by shift / mask, an exact f32 subtraction — twice.
static __device__ __forceinline__ void split3_pair(float lo, float hi, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    const f2v v = {lo, hi};
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f2v f1 = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    const f2v r1 = v - f1;
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    const f2v f2 = {__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xffff0000u)};
    const f2v r2 = r1 - f2;
    p3 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
}

// (lo, hi) -> two packed bf16 pairs (y1, y2) and the f32 residuals r1 = y - y1 (exact): y = y1 + y2 + r2 with |r2| <= 2^-18 |y|.
static __device__ __forceinline__ void split2_pair(float lo, float hi, uint32_t &p1, uint32_t &p2, float &r_lo, float &r_hi) {
    const f2v v = {lo, hi};
    p1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f2v f1 = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
    const f2v r1 = v - f1;
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
    r_lo = r1[0];
    r_hi = r1[1];
}

}  // namespace cleora
