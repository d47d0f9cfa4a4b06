// exp_spmm.hip — EXPERIMENT harness (not product code): variants of the d = 256 SpMM+L2 kernel
// timed against each other in one process.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
// -shared -fPIC scripts/exp_spmm.hip -o gpurun_out/libexp.so ; driven by scripts/exp_spmm.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v4f __attribute__((ext_vector_type(4)));

struct Args {
    const uint64_t *rowptr;
    const uint32_t *col;
    const float *val;
    const float *x;
    float *y;
    uint64_t n;
    uint32_t hub_threshold;
    const uint32_t *order;  // optional row order
};

__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ float rlf(float v, uint32_t l) { return __uint_as_float(rl(__float_as_uint(v), l)); }

template <bool NT>
__device__ __forceinline__ v4f ldx(const float *p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    else return *reinterpret_cast<const v4f *>(p);
}

__device__ __forceinline__ void fma_sep(v4f &acc, float w, v4f r) {
    acc.x = __fadd_rn(acc.x, __fmul_rn(w, r.x));
    acc.y = __fadd_rn(acc.y, __fmul_rn(w, r.y));
    acc.z = __fadd_rn(acc.z, __fmul_rn(w, r.z));
    acc.w = __fadd_rn(acc.w, __fmul_rn(w, r.w));
}

__device__ __forceinline__ void finish(v4f acc, float *yrow, int lane, bool nt_store) {
    float s = 0.f;
    float sq[4] = {__fmul_rn(acc.x, acc.x), __fmul_rn(acc.y, acc.y), __fmul_rn(acc.z, acc.z), __fmul_rn(acc.w, acc.w)};
    for (uint32_t g = 0; g < 64; ++g) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s = __fadd_rn(s, rlf(sq[q], g));
    }
    const float inv = 1.0f / fmaxf(sqrtf(s), 1e-10f);
    v4f o = {__fmul_rn(acc.x, inv), __fmul_rn(acc.y, inv), __fmul_rn(acc.z, inv), __fmul_rn(acc.w, inv)};
    if (nt_store) __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(yrow + lane * 4));
    else *reinterpret_cast<v4f *>(yrow + lane * 4) = o;
}

// U loads in flight; NTS: nt col/val loads + nt Y store; NTX: nt X gathers; BLK: threads per block
template <int U, bool NTS, bool NTX, int BLK>
__global__ __launch_bounds__(BLK) void k_basic(const Args a) {
    const int lane = threadIdx.x & 63;
    uint64_t item = (uint64_t)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
    if (item >= a.n) return;
    item = uni64(item);
    const uint64_t row = a.order ? (uint64_t)a.order[item] : item;
    uint64_t beg = uni64(a.rowptr[row]), end = uni64(a.rowptr[row + 1]);
    if (end - beg > a.hub_threshold) return;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t e = beg; e < end; e += 64) {
        const uint32_t cnt = (end - e) < 64 ? (uint32_t)(end - e) : 64u;
        uint32_t cv = 0;
        float wv = 0.f;
        if ((uint32_t)lane < cnt) {
            if constexpr (NTS) { cv = __builtin_nontemporal_load(a.col + e + lane); wv = __builtin_nontemporal_load(a.val + e + lane); }
            else { cv = a.col[e + lane]; wv = a.val[e + lane]; }
        }
        uint32_t k = 0;
        for (; k + U <= cnt; k += U) {
            v4f r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = ldx<NTX>(a.x + (uint64_t)rl(cv, k + u) * 256 + lane * 4);
#pragma unroll
            for (int u = 0; u < U; ++u) fma_sep(acc, rlf(wv, k + u), r[u]);
        }
        for (; k < cnt; ++k) fma_sep(acc, rlf(wv, k), ldx<NTX>(a.x + (uint64_t)rl(cv, k) * 256 + lane * 4));
    }
    finish(acc, a.y + row * 256, lane, NTS);
}

// Rolling pipeline: keep U loads in flight continuously within a 64-edge chunk
template <int U>
__global__ __launch_bounds__(256) void k_rolling(const Args a) {
    const int lane = threadIdx.x & 63;
    uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n) return;
    row = uni64(row);
    uint64_t beg = uni64(a.rowptr[row]), end = uni64(a.rowptr[row + 1]);
    if (end - beg > a.hub_threshold) return;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t e = beg; e < end; e += 64) {
        const uint32_t cnt = (end - e) < 64 ? (uint32_t)(end - e) : 64u;
        uint32_t cv = 0;
        float wv = 0.f;
        if ((uint32_t)lane < cnt) { cv = a.col[e + lane]; wv = a.val[e + lane]; }
        v4f r[U];
        // prologue
#pragma unroll
        for (int u = 0; u < U; ++u)
            if ((uint32_t)u < cnt) r[u] = ldx<false>(a.x + (uint64_t)rl(cv, u) * 256 + lane * 4);
        for (uint32_t k = 0; k < cnt; k += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k + u < cnt) {
                    fma_sep(acc, rlf(wv, k + u), r[u]);
                    if (k + u + U < cnt) r[u] = ldx<false>(a.x + (uint64_t)rl(cv, k + u + U) * 256 + lane * 4);
                }
            }
        }
    }
    finish(acc, a.y + row * 256, lane, false);
}

// LDS-staged: a 256-thread block owns R consecutive rows; their contiguous (col,val) slice is staged
// into LDS with coalesced loads, waves then pull rows from an LDS work counter.
template <int R, int CAP>
__global__ __launch_bounds__(256) void k_lds(const Args a) {
    __shared__ uint32_t s_col[CAP];
    __shared__ float s_val[CAP];
    __shared__ uint64_t s_rp[R + 1];
    __shared__ int s_next;
    const int lane = threadIdx.x & 63;
    const uint64_t r0 = (uint64_t)blockIdx.x * R;
    if (r0 >= a.n) return;
    const int nr = (a.n - r0) < R ? (int)(a.n - r0) : R;
    if (threadIdx.x <= nr) s_rp[threadIdx.x] = a.rowptr[r0 + threadIdx.x];
    if (threadIdx.x == 0) s_next = 0;
    __syncthreads();
    const uint64_t e0 = s_rp[0], e1 = s_rp[nr];
    const bool fits = (e1 - e0) <= CAP;
    if (fits) {
        for (uint64_t i = threadIdx.x; i < e1 - e0; i += 256) { s_col[i] = a.col[e0 + i]; s_val[i] = a.val[e0 + i]; }
    }
    __syncthreads();
    for (;;) {
        int j = 0;
        if (lane == 0) j = atomicAdd(&s_next, 1);
        j = __builtin_amdgcn_readfirstlane(j);
        if (j >= nr) break;
        const uint64_t beg = s_rp[j], end = s_rp[j + 1];
        if (end - beg > a.hub_threshold) continue;
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        constexpr int U = 8;
        uint64_t k = beg;
        if (fits) {
            for (; k + U <= end; k += U) {
                v4f r[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t c = __builtin_amdgcn_readfirstlane((int)s_col[k - e0 + u]);
                    r[u] = ldx<false>(a.x + (uint64_t)c * 256 + lane * 4);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) fma_sep(acc, s_val[k - e0 + u], r[u]);
            }
            for (; k < end; ++k) {
                const uint32_t c = __builtin_amdgcn_readfirstlane((int)s_col[k - e0]);
                fma_sep(acc, s_val[k - e0], ldx<false>(a.x + (uint64_t)c * 256 + lane * 4));
            }
        } else {
            for (; k < end; ++k) fma_sep(acc, a.val[k], ldx<false>(a.x + (uint64_t)a.col[k] * 256 + lane * 4));
        }
        finish(acc, a.y + (r0 + j) * 256, lane, false);
    }
}

// One 1 KiB row through a buffer descriptor built from the (wave-uniform) row pointer; the cache
// policy is an immediate of the instruction, so the two flavours cannot be merged by the compiler.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int AUX_A = 2, int AUX_B = 0>
__device__ __forceinline__ v4f ld_policy(const float *rowp, int lane, bool first) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)rowp, 0, 1024, 0x00020000);
    v4u t;
    if (first) t = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, AUX_A);   // aux: 1 = sc0, 2 = nt, 16 = sc1
    else t = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, 0, AUX_B);
    v4f r;
    r.x = __uint_as_float(t.x); r.y = __uint_as_float(t.y); r.z = __uint_as_float(t.z); r.w = __uint_as_float(t.w);
    return r;
}

// Same idea with plain global loads: the nt flavour reads through a pointer laundered by an empty asm so
// the compiler cannot merge it with the default-policy load of the other branch (it would drop `nt`).
__device__ __forceinline__ v4f ld_global_policy(const float *p, bool nt) {
    if (nt) {
        const float *q = p;
        asm volatile("" : "+v"(q));
        return __builtin_nontemporal_load(reinterpret_cast<const v4f *>(q));
    }
    return *reinterpret_cast<const v4f *>(p);
}

template <int U>
__global__ __launch_bounds__(256) void k_hot_global(const Args a) {
    const int lane = threadIdx.x & 63;
    uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n) return;
    row = uni64(row);
    uint64_t beg = uni64(a.rowptr[row]), end = uni64(a.rowptr[row + 1]);
    if (end - beg > a.hub_threshold) return;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t e = beg; e < end; e += 64) {
        const uint32_t cnt = (end - e) < 64 ? (uint32_t)(end - e) : 64u;
        uint32_t cv = 0;
        float wv = 0.f;
        if ((uint32_t)lane < cnt) { cv = a.col[e + lane]; wv = a.val[e + lane]; }
        uint32_t k = 0;
        for (; k + U <= cnt; k += U) {
            v4f r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t c = rl(cv, k + u);
                r[u] = ld_global_policy(a.x + (uint64_t)(c & 0x7fffffffu) * 256 + lane * 4, (c >> 31) != 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) fma_sep(acc, rlf(wv, k + u), r[u]);
        }
        for (; k < cnt; ++k) {
            const uint32_t c = rl(cv, k);
            fma_sep(acc, rlf(wv, k), ld_global_policy(a.x + (uint64_t)(c & 0x7fffffffu) * 256 + lane * 4, (c >> 31) != 0));
        }
    }
    finish(acc, a.y + row * 256, lane, false);
}

// Hot/cold split: bit 31 of the column index marks a "hot" (high in-degree) row; hot rows are gathered
// with the default cache policy, cold rows non-temporally, to keep the hot set resident in L2/MALL.
// EXTRA bit 0: Y rows stored non-temporally; bit 1: the CSR col / val streams loaded non-temporally.
template <int U, bool COLD_NT, int AUX_A = 2, int AUX_B = 0, int EXTRA = 0>
__global__ __launch_bounds__(256) void k_hot(const Args a) {
    const int lane = threadIdx.x & 63;
    uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n) return;
    row = uni64(row);
    uint64_t beg = uni64(a.rowptr[row]), end = uni64(a.rowptr[row + 1]);
    if (end - beg > a.hub_threshold) return;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t e = beg; e < end; e += 64) {
        const uint32_t cnt = (end - e) < 64 ? (uint32_t)(end - e) : 64u;
        uint32_t cv = 0;
        float wv = 0.f;
        if ((uint32_t)lane < cnt) {
            if constexpr (EXTRA & 2) { cv = __builtin_nontemporal_load(a.col + e + lane); wv = __builtin_nontemporal_load(a.val + e + lane); }
            else { cv = a.col[e + lane]; wv = a.val[e + lane]; }
        }
        uint32_t k = 0;
        for (; k + U <= cnt; k += U) {
            v4f r[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t c = rl(cv, k + u);
                r[u] = ld_policy<AUX_A, AUX_B>(a.x + (uint64_t)(c & 0x7fffffffu) * 256, lane, (c >> 31) != (COLD_NT ? 1u : 0u));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) fma_sep(acc, rlf(wv, k + u), r[u]);
        }
        for (; k < cnt; ++k) {
            const uint32_t c = rl(cv, k);
            v4f r = ld_policy<AUX_A, AUX_B>(a.x + (uint64_t)(c & 0x7fffffffu) * 256, lane, (c >> 31) != (COLD_NT ? 1u : 0u));
            fma_sep(acc, rlf(wv, k), r);
        }
    }
    finish(acc, a.y + row * 256, lane, (EXTRA & 1) != 0);
}

extern "C" int exp_launch(int variant, const uint64_t *rowptr, const uint32_t *col, const float *val,
                          const float *x, float *y, uint64_t n, uint32_t hub_threshold,
                          const uint32_t *order, void *stream) {
    Args a{rowptr, col, val, x, y, n, hub_threshold, order};
    hipStream_t s = (hipStream_t)stream;
    const unsigned g4 = (unsigned)((n + 3) / 4);
    switch (variant) {
        case 0: hipLaunchKernelGGL((k_basic<8, false, false, 256>), dim3(g4), dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL((k_basic<8, true, false, 256>), dim3(g4), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((k_basic<8, true, true, 256>), dim3(g4), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((k_basic<8, false, false, 64>), dim3((unsigned)n), dim3(64), 0, s, a); break;
        case 4: hipLaunchKernelGGL((k_basic<16, false, false, 256>), dim3(g4), dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL((k_basic<4, false, false, 256>), dim3(g4), dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL((k_rolling<8>), dim3(g4), dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL((k_lds<32, 2048>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, a); break;
        case 8: hipLaunchKernelGGL((k_lds<64, 4096>), dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, a); break;
        case 9: hipLaunchKernelGGL((k_basic<8, false, false, 128>), dim3((unsigned)((n + 1) / 2)), dim3(128), 0, s, a); break;
        case 10: hipLaunchKernelGGL((k_hot<8, false>), dim3(g4), dim3(256), 0, s, a); break;  // HOT rows nt (control)
        case 11: hipLaunchKernelGGL((k_hot<8, true>), dim3(g4), dim3(256), 0, s, a); break;   // COLD rows nt, hot default
        // k_hot<U, true, A, B>: COLD rows (bit 31 clear) get AUX_A, hot rows AUX_B
        case 12: hipLaunchKernelGGL((k_hot<8, true, 2, 1>), dim3(g4), dim3(256), 0, s, a); break;    // cold nt, hot sc0
        case 13: hipLaunchKernelGGL((k_hot<8, true, 2, 16>), dim3(g4), dim3(256), 0, s, a); break;   // cold nt, hot sc1
        case 14: hipLaunchKernelGGL((k_hot<8, true, 18, 0>), dim3(g4), dim3(256), 0, s, a); break;   // cold nt+sc1
        case 15: hipLaunchKernelGGL((k_hot<8, true, 16, 0>), dim3(g4), dim3(256), 0, s, a); break;   // cold sc1
        case 16: hipLaunchKernelGGL((k_hot<8, true, 1, 0>), dim3(g4), dim3(256), 0, s, a); break;    // cold sc0
        case 17: hipLaunchKernelGGL((k_hot<8, true, 3, 0>), dim3(g4), dim3(256), 0, s, a); break;    // cold nt+sc0
        case 18: hipLaunchKernelGGL((k_hot<8, true, 0, 0>), dim3(g4), dim3(256), 0, s, a); break;    // buffer loads, all default
        case 19: hipLaunchKernelGGL((k_hot_global<8>), dim3(g4), dim3(256), 0, s, a); break;       // global loads, hot nt (control)
        case 20: hipLaunchKernelGGL((k_hot<8, true, 2, 0, 1>), dim3(g4), dim3(256), 0, s, a); break;  // cold nt + nt stores
        case 21: hipLaunchKernelGGL((k_hot<8, true, 2, 0, 2>), dim3(g4), dim3(256), 0, s, a); break;  // cold nt + nt CSR
        case 22: hipLaunchKernelGGL((k_hot<8, true, 2, 0, 3>), dim3(g4), dim3(256), 0, s, a); break;  // cold nt + both
        case 23: hipLaunchKernelGGL((k_hot<16, true, 2, 0, 0>), dim3(g4), dim3(256), 0, s, a); break; // cold nt, 16 in flight
        case 24: hipLaunchKernelGGL((k_hot<4, true, 2, 0, 0>), dim3(g4), dim3(256), 0, s, a); break;  // cold nt, 4 in flight
        default: return -1;
    }
    return (int)hipGetLastError();
}
