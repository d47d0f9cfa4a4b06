// side_load.hip — synthetic kernels to run BESIDE the SpMM on another stream (scripts/r03_probe.py side_load): what does a
// co-resident kernel cost the SpMM when it (a) only occupies wave slots, (b) keeps the matrix cores busy, (c) streams memory?
// Built by scripts/probes/build.sh into scripts/probes/libside_load.so (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f16v;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf8v;

__global__ void spin_kernel(uint64_t cycles, uint32_t *sink) {
    const uint64_t t0 = __builtin_readcyclecounter();
    uint32_t x = threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < cycles) {
        x = x * 1664525u + 1013904223u;
        __builtin_amdgcn_s_sleep(8);
    }
    if (x == 0xdeadbeefu) sink[0] = x;
}

// bf16 matrix cores at full tilt from registers only (no memory traffic) for `iters` rounds of 16 MFMAs
__global__ void mfma_kernel(uint32_t iters, float *sink) {
    f16v acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    bf8v a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)(float)(e + 1); }
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) s += acc[q][0];
    if (s == 12345.678f) sink[0] = s;
}

// f32 matrix cores (v_mfma_f32_32x32x2_f32), registers only
__global__ void mfma32_kernel(uint32_t iters, float *sink) {
    f16v acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const float a = (float)(threadIdx.x & 7), b = 1.5f;
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) s += acc[q][0];
    if (s == 12345.678f) sink[0] = s;
}

// vector ALU only (fixed work, no sleep, no memory): 8 independent fma chains per lane
__global__ void valu_kernel(uint32_t iters, float *sink) {
    float v[8];
    for (int q = 0; q < 8; ++q) v[q] = (float)(threadIdx.x + q);
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = __builtin_fmaf(v[q], 1.0000001f, 0.5f);
    }
    float s = 0.f;
    for (int q = 0; q < 8; ++q) s += v[q];
    if (s == 12345.678f) sink[0] = s;
}

// streaming read of `bytes` (16 B per lane per step), `passes` times
__global__ void read_kernel(const float4 *x, uint64_t n16, uint32_t passes, float *sink) {
    float s = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint32_t p = 0; p < passes; ++p)
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            const float4 v = x[i];
            s += v.x + v.y + v.z + v.w;
        }
    if (s == 12345.678f) sink[0] = s;
}

extern "C" {
int side_spin(uint64_t cycles, uint32_t blocks, uint32_t threads, void *sink, void *stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, cycles, (uint32_t *)sink);
    return (int)hipGetLastError();
}
int side_mfma(uint32_t iters, uint32_t blocks, uint32_t threads, void *sink, void *stream) {
    hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, iters, (float *)sink);
    return (int)hipGetLastError();
}
int side_valu(uint32_t iters, uint32_t blocks, uint32_t threads, void *sink, void *stream) {
    hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, iters, (float *)sink);
    return (int)hipGetLastError();
}
int side_mfma32(uint32_t iters, uint32_t blocks, uint32_t threads, void *sink, void *stream) {
    hipLaunchKernelGGL(mfma32_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, iters, (float *)sink);
    return (int)hipGetLastError();
}
int side_read(const void *x, uint64_t bytes, uint32_t passes, uint32_t blocks, void *sink, void *stream) {
    hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4 *)x, bytes / 16, passes, (float *)sink);
    return (int)hipGetLastError();
}
}
