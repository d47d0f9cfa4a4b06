// asm_load_probe.hip — is "global_load_dwordx4 vdst, voffset32, s[base]" issued from inline asm usable the way project_f16.hip uses it?
//   hipcc --offload-arch=gfx950 -O3 scripts/r06/asm_load_probe.hip -o /tmp/asm_load_probe && /tmp/asm_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v load16_untracked(const void *base_uniform, uint32_t byte_offset) {
    f4v v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_offset), "s"(base_uniform) : "memory");
    return v;
}
__device__ __forceinline__ float load4_untracked(const void *base_uniform, uint32_t byte_offset) {
    float v;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(byte_offset), "s"(base_uniform) : "memory");
    return v;
}
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *q) {
    const uint64_t v = reinterpret_cast<uint64_t>(q);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}
__global__ void k(const float *x, float *out, float *out1, int mode) {
    const int t = threadIdx.x;
    const float *base = uniform_ptr(x + (size_t)blockIdx.x * 1024);
    f4v v = load16_untracked(base, (uint32_t)t * 16u);
    float s = load4_untracked(mode ? (const void *)x : (const void *)base, (uint32_t)t * 4u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    reinterpret_cast<f4v *>(out)[(size_t)blockIdx.x * 256 + t] = v;
    out1[(size_t)blockIdx.x * 256 + t] = s;
}
int main() {
    const int B = 64;
    std::vector<float> h((size_t)B * 1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *x, *o, *o1;
    hipMalloc(&x, h.size() * 4); hipMalloc(&o, h.size() * 4); hipMalloc(&o1, (size_t)B * 256 * 4);
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(B), dim3(256), 0, 0, x, o, o1, mode);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> r(h.size()), r1((size_t)B * 256);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0, bad1 = 0;
        for (size_t i = 0; i < r.size(); ++i) bad += r[i] != h[i];
        for (size_t i = 0; i < r1.size(); ++i) bad1 += r1[i] != (mode ? (float)(i % 256) : (float)((i / 256) * 1024 + i % 256));
        printf("mode %d: %s, dwordx4 mismatches %zu, dword mismatches %zu\n", mode, hipGetErrorString(e), bad, bad1);
    }
    return 0;
}
