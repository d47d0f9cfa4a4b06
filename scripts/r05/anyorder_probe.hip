// Does hipExtAnyOrderLaunch (AQL barrier bit cleared) let two kernels of ONE stream overlap on gfx950?  Two 4-block kernels that
// spin for ~5 ms each: in order they take ~10 ms, overlapped ~5.  hipcc --offload-arch=gfx950 anyorder_probe.hip -o /tmp/anyorder && /tmp/anyorder
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, int *out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}
int main() {
    int *d;
    hipMalloc(&d, 4096);
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned long long ticks = 500000;   // 100 MHz wall clock: 5 ms
    for (int mode = 0; mode < 3; ++mode) {
        hipDeviceSynchronize();
        hipEventRecord(e0, s);
        hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, s, ticks, d);
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, s, ticks, d + 64);
        else hipExtLaunchKernelGGL(spin, dim3(4), dim3(64), 0, s, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, ticks, d + 64);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000ull, d + 128);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.2f ms\n", mode, mode == 0 ? "plain launches" : mode == 1 ? "second kernel hipExtAnyOrderLaunch" : "hipExtLaunch, flags 0", ms);
    }
    return 0;
}
