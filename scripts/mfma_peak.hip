// mfma_peak.hip — measured ceilings of the two MFMA shapes the whitening kernels use, on this chip:
//   v_mfma_f64_16x16x4_f64  (gram_kernel)      v_mfma_f32_32x32x2_f32  (project_kernel)
// One wave per SIMD (256-thread blocks, 1 block/CU) and 2 waves per SIMD (2 blocks/CU); N independent
// accumulators per wave so the dependent-issue latency is covered.  Prints TFLOP/s and the effective clock
// implied by the known cycles per instruction.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f64(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float *out, int iters, float a0, float b0) {
    f16v acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// f64 VALU FMA ceiling (v_fma_f64): NACC independent chains per lane
template <int NACC>
__global__ __launch_bounds__(256) void k_valu64(double *out, int iters, double a0, double b0) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const double a = a0 + threadIdx.x * 1e-9, b = b0 * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// effective shader clock under a given load: s_memtime ticks per 100 MHz wall-clock tick, sampled by one wave
template <int NACC>
__global__ __launch_bounds__(256) void k_f64_clock(double *out, int iters, double a0, double b0, unsigned long long *ticks) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }
}

// Do the f64 matrix instruction and the f64 vector FMA share one datapath?  512-thread blocks, one per CU: waves 0-3
// run the MFMA loop, waves 4-7 the v_fma_f64 loop (both halves land on all four SIMDs), each for a fixed number of
// instructions; flops of both halves over the kernel's duration.
__global__ __launch_bounds__(512) void k_mixed64(double *out, int it_mfma, int it_valu, double a0, double b0) {
    const int w = threadIdx.x >> 6;
    double s = 0;
    if (w < 4) {
        d4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (d4){0, 0, 0, 0};
        const double a = a0 + threadIdx.x * 1e-9, b = b0;
        for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3 + i;
        const double a = 1.0000001 + threadIdx.x * 1e-9, b = b0 * 1e-3;
        for (int it = 0; it < it_valu; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <class K, class T>
double run(K kernel, int blocks, int iters, T *out, T a, T b) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters / 8, a, b);   // warm
    hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, a, b);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    void *buf;
    hipMalloc(&buf, (size_t)cus * 4 * 256 * 8);
    const int iters = 20000;
    for (int bpc : {1, 2}) {
        const int blocks = cus * bpc;
        {
            const double ms = run(k_f64<8>, blocks, iters, (double *)buf, 1.0, 0.5);
            const double mf = (double)blocks * 4 * iters * 8;             // wave-level MFMA instructions
            const double tf = mf * 2.0 * 16 * 16 * 4 / (ms * 1e-3) / 1e12;
            // 64 cycles per instruction per SIMD if the f64 matrix rate is 32 flop/clk/SIMD
            printf("f64 16x16x4  %d wave(s)/SIMD: %8.3f ms  %7.2f TFLOP/s   (%.0f cycles/instr/SIMD at 2.4 GHz)\n", bpc, ms, tf,
                   ms * 1e-3 * 2.4e9 / (mf / (cus * 4.0)));
        }
        {
            const double ms = run(k_f32<4>, blocks, iters, (float *)buf, 1.0f, 0.5f);
            const double mf = (double)blocks * 4 * iters * 4;
            const double tf = mf * 2.0 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
            printf("f32 32x32x2  %d wave(s)/SIMD: %8.3f ms  %7.2f TFLOP/s   (%.0f cycles/instr/SIMD at 2.4 GHz)\n", bpc, ms, tf,
                   ms * 1e-3 * 2.4e9 / (mf / (cus * 4.0)));
        }
    }
    // more waves per SIMD and other accumulator counts for the f64 matrix instruction; the f64 vector FMA beside it
    for (int bpc : {1, 2, 4}) {
        const int blocks = cus * bpc;
        const double ms16 = run(k_f64<16>, blocks, iters / 2, (double *)buf, 1.0, 0.5);
        const double ms4 = run(k_f64<4>, blocks, iters * 2, (double *)buf, 1.0, 0.5);
        const double mf = (double)blocks * 4 * iters * 8;
        printf("f64 16x16x4  %d wave(s)/SIMD: 16 accumulators %7.2f TFLOP/s   4 accumulators %7.2f TFLOP/s\n", bpc,
               mf * 2048.0 / (ms16 * 1e-3) / 1e12, mf * 2048.0 / (ms4 * 1e-3) / 1e12);
        const double msv = run(k_valu64<16>, blocks, iters * 4, (double *)buf, 1.0000001, 0.5);
        printf("f64 v_fma    %d wave(s)/SIMD: %7.2f TFLOP/s\n", bpc, (double)blocks * 256 * iters * 4 * 16 * 2.0 / (msv * 1e-3) / 1e12);
    }
    {
        // each half sized to ~the same time when alone (47 TF vs 58 TF): mfma 8192 flops / (4 acc) per iteration per wave,
        // valu 64 lanes * 16 * 2 = 2048 flops per iteration per wave
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int mode = 0; mode < 3; ++mode) {
            const int itm = mode == 1 ? 0 : 40000, itv = mode == 0 ? 0 : 200000;
            hipLaunchKernelGGL(k_mixed64, dim3(cus), dim3(512), 0, 0, (double *)buf, itm / 8, itv / 8, 1.0, 0.5);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_mixed64, dim3(cus), dim3(512), 0, 0, (double *)buf, itm, itv, 1.0, 0.5);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double fm = (double)cus * 4 * itm * 4 * 2048.0, fv = (double)cus * 4 * 64 * itv * 16 * 2.0;
            printf("f64 mixed (%s): %8.3f ms   mfma %6.2f + v_fma %6.2f = %6.2f TFLOP/s\n",
                   mode == 0 ? "mfma half only" : mode == 1 ? "v_fma half only" : "both halves", ms, fm / (ms * 1e-3) / 1e12,
                   fv / (ms * 1e-3) / 1e12, (fm + fv) / (ms * 1e-3) / 1e12);
        }
    }
    unsigned long long *ticks;
    hipMalloc(&ticks, 16);
    for (int bpc : {1, 2}) {
        hipLaunchKernelGGL(k_f64_clock<8>, dim3(cus * bpc), dim3(256), 0, 0, (double *)buf, iters, 1.0, 0.5, ticks);
        unsigned long long h[2];
        hipMemcpy(h, ticks, 16, hipMemcpyDeviceToHost);
        printf("f64 16x16x4  %d wave(s)/SIMD: %.1f s_memtime ticks per MFMA per wave, %.3f ticks per 10 ns of wall clock\n", bpc,
               (double)h[0] / ((double)iters * 8), (double)h[0] / (double)h[1]);
    }
    hipFree(buf);
    return 0;
}
