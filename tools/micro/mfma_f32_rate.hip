// Issue rate of the fp32 MFMA shapes on gfx950: cycles (s_memtime) per instruction for 1, 2 and 4 waves per SIMD, each wave
// running NACC independent accumulators.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters) {
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0, t1;
    if (SHAPE == 16) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; i++)
            for (int e = 0; e < 16; e++) acc[i][e] = 0;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][15];
        t1 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int SHAPE, int NACC>
void run(int waves_per_simd, float* out, long long* cyc) {
    const int iters = 2000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * NACC;                 // MFMAs per wave
    const double flop = (SHAPE == 16 ? 2048.0 : 4096.0) * n * waves_per_simd * 4 * 256;
    printf("%s  acc %d  waves/SIMD %d: %.1f cycles per MFMA per SIMD (wave: %.1f), wall %.3f ms, %.1f TFLOP/s, clock %.2f GHz\n",
           SHAPE == 16 ? "16x16x4 " : "32x32x2 ", NACC, waves_per_simd, c / (n * waves_per_simd), c / n, ms, flop / ms * 1e-9,
           c / ms * 1e-6);
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 8);
    for (int w : {1, 2, 4}) {
        run<16, 2>(w, out, cyc);
        run<16, 3>(w, out, cyc);
        run<16, 4>(w, out, cyc);
        run<16, 6>(w, out, cyc);
        run<16, 8>(w, out, cyc);
        run<32, 2>(w, out, cyc);
        run<32, 4>(w, out, cyc);
    }
    return 0;
}
