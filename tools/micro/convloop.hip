// Micro-benchmark of the 3x3-conv MFMA loop of net_mfma.hip in isolation: 8 waves (2 per SIMD), A fragments from an
// LDS "ring" region, B operands from an LDS image through shifted rows, v_mfma_f32_32x32x16_f16, fp32 accumulate.
// Variants (template): LOOK = k-steps of operand look-ahead, BAR = workgroup barrier every 12 k-steps (a ring stage),
// NROW x NC output tiles per wave.  Prints cycles per MFMA per SIMD (32 = matrix peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
constexpr int RS = 144, KS = 4, TAPS = 9, KK = 36, SK = 12, NROT = 2;

template <int LOOK, int BAR, int NROW, int NC, int PRIO>
__global__ __launch_bounds__(512, 2) void k_loop(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                 // 3 x 24 KiB
    char* img = smem + 3 * 24 * 1024;  // 320 rows x 144 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
    for (int i = tid; i < (3 * 24 * 1024 + 320 * RS) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
    __syncthreads();
    if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    f16v acc[NROW][NC];
    for (int a = 0; a < NROW; a++) for (int b = 0; b < NC; b++) for (int i = 0; i < 16; i++) acc[a][b][i] = 0.f;
    struct Ops { h8 a[NROW], b[NC]; };
    const int col0 = 16 + 32 * ((wave * NC) % 8);
    auto load = [&](int kk, int slot, Ops& o) {
        const int t = kk / KS, ks = kk % KS;
        const int d = (t / 3 - 1) * 9 + (t % 3 - 1);
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
            o.b[ct] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + (col0 + 32 * ct + r + d) * RS + (16 * ks + 8 * h) * 2));
#pragma unroll
        for (int ro = 0; ro < NROW; ro++)
            o.a[ro] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(ring + slot * 24 * 1024 + ((kk % SK) * NROT + ro) * 1024 + lane * 16));
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        Ops o[LOOK + 1];
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            if (kk % SK == 0) {
                if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int j = 0; j < LOOK; j++) load(kk + j, (kk / SK) % 3, o[j % (LOOK + 1)]);
            }
            const int nx = kk + LOOK;
            if (nx < KK && nx / SK == kk / SK) load(nx, (kk / SK) % 3, o[nx % (LOOK + 1)]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ro = 0; ro < NROW; ro++)
#pragma unroll
                for (int ct = 0; ct < NC; ct++) acc[ro][ct] = MFMA(o[kk % (LOOK + 1)].a[ro], o[kk % (LOOK + 1)].b[ct], acc[ro][ct]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NROW; a++) for (int b = 0; b < NC; b++) for (int i = 0; i < 16; i++) s += acc[a][b][i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int LOOK, int BAR, int NROW, int NC, int PRIO>
void run(const char* name, float* out, unsigned long long* cyc, int grid) {
    const int iters = 200;
    const size_t lds = 3 * 24 * 1024 + 320 * RS;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<LOOK, BAR, NROW, NC, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 6; rep++) {   // the clock settles over the first launches (DVFS): the last one is reported
        hipEventRecord(a);
        hipLaunchKernelGGL((k_loop<LOOK, BAR, NROW, NC, PRIO>), dim3(grid), dim3(512), lds, 0, out, cyc, iters);
        hipEventRecord(b);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(grid * 8);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mx0 = 0, mx4 = 0;
    for (int g = 0; g < grid; g++) { mx0 += c[g * 8 + 0]; mx4 += c[g * 8 + 7]; }
    const double mf = (double)iters * KK * NROW * NC;   // MFMAs per wave
    const double flops = (double)grid * 8 * mf * 32768.0;
    printf("%-28s %7.3f ms  %7.1f TFLOP/s  cycles/MFMA/SIMD: wave0 %.1f wave7 %.1f  (shader clock %.2f GHz)\n", name, ms, flops / ms / 1e9,
           mx0 / grid / mf / 2, mx4 / grid / mf / 2, mx4 / grid * (grid / 256.0) / (ms * 1e6));
}

int main() {
    float* out; unsigned long long* cyc;
    const int grid = 256 * 4;
    hipMalloc(&out, grid * 512 * 4); hipMalloc(&cyc, grid * 8 * 8);
    run<1, 1, 2, 1, 0>("look1 bar 2x1", out, cyc, grid);
    run<2, 1, 2, 1, 0>("look2 bar 2x1", out, cyc, grid);
    run<3, 1, 2, 1, 0>("look3 bar 2x1", out, cyc, grid);
    run<1, 0, 2, 1, 0>("look1 nobar 2x1", out, cyc, grid);
    run<2, 0, 2, 1, 0>("look2 nobar 2x1", out, cyc, grid);
    run<1, 1, 2, 1, 1>("look1 bar 2x1 prio", out, cyc, grid);
    run<2, 1, 2, 1, 1>("look2 bar 2x1 prio", out, cyc, grid);
    run<1, 1, 1, 2, 0>("look1 bar 1x2", out, cyc, grid);
    run<1, 1, 2, 2, 0>("look1 bar 2x2", out, cyc, grid);
    run<2, 1, 2, 2, 0>("look2 bar 2x2", out, cyc, grid);
    return 0;
}
