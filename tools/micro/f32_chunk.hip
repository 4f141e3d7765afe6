// The chunk loop of the exact-fp32 conv (net_fp32.hip:conv_mfma, MT = 1, NTW = 3) in isolation: 48 x v_mfma_f32_16x16x4_f32 per
// chunk, A operands by 4 global_load_dwordx4 a chunk ahead, B operands by 3 ds_read_b128 a group ahead.  Variants switch the
// operand traffic off / make the operands constant to find what keeps a lone wave from 32 cycles per MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -o f32_chunk f32_chunk.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ float lds[];

template <int MODE, int DEPTH = 1>   // bit 0: B from LDS; bit 1: A from global; bit 2: random data instead of zeros;
                                     // bit 3: a new tap every chunk (neighbour table lookup + base select), groups fenced by sched_barrier
__global__ __launch_bounds__(512) void k(const float* __restrict__ W, float* out, long long* cyc, int chunks, int wstride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 144 * 81; i += blockDim.x) lds[i] = (MODE & 4) ? (float)((i * 2654435761u) >> 8) * 1e-8f - 0.08f : 0.f;
    __syncthreads();
    typedef const __attribute__((address_space(1))) f32x4* gptr_t;
    gptr_t wl = (gptr_t)W + lane;
    const int u0 = wave + (wstride > 64 ? blockIdx.x * 61 : 0);   // wstride > 64: every workgroup at its own phase of a 4.7 MB sweep
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 a0[4], a1[4], a2[4], bv[3];
    const int jl = lane & 15, kq = lane >> 4;
    int base[3], gstr[3];
    short* nbt = (short*)(lds + 144 * 81);
    for (int i = threadIdx.x; i < 9 * 96; i += blockDim.x) nbt[i] = (short)((i % 96) < 81 ? ((i % 96) + i / 96) % 81 : -1);
    __syncthreads();
    for (int nt = 0; nt < 3; nt++) base[nt] = 16 * (kq * 81 + (16 * nt + jl) % 81), gstr[nt] = 64 * 81;
    for (int j = 0; j < 4; j++) a0[j] = (MODE & 2) ? wl[(u0 % wstride) * 256 + j * 64] : (f32x4){1.f, 2.f, 3.f, 4.f};
    if (DEPTH == 2)
        for (int j = 0; j < 4; j++) a1[j] = (MODE & 2) ? wl[((8 + u0) % wstride) * 256 + j * 64] : (f32x4){1.f, 2.f, 3.f, 4.f};
    for (int nt = 0; nt < 3; nt++) bv[nt] = (MODE & 1) ? *(const f32x4*)((const char*)lds + base[nt]) : (f32x4){1.f, 1.f, 1.f, 1.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    auto chunk = [&](f32x4 (&a)[4], f32x4 (&an)[4], int c) {
        if (MODE & 2)
            for (int j = 0; j < 4; j++) an[j] = wl[(((c + DEPTH) * 8 + u0) % wstride) * 256 + j * 64];
        int qn[3], basen[3], gstrn[3];
        if (MODE & 8)
            for (int nt = 0; nt < 3; nt++) qn[nt] = nbt[((c + 1) % 9) * 96 + 16 * nt + jl];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            f32x4 bn[3];
            if (MODE & 8) {
                if (g == 3)
                    for (int nt = 0; nt < 3; nt++) base[nt] = basen[nt], gstr[nt] = gstrn[nt];
            }
            if (MODE & 64) {
#pragma unroll
                for (int nt = 0; nt < 3; nt++) bv[nt] = *(const f32x4*)((const char*)lds + (__umul24(g, 64 * 81) + base[nt]));
            } else if (MODE & 1) {
                const int G = (g + 1) & 3;
#pragma unroll
                for (int nt = 0; nt < 3; nt++) bn[nt] = *(const f32x4*)((const char*)lds + (__umul24(G, (MODE & 8) ? gstr[nt] : 64 * 81) + base[nt]));
            }
            if ((MODE & 8) && g == 0)
                for (int nt = 0; nt < 3; nt++) {
                    basen[nt] = qn[nt] >= 0 ? 16 * (kq * 81 + qn[nt]) : 4 * 144 * 81 + 9 * 96 * 2;
                    gstrn[nt] = qn[nt] >= 0 ? 64 * 81 : 0;
                }
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int nt = 0; nt < 3; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][s], bv[nt][s], acc[nt], 0, 0, 0);
            if ((MODE & 65) == 1) {
#pragma unroll
                for (int nt = 0; nt < 3; nt++) bv[nt] = bn[nt];
            }
            if ((MODE & 56) == 8) __builtin_amdgcn_sched_barrier(0);   // bit 5: no fence at all
            if (MODE & 16) {   // pin: an MFMA, then a slice of the other work, 12 times
#pragma unroll
                for (int i = 0; i < 12; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    if (i < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (g == 0 && i >= 3 && i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (g == 0 && i >= 6 && i < 10) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    };
    if (DEPTH == 1) {
        for (int c = 0; c < chunks; c += 2) {
            chunk(a0, a1, c);
            chunk(a1, a0, c + 1);
        }
    } else {
        for (int c = 0; c + 2 < chunks; c += 3) {
            chunk(a0, a2, c);
            chunk(a1, a0, c + 1);
            chunk(a2, a1, c + 2);
        }
    }
    float s = 0;
    for (int nt = 0; nt < 3; nt++) s += acc[nt][0] + acc[nt][3];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int DEPTH = 1>
void run(int threads, const float* W, float* out, long long* cyc, int wstride = 64) {
    const int chunks = 510;
    hipFuncSetAttribute((const void*)k<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(threads), 64 * 1024, 0, W, out, cyc, chunks, wstride);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, DEPTH>), dim3(256), dim3(threads), 64 * 1024, 0, W, out, cyc, chunks, wstride);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = chunks * 48.0;
    printf("mode %2d depth %d units %4d B-lds %d A-global %d data %d  waves/SIMD %d: wave 0 %.1f cycles per own MFMA; wall %.3f ms = %.1f TFLOP/s\n", MODE, DEPTH, wstride, MODE & 1,
           (MODE >> 1) & 1, (MODE >> 2) & 1, threads / 256, c / n, ms, 2048.0 * n * (threads / 64) * 256 / ms * 1e-9);
}

int main() {
    float *W, *out;
    long long* cyc;
    hipMalloc(&W, 64 * 16 * 1024 * 8 + 4096);
    hipMemset(W, 0, 64 * 16 * 1024 * 8 + 4096);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    for (int th : {256, 512}) {
        run<0>(th, W, out, cyc);
        run<1>(th, W, out, cyc);
        run<2>(th, W, out, cyc);
        run<3>(th, W, out, cyc);
        run<7>(th, W, out, cyc);
        run<7>(th, W, out, cyc, 1152);
        run<7, 2>(th, W, out, cyc, 1152);
        run<7, 2>(th, W, out, cyc, 64);
        run<15>(th, W, out, cyc, 1152);
        run<31>(th, W, out, cyc, 1152);
        run<47>(th, W, out, cyc, 1152);
        run<71>(th, W, out, cyc, 1152);
        run<2>(th, W, out, cyc, 1152);
    }
    return 0;
}
