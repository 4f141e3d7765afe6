// Micro-benchmark of the 3x3-conv MFMA loop of net_mfma.hip in isolation: 8 waves (2 per SIMD), A fragments from an
// LDS "ring" region, B operands from an LDS image through shifted rows, v_mfma_f32_32x32x16_f16, fp32 accumulate.
// Variants (template): LOOK = k-steps of operand look-ahead, BAR = workgroup barrier every 12 k-steps (a ring stage),
// NROW x NC output tiles per wave.  Prints cycles per MFMA per SIMD (32 = matrix peak).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
constexpr int RS = 144, KS = 4, TAPS = 9, KK = 36, SK = 12, NROT = 2;

// DMA: 0 none; 1 the weight stream as net_mfma.hip issues it (three 1-KiB global_load_lds_dwordx4 per wave right after the
// stage barrier, per-lane 64-bit address, vmcnt(0) at every barrier); 2 the same with a scalar base + 32-bit lane offset;
// 3 scalar base, the pieces spread over the stage: wave w issues at k-steps w/2, w/2 + 4, w/2 + 8 (two pieces per k-step
// in the whole workgroup, SIMD partners two k-steps apart), skipped by a branch inside the asm statement, vmcnt(3) at the
// barrier; 4 scalar base, every wave at k-steps 1, 5, 9 (no stagger), vmcnt(3).
// IDLE: what follows every conv (36 k-steps) - 0 nothing; 1 s_sleep (~IDLEN x 64 cycles); 2 a dependent v_pk_fma_f32 chain of
// IDLEN x 16 instructions on 4 registers (LayerNorm-like VALU work); 3 IDLEN x (ds_write_b64 + ds_read_b128) on the image.
template <int LOOK, int BAR, int NROW, int NC, int PRIO, int DMA = 0, int IDLE = 0, int IDLEN = 0>
__global__ __launch_bounds__(512, 2) void k_loop(float* out, unsigned long long* cyc, int iters, const char* wstream = nullptr) {
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
    const unsigned ringoff = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
    const unsigned lane16 = lane * 16;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const int cls = uw >> 1;
    int conv = 0;                                   // position in the 50-conv weight stream (3.6 MB, L2-resident)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        Ops o[LOOK + 1];
        const char* wbase = wstream + (size_t)conv * (72 * 1024);
        conv = conv == 49 ? 0 : conv + 1;
#pragma unroll
        for (int kk = 0; kk < KK; kk++) {
            const int st = kk / SK;
            // piece i of this wave for the stage two ahead: fragment f = wave + 8 i of slot (st + 2) % 3
            auto piece = [&](int i, bool guard, int want) {
                const unsigned f = uw + 8 * i;
                const char* sp = wbase + ((st + 2) % 3) * 24 * 1024 + f * 1024;   // (the same 72 KB again: stream position is immaterial)
                const unsigned m0v = ringoff + ((st + 2) % 3) * 24 * 1024 + f * 1024;
                if (DMA == 1) {
                    const char* gp = sp + lane16;
                    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(m0v) : "memory");
                } else if (!guard) {
                    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(lane16), "s"(m0v), "s"(sp) : "memory");
                } else {
                    asm volatile("s_cmp_lg_u32 %3, %4\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, %2\n1:"
                                 ::"v"(lane16), "s"(m0v), "s"(sp), "s"(cls), "s"(want) : "memory", "scc");
                }
            };
            if (kk % SK == 0) {
                if (BAR) {
                    if (DMA == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else if (DMA <= 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                if (DMA == 1 || DMA == 2)
                    for (int i = 0; i < 3; i++) piece(i, false, 0);
#pragma unroll
                for (int j = 0; j < LOOK; j++) load(kk + j, (kk / SK) % 3, o[j % (LOOK + 1)]);
            }
            const int nx = kk + LOOK;
            if (nx < KK && nx / SK == kk / SK) load(nx, (kk / SK) % 3, o[nx % (LOOK + 1)]);
            if (DMA == 4 && (kk % SK) % 4 == 1) piece((kk % SK) / 4, false, 0);
            if (DMA == 3) piece((kk % SK) / 4, true, (kk % SK) % 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ro = 0; ro < NROW; ro++)
#pragma unroll
                for (int ct = 0; ct < NC; ct++) acc[ro][ct] = MFMA(o[kk % (LOOK + 1)].a[ro], o[kk % (LOOK + 1)].b[ct], acc[ro][ct]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (IDLE == 1) {
            for (int j = 0; j < IDLEN; j++) __builtin_amdgcn_s_sleep(1);
        } else if (IDLE == 2) {
            typedef __attribute__((ext_vector_type(2))) float f2;
            f2 p = {acc[0][0][0], acc[0][0][1]}, q = {acc[0][0][2], acc[0][0][3]};
            for (int j = 0; j < IDLEN; j++) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    p = p * q + p;
                    q = q * p + q;
                }
            }
            acc[0][0][0] += 1e-30f * (p.x + p.y + q.x + q.y);
        } else if (IDLE == 3) {
            u4 v = {0, 0, 0, 0};
            for (int j = 0; j < IDLEN; j++) {
                *reinterpret_cast<unsigned long long*>(img + (16 + tid / 2) * RS + (j & 7) * 16 + (tid & 1) * 8) = v[0] + j;
                u4 w = *reinterpret_cast<const u4*>(img + (16 + (tid + j) % 256) * RS + (j & 7) * 16);
                v[0] ^= w[0]; v[1] ^= w[1];
            }
            acc[0][0][0] += 1e-30f * (float)(v[0] + v[1]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NROW; a++) for (int b = 0; b < NC; b++) for (int i = 0; i < 16; i++) s += acc[a][b][i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// The 2x1 loop with the k-steps taken in pairs and the two MFMAs of one accumulator back to back (r0 r0 r1 r1 instead of
// r0 r1 r0 r1): does a chained accumulator cost less energy (wall time at the power cap)?
template <int CHAIN>
__global__ __launch_bounds__(512, 2) void k_pair(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* img = smem + 3 * 24 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
    for (int i = tid; i < (3 * 24 * 1024 + 320 * RS) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
    __syncthreads();
    f16v acc[2];
    for (int a = 0; a < 2; a++) for (int i = 0; i < 16; i++) acc[a][i] = 0.f;
    struct Ops { h8 a[2], b; };
    const int col0 = 16 + 32 * (wave % 8);
    auto load = [&](int kk, int slot, Ops& o) {
        const int t = kk / KS, ks = kk % KS;
        const int d = (t / 3 - 1) * 9 + (t % 3 - 1);
        o.b = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + (col0 + r + d) * RS + (16 * ks + 8 * h) * 2));
#pragma unroll
        for (int ro = 0; ro < 2; ro++)
            o.a[ro] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(ring + slot * 24 * 1024 + ((kk % SK) * NROT + ro) * 1024 + lane * 16));
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        Ops o[4];
#pragma unroll
        for (int kk = 0; kk < KK; kk += 2) {
            if (kk % SK == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                load(kk, (kk / SK) % 3, o[kk % 4]);
                load(kk + 1, (kk / SK) % 3, o[(kk + 1) % 4]);
            }
            if ((kk + 2) % SK != 0) {
                load(kk + 2, (kk / SK) % 3, o[(kk + 2) % 4]);
                load(kk + 3, (kk / SK) % 3, o[(kk + 3) % 4]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const Ops &p = o[kk % 4], &q = o[(kk + 1) % 4];
            if (CHAIN) {
                acc[0] = MFMA(p.a[0], p.b, acc[0]);
                acc[0] = MFMA(q.a[0], q.b, acc[0]);
                acc[1] = MFMA(p.a[1], p.b, acc[1]);
                acc[1] = MFMA(q.a[1], q.b, acc[1]);
            } else {
                acc[0] = MFMA(p.a[0], p.b, acc[0]);
                acc[1] = MFMA(p.a[1], p.b, acc[1]);
                acc[0] = MFMA(q.a[0], q.b, acc[0]);
                acc[1] = MFMA(q.a[1], q.b, acc[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 2; a++) for (int i = 0; i < 16; i++) s += acc[a][i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int CHAIN>
void run_pair(const char* name, float* out, unsigned long long* cyc, int grid) {
    const int iters = 200;
    const size_t lds = 3 * 24 * 1024 + 320 * RS;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair<CHAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_pair<CHAIN>), dim3(grid), dim3(512), lds, 0, out, cyc, iters);
        hipEventRecord(b);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(grid * 8);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int g = 0; g < grid; g++) mx += c[g * 8 + 7];
    const double mf = (double)iters * KK * 2;
    printf("%-28s %7.3f ms  %7.1f TFLOP/s  cycles/MFMA/SIMD %.1f  (shader clock %.2f GHz)\n", name, ms, (double)grid * 8 * mf * 32768.0 / ms / 1e9,
           mx / grid / mf / 2, mx / grid * (grid / 256.0) / (ms * 1e6));
}


// The same loop on v_mfma_f32_16x16x32_f16 at the SAME output tile per wave (MI355X_MICROARCH.md DVFS
// give-back item 7): a wave's NROW x NC blocks of 32 channels x 32 columns are 2NROW x 2NC tiles of 16 x 16, a k-step is 32
// input channels (18 k-steps per 3x3 conv on 64 channels, 6 per ring stage); per k-step 2NROW A fragments (16 rows x 32 k =
// 1 KiB each, lane * 16) and 2NC B operands (lane (r = lane & 15, q = lane >> 4) reads channels 32 ks + 8 q .. + 7 of image row
// col + 16 j + r + d): the same LDS bytes per FLOP as the 32x32x16 loop, twice the MFMA instructions at half the cycles each.
typedef __attribute__((ext_vector_type(4))) float f4v;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
template <int NROW, int NC, int DMA, int RS16>
__global__ __launch_bounds__(512, 2) void k_loop16(float* out, unsigned long long* cyc, int iters, const char* wstream = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* img = smem + 3 * 24 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, r = lane & 15;
    for (int i = tid; i < (3 * 24 * 1024 + 320 * RS16) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
    __syncthreads();
    constexpr int KS2 = 2, KK2 = TAPS * KS2, SK2 = 6, NA = 2 * NROW, NB = 2 * NC;
    f4v acc[NA][NB];
    for (int a = 0; a < NA; a++) for (int b = 0; b < NB; b++) for (int i = 0; i < 4; i++) acc[a][b][i] = 0.f;
    struct Ops { h8 a[NA], b[NB]; };
    const int col0 = 16 + 32 * ((wave * NC) % 8);
    auto load = [&](int kk, int slot, Ops& o) {
        const int t = kk / KS2, ks = kk % KS2;
        const int d = (t / 3 - 1) * 9 + (t % 3 - 1);
#pragma unroll
        for (int cj = 0; cj < NB; cj++)
            o.b[cj] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + (col0 + 16 * cj + r + d) * RS16 + (32 * ks + 8 * q) * 2));
#pragma unroll
        for (int ri = 0; ri < NA; ri++)
            o.a[ri] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(ring + slot * 24 * 1024 + ((kk % SK2) * (2 * NROT) + ri) * 1024 + lane * 16));
    };
    const unsigned ringoff = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
    const unsigned lane16 = lane * 16;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    int conv = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        Ops o[2];
        const char* wbase = wstream + (size_t)conv * (72 * 1024);
        conv = conv == 49 ? 0 : conv + 1;
#pragma unroll
        for (int kk = 0; kk < KK2; kk++) {
            const int st = kk / SK2;
            if (kk % SK2 == 0) {
                if (DMA == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (DMA)
                    for (int i = 0; i < 3; i++) {
                        const unsigned f = uw + 8 * i;
                        const char* sp = wbase + ((st + 2) % 3) * 24 * 1024 + f * 1024;
                        const unsigned m0v = ringoff + ((st + 2) % 3) * 24 * 1024 + f * 1024;
                        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(lane16), "s"(m0v), "s"(sp) : "memory");
                    }
                load(kk, st % 3, o[kk % 2]);
            }
            const int nx = kk + 1;
            if (nx < KK2 && nx / SK2 == kk / SK2) load(nx, st % 3, o[nx % 2]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ri = 0; ri < NA; ri++)
#pragma unroll
                for (int cj = 0; cj < NB; cj++) acc[ri][cj] = MFMA16(o[kk % 2].a[ri], o[kk % 2].b[cj], acc[ri][cj]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NA; a++) for (int b = 0; b < NB; b++) for (int i = 0; i < 4; i++) s += acc[a][b][i];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int NROW, int NC, int DMA, int RS16 = 160>
double run16(const char* name, float* out, unsigned long long* cyc, int grid, const char* wstream = nullptr) {
    const int iters = 200;
    const size_t lds = 3 * 24 * 1024 + 320 * RS16;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop16<NROW, NC, DMA, RS16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_loop16<NROW, NC, DMA, RS16>), dim3(grid), dim3(512), lds, 0, out, cyc, iters, wstream);
        hipEventRecord(b);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(grid * 8);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mx0 = 0, mx4 = 0;
    for (int g = 0; g < grid; g++) { mx0 += c[g * 8 + 0]; mx4 += c[g * 8 + 7]; }
    const double mf = (double)iters * 18 * 4 * NROW * NC;   // 16x16x32 MFMAs per wave
    const double flops = (double)grid * 8 * mf * 16384.0;
    // cycles are quoted per 32768 FLOP (one 32x32x16 MFMA = two 16x16x32), so 32 = matrix peak in both tables
    printf("%-28s %7.3f ms  %7.1f TFLOP/s  cycles/32768 FLOP/SIMD: wave0 %.1f wave7 %.1f  (shader clock %.2f GHz, %.0f cycles per conv)\n", name, ms,
           flops / ms / 1e9, mx0 / grid / mf, mx4 / grid / mf, mx4 / grid * (grid / 256.0) / (ms * 1e6), mx4 / grid / iters);
    return ms;
}


// Both shapes in one template, with the operand look-ahead optionally running ACROSS the ring-stage barriers (XB): the A
// fragments of stage st + 1 are already published when stage st runs (every wave waited vmcnt(0) for its share of both in-flight
// stages before barrier st), so the first k-step of the next stage can be requested before barrier st + 1, and the MFMAs behind
// the barrier start at once instead of after 8 waves x 6-12 KiB of LDS reads.  The barrier then only frees slot st for the DMA.
template <int SHAPE, int NROW, int NC, int DMA, int XB, int RSI, int INTER = 0>
__global__ __launch_bounds__(512, 2) void k_loopx(float* out, unsigned long long* cyc, int iters, const char* wstream = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* img = smem + 3 * 24 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (3 * 24 * 1024 + 320 * RSI) / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u ^ (i * 2654435761u & 0x03ff03ffu);
    __syncthreads();
    constexpr bool S16 = SHAPE == 16;
    constexpr int KSX = S16 ? 2 : 4, KKX = TAPS * KSX, SKX = S16 ? 6 : 12, NA = S16 ? 2 * NROW : NROW, NB = S16 ? 2 * NC : NC;
    constexpr int FPK = S16 ? 4 : 2;   // fragments per k-step in the ring (NROT = 2 row tiles of 32)
    const int q = S16 ? lane >> 4 : lane >> 5, r = S16 ? lane & 15 : lane & 31;
    f4v acc16[S16 ? NA : 1][S16 ? NB : 1];
    f16v acc32[S16 ? 1 : NA][S16 ? 1 : NB];
    if (S16) { for (int a = 0; a < NA; a++) for (int b = 0; b < NB; b++) for (int i = 0; i < 4; i++) acc16[a][b][i] = 0.f; }
    else { for (int a = 0; a < NA; a++) for (int b = 0; b < NB; b++) for (int i = 0; i < 16; i++) acc32[a][b][i] = 0.f; }
    struct Ops { h8 a[NA], b[NB]; };
    const int col0 = 16 + 32 * ((wave * NC) % 8);
    auto load = [&](int kk, Ops& o) {
        kk %= KKX;
        const int slot = (kk / SKX) % 3;
        const int t = kk / KSX, ks = kk % KSX;
        const int d = (t / 3 - 1) * 9 + (t % 3 - 1);
#pragma unroll
        for (int cj = 0; cj < NB; cj++)
            o.b[cj] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(img + (col0 + (S16 ? 16 : 32) * cj + r + d) * RSI + ((S16 ? 32 : 16) * ks + 8 * q) * 2));
#pragma unroll
        for (int ri = 0; ri < NA; ri++)
            o.a[ri] = __builtin_bit_cast(h8, *reinterpret_cast<const u4*>(ring + slot * 24 * 1024 + ((kk % SKX) * FPK + ri) * 1024 + lane * 16));
    };
    const unsigned ringoff = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
    const unsigned lane16 = lane * 16;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    int conv = 0;
    Ops o[2];
    if (XB) load(0, o[0]);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const char* wbase = wstream + (size_t)conv * (72 * 1024);
        conv = conv == 49 ? 0 : conv + 1;
#pragma unroll
        for (int kk = 0; kk < KKX; kk++) {
            const int st = kk / SKX;
            if (kk % SKX == 0) {
                if (XB) {
                    if (DMA == 0) asm volatile("s_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    if (DMA == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                if (DMA)
                    for (int i = 0; i < 3; i++) {
                        const unsigned f = uw + 8 * i;
                        const char* sp = wbase + ((st + 2) % 3) * 24 * 1024 + f * 1024;
                        const unsigned m0v = ringoff + ((st + 2) % 3) * 24 * 1024 + f * 1024;
                        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(lane16), "s"(m0v), "s"(sp) : "memory");
                    }
                if (!XB) load(kk, o[kk % 2]);
            }
            const int nx = kk + 1;
            if (INTER) __builtin_amdgcn_sched_barrier(0);
            if (XB || (nx < KKX && nx / SKX == kk / SKX)) load(nx, o[nx % 2]);
            if (!INTER) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ri = 0; ri < NA; ri++)
#pragma unroll
                for (int cj = 0; cj < NB; cj++) {
                    if constexpr (S16) acc16[ri][cj] = MFMA16(o[kk % 2].a[ri], o[kk % 2].b[cj], acc16[ri][cj]);
                    else acc32[ri][cj] = MFMA(o[kk % 2].a[ri], o[kk % 2].b[cj], acc32[ri][cj]);
                }
            if (INTER) {   // the next k-step's reads dealt out between this k-step's MFMAs instead of in a block in front of them
                constexpr int M = NA * NB, R = NA + NB;
#pragma unroll
                for (int m = 0; m < M; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    const int nr = (m + 1) * R / M - m * R / M;
                    if (nr == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    else if (nr == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else if (nr == 3) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (S16) { for (int a = 0; a < NA; a++) for (int b = 0; b < NB; b++) for (int i = 0; i < 4; i++) s += acc16[a][b][i]; }
    else { for (int a = 0; a < NA; a++) for (int b = 0; b < NB; b++) for (int i = 0; i < 16; i++) s += acc32[a][b][i]; }
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int SHAPE, int NROW, int NC, int DMA, int XB, int RSI, int INTER = 0>
double runx(const char* name, float* out, unsigned long long* cyc, int grid, const char* wstream = nullptr) {
    const int iters = 200;
    const size_t lds = 3 * 24 * 1024 + 320 * RSI;
    auto kern = &k_loopx<SHAPE, NROW, NC, DMA, XB, RSI, INTER>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, out, cyc, iters, wstream);
        hipEventRecord(b);
        hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> c(grid * 8);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mx0 = 0, mx4 = 0;
    for (int g = 0; g < grid; g++) { mx0 += c[g * 8 + 0]; mx4 += c[g * 8 + 7]; }
    const double mf = (double)iters * 36 * NROW * NC;   // in units of 32768 FLOP per wave
    const double flops = (double)grid * 8 * mf * 32768.0;
    printf("%-34s %7.3f ms  %7.1f TFLOP/s  cycles/32768 FLOP/SIMD: wave0 %.1f wave7 %.1f  (shader clock %.2f GHz, %.0f cycles per conv)\n", name, ms,
           flops / ms / 1e9, mx0 / grid / mf / 2, mx4 / grid / mf / 2, mx4 / grid * (grid / 256.0) / (ms * 1e6), mx4 / grid / iters);
    return ms;
}

template <int LOOK, int BAR, int NROW, int NC, int PRIO, int DMA = 0, int IDLE = 0, int IDLEN = 0>
void run(const char* name, float* out, unsigned long long* cyc, int grid, const char* wstream = nullptr) {
    const int iters = 200;
    const size_t lds = 3 * 24 * 1024 + 320 * RS;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<LOOK, BAR, NROW, NC, PRIO, DMA, IDLE, IDLEN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 6; rep++) {   // the clock settles over the first launches (DVFS): the last one is reported
        hipEventRecord(a);
        hipLaunchKernelGGL((k_loop<LOOK, BAR, NROW, NC, PRIO, DMA, IDLE, IDLEN>), dim3(grid), dim3(512), lds, 0, out, cyc, iters, wstream);
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
    printf("%-28s %7.3f ms  %7.1f TFLOP/s  cycles/MFMA/SIMD: wave0 %.1f wave7 %.1f  (shader clock %.2f GHz, %.0f cycles per conv)\n", name, ms,
           flops / ms / 1e9, mx0 / grid / mf / 2, mx4 / grid / mf / 2, mx4 / grid * (grid / 256.0) / (ms * 1e6), mx4 / grid / iters);
}

// the two MFMA shapes at the same output tile per wave, interleaved rounds in one process on random data
int main_shapes() {
    float* out; unsigned long long* cyc;
    const int grid = 256 * 4;
    hipMalloc(&out, grid * 512 * 4); hipMalloc(&cyc, grid * 8 * 8);
    char* ws; hipMalloc(&ws, 51 * 72 * 1024); hipMemset(ws, 0x3c, 51 * 72 * 1024);
    for (int round = 0; round < 4; round++) {
        printf("-- round %d\n", round);
        run<1, 1, 2, 1, 0>("32x32x16 2x1", out, cyc, grid);
        run16<2, 1, 0, 144>("16x16x32 2x1 rs144 (2-way B)", out, cyc, grid);
        run16<2, 1, 0>("16x16x32 2x1 rs160", out, cyc, grid);
        run<1, 1, 2, 1, 0, 2>("32x32x16 2x1 + weight DMA", out, cyc, grid, ws);
        run16<2, 1, 2>("16x16x32 2x1 rs160 + DMA", out, cyc, grid, ws);
        runx<32, 2, 1, 0, 1, 144>("32x32x16 2x1 xbar", out, cyc, grid);
        runx<16, 2, 1, 0, 1, 160>("16x16x32 2x1 rs160 xbar", out, cyc, grid);
        runx<32, 2, 1, 2, 0, 144>("32x32x16 2x1 + DMA (k_loopx)", out, cyc, grid, ws);
        runx<32, 2, 1, 2, 1, 144>("32x32x16 2x1 + DMA xbar", out, cyc, grid, ws);
        runx<16, 2, 1, 2, 1, 160>("16x16x32 2x1 rs160 + DMA xbar", out, cyc, grid, ws);
        runx<32, 2, 1, 2, 1, 144, 1>("32x32x16 2x1 + DMA xbar inter", out, cyc, grid, ws);
        runx<16, 2, 1, 2, 1, 160, 1>("16x16x32 2x1 rs160 + DMA xbar inter", out, cyc, grid, ws);
        runx<16, 2, 1, 0, 1, 160, 1>("16x16x32 2x1 rs160 xbar inter", out, cyc, grid, ws);
        run<1, 1, 2, 2, 0>("32x32x16 2x2", out, cyc, grid);
        run16<2, 2, 0>("16x16x32 2x2 rs160", out, cyc, grid);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "shapes")) return main_shapes();
    float* out; unsigned long long* cyc;
    const int grid = 256 * 4;
    hipMalloc(&out, grid * 512 * 4); hipMalloc(&cyc, grid * 8 * 8);
    char* ws; hipMalloc(&ws, 51 * 72 * 1024); hipMemset(ws, 0x3c, 51 * 72 * 1024);
    run<1, 1, 2, 1, 0>("look1 bar 2x1", out, cyc, grid);
    run_pair<0>("pairs r0 r1 r0 r1", out, cyc, grid);
    run_pair<1>("pairs r0 r0 r1 r1 (chained)", out, cyc, grid);
    run_pair<0>("pairs r0 r1 r0 r1", out, cyc, grid);
    run_pair<1>("pairs r0 r0 r1 r1 (chained)", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 1, 8>("  + sleep 8 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 1, 16>("  + sleep 16 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 1, 32>("  + sleep 32 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 2, 8>("  + valu 128 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 2, 16>("  + valu 256 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 2, 32>("  + valu 512 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 3, 16>("  + lds 16 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 0, 3, 32>("  + lds 32 per conv", out, cyc, grid);
    run<1, 1, 2, 1, 0, 1>("look1 bar 2x1 dma1 (as is)", out, cyc, grid, ws);
    run<1, 1, 2, 1, 0, 2>("look1 bar 2x1 dma2 (saddr)", out, cyc, grid, ws);
    run<1, 1, 2, 1, 0, 3>("look1 bar 2x1 dma3 (spread)", out, cyc, grid, ws);
    run<1, 1, 2, 1, 0, 4>("look1 bar 2x1 dma4 (1,5,9)", out, cyc, grid, ws);
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
