// fp32 forward of initialInference (f o h) and recurrentInference (f o g): one 256-thread workgroup
// per board, the whole board's activations resident in LDS from the first conv to the heads, so the
// only HBM traffic per sample is the parent hidden state in, the child hidden state out and the
// (L2-resident) weights.  This is the exact-precision path (MZ_PRECISION_FP32); the bf16 MFMA path
// in net_mfma.hip is checked against it and against the CPU oracle.
//
// Reference math (platform/.../agent/d_model/djl/blocks/): RepresentationBlock.java:38-40,
// DynamicsBlock.java:40-42, StartResidualBlock.java:53-62, ResidualTower.java:30-40,
// BottleneckResidualBlock.java:55-79, BroadcastResidualBlock.java:56-77, PredictionBlock.java:46-84;
// onnx/.../blocks/BroadcastBlock.java:39-58, ext/RescaleBlockExt.java:49-73, ext/LayerNormExt.java:22-27;
// b_inference/InitialInferenceListTranslator.java:44-112 (softmax, value = valueSpan/2 * v).
#include "net.h"

#include <mutex>
#include <set>
#include <utility>

#include <math.h>

namespace mz {

namespace {

constexpr int NT = 512;   // 8 waves, two per SIMD: the partner wave covers LDS / L2 latency of the MFMA convs
constexpr int NW = NT / 64;
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
// Deterministic block reductions (fixed tree). `red` holds NW floats. Ends with a barrier.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) r += red[w];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r = fmaxf(r, red[w]);
    return r;
}
__device__ __forceinline__ float block_min(float v, float* red) {
    v = wave_min_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r = fminf(r, red[w]);
    return r;
}

// LayerNorm over all n = channels*HW elements of one sample (eps 1e-5, affine [c][hw]) + ReLU.
__device__ __forceinline__ void ln_relu(const float* in, float* out, int n, const float* __restrict__ gamma,
                        const float* __restrict__ beta, float* red) {
    // Fast path (every tower LayerNorm: n = channels * HW with channels % 4 == 0, at most LNI float4 items per thread): the
    // thread's items are read once and stay in registers through both statistics passes and the normalisation, and its
    // gamma/beta entries are requested first, so their L2 round trip runs under the two block reductions.
    constexpr int LNI = 6;
    const bool vec = (n & 3) == 0 && (n >> 2) <= LNI * NT &&
                     ((reinterpret_cast<size_t>(gamma) | reinterpret_cast<size_t>(beta) | reinterpret_cast<size_t>(in) |
                       reinterpret_cast<size_t>(out)) & 15) == 0;
    if (vec) {
        typedef const __attribute__((address_space(1))) f32x4v* g4ptr_t;
        const float4* in4 = reinterpret_cast<const float4*>(in);
        g4ptr_t g4 = (g4ptr_t)gamma;
        g4ptr_t b4 = (g4ptr_t)beta;
        const int n4 = n >> 2;
        float4 v[LNI];
        f32x4v g[LNI], bb[LNI];
#pragma unroll
        for (int j = 0; j < LNI; j++) {
            const int i = threadIdx.x + j * NT;
            const int ic = i < n4 ? i : 0;
            g[j] = g4[ic];
            bb[j] = b4[ic];
            v[j] = in4[ic];
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LNI; j++)
            if (threadIdx.x + j * NT < n4) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mean = block_sum(s, red) / (float)n;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LNI; j++)
            if (threadIdx.x + j * NT < n4) {
                const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        const float var = block_sum(q, red) / (float)n;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int j = 0; j < LNI; j++) {
            const int i = threadIdx.x + j * NT;
            if (i < n4) {
                float4 y;
                y.x = fmaxf((v[j].x - mean) * rstd * g[j].x + bb[j].x, 0.f);
                y.y = fmaxf((v[j].y - mean) * rstd * g[j].y + bb[j].y, 0.f);
                y.z = fmaxf((v[j].z - mean) * rstd * g[j].z + bb[j].z, 0.f);
                y.w = fmaxf((v[j].w - mean) * rstd * g[j].w + bb[j].w, 0.f);
                reinterpret_cast<float4*>(out)[i] = y;
            }
        }
        __syncthreads();
        return;
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) s += in[i];
    const float mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) {
        float dlt = in[i] - mean;
        q += dlt * dlt;
    }
    const float var = block_sum(q, red) / (float)n;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    for (int i = threadIdx.x; i < n; i += NT) {
        float y = (in[i] - mean) * rstd * gamma[i] + beta[i];
        out[i] = y > 0.f ? y : 0.f;
    }
    __syncthreads();
}

// out[co][p] (+)= sum_ci sum_tap in[ci][nb(p,tap)] * Wt[ci][tap][co]   (Wt: co contiguous, padded to 4)
__device__ __forceinline__ void conv_valu(const float* in, float* out, const float* __restrict__ Wt, int Cin, int CoutPad, int Cout,
                          int taps, const short* nb, int HW, bool accumulate) {
    const int nco4 = CoutPad >> 2;
    const int items = nco4 * HW;
    for (int it = threadIdx.x; it < items; it += NT) {
        const int co4 = it / HW;
        const int p = it - co4 * HW;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (taps == 1) {
            const float* w = Wt + co4 * 4;
            for (int ci = 0; ci < Cin; ci++) {
                const float x = in[ci * HW + p];
                const float4 wv = *reinterpret_cast<const float4*>(w + (long long)ci * CoutPad);
                acc.x = fmaf(x, wv.x, acc.x);
                acc.y = fmaf(x, wv.y, acc.y);
                acc.z = fmaf(x, wv.z, acc.z);
                acc.w = fmaf(x, wv.w, acc.w);
            }
        } else {
            int q[9];
#pragma unroll
            for (int t = 0; t < 9; t++) q[t] = nb[t * HW + p];
            const float* w = Wt + co4 * 4;
            for (int ci = 0; ci < Cin; ci++) {
                const float* row = in + ci * HW;
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    if (q[t] >= 0) {
                        const float x = row[q[t]];
                        const float4 wv = *reinterpret_cast<const float4*>(w + (long long)(ci * 9 + t) * CoutPad);
                        acc.x = fmaf(x, wv.x, acc.x);
                        acc.y = fmaf(x, wv.y, acc.y);
                        acc.z = fmaf(x, wv.z, acc.z);
                        acc.w = fmaf(x, wv.w, acc.w);
                    }
                }
            }
        }
        const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int co = co4 * 4 + k;
            if (co < Cout) {
                float* o = out + co * HW + p;
                *o = accumulate ? (*o + r[k]) : r[k];
            }
        }
    }
    __syncthreads();
}

// The same conv on the matrix cores, exact fp32: v_mfma_f32_16x16x4_f32, D[out-ch][cell] += W[out-ch][k] * in[k][cell].
// A wave owns MT adjacent row tiles of 16 output channels and all (<= 6) column tiles of 16 cells; K runs tap-major
// (the tap is wave-uniform, so the neighbour cell of a lane's column is looked up once per tap), four input channels
// per MFMA, one per 16-lane group.  A operands: Wt[ci][tap][co] straight from global/L2 (16 consecutive floats per lane
// group), requested a chunk of CH k-steps ahead; B operands: one ds_read_b32 per (k-step, column tile) from the [ch][HW]
// image in LDS, 0 for off-board neighbours, padded columns and channels beyond Cin.
constexpr int MAXNT = 6;   // column tiles: HW <= 96
extern __shared__ float lds_f32[];   // the kernel's dynamic LDS; the MFMA conv addresses it by offset so that its reads are ds_reads

// A work item = MT adjacent row tiles x NTW column tiles starting at tile nt0 (a wave takes items wave, wave + NW, ...;
// the column tiles are split in two when there are fewer row-tile groups than waves).  zero_off = an LDS word that holds 0.0f: off-board neighbours and padded
// columns read it (address = ci * 0 + zero_off), so no B operand needs a select; channels beyond Cin read channel Cin - 1
// against a zeroed A operand.  Loads are unconditional and their results pass through an empty asm statement before the
// select: hipcc otherwise sinks a load under its condition and puts an EXEC-masked branch and a full s_waitcnt in front of
// every MFMA.
template <int MT, int NTW>
__device__ __forceinline__ void conv_mfma(int in_off, int out_off, const float* __restrict__ Wt, int Cin, int Cout, int taps,
                                          int nb_off, int zero_off, int HW, bool accumulate, int nsplit) {
    constexpr int CH = 16 / MT;                 // k-steps per chunk: 16 A registers in use, 16 in flight
    const short* nb = reinterpret_cast<const short*>(lds_f32 + nb_off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jl = lane & 15, kq = lane >> 4;
    const int nmt = Cout >> 4;
    const int KS = (Cin + 3) >> 2;              // k-steps per tap
    const int nc16 = (Cin + 63) >> 6;           // 4 KiB weight units per (tap, row tile): api.cpp:pack_conv
    const int ngroups = nmt / MT;
    for (int it = wave; it < ngroups * nsplit; it += NW) {
        const int mt0 = (it % ngroups) * MT, nt0 = (it / ngroups) * NTW;
        f32x4v acc[MT][NTW];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) acc[m][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        // (an explicit global pointer: in this out-of-line function Wt is a generic pointer, and a flat_load counts on
        // lgkmcnt as well, so every wait for a B operand would also wait for the A chunk requested ahead)
        typedef const __attribute__((address_space(1))) f32x4v* gptr_t;
        gptr_t wlane = (gptr_t)(Wt) + lane;
        // the chunk's A operands: CH / 4 sixteen-byte loads per row tile, each 1 KiB contiguous over the wave; channels beyond
        // Cin are zeros in the packed units
        auto load_a = [&](int t, int s0, float (&a)[CH][MT]) {
#pragma unroll
            for (int j = 0; j < CH / 4; j++) {
                const int qd = (s0 >> 2) + j;
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const f32x4v v = wlane[(((t * nmt + mt0 + m) * nc16 + (qd >> 2)) * 4 + (qd & 3)) * 64];
#pragma unroll
                    for (int e = 0; e < 4; e++) a[4 * j + e][m] = v[e];
                }
            }
        };
        float a0[CH][MT], a1[CH][MT];
        load_a(0, 0, a0);
        int qoff[NTW], hwm[NTW];
        int t = 0, s0 = 0;
        // one chunk: request the next chunk's A operands into `an`, run this chunk's MFMAs from `a`; returns false after the last
        // (a third set, two chunks in flight, measured 7 % slower)
        auto chunk = [&](float (&a)[CH][MT], float (&an)[CH][MT]) -> bool {
            int tn = t, sn = s0 + CH;
            if (sn >= KS) {
                sn = 0;
                tn = t + 1;
            }
            const bool more = tn < taps;
            if (more) load_a(tn, sn, an);
            if (s0 == 0) {
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) {
                    const int p = 16 * (nt0 + nt) + jl;
                    const int q = p < HW ? (taps == 1 ? p : (int)nb[t * HW + p]) : -1;
                    qoff[nt] = 4 * (q >= 0 ? in_off + q : zero_off);   // byte address of channel 0
                    hwm[nt] = q >= 0 ? 4 * HW : 0;                     // bytes per channel
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < CH; g4 += 4) {          // groups of four k-steps; a group beyond the tap's last step is skipped
                if (s0 + g4 < KS) {
                    float bv[4][NTW];
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        int ci = 4 * (s0 + g4 + s) + kq;
                        ci = ci < Cin ? ci : Cin - 1;
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++)   // v_mad_u32_u24: one full-rate instruction per address
                            bv[s][nt] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds_f32) +
                                                                        (__umul24(ci, hwm[nt]) + qoff[nt]));
                    }
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                            for (int m = 0; m < MT; m++)
                                acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g4 + s][m], bv[s][nt], acc[m][nt], 0, 0, 0);
                }
            }
            if (!more) return false;
            t = tn;
            s0 = sn;
            return true;
        };
        while (chunk(a0, a1) && chunk(a1, a0)) {
        }
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
                if (16 * (nt0 + nt) + jl < HW) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float* o = lds_f32 + out_off + (16 * (mt0 + m) + 4 * kq + r) * HW + 16 * (nt0 + nt) + jl;
                        *o = accumulate ? (*o + acc[m][nt][r]) : acc[m][nt][r];
                    }
                }
    }
    __syncthreads();
}

template <int MT>
__device__ __forceinline__ void conv_mfma_nt(int in_off, int out_off, const float* __restrict__ Wt, int Cin, int Cout, int taps,
                                             int nb_off, int zero_off, int HW, bool accumulate) {
    const int ntl = (HW + 15) >> 4;
    const int nsplit = (ntl >= 2 && (Cout >> 4) / MT < NW) ? 2 : 1;
    switch ((ntl + nsplit - 1) / nsplit) {
        case 1: conv_mfma<MT, 1>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, zero_off, HW, accumulate, nsplit); break;
        case 2: conv_mfma<MT, 2>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, zero_off, HW, accumulate, nsplit); break;
        case 3: conv_mfma<MT, 3>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, zero_off, HW, accumulate, nsplit); break;
        case 4: conv_mfma<MT, 4>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, zero_off, HW, accumulate, nsplit); break;
        case 5: conv_mfma<MT, 5>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, zero_off, HW, accumulate, nsplit); break;
        default: conv_mfma<MT, 6>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, zero_off, HW, accumulate, nsplit); break;
    }
}

// Convs whose output channels fill whole 16-row tiles go to the matrix cores; the rest (the head convs with 1 and 2
// output channels, toy networks) stay on the vector ALU.
__device__ __noinline__ void conv_lds(const float* in, float* out, const float* __restrict__ Wt, int Cin, int CoutPad, int Cout, int taps,
                         const short* nb, int HW, bool accumulate) {
    if (fp32_conv_on_mfma(Cout, CoutPad, HW)) {
        const int io = (int)(in - lds_f32), oo = (int)(out - lds_f32);
        const int no = (int)(reinterpret_cast<const float*>(nb) - lds_f32);
        const int zo = no + (9 * HW + 1) / 2;   // the zero word behind the neighbour table (k_net_fp32 clears it)
        if ((Cout & 31) == 0 && Cout >= 128) conv_mfma_nt<2>(io, oo, Wt, Cin, Cout, taps, no, zo, HW, accumulate);
        else conv_mfma_nt<1>(io, oo, Wt, Cin, Cout, taps, no, zo, HW, accumulate);
    } else {
        conv_valu(in, out, Wt, Cin, CoutPad, Cout, taps, nb, HW, accumulate);
    }
}

// BroadcastBlock: per channel, out[c][q] = sum_p in[c][p] * Wt[p][q] + b[q]
__device__ __forceinline__ void broadcast_lds(const float* in, float* out, const float* __restrict__ Wt, const float* __restrict__ b,
                              int C, int HW) {
    for (int it = threadIdx.x; it < C * HW; it += NT) {
        const int c = it / HW;
        const int q = it - c * HW;
        float acc = b[q];
        const float* row = in + c * HW;
        for (int p = 0; p < HW; p++) acc = fmaf(row[p], Wt[p * HW + q], acc);
        out[it] = acc;
    }
    __syncthreads();
}

#if defined(MZ_FP32_STAMPS)
#define TC(x) { long long t_ = __builtin_amdgcn_s_memtime(); x; tconv += __builtin_amdgcn_s_memtime() - t_; }
#define TL(x) { long long t_ = __builtin_amdgcn_s_memtime(); x; tln += __builtin_amdgcn_s_memtime() - t_; }
#define TB(x) { long long t_ = __builtin_amdgcn_s_memtime(); x; tbc += __builtin_amdgcn_s_memtime() - t_; }
#else
#define TC(x) x
#define TL(x) x
#define TB(x) x
#endif
// BroadcastBlock on the matrix cores: D[q][c] = sum_p Wt[p][q] * in[c][p] (+ b[q]): row tiles over the output cells q, column
// tiles over the channels (a wave takes channel tiles wave, wave + NW, ...), four cells p per MFMA.
template <int NTQ>
__device__ __forceinline__ void broadcast_mfma(int in_off, int out_off, const float* __restrict__ Wt, const float* __restrict__ b,
                                               int C, int HW) {
    typedef const __attribute__((address_space(1))) float* gptr_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jl = lane & 15, kq = lane >> 4;
    const int KS = (HW + 3) >> 2;
    gptr_t wg = (gptr_t)Wt;
    for (int ct = wave; ct < (C >> 4); ct += NW) {
        f32x4v acc[NTQ];
#pragma unroll
        for (int mt = 0; mt < NTQ; mt++) acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        const int crow = in_off + (16 * ct + jl) * HW;
        for (int ks = 0; ks < KS; ks++) {
            const int p = 4 * ks + kq;
            const bool pok = p < HW;
            const int pc = pok ? p : HW - 1;
            float bvl = lds_f32[crow + pc];
            float av[NTQ];
#pragma unroll
            for (int mt = 0; mt < NTQ; mt++) {
                const int q = 16 * mt + jl;
                av[mt] = wg[pc * HW + (q < HW ? q : HW - 1)];
            }
            asm volatile("" : "+v"(bvl));
            bvl = pok ? bvl : 0.f;
#pragma unroll
            for (int mt = 0; mt < NTQ; mt++) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bvl, acc[mt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < NTQ; mt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int q = 16 * mt + 4 * kq + r;
                if (q < HW) lds_f32[out_off + (16 * ct + jl) * HW + q] = acc[mt][r] + b[q];
            }
    }
    __syncthreads();
}
__device__ __noinline__ void broadcast_any(const float* in, float* out, const float* __restrict__ Wt, const float* __restrict__ b,
                                           int C, int HW) {
    const int io = (int)(in - lds_f32), oo = (int)(out - lds_f32);
    if ((C & 15) == 0 && HW <= 16 * MAXNT) {
        switch ((HW + 15) >> 4) {
            case 1: broadcast_mfma<1>(io, oo, Wt, b, C, HW); break;
            case 2: broadcast_mfma<2>(io, oo, Wt, b, C, HW); break;
            case 3: broadcast_mfma<3>(io, oo, Wt, b, C, HW); break;
            case 4: broadcast_mfma<4>(io, oo, Wt, b, C, HW); break;
            case 5: broadcast_mfma<5>(io, oo, Wt, b, C, HW); break;
            default: broadcast_mfma<6>(io, oo, Wt, b, C, HW); break;
        }
    } else {
        broadcast_lds(in, out, Wt, b, C, HW);
    }
}

template <bool INITIAL>
__global__ __launch_bounds__(NT) void k_net_fp32(NetDims d, const NetProgramF32* __restrict__ prog,
                                                 const float* __restrict__ wb, NetIO io, int buf_floats) {
    float* smem = lds_f32;
    const int g = blockIdx.x;
#if defined(MZ_FP32_STAMPS)
    long long tconv = 0, tln = 0, tbc = 0;
    const long long tstart = __builtin_amdgcn_s_memtime();
#endif
    if (io.active && !io.active[g]) return;
    const int HW = d.HW, C = d.C, Cb = d.Cb, A = d.A;
    float* X = smem;
    float* T1 = X + buf_floats;
    float* T2 = T1 + buf_floats;
    float* red = T2 + buf_floats;
    short* nb = reinterpret_cast<short*>(red + 8);
    const int tid = threadIdx.x;
    const int c_pad = prog->c_pad, cb_pad = prog->cb_pad;

    for (int i = tid; i < 9 * HW; i += NT) {
        const int t = i / HW, p = i - t * HW;
        const int y = p / d.Wd + t / 3 - 1, x = p % d.Wd + t % 3 - 1;
        nb[i] = (y >= 0 && y < d.H && x >= 0 && x < d.Wd) ? (short)(y * d.Wd + x) : (short)-1;
    }
    if (tid == 0) reinterpret_cast<float*>(nb)[(9 * HW + 1) / 2] = 0.f;   // what off-board taps of the MFMA convs read
    const TowerOff& tw = INITIAL ? prog->rep : prog->dyn;
    if (INITIAL) {
        const float* src = io.obs + (long long)g * d.O * HW;
        for (int i = tid; i < d.O * HW; i += NT) T1[i] = src[i];
    } else {
        const float* src = io.pool + (long long)io.src_slot[g] * C * HW;
        for (int i = tid; i < C * HW; i += NT) T1[i] = src[i];
        const int cell = io.action_cells[io.action[g]];
        for (int i = tid; i < d.a * HW; i += NT) T1[C * HW + i] = (i == cell) ? 1.f : 0.f;  // ConcatInputsBlock
    }
    __syncthreads();
    TC(conv_lds(T1, X, wb + tw.conv0, tw.cin0, c_pad, C, 9, nb, HW, false);)
    if (!INITIAL) {  // StartResidualBlock: conv3x3(cat[s, a]) + s
        for (int i = tid; i < C * HW; i += NT) X[i] += T1[i];
        __syncthreads();
    }
    for (int r = 0; r < d.R; r++) {
        const BlockOff& b = tw.blocks[r];
        if (b.type == 0) {
            TL(ln_relu(X, T1, C * HW, wb + b.ln1g, wb + b.ln1b, red);)
            TC(conv_lds(T1, T2, wb + b.c1, C, cb_pad, Cb, 1, nb, HW, false);)
            TL(ln_relu(T2, T2, Cb * HW, wb + b.ln2g, wb + b.ln2b, red);)
            TC(conv_lds(T2, T1, wb + b.c2, Cb, cb_pad, Cb, 9, nb, HW, false);)
            TL(ln_relu(T1, T1, Cb * HW, wb + b.ln3g, wb + b.ln3b, red);)
            TC(conv_lds(T1, T2, wb + b.c3, Cb, cb_pad, Cb, 9, nb, HW, false);)
            TL(ln_relu(T2, T2, Cb * HW, wb + b.ln4g, wb + b.ln4b, red);)
            TC(conv_lds(T2, X, wb + b.c4, Cb, c_pad, C, 1, nb, HW, true);)
        } else {
            TL(ln_relu(X, T1, C * HW, wb + b.ln1g, wb + b.ln1b, red);)
            TC(conv_lds(T1, T2, wb + b.c1, C, c_pad, C, 1, nb, HW, false);)
            TL(ln_relu(T2, T2, C * HW, wb + b.ln2g, wb + b.ln2b, red);)
            TB(broadcast_any(T2, T1, wb + b.bw, wb + b.bb, C, HW);)
            TL(ln_relu(T1, T1, C * HW, wb + b.ln3g, wb + b.ln3b, red);)
            TC(conv_lds(T1, X, wb + b.c2, C, c_pad, C, 1, nb, HW, true);)
        }
    }
    // RescaleBlockExt: (x - min) / max(max - min, 1e-5) over the whole sample
    {
        float mn = INFINITY, mx = -INFINITY;
        for (int i = tid; i < C * HW; i += NT) {
            mn = fminf(mn, X[i]);
            mx = fmaxf(mx, X[i]);
        }
        mn = block_min(mn, red);
        mx = block_max(mx, red);
        const float dd = fmaxf(mx - mn, 1e-5f);
        float* dst = io.pool + (long long)io.dst_slot[g] * C * HW;
        for (int i = tid; i < C * HW; i += NT) {
            const float y = (X[i] - mn) / dd;
            X[i] = y;
            dst[i] = y;
        }
        __syncthreads();
    }
    // PredictionBlock
    const HeadOff& h = prog->head;
    TC(conv_lds(X, T1, wb + h.pconv, C, 4, 2, 1, nb, HW, false);)
    TL(ln_relu(T1, T1, 2 * HW, wb + h.plng, wb + h.plnb, red);)
    float* lgt = T2;  // logits staged in LDS for the softmax
    for (int a = tid; a < A; a += NT) {
        float acc = wb[h.pfcb + a];
        for (int i = 0; i < 2 * HW; i++) acc = fmaf(T1[i], wb[h.pfcw + (long long)i * A + a], acc);
        lgt[a] = acc;
    }
    __syncthreads();
    {
        float mx = -INFINITY;
        for (int a = tid; a < A; a += NT) mx = fmaxf(mx, lgt[a]);
        mx = block_max(mx, red);
        float s = 0.f;
        for (int a = tid; a < A; a += NT) s += expf(lgt[a] - mx);
        s = block_sum(s, red);
        for (int a = tid; a < A; a += NT) {
            io.logits[(long long)g * A + a] = lgt[a];
            io.policy[(long long)g * A + a] = expf(lgt[a] - mx) / s;
        }
        __syncthreads();
    }
    TC(conv_lds(X, T1, wb + h.vconv, C, 4, 1, 1, nb, HW, false);)
    TL(ln_relu(T1, T1, HW, wb + h.vlng, wb + h.vlnb, red);)
    float part = 0.f;
    for (int c = tid; c < C; c += NT) {
        float acc = wb[h.vfc1b + c];
        for (int p = 0; p < HW; p++) acc = fmaf(T1[p], wb[h.vfc1w + (long long)p * C + c], acc);
        acc = acc > 0.f ? acc : 0.f;
        part = fmaf(acc, wb[h.vfc2w + c], part);
    }
    float v = block_sum(part, red) + wb[h.vfc2b];
    if (d.two_players) v = tanhf(v);
    if (tid == 0) io.value[g] = d.value_scale * (double)v;
#if defined(MZ_FP32_STAMPS)
    if (g == 0 && tid == 0)
        printf("fp32 board: total %lld conv %lld ln %lld bcast %lld\n", (long long)__builtin_amdgcn_s_memtime() - tstart, tconv, tln, tbc);
#endif
}

}  // namespace

size_t fp32_buf_floats(const NetDims& d) {
    int extra = d.a > d.O ? d.a : d.O;
    size_t n = (size_t)(d.C + extra) * d.HW;
    size_t head = (size_t)(d.A > 2 * d.HW ? d.A : 2 * d.HW);
    return ((n > head ? n : head) + 3) & ~(size_t)3;   // 16-byte aligned buffers (float4 LayerNorm passes)
}

size_t fp32_lds_bytes(const NetDims& d) {
    return (3 * fp32_buf_floats(d) + 8) * sizeof(float) + (size_t)9 * d.HW * sizeof(short) + 16;
}

namespace {
std::mutex g_optin_mu;
std::set<std::pair<const void*, int>> g_optin_done;
thread_local hipError_t g_optin_err = hipSuccess;
}  // namespace

void lds_optin(const void* fn) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_optin_mu);
        if (g_optin_done.count({fn, dev})) return;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) g_optin_done.insert({fn, dev});
    }
    if (e != hipSuccess) g_optin_err = e;
}
hipError_t take_optin_error() {
    const hipError_t e = g_optin_err;
    g_optin_err = hipSuccess;
    return e;
}

hipError_t launch_net_fp32(hipStream_t st, bool initial, const NetDims& d, const NetProgramF32* prog_dev, const float* wbuf,
                           const NetIO& io) {
    const size_t lds = fp32_lds_bytes(d);
    const int bf = (int)fp32_buf_floats(d);
    if (initial) {
        lds_optin(reinterpret_cast<const void*>(&k_net_fp32<true>));
        hipLaunchKernelGGL(k_net_fp32<true>, dim3(io.B), dim3(NT), lds, st, d, prog_dev, wbuf, io, bf);
    } else {
        lds_optin(reinterpret_cast<const void*>(&k_net_fp32<false>));
        hipLaunchKernelGGL(k_net_fp32<false>, dim3(io.B), dim3(NT), lds, st, d, prog_dev, wbuf, io, bf);
    }
    return take_optin_error();
}

}  // namespace mz
