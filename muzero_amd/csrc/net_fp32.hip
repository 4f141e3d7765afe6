// fp32 forward of initialInference (f o h) and recurrentInference (f o g): one 512-thread workgroup
// per board, the whole board's activations resident in LDS from the first conv to the heads, so the
// only HBM traffic per sample is the parent hidden state in, the child hidden state out and the
// (mostly L2-resident) weights.  This is the exact-precision path (MZ_PRECISION_FP32: fp32 operands,
// v_mfma_f32_16x16x4_f32 convs); the 16-bit MFMA paths in net_mfma.hip are checked against it and
// against the CPU oracle.  DESIGN.md §5 describes the layout (Q16 order with a zero cell per plane)
// and what was measured.
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

#if defined(MZ_X_NT)   // experiment: waves per workgroup
constexpr int NT = MZ_X_NT;
#else
constexpr int NT = 512;   // 8 waves, two per SIMD: the partner wave covers LDS / L2 latency of the MFMA convs
#endif
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

// LayerNorm over all channels*HW elements of one sample (eps 1e-5, affine [c][hw]) + ReLU.  n = the elements stored; hwp > 0:
// a Q16 image (net.h) whose planes carry a zero cell behind the board (cell hwp - 1 of every hwp): those cells hold 0 and
// have gamma = beta = 0, so they add nothing to the sum and come out as 0, but they are not samples: `count` = the real
// element count, and the centred squares skip them (padmask: bit j = this thread's j-th float4 of the fast path is one).
// presum != nullptr: the conv that wrote `in` left the NW per-wave sums of what it stored there (conv_mfma), published by
// its closing barrier: the mean comes from those and the first block reduction (two barriers) is not run.
__device__ __forceinline__ void ln_relu(const float* in, float* out, int n, const float* __restrict__ gamma,
                        const float* __restrict__ beta, float* red, int count = 0, int hwp = 0, unsigned padmask = 0,
                        const float* presum = nullptr) {
    if (count == 0) count = n;
    // Fast path (every tower LayerNorm: n = channels * HW with channels % 4 == 0, at most LNI float4 items per thread): the
    // thread's items are read once and stay in registers through both statistics passes and the normalisation, and its
    // gamma/beta entries are requested first, so their L2 round trip runs under the two block reductions.
    constexpr int LNI = 3072 / NT;
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
#if defined(MZ_X_NOGB)   // experiment: no gamma / beta traffic
            g[j] = (f32x4v){1.f, 1.f, 1.f, 1.f};
            bb[j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#else
            g[j] = g4[ic];
            bb[j] = b4[ic];
#endif
            v[j] = in4[ic];
        }
        float mean;
        if (presum) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w++) t += presum[w];
            mean = t / (float)count;
        } else {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < LNI; j++)
                if (threadIdx.x + j * NT < n4) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            mean = block_sum(s, red) / (float)count;
        }
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LNI; j++)
            if (threadIdx.x + j * NT < n4 && !((padmask >> j) & 1)) {
                const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        const float var = block_sum(q, red) / (float)count;
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
    const float mean = block_sum(s, red) / (float)count;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += NT) {
        float dlt = in[i] - mean;
        if (hwp > 0 && (i >> 2) % hwp == hwp - 1) dlt = 0.f;
        q += dlt * dlt;
    }
    const float var = block_sum(q, red) / (float)count;
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
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    if (q[t] >= 0) {
                        const float x = in[ci * HW + q[t]];
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
// A wave owns MT adjacent row tiles of 16 output channels and NTW (<= 6) column tiles of 16 cells; K runs tap-major
// (the tap is wave-uniform, so the neighbour cell of a lane's column is looked up once per tap), four input channels
// per MFMA, one per 16-lane group.  A operands: 4 KiB units of api.cpp:pack_conv straight from global/L2 (a lane's operands of
// four k-steps are one 16-byte load), requested a chunk of CH k-steps ahead; B operands: the image is in Q16 order (net.h),
// so a lane's operands of the four k-steps of a group of 16 input channels are ONE ds_read_b128.  Every plane of the image
// carries a zero cell behind the board (planes of HWp = HW + 1 cells): the neighbour table names it for off-board taps and
// for the padded columns of the last tile, so a B address is base + group * stride with no select anywhere - the loop has
// ONE MFMA (32 cycles) of slack per wave, and every vector instruction between two MFMAs beyond that idles the matrix pipe
// (tools/micro/f32_chunk.hip).  Input and output images are both Q16; Cin counts in whole groups of 16 (the caller
// zero-fills the image up to there, the packed A operands are zero beyond Cin anyway).
constexpr int MAXNT = 6;   // column tiles: HW <= 96
extern __shared__ float lds_f32[];   // the kernel's dynamic LDS; the MFMA conv addresses it by offset so that its reads are ds_reads

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// A work item = MT adjacent row tiles x NTW column tiles starting at tile nt0 (a wave takes items wave, wave + NW, ...;
// the column tiles are split when there are fewer row-tile groups than waves).  nb = the neighbour table [tap][NP] of
// k_net_fp32 (NP = 16 column tiles' worth of columns; row 4, the centre tap, serves the 1x1 convs).
// readfirstlane keeps the wave-uniform arguments in SGPRs whatever the inliner does, so that the loop control is scalar
// branches and not EXEC masks.
template <int MT, int NTW>
__device__ __forceinline__ void conv_mfma(int in_off, int out_off, const float* __restrict__ Wt, int Cin, int Cout, int taps,
                                          int nb_off, int NP, int HW, bool accumulate, int nsplit, int csum_off) {
    constexpr int CH = 16 / MT;                 // k-steps per chunk: 16 A registers in use, 16 in flight
    constexpr int NG = CH / 4;                  // groups of 16 input channels per chunk
    in_off = uni(in_off), out_off = uni(out_off), Cin = uni(Cin), Cout = uni(Cout), taps = uni(taps), nb_off = uni(nb_off);
    NP = uni(NP), HW = uni(HW), nsplit = uni(nsplit);
    const int HWp = HW + 1, gstride = 64 * HWp;   // cells per plane, bytes per group of 16 channels
    accumulate = uni((int)accumulate) != 0;
    {
        const unsigned long long w = reinterpret_cast<unsigned long long>(Wt);
        Wt = reinterpret_cast<const float*>(((unsigned long long)(unsigned)uni((int)(w >> 32)) << 32) | (unsigned)uni((int)w));
    }
    const short* nb = reinterpret_cast<const short*>(lds_f32 + nb_off);
    const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
    const int jl = lane & 15, kq = lane >> 4;
    const int nmt = Cout >> 4;
    const int KS = ((Cin + 15) >> 4) << 2;      // k-steps per tap: whole groups
    const int nc16 = (Cin + 63) >> 6;           // 4 KiB weight units per (tap, row tile): api.cpp:pack_conv
    const int ngroups = nmt / MT;
    float wsum = 0.f;   // sum of everything this wave stores: the next LayerNorm's mean (ln_relu presum)
    for (int it = wave; it < ngroups * nsplit; it += NW) {
        const int mt0 = (it % ngroups) * MT, nt0 = (it / ngroups) * NTW;
        f32x4v acc[MT][NTW];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++) acc[m][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        // (an explicit global pointer: a flat_load counts on lgkmcnt as well, so every wait for a B operand would also
        // wait for the A chunk requested ahead)
        typedef const __attribute__((address_space(1))) f32x4v* gptr_t;
        gptr_t wlane = (gptr_t)(Wt) + lane;
        // the chunk's A operands: CH / 4 sixteen-byte loads per row tile, each 1 KiB contiguous over the wave; channels beyond
        // Cin are zeros in the packed units
        auto load_a = [&](int t, int s0, float (&a)[CH][MT]) {
#pragma unroll
            for (int j = 0; j < NG; j++) {
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
        int base[NTW];              // byte address of the lane's 16 bytes of group 0
        const int cbase = 4 * in_off + 16 * kq * HWp;
        int t = 0, s0 = 0;
        auto read_b = [&](int G, int nt) -> f32x4v {   // s_mul + v_add + ds_read_b128
#if defined(MZ_X_NOB)   // experiment: no B traffic
            const float z = (float)(G + base[nt]);
            return (f32x4v){z, z, z, z};
#else
            return *reinterpret_cast<const f32x4v*>(reinterpret_cast<const char*>(lds_f32) + (G * gstride + base[nt]));
#endif
        };
        // one chunk: request the next chunk's A operands into `an`, run this chunk's MFMAs from `a`; returns false after the last
        // (a third set, two chunks in flight, measured 7 % slower)
        auto chunk = [&](float (&a)[CH][MT], float (&an)[CH][MT]) -> bool {
            int tn = t, sn = s0 + CH;
            if (sn >= KS) {
                sn = 0;
                tn = t + 1;
            }
            const bool more = tn < taps;
#if defined(MZ_X_NOA)   // experiment: no A traffic after the first chunk
            if (more && false) load_a(tn, sn, an);
#else
            if (more) load_a(tn, sn, an);
#endif
            if (s0 == 0) {
                const int trow = (taps == 1 ? 4 : t) * NP + 16 * nt0 + jl;
#pragma unroll
                for (int nt = 0; nt < NTW; nt++) base[nt] = cbase + 16 * (int)nb[trow + 16 * nt];
            }
            const int G0 = s0 >> 2;
            if (s0 + CH <= KS) {                          // a whole chunk: every read ahead of the first MFMA
                f32x4v bv[NG][NTW];
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int nt = 0; nt < NTW; nt++) bv[g][nt] = read_b(G0 + g, nt);
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                            for (int m = 0; m < MT; m++)
                                acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * g + s][m], bv[g][nt][s], acc[m][nt], 0, 0, 0);
            } else {                                       // the tap's last chunk ends early: groups beyond KS are skipped
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    if (s0 + 4 * g < KS) {
                        f32x4v bv[NTW];
#pragma unroll
                        for (int nt = 0; nt < NTW; nt++) bv[nt] = read_b(G0 + g, nt);
#pragma unroll
                        for (int s = 0; s < 4; s++)
#pragma unroll
                            for (int nt = 0; nt < NTW; nt++)
#pragma unroll
                                for (int m = 0; m < MT; m++)
                                    acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4 * g + s][m], bv[nt][s], acc[m][nt], 0, 0, 0);
                    }
                }
            }
            if (!more) return false;
            t = tn;
            s0 = sn;
            return true;
        };
        while (chunk(a0, a1) && chunk(a1, a0)) {
        }
        // D[16 (mt0 + m) + 4 kq + r][cell] -> Q16: the wave's 64 lanes write 64 consecutive words per (m, nt, r)
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int nt = 0; nt < NTW; nt++)
                if (16 * (nt0 + nt) + jl < HW) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float* o = lds_f32 + out_off + (((mt0 + m) * 4 + r) * HWp + 16 * (nt0 + nt) + jl) * 4 + kq;
                        const float val = accumulate ? (*o + acc[m][nt][r]) : acc[m][nt][r];
                        *o = val;
                        wsum += val;
                    }
                }
    }
    wsum = wave_sum_f(wsum);
    if (lane == 0) lds_f32[csum_off + wave] = wsum;
    __syncthreads();
}

template <int MT>
__device__ __forceinline__ void conv_mfma_nt(int in_off, int out_off, const float* __restrict__ Wt, int Cin, int Cout, int taps,
                                             int nb_off, int NP, int HW, bool accumulate, int csum_off) {
    const int ntl = (HW + 15) >> 4;
    const int ngr = (Cout >> 4) / MT;
    int nsplit = (NW + ngr - 1) / ngr;            // enough work items for every wave
    nsplit = nsplit < 1 ? 1 : (nsplit > ntl ? ntl : nsplit);
    switch ((ntl + nsplit - 1) / nsplit) {
        case 1: conv_mfma<MT, 1>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, NP, HW, accumulate, nsplit, csum_off); break;
        case 2: conv_mfma<MT, 2>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, NP, HW, accumulate, nsplit, csum_off); break;
        case 3: conv_mfma<MT, 3>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, NP, HW, accumulate, nsplit, csum_off); break;
        case 4: conv_mfma<MT, 4>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, NP, HW, accumulate, nsplit, csum_off); break;
        case 5: conv_mfma<MT, 5>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, NP, HW, accumulate, nsplit, csum_off); break;
        default: conv_mfma<MT, 6>(in_off, out_off, Wt, Cin, Cout, taps, nb_off, NP, HW, accumulate, nsplit, csum_off); break;
    }
}

// The convs of a Q16 network (net.h) whose output channels fill whole 16-row tiles go to the matrix cores; the rest (the
// head convs with 1 and 2 output channels, toy networks) stay on the vector ALU.
__device__ __forceinline__ void conv_lds(const float* in, float* out, const float* __restrict__ Wt, int Cin, int CoutPad, int Cout, int taps,
                         const short* nb, int HW, bool accumulate, bool q16, const float* csum) {
    if (fp32_conv_on_mfma(q16, Cout)) {
        const int co = (int)(csum - lds_f32);
        const int io = (int)(in - lds_f32), oo = (int)(out - lds_f32);
        const int no = (int)(reinterpret_cast<const float*>(nb) - lds_f32);
        const int NP = ((HW + 15) >> 4) << 4;
        // Two row tiles per wave halve the B reads per MFMA, more column tiles per wave the (dearer) A loads: two row tiles
        // only where that still leaves a wave three column tiles, or as many as one row tile would (Go 5x5, 256 channels on
        // 2 column tiles: 15.5 -> 13.6 ms per 4096 boards with one row tile x 2 column tiles instead of 2 x 1).
        const int ntl = NP >> 4, nmt = Cout >> 4;
        auto ntw_of = [&](int mt) {
            const int ngr = nmt / mt;
            int ns = (NW + ngr - 1) / ngr;
            ns = ns < 1 ? 1 : (ns > ntl ? ntl : ns);
            return (ntl + ns - 1) / ns;
        };
        const int n1 = ntw_of(1);
        const bool two = (Cout & 31) == 0 && Cout >= 128 && ntw_of(2) >= (n1 < 3 ? n1 : 3);
        if (two) conv_mfma_nt<2>(io, oo, Wt, Cin, Cout, taps, no, NP, HW, accumulate, co);
        else conv_mfma_nt<1>(io, oo, Wt, Cin, Cout, taps, no, NP, HW, accumulate, co);
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
#define TC(...) { long long t_ = __builtin_amdgcn_s_memtime(); __VA_ARGS__; tconv += __builtin_amdgcn_s_memtime() - t_; }
#define TL(...) { long long t_ = __builtin_amdgcn_s_memtime(); __VA_ARGS__; tln += __builtin_amdgcn_s_memtime() - t_; }
#define TB(...) { long long t_ = __builtin_amdgcn_s_memtime(); __VA_ARGS__; tbc += __builtin_amdgcn_s_memtime() - t_; }
#define TH(...) { long long t_ = __builtin_amdgcn_s_memtime(); __VA_ARGS__; thd += __builtin_amdgcn_s_memtime() - t_; }
#else
#define TC(...) __VA_ARGS__
#define TL(...) __VA_ARGS__
#define TB(...) __VA_ARGS__
#define TH(...) __VA_ARGS__
#endif
// BroadcastBlock on the matrix cores: D[q][c] = sum_p Wt[p][q] * in[c][p] (+ b[q]): row tiles over the output cells q, column
// tiles over the channels (a wave takes channel tiles wave, wave + NW, ...), four cells p per MFMA.
template <int NTQ>
__device__ __forceinline__ void broadcast_mfma(int in_off, int out_off, const float* __restrict__ Wt, const float* __restrict__ b,
                                               int C, int HW, bool q16) {
    typedef const __attribute__((address_space(1))) float* gptr_t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jl = lane & 15, kq = lane >> 4;
    const int KS = (HW + 3) >> 2;
    gptr_t wg = (gptr_t)Wt;
    for (int ct = wave; ct < (C >> 4); ct += NW) {
        f32x4v acc[NTQ];
#pragma unroll
        for (int mt = 0; mt < NTQ; mt++) acc[mt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        const int ch = 16 * ct + jl;
        for (int ks = 0; ks < KS; ks++) {
            const int p = 4 * ks + kq;
            const bool pok = p < HW;
            const int pc = pok ? p : HW - 1;
            float bvl = lds_f32[in_off + (q16 ? fp32_q16_index(ch, pc, HW + 1) : ch * HW + pc)];
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
                if (q < HW) lds_f32[out_off + (q16 ? fp32_q16_index(ch, q, HW + 1) : ch * HW + q)] = acc[mt][r] + b[q];
            }
    }
    __syncthreads();
}
__device__ __forceinline__ void broadcast_any(const float* in, float* out, const float* __restrict__ Wt, const float* __restrict__ b,
                                           int C, int HW, bool q16) {
    const int io = (int)(in - lds_f32), oo = (int)(out - lds_f32);
    if (q16 || ((C & 15) == 0 && HW <= 16 * MAXNT)) {
        switch ((HW + 15) >> 4) {
            case 1: broadcast_mfma<1>(io, oo, Wt, b, C, HW, q16); break;
            case 2: broadcast_mfma<2>(io, oo, Wt, b, C, HW, q16); break;
            case 3: broadcast_mfma<3>(io, oo, Wt, b, C, HW, q16); break;
            case 4: broadcast_mfma<4>(io, oo, Wt, b, C, HW, q16); break;
            case 5: broadcast_mfma<5>(io, oo, Wt, b, C, HW, q16); break;
            default: broadcast_mfma<6>(io, oo, Wt, b, C, HW, q16); break;
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
    long long tconv = 0, tln = 0, tbc = 0, thd = 0, tmark = 0, tstage = 0, trescale = 0, tfc = 0;
    const long long tstart = __builtin_amdgcn_s_memtime();
#endif
    if (io.active && !io.active[g]) return;
    const int HW = d.HW, C = d.C, Cb = d.Cb, A = d.A;
    float* X = smem;
    float* T1 = X + buf_floats;
    float* T2 = T1 + buf_floats;
    float* red = T2 + buf_floats;
    float* csum = red + 8;   // per-wave sums of the last MFMA conv's output (conv_mfma -> ln_relu)
    short* nb = reinterpret_cast<short*>(red + 16);
    const int tid = threadIdx.x;
    const int c_pad = prog->c_pad, cb_pad = prog->cb_pad;
    const bool q16 = fp32_q16(C, Cb, HW);   // activations, hidden-state slots and gamma / beta in Q16 order (net.h)

    // Q16 images: planes of HWp = HW + 1 cells, the last one the zero cell (conv_mfma); the neighbour table [tap][NP] names it
    // for off-board taps and for the columns that pad the last tile.  Otherwise [tap][HW] with -1 for off-board (conv_valu).
    const int HWp = q16 ? HW + 1 : HW;
    const int NP = q16 ? ((HW + 15) >> 4) << 4 : HW;
    for (int i = tid; i < 9 * NP; i += NT) {
        const int t = i / NP, p = i - t * NP;
        const int y = p / d.Wd + t / 3 - 1, x = p % d.Wd + t % 3 - 1;
        const bool on = p < HW && y >= 0 && y < d.H && x >= 0 && x < d.Wd;
        nb[i] = on ? (short)(y * d.Wd + x) : (short)(q16 ? HW : -1);
    }
    unsigned padmask = 0;   // ln_relu: which of this thread's float4 items are zero cells (item j = float4 tid + j NT of a buffer)
    if (q16) {
        for (int j = 0; j < 3072 / NT; j++) padmask |= ((tid + j * NT) % HWp == HW ? 1u : 0u) << j;
        const int planes = (buf_floats >> 2) / HWp;   // the zero cells of X, T1, T2: nothing below writes anything but 0 there
        for (int i = tid; i < 3 * planes; i += NT) {
            const int bf = i / planes, pl = i - bf * planes;
            reinterpret_cast<float4*>(smem + bf * buf_floats)[pl * HWp + HW] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const TowerOff& tw = INITIAL ? prog->rep : prog->dyn;
    // element j of a Q16 image in LDS (planes of HWp cells) -> its channel and cell (cell HW = the zero cell)
    auto unq16 = [&](int j, int& c, int& p) {
        const int G = j / (16 * HWp), r = j - G * 16 * HWp, c4 = r / (4 * HWp), r2 = r - c4 * 4 * HWp;
        p = r2 >> 2;
        c = 16 * G + 4 * (r2 & 3) + c4;
    };
    if (INITIAL) {
        const float* src = io.obs + (long long)g * d.O * HW;
        if (q16) {   // the observation planes, zero-filled to whole groups of 16 channels
            for (int j = tid; j < ((d.O + 15) & ~15) * HWp; j += NT) {
                int c, p;
                unq16(j, c, p);
                T1[j] = (c < d.O && p < HW) ? src[c * HW + p] : 0.f;
            }
        } else {
            for (int i = tid; i < d.O * HW; i += NT) T1[i] = src[i];
        }
    } else {
        const float* src = io.pool + (long long)io.src_slot[g] * C * HW;
        if (((C * HW) & 3) == 0) {   // the parent state: every load of a batch in flight before the first LDS store
            const float4* src4 = reinterpret_cast<const float4*>(src);
            const int n4 = (C * HW) >> 2;
            for (int i0 = 0; i0 < n4; i0 += 6 * NT) {
                float4 v[6];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const int i = i0 + tid + j * NT;
                    v[j] = src4[i < n4 ? i : 0];
                }
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const int i = i0 + tid + j * NT;   // slot order: planes of HW cells; LDS: planes of HWp
                    if (i < n4) reinterpret_cast<float4*>(T1)[q16 ? i + i / HW : i] = v[j];
                }
            }
        } else {
            for (int i = tid; i < C * HW; i += NT) T1[i] = src[i];
        }
        const int cell = io.action_cells[io.action[g]];
        if (q16) {   // ConcatInputsBlock: the action planes are channels C .. C + a - 1 of group C / 16
            for (int j = tid; j < ((d.a + 15) & ~15) * HWp; j += NT) {
                int c, p;
                unq16(j, c, p);
                T1[C * HWp + j] = (c < d.a && p < HW && c * HW + p == cell) ? 1.f : 0.f;
            }
        } else {
            for (int i = tid; i < d.a * HW; i += NT) T1[C * HW + i] = (i == cell) ? 1.f : 0.f;  // ConcatInputsBlock
        }
    }
    __syncthreads();
#if defined(MZ_FP32_STAMPS)
    tstage = __builtin_amdgcn_s_memtime() - tstart;
#endif
    // The tower as a list of steps [LayerNorm + ReLU] -> [broadcast] -> [conv]: ONE inlined copy of each routine in a loop over
    // the steps, their operands picked by wave-uniform selects.  (Out-of-line routines called from ten sites cost more than
    // they saved: every call spilled and refilled ~100 callee-saved VGPRs per lane through scratch memory.)
    // step (-1, 0) = the input conv; a bottleneck block = 4 steps, a broadcast block = 3.
    const int oX = 0, oT1 = buf_floats, oT2 = 2 * buf_floats;
    bool sum_ready = false;   // csum holds the sums of the buffer the next LayerNorm reads
    for (int r = -1; r < d.R; r++) {
        const BlockOff& b = tw.blocks[r < 0 ? 0 : r];
        const int btype = r < 0 ? -1 : b.type;
        const int nj = btype < 0 ? 1 : (btype == 0 ? 4 : 3);
        for (int j = 0; j < nj; j++) {
            int ln_in = 0, ln_out = 0, ln_c = 0, cv_in = oT1, cv_out = oX, cv_cin = tw.cin0, cv_cout = C, cv_pad = c_pad, cv_taps = 9;
            long long ln_g = 0, ln_b = 0, cv_w = tw.conv0;
            bool do_conv = true, cv_acc = false;
            if (btype == 0) {          // BottleneckResidualBlock.java:55-79
                ln_in = j == 0 ? oX : (j == 2 ? oT1 : oT2);
                ln_out = (j & 1) ? oT2 : oT1;
                ln_c = j == 0 ? C : Cb;
                ln_g = j == 0 ? b.ln1g : j == 1 ? b.ln2g : j == 2 ? b.ln3g : b.ln4g;
                ln_b = j == 0 ? b.ln1b : j == 1 ? b.ln2b : j == 2 ? b.ln3b : b.ln4b;
                cv_in = ln_out;
                cv_out = j == 3 ? oX : ((j & 1) ? oT1 : oT2);
                cv_cin = j == 0 ? C : Cb;
                cv_cout = j == 3 ? C : Cb;
                cv_pad = j == 3 ? c_pad : cb_pad;
                cv_taps = (j == 1 || j == 2) ? 9 : 1;
                cv_w = j == 0 ? b.c1 : j == 1 ? b.c2 : j == 2 ? b.c3 : b.c4;
                cv_acc = j == 3;
            } else if (btype == 1) {   // BroadcastResidualBlock.java:56-77
                ln_in = j == 0 ? oX : (j == 1 ? oT2 : oT1);
                ln_out = j == 1 ? oT2 : oT1;
                ln_c = C;
                ln_g = j == 0 ? b.ln1g : j == 1 ? b.ln2g : b.ln3g;
                ln_b = j == 0 ? b.ln1b : j == 1 ? b.ln2b : b.ln3b;
                do_conv = j != 1;
                cv_in = oT1;
                cv_out = j == 0 ? oT2 : oX;
                cv_cin = C, cv_cout = C, cv_pad = c_pad, cv_taps = 1;
                cv_w = j == 0 ? b.c1 : b.c2;
                cv_acc = j == 2;
            }
            if (btype >= 0) TL(ln_relu(smem + ln_in, smem + ln_out, ln_c * HWp, wb + ln_g, wb + ln_b, red, ln_c * HW, q16 ? HWp : 0, padmask,
                                       sum_ready ? csum : nullptr);)
            sum_ready = false;
            if (btype == 1 && j == 1) TB(broadcast_any(T2, T1, wb + b.bw, wb + b.bb, C, HW, q16);)
            if (do_conv) {
                TC(conv_lds(smem + cv_in, smem + cv_out, wb + cv_w, cv_cin, cv_pad, cv_cout, cv_taps, nb, HW, cv_acc, q16, csum);)
                sum_ready = fp32_conv_on_mfma(q16, cv_cout) && !(btype < 0 && !INITIAL);   // (the StartResidual add changes X)
            }
            if (btype < 0 && !INITIAL) {  // StartResidualBlock: conv3x3(cat[s, a]) + s
                for (int i = tid; i < C * HWp; i += NT) X[i] += T1[i];
                __syncthreads();
            }
        }
    }
    // RescaleBlockExt: (x - min) / max(max - min, 1e-5) over the whole sample
#if defined(MZ_FP32_STAMPS)
    tmark = __builtin_amdgcn_s_memtime();
#endif
    {
        float mn = INFINITY, mx = -INFINITY;
        float* dst = io.pool + (long long)io.dst_slot[g] * C * HW;
        if (q16) {   // float4 i of the state in slot order (planes of HW cells) = float4 i + i / HW of X (planes of HWp)
            const int n4 = (C * HW) >> 2;
            const float4* X4 = reinterpret_cast<const float4*>(X);
            for (int i = tid; i < n4; i += NT) {
                const float4 x = X4[i + i / HW];
                mn = fminf(mn, fminf(fminf(x.x, x.y), fminf(x.z, x.w)));
                mx = fmaxf(mx, fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w)));
            }
            mn = block_min(mn, red);
            mx = block_max(mx, red);
            const float dd = fmaxf(mx - mn, 1e-5f);
            for (int i = tid; i < n4; i += NT) {
                const int k = i + i / HW;
                const float4 x = X4[k];
                const float4 y = make_float4((x.x - mn) / dd, (x.y - mn) / dd, (x.z - mn) / dd, (x.w - mn) / dd);
                reinterpret_cast<float4*>(X)[k] = y;
                reinterpret_cast<float4*>(dst)[i] = y;
            }
        } else {
            for (int i = tid; i < C * HW; i += NT) {
                mn = fminf(mn, X[i]);
                mx = fmaxf(mx, X[i]);
            }
            mn = block_min(mn, red);
            mx = block_max(mx, red);
            const float dd = fmaxf(mx - mn, 1e-5f);
            for (int i = tid; i < C * HW; i += NT) {
                const float y = (X[i] - mn) / dd;
                X[i] = y;
                dst[i] = y;
            }
        }
        __syncthreads();
    }
#if defined(MZ_FP32_STAMPS)
    trescale = __builtin_amdgcn_s_memtime() - tmark;
#endif
    // PredictionBlock (PredictionBlock.java:46-84).  Both 1x1 head convs in one pass over X, then the two small LayerNorms,
    // the policy and value linears.  Every reduction dimension is dealt over NPART groups of threads (partial sums in LDS,
    // added in a fixed order): one thread per output would leave most of the 512 idle on chains of 128..162 dependent loads.
    const HeadOff& h = prog->head;
    float* pc = T1;              // policy conv -> LayerNorm -> [2][HW]
    float* vc = T1 + 2 * HW;     // value conv -> LayerNorm -> [HW]
    float* part_buf = T2;        // partial sums
    {
        const int np = max(1, min(min(NT / HW, buf_floats / (3 * HW)), C));   // buf_floats >= 3 HW (fp32_buf_floats)
        TH(
        for (int it = tid; it < np * HW; it += NT) {
            const int part = it / HW, p = it - part * HW;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int c = part; c < C; c += np) {
                const float x = X[q16 ? fp32_q16_index(c, p, HWp) : c * HW + p];
                const float4 wp = *reinterpret_cast<const float4*>(wb + h.pconv + (long long)c * 4);
                const float wv = wb[h.vconv + (long long)c * 4];
                a0 = fmaf(x, wp.x, a0);
                a1 = fmaf(x, wp.y, a1);
                a2 = fmaf(x, wv, a2);
            }
            part_buf[(part * 3 + 0) * HW + p] = a0;
            part_buf[(part * 3 + 1) * HW + p] = a1;
            part_buf[(part * 3 + 2) * HW + p] = a2;
        }
        __syncthreads();
        for (int it = tid; it < 3 * HW; it += NT) {
            float a = 0.f;
            for (int part = 0; part < np; part++) a += part_buf[part * 3 * HW + it];
            T1[it] = a;   // [p0 | p1 | v] = pc, vc
        }
        __syncthreads();
        )
    }
    TL(ln_relu(pc, pc, 2 * HW, wb + h.plng, wb + h.plnb, red);)
    TL(ln_relu(vc, vc, HW, wb + h.vlng, wb + h.vlnb, red);)
#if defined(MZ_FP32_STAMPS)
    tmark = __builtin_amdgcn_s_memtime();
#endif
    float* lgt = T2;                 // logits staged in LDS for the softmax
    float* fc_part = T2 + ((A + 3) & ~3);
    {
        const int np = max(1, min(min(NT / A, (buf_floats - ((A + 3) & ~3)) / A), 2 * HW));
        for (int it = tid; it < np * A; it += NT) {
            const int part = it / A, a = it - part * A;
            float acc = 0.f;
            for (int i = part; i < 2 * HW; i += np) acc = fmaf(pc[i], wb[h.pfcw + (long long)i * A + a], acc);
            fc_part[it] = acc;
        }
        __syncthreads();
        for (int a = tid; a < A; a += NT) {
            float acc = wb[h.pfcb + a];
            for (int part = 0; part < np; part++) acc += fc_part[part * A + a];
            lgt[a] = acc;
        }
        __syncthreads();
    }
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
    float part = 0.f;
    {
        const int np = max(1, min(min(NT / C, buf_floats / C), HW));
        for (int it = tid; it < np * C; it += NT) {
            const int pt = it / C, c = it - pt * C;
            float acc = 0.f;
            for (int p = pt; p < HW; p += np) acc = fmaf(vc[p], wb[h.vfc1w + (long long)p * C + c], acc);
            part_buf[it] = acc;
        }
        __syncthreads();
        for (int c = tid; c < C; c += NT) {
            float acc = wb[h.vfc1b + c];
            for (int pt = 0; pt < np; pt++) acc += part_buf[pt * C + c];
            acc = acc > 0.f ? acc : 0.f;
            part = fmaf(acc, wb[h.vfc2w + c], part);
        }
    }
    float v = block_sum(part, red) + wb[h.vfc2b];
    if (d.two_players) v = tanhf(v);
    if (tid == 0) io.value[g] = d.value_scale * (double)v;
#if defined(MZ_FP32_STAMPS)
    if (g == 0 && tid == 0)
        printf("fp32 board: total %lld conv %lld ln %lld bcast %lld headconv %lld staging %lld rescale %lld fc+softmax %lld\n",
               (long long)__builtin_amdgcn_s_memtime() - tstart, tconv, tln, tbc, thd, tstage, trescale, (long long)__builtin_amdgcn_s_memtime() - tmark);
#endif
}

}  // namespace

size_t fp32_buf_floats(const NetDims& d) {
    int extra = d.a > d.O ? d.a : d.O;
    size_t n = (size_t)(d.C + extra) * d.HW;
    if (fp32_q16(d.C, d.Cb, d.HW)) {   // conv0's input image in whole groups of 16 channels
        const size_t rec = (size_t)(d.C + ((d.a + 15) & ~15)) * (d.HW + 1), ini = (size_t)((d.O + 15) & ~15) * (d.HW + 1);   // + zero cells
        n = rec > ini ? rec : ini;
    }
    const size_t a4 = ((size_t)d.A + 3) & ~(size_t)3;
    size_t head = 2 * a4 > (size_t)3 * d.HW ? 2 * a4 : (size_t)3 * d.HW;   // the heads' staging: [3][HW] partial sums, logits + partials
    if (head < (size_t)d.C) head = d.C;
    return ((n > head ? n : head) + 3) & ~(size_t)3;   // 16-byte aligned buffers (float4 LayerNorm passes)
}

size_t fp32_lds_bytes(const NetDims& d) {
    return (3 * fp32_buf_floats(d) + 16) * sizeof(float) + (size_t)9 * (((d.HW + 15) >> 4) << 4) * sizeof(short) + 16;   // + the neighbour table
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
