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

constexpr int NT = 256;
constexpr int NW = NT / 64;

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
__device__ float block_sum(float v, float* red) {
    v = wave_sum_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w++) r += red[w];
    return r;
}
__device__ float block_max(float v, float* red) {
    v = wave_max_f(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r = fmaxf(r, red[w]);
    return r;
}
__device__ float block_min(float v, float* red) {
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
__device__ void ln_relu(const float* in, float* out, int n, const float* __restrict__ gamma,
                        const float* __restrict__ beta, float* red) {
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
__device__ void conv_lds(const float* in, float* out, const float* __restrict__ Wt, int Cin, int CoutPad, int Cout,
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

// BroadcastBlock: per channel, out[c][q] = sum_p in[c][p] * Wt[p][q] + b[q]
__device__ void broadcast_lds(const float* in, float* out, const float* __restrict__ Wt, const float* __restrict__ b,
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

template <bool INITIAL>
__global__ __launch_bounds__(NT) void k_net_fp32(NetDims d, const NetProgramF32* __restrict__ prog,
                                                 const float* __restrict__ wb, NetIO io, int buf_floats) {
    extern __shared__ float smem[];
    const int g = blockIdx.x;
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
    conv_lds(T1, X, wb + tw.conv0, tw.cin0, c_pad, C, 9, nb, HW, false);
    if (!INITIAL) {  // StartResidualBlock: conv3x3(cat[s, a]) + s
        for (int i = tid; i < C * HW; i += NT) X[i] += T1[i];
        __syncthreads();
    }
    for (int r = 0; r < d.R; r++) {
        const BlockOff& b = tw.blocks[r];
        if (b.type == 0) {
            ln_relu(X, T1, C * HW, wb + b.ln1g, wb + b.ln1b, red);
            conv_lds(T1, T2, wb + b.c1, C, cb_pad, Cb, 1, nb, HW, false);
            ln_relu(T2, T2, Cb * HW, wb + b.ln2g, wb + b.ln2b, red);
            conv_lds(T2, T1, wb + b.c2, Cb, cb_pad, Cb, 9, nb, HW, false);
            ln_relu(T1, T1, Cb * HW, wb + b.ln3g, wb + b.ln3b, red);
            conv_lds(T1, T2, wb + b.c3, Cb, cb_pad, Cb, 9, nb, HW, false);
            ln_relu(T2, T2, Cb * HW, wb + b.ln4g, wb + b.ln4b, red);
            conv_lds(T2, X, wb + b.c4, Cb, c_pad, C, 1, nb, HW, true);
        } else {
            ln_relu(X, T1, C * HW, wb + b.ln1g, wb + b.ln1b, red);
            conv_lds(T1, T2, wb + b.c1, C, c_pad, C, 1, nb, HW, false);
            ln_relu(T2, T2, C * HW, wb + b.ln2g, wb + b.ln2b, red);
            broadcast_lds(T2, T1, wb + b.bw, wb + b.bb, C, HW);
            ln_relu(T1, T1, C * HW, wb + b.ln3g, wb + b.ln3b, red);
            conv_lds(T1, X, wb + b.c2, C, c_pad, C, 1, nb, HW, true);
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
    conv_lds(X, T1, wb + h.pconv, C, 4, 2, 1, nb, HW, false);
    ln_relu(T1, T1, 2 * HW, wb + h.plng, wb + h.plnb, red);
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
    conv_lds(X, T1, wb + h.vconv, C, 4, 1, 1, nb, HW, false);
    ln_relu(T1, T1, HW, wb + h.vlng, wb + h.vlnb, red);
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
}

}  // namespace

size_t fp32_buf_floats(const NetDims& d) {
    int extra = d.a > d.O ? d.a : d.O;
    size_t n = (size_t)(d.C + extra) * d.HW;
    size_t head = (size_t)(d.A > 2 * d.HW ? d.A : 2 * d.HW);
    return n > head ? n : head;
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
