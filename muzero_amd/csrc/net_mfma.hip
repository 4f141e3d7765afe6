// bf16 MFMA forward of initialInference (f o h) and recurrentInference (f o g) for gfx950.
//
// One 256-thread workgroup (4 wavefronts, one per SIMD, up to 512 VGPRs each) carries a tile of NB
// whole boards through the ENTIRE network in one launch.  Every conv is the transposed GEMM
//     Y[out-channel][column] = W[out-channel][k] * Act[k][column],   column = (board, cell),
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   * each wave owns 64 columns (2 column tiles of 32) and ALL channels of them, so the fp32 residual
//     stream X lives in that wave's accumulator registers for the whole tower and the closing 1x1 conv
//     of a block accumulates straight into it;
//   * 1x1 convs that follow a LayerNorm take their B operand directly from the accumulator registers
//     (no LDS, no lane movement; the weights are pre-permuted in k to the accumulator's row order);
//   * 3x3 convs read their B operand from a bf16 [column][channel] image in LDS through 9 shifted
//     neighbour offsets (zero row for off-board cells) - the implicit GEMM, never materialised;
//   * LayerNorm over (C,H,W) of a sample = per-column partial sums -> LDS -> per-board reduction,
//     fused into the epilogue of the producing conv; ReLU, residual add, min-max rescale, softmax
//     and the value scaling are fused too;
//   * weights are packed on the host into 1 KiB MFMA A-fragments in exact execution order: one linear
//     stream per tower that each wave prefetches through a 16-deep register ring (16 KiB in flight per
//     wave) across layer and block boundaries.
// HBM traffic per sample: parent hidden state in, child hidden state out, logits/policy/value out.
//
// Reference math: see net_fp32.hip / oracle/net.py (same citations).
#include "net_mfma.h"

#include <math.h>

#include <cstring>
#include <vector>

namespace mz {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int NT = 256;        // threads per workgroup
constexpr int TC = 2;          // column tiles (of 32) per wave
constexpr int COLS = 256;      // columns per workgroup
constexpr int ZROW = COLS;     // LDS row of zeros (off-board neighbours, padded columns)
constexpr int RING = 16;       // A-fragment prefetch depth per wave
constexpr int MAXNB = 32;      // boards per workgroup upper bound (COLS / 9 = 28)

struct TowerDev {
    const u32x4* stream;   // A fragments in execution order (+ RING fragments of slack)
    const u32x4* gb;       // LayerNorm gamma/beta, fp16, [p][Cn/4]{g0..3,b0..3}
    const u32x4* wbf;      // broadcast-linear B fragments (bf16)
    const float* bb;       // broadcast-linear bias, padded to QT*32
    int type[kMaxResiduals];
    long long gboff[kMaxResiduals];  // in u32x4 units
    int wboff[kMaxResiduals];        // broadcast blocks: offset into wbf (u32x4 units) / bb (floats)
    int bboff[kMaxResiduals];
};
struct HeadDev {
    const float* hconv; // [3][C]: policy conv rows 0,1 and the value conv row
    const float* plng;  // [2][HW]
    const float* plnb;
    const float* pfcw;  // [2HW][A]
    const float* pfcb;  // [A]
    const float* vlng;  // [HW]
    const float* vlnb;
    const float* vfc1w; // [HW][C]
    const float* vfc1b; // [C]
    const float* vfc2w; // [C]
    const float* vfc2b; // [1]
};
struct LdsPlan {
    int act_off, tb_off, colsum_off, bstat_off, meta_off, nbt_off, head_off, total;
    int tbs;  // TB row stride in bytes
    int kp, qt;
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8 pack8(const float* y) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = (__bf16)y[i];
    return r;
}
__device__ __forceinline__ bf16x4 pack4(const float* y) {
    bf16x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = (__bf16)y[i];
    return r;
}

// Register ring over the linear A-fragment stream. Indices passed to get() are compile-time
// constants after unrolling; advance() moves the origin by a multiple of RING fragments.
struct Ring {
    u32x4 s[RING];
    const u32x4* p;
    __device__ __forceinline__ void init(const u32x4* base, int lane) {
        p = base + lane;
#pragma unroll
        for (int i = 0; i < RING; i++) s[i] = p[i * 64];
    }
    __device__ __forceinline__ bf16x8 get(int i) {
        u32x4 r = s[i % RING];
        s[i % RING] = p[(i + RING) * 64];
        return __builtin_bit_cast(bf16x8, r);
    }
    __device__ __forceinline__ void advance(int nfrags) { p += (long long)nfrags * 64; }
};

struct Cols {
    int col[TC];     // column index in the workgroup
    int p[TC];       // cell within the board (0 for padded columns)
    int board[TC];   // board within the tile (MAXNB for padded columns -> neutral statistics)
    bool valid[TC];  // col < NB*HW
    bool live[TC];   // valid and the board is an active sample
    int nb[TC][9];   // LDS row of each 3x3 neighbour (ZROW if off-board)
};

struct Ctx {
    char* smem;
    LdsPlan L;
    int HW, NB, lane, wave, h, r;
    float* colsum;   // [COLS][2]
    float* bstat;    // [MAXNB+1][2]
};

// Per-board reduction of per-column partials. MODE 0: (sum, sumsq) -> (mean, rstd); MODE 1: (min, max).
template <int MODE>
__device__ __forceinline__ void board_reduce(const Ctx& c, const Cols& k, float (&a)[TC], float (&b)[TC], float n_inv) {
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        float oa = __shfl_xor(a[ct], 32), ob = __shfl_xor(b[ct], 32);
        if (MODE == 0) {
            a[ct] += oa;
            b[ct] += ob;
        } else {
            a[ct] = fminf(a[ct], oa);
            b[ct] = fmaxf(b[ct], ob);
        }
        if (c.h == 0) {
            c.colsum[2 * k.col[ct]] = a[ct];
            c.colsum[2 * k.col[ct] + 1] = b[ct];
        }
    }
    lds_barrier();
    {
        const int t = threadIdx.x;
        const int bd = t >> 3, part = t & 7;
        float x = MODE == 0 ? 0.f : INFINITY, y = MODE == 0 ? 0.f : -INFINITY;
        if (bd < c.NB) {
            for (int i = part; i < c.HW; i += 8) {
                float u = c.colsum[2 * (bd * c.HW + i)], v = c.colsum[2 * (bd * c.HW + i) + 1];
                if (MODE == 0) {
                    x += u;
                    y += v;
                } else {
                    x = fminf(x, u);
                    y = fmaxf(y, v);
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            float u = __shfl_xor(x, o), v = __shfl_xor(y, o);
            if (MODE == 0) {
                x += u;
                y += v;
            } else {
                x = fminf(x, u);
                y = fmaxf(y, v);
            }
        }
        if (bd < c.NB && part == 0) {
            if (MODE == 0) {
                float mean = x * n_inv;
                float var = fmaxf(y * n_inv - mean * mean, 0.f);
                c.bstat[2 * bd] = mean;
                c.bstat[2 * bd + 1] = 1.0f / sqrtf(var + 1e-5f);
            } else {
                c.bstat[2 * bd] = x;
                c.bstat[2 * bd + 1] = y;
            }
        }
        if (t == 0) {  // neutral entry for padded columns
            c.bstat[2 * MAXNB] = 0.f;
            c.bstat[2 * MAXNB + 1] = MODE == 0 ? 0.f : 1.f;
        }
    }
    lds_barrier();
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        a[ct] = c.bstat[2 * k.board[ct]];
        b[ct] = c.bstat[2 * k.board[ct] + 1];
    }
}

// LayerNorm statistics of NRT row tiles (all channels) x own columns -> per-column (mean, rstd).
template <int NRT>
__device__ __forceinline__ void ln_stats(const Ctx& c, const Cols& k, const f32x16 (&v)[NRT][TC], int nch,
                                         float (&mean)[TC], float (&rstd)[TC]) {
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int rt = 0; rt < NRT; rt++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float x = v[rt][ct][i];
                s += x;
                q = fmaf(x, x, q);
            }
        mean[ct] = k.valid[ct] ? s : 0.f;
        rstd[ct] = k.valid[ct] ? q : 0.f;
    }
    board_reduce<0>(c, k, mean, rstd, 1.0f / (float)(nch * c.HW));
}

// gamma/beta of row tile rt for both column tiles: 4 x 16-byte loads per column.
__device__ __forceinline__ void gb_load(const u32x4* gb, int cn4, const Ctx& c, const Cols& k, int rt,
                                        u32x4 (&g)[TC][4]) {
#pragma unroll
    for (int ct = 0; ct < TC; ct++)
#pragma unroll
        for (int qd = 0; qd < 4; qd++) g[ct][qd] = gb[(long long)k.p[ct] * cn4 + 8 * rt + 2 * qd + c.h];
}

// y = relu((v - mean) * rstd * gamma + beta) for the 16 accumulator rows of one (row tile, column tile).
__device__ __forceinline__ void ln_apply(const f32x16& v, float mean, float rstd, const u32x4 (&g)[4], float (&y)[16]) {
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        f16x8 hv = __builtin_bit_cast(f16x8, g[qd]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float t = (v[4 * qd + e] - mean) * rstd;
            t = fmaf(t, (float)hv[e], (float)hv[4 + e]);
            y[4 * qd + e] = fmaxf(t, 0.f);
        }
    }
}

// Write the 16 rows of (row tile rt, column tile) as bf16 into the [column][channel] LDS image.
__device__ __forceinline__ void act_store(char* act, int rs, int col, int rt, int h, const float (&y)[16]) {
#pragma unroll
    for (int qd = 0; qd < 4; qd++)
        *reinterpret_cast<bf16x4*>(act + col * rs + (32 * rt + 8 * qd + 4 * h) * 2) = pack4(&y[4 * qd]);
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// out[ro][ct] += W * relu(LN(in)) with the B operand taken from the accumulator registers `in`.
// Fragment order in the stream: (rt_in, s, ro).
template <int NRI, int NRO>
__device__ __forceinline__ void conv1x1_regfed(const Ctx& c, const Cols& k, Ring& ring, int& fi, const u32x4* gb,
                                               const f32x16 (&in)[NRI][TC], f32x16 (&out)[NRO][TC]) {
    float mean[TC], rstd[TC];
    ln_stats<NRI>(c, k, in, NRI * 32, mean, rstd);
    u32x4 g[TC][4];
    gb_load(gb, NRI * 8, c, k, 0, g);
#pragma unroll
    for (int rt = 0; rt < NRI; rt++) {
        bf16x8 bf[TC][2];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            float y[16];
            ln_apply(in[rt][ct], mean[ct], rstd[ct], g[ct], y);
            bf[ct][0] = pack8(&y[0]);
            bf[ct][1] = pack8(&y[8]);
        }
        if (rt + 1 < NRI) gb_load(gb, NRI * 8, c, k, rt + 1, g);
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int ro = 0; ro < NRO; ro++) {
                bf16x8 a = ring.get(fi++);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) out[ro][ct] = MFMA(a, bf[ct][s], out[ro][ct]);
            }
    }
}

// LN + ReLU of `in` (NRI row tiles) into the LDS image, then out = conv(image) over TAPS taps.
// Fragment order: (tap, ks, ro).  TAPS == 9: 3x3 with neighbour rows k.nb; TAPS == 1: centre tap only.
template <int NRI, int NRO, int TAPS>
__device__ __forceinline__ void conv_ldsfed(const Ctx& c, const Cols& k, Ring& ring, int& fi, const u32x4* gb,
                                            const f32x16 (&in)[NRI][TC], f32x16 (&out)[NRO][TC]) {
    constexpr int RS = NRI * 64 + 16;
    char* act = c.smem + c.L.act_off;
    float mean[TC], rstd[TC];
    ln_stats<NRI>(c, k, in, NRI * 32, mean, rstd);
#pragma unroll
    for (int rt = 0; rt < NRI; rt++) {
        u32x4 g[TC][4];
        gb_load(gb, NRI * 8, c, k, rt, g);
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            float y[16];
            ln_apply(in[rt][ct], mean[ct], rstd[ct], g[ct], y);
            if (!k.valid[ct]) {
#pragma unroll
                for (int i = 0; i < 16; i++) y[i] = 0.f;
            }
            act_store(act, RS, k.col[ct], rt, c.h, y);
        }
    }
    lds_barrier();
#pragma unroll
    for (int t = 0; t < TAPS; t++)
#pragma unroll
        for (int ks = 0; ks < NRI * 2; ks++) {
            bf16x8 b[TC];
#pragma unroll
            for (int ct = 0; ct < TC; ct++) {
                const int row = TAPS == 9 ? k.nb[ct][t] : k.col[ct];
                b[ct] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(act + row * RS + (16 * ks + 8 * c.h) * 2));
            }
#pragma unroll
            for (int ro = 0; ro < NRO; ro++) {
                bf16x8 a = ring.get(fi++);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) out[ro][ct] = MFMA(a, b[ct], out[ro][ct]);
            }
        }
}

template <int NR>
__device__ __forceinline__ void zero_acc(f32x16 (&a)[NR][TC]) {
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int ct = 0; ct < TC; ct++)
#pragma unroll
            for (int i = 0; i < 16; i++) a[r][ct][i] = 0.f;
}

// BottleneckResidualBlock: x + c4(relu(LN(c3(relu(LN(c2(relu(LN(c1(relu(LN(x))))))))))))
template <int C, int CB>
__device__ __forceinline__ void bottleneck(const Ctx& c, const Cols& k, Ring& ring, const u32x4* gb,
                                           f32x16 (&X)[C / 32][TC]) {
    constexpr int NRC = C / 32, NRB = CB / 32;
    const long long s1 = (long long)c.HW * (C / 4), sb = (long long)c.HW * (CB / 4);
    int fi = 0;
    f32x16 a1[NRB][TC], a2[NRB][TC];
    zero_acc<NRB>(a1);
    conv1x1_regfed<NRC, NRB>(c, k, ring, fi, gb, X, a1);
    zero_acc<NRB>(a2);
    conv_ldsfed<NRB, NRB, 9>(c, k, ring, fi, gb + s1, a1, a2);
    zero_acc<NRB>(a1);
    conv_ldsfed<NRB, NRB, 9>(c, k, ring, fi, gb + s1 + sb, a2, a1);
    conv1x1_regfed<NRB, NRC>(c, k, ring, fi, gb + s1 + 2 * sb, a1, X);
    ring.advance(fi);  // 2*NRC*NRB*2 + 2*9*2*NRB*NRB fragments: a multiple of RING for the supported shapes
}

// BroadcastResidualBlock: x + c2(relu(LN(bcast(relu(LN(c1(relu(LN(x)))))))))
template <int C>
__device__ __forceinline__ void broadcast_block(const Ctx& c, const Cols& k, Ring& ring, const u32x4* gb,
                                                const u32x4* wbf, const float* bb, f32x16 (&X)[C / 32][TC]) {
    constexpr int NRC = C / 32;
    constexpr int RS = C * 2 + 16;
    const long long s1 = (long long)c.HW * (C / 4);
    const int KP = c.L.kp, QT = c.L.qt, TBS = c.L.tbs;
    char* tb = c.smem + c.L.tb_off;
    char* act = c.smem + c.L.act_off;
    int fi = 0;
    {
        f32x16 a1[NRC][TC];
        zero_acc<NRC>(a1);
        conv1x1_regfed<NRC, NRC>(c, k, ring, fi, gb, X, a1);
        // LN2 + ReLU -> bf16 -> transposed image TB[channel][board*KP + cell]
        float mean[TC], rstd[TC];
        ln_stats<NRC>(c, k, a1, C, mean, rstd);
        for (int i = threadIdx.x; i < C * c.NB * (KP - c.HW); i += NT) {  // zero the K padding of every board
            const int padw = KP - c.HW;
            const int ch = i / (c.NB * padw), rem = i - ch * (c.NB * padw);
            const int bd = rem / padw, pp = rem - bd * padw;
            *reinterpret_cast<__bf16*>(tb + ch * TBS + (bd * KP + c.HW + pp) * 2) = (__bf16)0.f;
        }
#pragma unroll
        for (int rt = 0; rt < NRC; rt++) {
            u32x4 g[TC][4];
            gb_load(gb + s1, C / 4, c, k, rt, g);
#pragma unroll
            for (int ct = 0; ct < TC; ct++) {
                float y[16];
                ln_apply(a1[rt][ct], mean[ct], rstd[ct], g[ct], y);
                if (k.valid[ct]) {
                    const int cbase = k.board[ct] * KP + k.p[ct];
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int ch = 32 * rt + (i & 3) + 8 * (i >> 2) + 4 * c.h;
                        *reinterpret_cast<__bf16*>(tb + ch * TBS + cbase * 2) = (__bf16)y[i];
                    }
                }
            }
        }
    }
    lds_barrier();
    // BroadcastBlock: out[ch][q] = sum_p in[ch][p] * Wb[q][p] + b[q], per board; wave w owns row tiles w, w+4, ..
    // Pass 0 accumulates the LN3 statistics, pass 1 recomputes and writes relu(LN3(.)) to the act image.
    float* wpart = c.bstat + 2 * (MAXNB + 1);  // [4 waves][MAXNB][2]
    float bmean = 0.f, brstd = 0.f;
    for (int pass = 0; pass < 2; pass++) {
        for (int bd = 0; bd < c.NB; bd++) {
            float ps = 0.f, pq = 0.f;
            if (pass == 1) {
                bmean = c.bstat[2 * bd];
                brstd = c.bstat[2 * bd + 1];
            }
            for (int rtw = c.wave; rtw < NRC; rtw += 4) {
                for (int qt = 0; qt < QT; qt++) {
                    f32x16 acc;
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = 0.f;
                    for (int ks = 0; ks < KP / 16; ks++) {
                        bf16x8 a = __builtin_bit_cast(
                            bf16x8, *reinterpret_cast<const u32x4*>(tb + (32 * rtw + c.r) * TBS + (bd * KP + 16 * ks + 8 * c.h) * 2));
                        bf16x8 b = __builtin_bit_cast(bf16x8, wbf[(ks * QT + qt) * 64 + c.lane]);
                        acc = MFMA(a, b, acc);
                    }
                    const int q = 32 * qt + c.r;
                    const bool qv = q < c.HW;
                    const float bias = bb[q];
                    if (pass == 0) {
                        if (qv) {
#pragma unroll
                            for (int i = 0; i < 16; i++) {
                                float x = acc[i] + bias;
                                ps += x;
                                pq = fmaf(x, x, pq);
                            }
                        }
                    } else if (qv) {
                        u32x4 g[4];
#pragma unroll
                        for (int qd = 0; qd < 4; qd++) g[qd] = (gb + 2 * s1)[(long long)q * (C / 4) + 8 * rtw + 2 * qd + c.h];
#pragma unroll
                        for (int i = 0; i < 16; i++) acc[i] += bias;
                        float y[16];
                        ln_apply(acc, bmean, brstd, g, y);
                        act_store(act, RS, bd * c.HW + q, rtw, c.h, y);
                    }
                }
            }
            if (pass == 0) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ps += __shfl_xor(ps, o);
                    pq += __shfl_xor(pq, o);
                }
                if (c.lane == 0) {
                    wpart[(c.wave * MAXNB + bd) * 2] = ps;
                    wpart[(c.wave * MAXNB + bd) * 2 + 1] = pq;
                }
            }
        }
        if (pass == 0) {
            lds_barrier();
            if ((int)threadIdx.x < c.NB) {
                const int bd = threadIdx.x;
                float s = 0.f, q = 0.f;
                for (int w = 0; w < 4; w++) {
                    s += wpart[(w * MAXNB + bd) * 2];
                    q += wpart[(w * MAXNB + bd) * 2 + 1];
                }
                const float ninv = 1.0f / (float)(C * c.HW);
                const float mean = s * ninv;
                const float var = fmaxf(q * ninv - mean * mean, 0.f);
                c.bstat[2 * bd] = mean;
                c.bstat[2 * bd + 1] = 1.0f / sqrtf(var + 1e-5f);
            }
            lds_barrier();
        }
    }
    lds_barrier();
    // c2: 1x1 conv C -> C on the act image (own columns), accumulated into X. Fragment order (ks, ro).
#pragma unroll
    for (int ks = 0; ks < C / 16; ks++) {
        bf16x8 b[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++)
            b[ct] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(act + k.col[ct] * RS + (16 * ks + 8 * c.h) * 2));
#pragma unroll
        for (int ro = 0; ro < NRC; ro++) {
            bf16x8 a = ring.get(fi++);
#pragma unroll
            for (int ct = 0; ct < TC; ct++) X[ro][ct] = MFMA(a, b[ct], X[ro][ct]);
        }
    }
    ring.advance(fi);
    lds_barrier();
}

template <int C, int CB, bool INITIAL>
__global__ __launch_bounds__(NT, 1) void k_net_mfma(NetDims d, TowerDev tw, HeadDev hd, NetIO io, LdsPlan L, int NB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NRC = C / 32;
    constexpr int CIN0 = INITIAL ? 32 : C + 16;     // input channels of conv0, padded to 16
    constexpr int KS0 = CIN0 / 16;
    constexpr int KS0P = (KS0 * NRC + RING - 1) / RING * RING / NRC;  // k-steps per tap in the stream (padded)
    constexpr int RS0 = CIN0 * 2 + 16;
    static_assert((KS0P * NRC) % RING == 0, "conv0 fragment count per tap must be a multiple of RING");
    const int tid = threadIdx.x;
    Ctx c;
    c.smem = smem;
    c.L = L;
    c.HW = d.HW;
    c.NB = NB;
    c.lane = tid & 63;
    c.wave = tid >> 6;
    c.h = c.lane >> 5;
    c.r = c.lane & 31;
    c.colsum = reinterpret_cast<float*>(smem + L.colsum_off);
    c.bstat = reinterpret_cast<float*>(smem + L.bstat_off);
    int* meta = reinterpret_cast<int*>(smem + L.meta_off);  // [4][MAXNB]: src, dst, live, cell
    short* nbt = reinterpret_cast<short*>(smem + L.nbt_off); // [9][COLS]
    const int HW = d.HW, A = d.A;
    const int g0 = blockIdx.x * NB;

    if (tid < MAXNB) {
        const int g = g0 + tid;
        const bool ok = tid < NB && g < io.B && (!io.active || io.active[g]);
        meta[tid] = (ok && !INITIAL) ? io.src_slot[g] : 0;
        meta[MAXNB + tid] = ok ? io.dst_slot[g] : 0;
        meta[2 * MAXNB + tid] = ok ? 1 : 0;
        meta[3 * MAXNB + tid] = (ok && !INITIAL) ? io.action_cells[io.action[g]] : -1;
    }
    for (int i = tid; i < 9 * COLS; i += NT) {
        const int t = i / COLS, col = i - t * COLS;
        int row = ZROW;
        if (col < NB * HW) {
            const int bd = col / HW, p = col - bd * HW;
            const int y = p / d.Wd + t / 3 - 1, x = p % d.Wd + t % 3 - 1;
            if (y >= 0 && y < d.H && x >= 0 && x < d.Wd) row = bd * HW + y * d.Wd + x;
        }
        nbt[i] = (short)row;
    }
    lds_barrier();
    {
        int any = 0;
        for (int b = 0; b < NB; b++) any |= meta[2 * MAXNB + b];
        if (!any) return;
    }
    Cols k;
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        const int col = 64 * c.wave + 32 * ct + c.r;
        k.col[ct] = col;
        k.valid[ct] = col < NB * HW;
        k.board[ct] = k.valid[ct] ? col / HW : MAXNB;
        k.p[ct] = k.valid[ct] ? col - k.board[ct] * HW : 0;
        k.live[ct] = k.valid[ct] && meta[2 * MAXNB + k.board[ct]];
#pragma unroll
        for (int t = 0; t < 9; t++) k.nb[ct][t] = nbt[t * COLS + col];
    }

    Ring ring;
    ring.init(tw.stream, c.lane);

    // ---------------- input image + conv0 ----------------
    f32x16 X[NRC][TC];
    char* act = smem + L.act_off;
    for (int i = tid; i < RS0 / 4; i += NT) reinterpret_cast<int*>(act + ZROW * RS0)[i] = 0;
    if (INITIAL) {
        zero_acc<NRC>(X);
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            const int g = g0 + (k.valid[ct] ? k.board[ct] : 0);
            float y[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int ch = 16 * c.h + i;
                y[i] = (k.live[ct] && ch < d.O) ? io.obs[((long long)g * d.O + ch) * HW + k.p[ct]] : 0.f;
            }
            *reinterpret_cast<bf16x8*>(act + k.col[ct] * RS0 + (16 * c.h) * 2) = pack8(&y[0]);
            *reinterpret_cast<bf16x8*>(act + k.col[ct] * RS0 + (16 * c.h + 8) * 2) = pack8(&y[8]);
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            const int bd = k.valid[ct] ? k.board[ct] : 0;
            const float* src = io.pool + (long long)meta[bd] * C * HW + k.p[ct];
#pragma unroll
            for (int rt = 0; rt < NRC; rt++) {
                float y[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int ch = 32 * rt + (i & 3) + 8 * (i >> 2) + 4 * c.h;
                    y[i] = k.live[ct] ? src[(long long)ch * HW] : 0.f;
                    X[rt][ct][i] = y[i];
                }
                act_store(act, RS0, k.col[ct], rt, c.h, y);
            }
            // ConcatInputsBlock: the one-hot action planes as channels C .. C+a-1 (padded to 16)
            if (c.h == ct) {
                float y[16];
                const int cell = meta[3 * MAXNB + bd];
#pragma unroll
                for (int i = 0; i < 16; i++) y[i] = (k.live[ct] && i < d.a && cell == i * HW + k.p[ct]) ? 1.f : 0.f;
                *reinterpret_cast<bf16x8*>(act + k.col[ct] * RS0 + C * 2) = pack8(&y[0]);
                *reinterpret_cast<bf16x8*>(act + k.col[ct] * RS0 + (C + 8) * 2) = pack8(&y[8]);
            }
        }
    }
    lds_barrier();
    for (int t = 0; t < 9; t++) {
        int row[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) row[ct] = nbt[t * COLS + k.col[ct]];
#pragma unroll
        for (int ks = 0; ks < KS0; ks++) {
            bf16x8 b[TC];
#pragma unroll
            for (int ct = 0; ct < TC; ct++)
                b[ct] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(act + row[ct] * RS0 + (16 * ks + 8 * c.h) * 2));
#pragma unroll
            for (int ro = 0; ro < NRC; ro++) {
                bf16x8 a = ring.get(ks * NRC + ro);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) X[ro][ct] = MFMA(a, b[ct], X[ro][ct]);
            }
        }
#pragma unroll
        for (int i = KS0 * NRC; i < KS0P * NRC; i++) (void)ring.get(i);  // padding fragments of the stream
        ring.advance(KS0P * NRC);
    }
    lds_barrier();
    // the tower's act image uses a different row stride: re-zero its zero row
    {
        constexpr int RSB = CB * 2 + 16, RSC = C * 2 + 16;
        for (int i = tid; i < RSC / 4; i += NT) {
            if (i < RSB / 4) reinterpret_cast<int*>(act + ZROW * RSB)[i] = 0;
            reinterpret_cast<int*>(act + ZROW * RSC)[i] = 0;
        }
    }
    lds_barrier();

    // ---------------- residual tower ----------------
    for (int blk = 0; blk < d.R; blk++) {
        const u32x4* gb = tw.gb + tw.gboff[blk];
        if (tw.type[blk] == 0) bottleneck<C, CB>(c, k, ring, gb, X);
        else {
            broadcast_block<C>(c, k, ring, gb, tw.wbf + tw.wboff[blk], tw.bb + tw.bboff[blk], X);
            constexpr int RSB = CB * 2 + 16;
            for (int i = tid; i < RSB / 4; i += NT) reinterpret_cast<int*>(act + ZROW * RSB)[i] = 0;
            lds_barrier();
        }
    }

    // ---------------- RescaleBlockExt + hidden-state store ----------------
    {
        float mn[TC], mx[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            float a = INFINITY, b = -INFINITY;
#pragma unroll
            for (int rt = 0; rt < NRC; rt++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    a = fminf(a, X[rt][ct][i]);
                    b = fmaxf(b, X[rt][ct][i]);
                }
            mn[ct] = k.valid[ct] ? a : INFINITY;
            mx[ct] = k.valid[ct] ? b : -INFINITY;
        }
        board_reduce<1>(c, k, mn, mx, 0.f);
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            const float dd = fmaxf(mx[ct] - mn[ct], 1e-5f);
            const int bd = k.valid[ct] ? k.board[ct] : 0;
            float* dst = io.pool + (long long)meta[MAXNB + bd] * C * HW + k.p[ct];
#pragma unroll
            for (int rt = 0; rt < NRC; rt++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int ch = 32 * rt + (i & 3) + 8 * (i >> 2) + 4 * c.h;
                    const float y = (X[rt][ct][i] - mn[ct]) / dd;
                    X[rt][ct][i] = y;
                    if (k.live[ct]) dst[(long long)ch * HW] = y;
                }
        }
    }

    // ---------------- PredictionBlock ----------------
    // 1x1 convs C -> {policy 0, policy 1, value} in fp32 on the VALU (3 output rows only): the per-board
    // LayerNorm over a single 81-value channel that follows amplifies bf16 rounding, so these stay exact.
    float* hs = reinterpret_cast<float*>(smem + L.head_off);
    float* pfeat = hs;                         // [NB][2*HW]
    float* vfeat = pfeat + MAXNB * 2 * HW;     // [NB][HW]
    float* lgt = vfeat + MAXNB * HW;           // [NB][A]
    float* hv = lgt + MAXNB * A;               // [NB][C]
    float* hwl = hv + MAXNB * C;               // [3][C] head conv weights
    for (int i = tid; i < 3 * C; i += NT) hwl[i] = hd.hconv[i];
    lds_barrier();
    f32x16 hacc[TC];
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        float p0 = 0.f, p1 = 0.f, v = 0.f;
#pragma unroll
        for (int rt = 0; rt < NRC; rt++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int ch = 32 * rt + (i & 3) + 8 * (i >> 2) + 4 * c.h;
                const float x = X[rt][ct][i];
                p0 = fmaf(x, hwl[ch], p0);
                p1 = fmaf(x, hwl[C + ch], p1);
                v = fmaf(x, hwl[2 * C + ch], v);
            }
        p0 += __shfl_xor(p0, 32);
        p1 += __shfl_xor(p1, 32);
        v += __shfl_xor(v, 32);
        hacc[ct][0] = p0;
        hacc[ct][1] = p1;
        hacc[ct][2] = v;
    }
    {
        // policy LN over (2, HW), value LN over (1, HW); rows 0..3 sit in regs 0..3 of lanes with h == 0
        float s[TC], q[TC], vs[TC], vq[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            const bool own = k.valid[ct] && c.h == 0;
            const float p0 = hacc[ct][0], p1 = hacc[ct][1], v = hacc[ct][2];
            s[ct] = own ? p0 + p1 : 0.f;
            q[ct] = own ? p0 * p0 + p1 * p1 : 0.f;
            vs[ct] = own ? v : 0.f;
            vq[ct] = own ? v * v : 0.f;
        }
        board_reduce<0>(c, k, s, q, 1.0f / (float)(2 * HW));
        board_reduce<0>(c, k, vs, vq, 1.0f / (float)HW);
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            if (k.valid[ct] && c.h == 0) {
                const int bd = k.board[ct], p = k.p[ct];
#pragma unroll
                for (int ch = 0; ch < 2; ch++) {
                    float t = (hacc[ct][ch] - s[ct]) * q[ct] * hd.plng[ch * HW + p] + hd.plnb[ch * HW + p];
                    pfeat[bd * 2 * HW + ch * HW + p] = fmaxf(t, 0.f);
                }
                float t = (hacc[ct][2] - vs[ct]) * vq[ct] * hd.vlng[p] + hd.vlnb[p];
                vfeat[bd * HW + p] = fmaxf(t, 0.f);
            }
        }
    }
    lds_barrier();
    for (int idx = tid; idx < NB * A; idx += NT) {
        const int bd = idx / A, a = idx - bd * A;
        float acc = hd.pfcb[a];
        const float* f = pfeat + bd * 2 * HW;
        for (int i = 0; i < 2 * HW; i++) acc = fmaf(f[i], hd.pfcw[(long long)i * A + a], acc);
        lgt[idx] = acc;
    }
    for (int idx = tid; idx < NB * C; idx += NT) {
        const int bd = idx / C, ch = idx - bd * C;
        float acc = hd.vfc1b[ch];
        const float* f = vfeat + bd * HW;
        for (int p = 0; p < HW; p++) acc = fmaf(f[p], hd.vfc1w[(long long)p * C + ch], acc);
        hv[idx] = fmaxf(acc, 0.f);
    }
    lds_barrier();
    for (int bd = c.wave; bd < NB; bd += 4) {
        if (!meta[2 * MAXNB + bd]) continue;
        const int g = g0 + bd;
        float part = 0.f;
        for (int ch = c.lane; ch < C; ch += 64) part = fmaf(hv[bd * C + ch], hd.vfc2w[ch], part);
        float mx = -INFINITY;
        for (int a = c.lane; a < A; a += 64) mx = fmaxf(mx, lgt[bd * A + a]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            part += __shfl_xor(part, o);
            mx = fmaxf(mx, __shfl_xor(mx, o));
        }
        float se = 0.f;
        for (int a = c.lane; a < A; a += 64) se += expf(lgt[bd * A + a] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
        for (int a = c.lane; a < A; a += 64) {
            const float l = lgt[bd * A + a];
            io.logits[(long long)g * A + a] = l;
            io.policy[(long long)g * A + a] = expf(l - mx) / se;
        }
        if (c.lane == 0) {
            float v = part + hd.vfc2b[0];
            if (d.two_players) v = tanhf(v);
            io.value[g] = d.value_scale * (double)v;
        }
    }
}

// ------------------------------- host side: packing -------------------------------------------
inline unsigned short f2bf(float f) {
    unsigned int u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline unsigned short f2h(float f) {
    _Float16 h = (_Float16)f;
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
}

struct FragWriter {
    std::vector<unsigned short>& out;
    // w(row, kprime) -> weight for MFMA row (0..31) and hardware k index (0..15) of this fragment
    template <typename F>
    void frag(F w) {
        size_t base = out.size();
        out.resize(base + 512);
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 8; j++) out[base + l * 8 + j] = f2bf(w(l & 31, 8 * (l >> 5) + j));
    }
    void pad_to(int mult) {
        while ((out.size() / 512) % mult) out.resize(out.size() + 512, 0);
    }
};

// register-fed 1x1 conv W[Cout][Cin]: order (rt_in, s, ro); hardware k' = 8h + j  <->  channel
// 32*rt_in + 16*s + 8*(j>>2) + 4*h + (j&3)
void pack_regfed(FragWriter& fw, const float* W, int Cout, int Cin, int rows_used = -1) {
    const int nro = (Cout + 31) / 32;
    for (int rt = 0; rt < Cin / 32; rt++)
        for (int s = 0; s < 2; s++)
            for (int ro = 0; ro < nro; ro++)
                fw.frag([&](int row, int kp) {
                    const int h = kp >> 3, j = kp & 7;
                    const int ci = 32 * rt + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
                    const int co = 32 * ro + row;
                    return (co < Cout && (rows_used < 0 || co < rows_used)) ? W[(size_t)co * Cin + ci] : 0.f;
                });
}
// LDS-fed conv W[Cout][Cin][taps]: order (tap, ks, ro) with natural k; ks padded to ksp per tap
void pack_ldsfed(FragWriter& fw, const float* W, int Cout, int Cin, int taps, int cin_pad, int ksp) {
    for (int t = 0; t < taps; t++)
        for (int ks = 0; ks < ksp; ks++)
            for (int ro = 0; ro < Cout / 32; ro++)
                fw.frag([&](int row, int kp) {
                    const int ci = 16 * ks + kp;
                    if (ks >= cin_pad / 16 || ci >= Cin) return 0.f;
                    return W[((size_t)(32 * ro + row) * Cin + ci) * taps + t];
                });
}
// gamma/beta [Cn][HW] -> fp16 [p][Cn/4]{g0..3, b0..3}
void pack_gb(std::vector<unsigned short>& out, const float* g, const float* b, int Cn, int HW) {
    for (int p = 0; p < HW; p++)
        for (int c4 = 0; c4 < Cn / 4; c4++) {
            for (int e = 0; e < 4; e++) out.push_back(f2h(g[(size_t)(4 * c4 + e) * HW + p]));
            for (int e = 0; e < 4; e++) out.push_back(f2h(b[(size_t)(4 * c4 + e) * HW + p]));
        }
}

struct TowerHost {
    std::vector<unsigned short> stream, gb, wbf;
    std::vector<float> bb;
    TowerDev dev{};
};

}  // namespace

struct MfmaNet {
    NetDims d{};
    int NB = 0;
    LdsPlan L{};
    void* bufs[16] = {nullptr};
    int nbufs = 0;
    TowerDev tw[2]{};
    HeadDev hd{};
    bool packed = false;
};

namespace {

bool shape_supported(const NetDims& d) { return d.C == 128 && d.Cb == 64 && d.O <= 32 && d.a <= 16 && d.HW <= 81 + 15; }

LdsPlan make_plan(const NetDims& d, int NB) {
    LdsPlan L{};
    const int C = d.C;
    const int rs0 = (C + 16) * 2 + 16, rsc = C * 2 + 16;
    L.kp = (d.HW + 15) / 16 * 16;
    L.qt = (d.HW + 31) / 32;
    L.tbs = (NB * L.kp + 8) * 2;
    const int act_tower = (COLS + 1) * rsc;
    const int act_conv0 = (COLS + 1) * rs0;
    L.act_off = 0;
    L.tb_off = (act_tower + 15) / 16 * 16;
    int end = L.tb_off + C * L.tbs;
    if (act_conv0 > end) end = act_conv0;
    end = (end + 15) / 16 * 16;
    L.colsum_off = end;
    end += COLS * 2 * 4;
    L.bstat_off = end;
    end += (2 * (MAXNB + 1) + 4 * MAXNB * 2) * 4;
    L.meta_off = end;
    end += 4 * MAXNB * 4;
    L.nbt_off = end;
    end += 9 * COLS * 2;
    end = (end + 15) / 16 * 16;
    // head scratch overlaps the (dead) act image
    L.head_off = 0;
    L.total = end;
    return L;
}

template <typename T>
int upload(MfmaNet* n, const std::vector<T>& v, const void** out, size_t slack_bytes, std::string* err) {
    void* p = nullptr;
    size_t bytes = v.size() * sizeof(T) + slack_bytes;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
        if (err) *err = "hipMalloc failed in mfma_pack_weights";
        return -1;
    }
    n->bufs[n->nbufs++] = p;
    if (slack_bytes) (void)hipMemset(p, 0, bytes);
    if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        if (err) *err = "hipMemcpy failed in mfma_pack_weights";
        return -1;
    }
    *out = p;
    return 0;
}

}  // namespace

MfmaNet* mfma_create(const NetDims& d, int /*max_batch*/, std::string* err) {
    if (!shape_supported(d)) {
        if (err)
            *err = "bf16 MFMA path supports numChannels=128 / numBottleneckChannels=64 (Go 9x9, PegSolitair); "
                   "use MZ_PRECISION_FP32 for other shapes";
        return nullptr;
    }
    MfmaNet* n = new MfmaNet();
    n->d = d;
    n->NB = COLS / d.HW;
    if (n->NB > MAXNB) n->NB = MAXNB;
    n->L = make_plan(d, n->NB);
    const size_t head_need = ((size_t)MAXNB * (3 * d.HW + d.A + d.C) + 3 * d.C) * 4;
    if (n->L.total > 160 * 1024 || head_need > (size_t)n->L.colsum_off) {
        if (err) *err = "LDS plan does not fit 160 KiB";
        delete n;
        return nullptr;
    }
    return n;
}

void mfma_destroy(MfmaNet* n) {
    if (!n) return;
    for (int i = 0; i < n->nbufs; i++) (void)hipFree(n->bufs[i]);
    delete n;
}

int mfma_pack_weights(MfmaNet* n, const NetDims& d, const float* blob, size_t count, std::string* err) {
    for (int i = 0; i < n->nbufs; i++) (void)hipFree(n->bufs[i]);
    n->nbufs = 0;
    n->packed = false;
    const int C = d.C, Cb = d.Cb, HW = d.HW, A = d.A;
    const int NRC = C / 32;
    size_t off = 0;
    auto take = [&](size_t cnt) {
        const float* p = blob + off;
        off += cnt;
        return p;
    };
    TowerHost th[2];
    const int cin0[2] = {d.O, C + d.a};
    const int cin0_pad[2] = {32, C + 16};
    for (int t = 0; t < 2; t++) {
        TowerHost& T = th[t];
        FragWriter fw{T.stream};
        const float* w0 = take((size_t)C * cin0[t] * 9);
        const int ks0 = cin0_pad[t] / 16;
        const int ks0p = (ks0 * NRC + RING - 1) / RING * RING / NRC;
        pack_ldsfed(fw, w0, C, cin0[t], 9, cin0_pad[t], ks0p);
        for (int i = 0; i < d.R; i++) {
            const bool bc = d.bcast_every_n > 0 && (i % d.bcast_every_n == d.bcast_every_n - 1);
            T.dev.type[i] = bc ? 1 : 0;
            T.dev.gboff[i] = (long long)(T.gb.size() / 8);
            if (!bc) {
                const float *g1 = take((size_t)C * HW), *b1 = take((size_t)C * HW), *c1 = take((size_t)Cb * C);
                const float *g2 = take((size_t)Cb * HW), *b2 = take((size_t)Cb * HW), *c2 = take((size_t)Cb * Cb * 9);
                const float *g3 = take((size_t)Cb * HW), *b3 = take((size_t)Cb * HW), *c3 = take((size_t)Cb * Cb * 9);
                const float *g4 = take((size_t)Cb * HW), *b4 = take((size_t)Cb * HW), *c4 = take((size_t)C * Cb);
                pack_regfed(fw, c1, Cb, C);
                pack_ldsfed(fw, c2, Cb, Cb, 9, Cb, Cb / 16);
                pack_ldsfed(fw, c3, Cb, Cb, 9, Cb, Cb / 16);
                pack_regfed(fw, c4, C, Cb);
                pack_gb(T.gb, g1, b1, C, HW);
                pack_gb(T.gb, g2, b2, Cb, HW);
                pack_gb(T.gb, g3, b3, Cb, HW);
                pack_gb(T.gb, g4, b4, Cb, HW);
            } else {
                const float *g1 = take((size_t)C * HW), *b1 = take((size_t)C * HW), *c1 = take((size_t)C * C);
                const float *g2 = take((size_t)C * HW), *b2 = take((size_t)C * HW);
                const float *wb = take((size_t)HW * HW), *bbv = take((size_t)HW);
                const float *g3 = take((size_t)C * HW), *b3 = take((size_t)C * HW), *c2 = take((size_t)C * C);
                pack_regfed(fw, c1, C, C);
                pack_ldsfed(fw, c2, C, C, 1, C, C / 16);
                pack_gb(T.gb, g1, b1, C, HW);
                pack_gb(T.gb, g2, b2, C, HW);
                pack_gb(T.gb, g3, b3, C, HW);
                // B fragments of the broadcast linear: lane (q = 32qt + r, h) holds Wb[q][16ks + 8h + j]
                const int KP = n->L.kp, QT = n->L.qt;
                T.dev.wboff[i] = (int)(T.wbf.size() / 8);
                T.dev.bboff[i] = (int)T.bb.size();
                T.bb.resize(T.bb.size() + (size_t)QT * 32, 0.f);
                for (int q = 0; q < HW; q++) T.bb[T.dev.bboff[i] + q] = bbv[q];
                for (int ks = 0; ks < KP / 16; ks++)
                    for (int qt = 0; qt < QT; qt++)
                        for (int l = 0; l < 64; l++)
                            for (int j = 0; j < 8; j++) {
                                const int q = 32 * qt + (l & 31), p = 16 * ks + 8 * (l >> 5) + j;
                                T.wbf.push_back((q < HW && p < HW) ? f2bf(wb[(size_t)q * HW + p]) : 0);
                            }
            }
            fw.pad_to(RING);
        }
    }
    // heads (shared by both towers): conv rows {policy0, policy1, value} appended to each stream
    const float *pconv = take((size_t)2 * C), *plng = take((size_t)2 * HW), *plnb = take((size_t)2 * HW);
    const float *pfcw = take((size_t)A * 2 * HW), *pfcb = take((size_t)A);
    const float *vconv = take((size_t)C), *vlng = take((size_t)HW), *vlnb = take((size_t)HW);
    const float *vfc1w = take((size_t)C * HW), *vfc1b = take((size_t)C), *vfc2w = take((size_t)C), *vfc2b = take(1);
    if (off != count) {
        if (err) *err = "weight blob walk mismatch (mfma)";
        return -1;
    }
    std::vector<float> hw3((size_t)3 * C);
    std::memcpy(hw3.data(), pconv, (size_t)2 * C * 4);
    std::memcpy(hw3.data() + 2 * C, vconv, (size_t)C * 4);
    for (int t = 0; t < 2; t++) {
        const void* p;
        if (upload(n, th[t].stream, &p, (size_t)RING * 1024 * 2, err)) return -1;
        th[t].dev.stream = (const u32x4*)p;
        if (upload(n, th[t].gb, &p, 64, err)) return -1;
        th[t].dev.gb = (const u32x4*)p;
        if (upload(n, th[t].wbf, &p, 64, err)) return -1;
        th[t].dev.wbf = (const u32x4*)p;
        if (th[t].bb.empty()) th[t].bb.assign(128, 0.f);
        if (upload(n, th[t].bb, &p, 64, err)) return -1;
        th[t].dev.bb = (const float*)p;
        n->tw[t] = th[t].dev;
    }
    // fp32 head parameters
    std::vector<float> hp;
    auto put = [&](const float* src, size_t cnt) {
        size_t o = hp.size();
        hp.insert(hp.end(), src, src + cnt);
        return o;
    };
    std::vector<float> pfcw_t((size_t)2 * HW * A), vfc1w_t((size_t)HW * C);
    for (int a = 0; a < A; a++)
        for (int i = 0; i < 2 * HW; i++) pfcw_t[(size_t)i * A + a] = pfcw[(size_t)a * 2 * HW + i];
    for (int ch = 0; ch < C; ch++)
        for (int p = 0; p < HW; p++) vfc1w_t[(size_t)p * C + ch] = vfc1w[(size_t)ch * HW + p];
    const size_t o_hconv = put(hw3.data(), hw3.size());
    const size_t o_plng = put(plng, 2 * HW), o_plnb = put(plnb, 2 * HW), o_pfcw = put(pfcw_t.data(), pfcw_t.size());
    const size_t o_pfcb = put(pfcb, A), o_vlng = put(vlng, HW), o_vlnb = put(vlnb, HW);
    const size_t o_vfc1w = put(vfc1w_t.data(), vfc1w_t.size()), o_vfc1b = put(vfc1b, C), o_vfc2w = put(vfc2w, C);
    const size_t o_vfc2b = put(vfc2b, 1);
    const void* p;
    if (upload(n, hp, &p, 64, err)) return -1;
    const float* hb = (const float*)p;
    n->hd = HeadDev{hb + o_hconv, hb + o_plng, hb + o_plnb, hb + o_pfcw, hb + o_pfcb, hb + o_vlng,
                    hb + o_vlnb, hb + o_vfc1w, hb + o_vfc1b, hb + o_vfc2w, hb + o_vfc2b};
    n->packed = true;
    return 0;
}

int mfma_forward(MfmaNet* n, hipStream_t st, bool initial, const NetDims& d, const NetIO& io, std::string* err) {
    if (!n || !n->packed) {
        if (err) *err = "mfma weights not packed";
        return -1;
    }
    const int grid = (io.B + n->NB - 1) / n->NB;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_net_mfma<128, 64, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_net_mfma<128, 64, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (initial)
        hipLaunchKernelGGL((k_net_mfma<128, 64, true>), dim3(grid), dim3(NT), n->L.total, st, d, n->tw[0], n->hd, io,
                           n->L, n->NB);
    else
        hipLaunchKernelGGL((k_net_mfma<128, 64, false>), dim3(grid), dim3(NT), n->L.total, st, d, n->tw[1], n->hd, io,
                           n->L, n->NB);
    return 0;
}

}  // namespace mz
