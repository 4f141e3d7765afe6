// bf16 MFMA forward of initialInference (f o h) and recurrentInference (f o g) for gfx950.
//
// One 256-thread workgroup (4 wavefronts, one per SIMD, up to 512 VGPRs each) carries a tile of NB
// whole boards through a run of layers in one launch.  Every conv is the transposed GEMM
//     Y[out-channel][column] = W[out-channel][k] * Act[k][column],   column = (board, cell),
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   * each wave owns 64 columns (2 column tiles of 32) and ALL channels of them, so the fp32 residual
//     stream X lives in that wave's accumulator registers for the whole run and the closing 1x1 conv
//     of a block accumulates straight into it;
//   * 1x1 convs that follow a LayerNorm take their B operand directly from the accumulator registers
//     (no LDS, no lane movement; the weights are pre-permuted in k to the accumulator's row order);
//   * 3x3 convs read their B operand from a bf16 [column][channel] image in LDS through 9 shifted
//     neighbour rows (a zero row for off-board cells) - the implicit GEMM, never materialised;
//   * LayerNorm over (C,H,W) of a sample = per-column partial sums -> LDS -> per-board reduction,
//     fused into the epilogue of the producing conv; ReLU, residual add, min-max rescale, softmax
//     and the value scaling are fused too;
//   * weights are packed on the host into 1 KiB MFMA A-fragments in exact execution order: one linear
//     stream per tower that each wave prefetches through a register ring (RING KiB in flight per wave)
//     across layer and block boundaries.
// A tower runs as 1..3 launches: [conv0 + bottleneck blocks] -> [broadcast block] -> [bottleneck
// blocks + rescale + heads].  The split exists for the register allocator (the broadcast block's
// live set spilled when it shared a kernel with the bottleneck blocks); between launches X travels
// through the sample's own destination hidden-state slot, which is overwritten at the end anyway.
// HBM traffic per sample: parent hidden state in, child hidden state out (+ X round trips at the
// broadcast block), logits/policy/value out.
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

#if defined(MZ_STAMPS)  // diagnostic build: per-phase cycle counts of workgroup 0 / wave 0, printed by the kernel
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += (unsigned)(t_ - st_last); st_last = t_; } while (0)
#define STAMP_DECL unsigned st_acc[12] = {0}; unsigned long long st_last = __builtin_amdgcn_s_memtime();
#else
#define STAMP(i)
#define STAMP_DECL
#endif

enum { IN_OBS = 0, IN_HIDDEN = 1, IN_X = 2 };
enum { OUT_X = 0, OUT_FINAL = 1 };

struct TowerDev {
    const u32x4* stream;   // A fragments in execution order (+ slack)
    const u32x4* gb;       // LayerNorm gamma/beta, fp16, [p][Cn/4]{g0..3,b0..3}
    const u32x4* wbf;      // broadcast-linear B fragments (bf16)
    const float* bb;       // broadcast-linear bias, padded to QT*32
    int type[kMaxResiduals];
    int foff[kMaxResiduals];         // first fragment of the block in the stream
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
    int act_off, tb_off, colsum_off, bstat_off, meta_off, head_off, total;
    int tbs;  // TB row stride in bytes
    int kp, qt;
};

#if defined(MZ_ABL_NOBARRIER)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

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

// Register ring over the linear A-fragment stream: wave-uniform position (SGPRs) + 32-bit lane offset.
// Indices passed to get() are compile-time constants after unrolling; advance() moves the origin.
struct Ring {
    u32x4 s[RING];
    const char* base;
    unsigned lane_off;
    __device__ __forceinline__ u32x4 ld(int i) const {
        return *reinterpret_cast<const u32x4*>(base + (size_t)i * 1024 + lane_off);
    }
    __device__ __forceinline__ void init(const u32x4* b, int lane) {
        base = reinterpret_cast<const char*>(b);
        lane_off = (unsigned)lane * 16u;
#pragma unroll
        for (int i = 0; i < RING; i++) s[i] = ld(i);
    }
    __device__ __forceinline__ bf16x8 get(int i) {
#if defined(MZ_ABL_NORING)
        return __builtin_bit_cast(bf16x8, s[i % RING]);
#endif
        u32x4 r = s[i % RING];
        s[i % RING] = ld(i + RING);
        // Pin the refill HERE: without this the scheduler sinks each load down to just before its use
        // (to shorten the live range) and the ring stops hiding any latency.  0x38F = everything except
        // vector-memory instructions may still be scheduled across this point.
        __builtin_amdgcn_sched_barrier(0x38F);
        return __builtin_bit_cast(bf16x8, r);
    }
    __device__ __forceinline__ void advance(int nfrags) { base += (size_t)nfrags * 1024; }
};

struct Cols {
    int col[TC];          // column index in the workgroup
    int p[TC];            // cell within the board (0 for padded columns)
    int board[TC];        // board within the tile (MAXNB for padded columns -> neutral statistics)
    bool valid[TC];       // col < NB*HW
    bool live[TC];        // valid and the board is an active sample
    unsigned nbmask[TC];  // bit t set: 3x3 neighbour t of this cell is on the board
};

struct Ctx {
    char* smem;
    LdsPlan L;
    int HW, NB, Wd, lane, wave, h, r;
    float* wpart;    // [2][4 waves][MAXNB][2] per-wave board partials (double buffered)
    int parity;
    float* bstat;    // [MAXNB+1][2] + [4][MAXNB][2] (broadcast kernel's own per-board statistics)
    int* meta;       // [4][MAXNB]: src slot, dst slot, live, action cell
};

// Wave-wide reductions on the DPP path (quad_perm / row_ror inside rows of 16 lanes, then v_readlane of the
// four row totals).  __shfl_xor compiles to ds_bpermute_b32, whose ~100-cycle latency in a dependent
// chain made every LayerNorm statistic cost thousands of cycles.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float rl(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
template <int MODE>  // 0 add, 1 min, 2 max; result is wave-uniform
__device__ __forceinline__ float wave_red(float v) {
#define MZ_OP(a, b) (MODE == 0 ? (a) + (b) : (MODE == 1 ? fminf((a), (b)) : fmaxf((a), (b))))
    v = MZ_OP(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
    v = MZ_OP(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
    v = MZ_OP(v, dpp_f<0x124>(v));  // row_ror:4
    v = MZ_OP(v, dpp_f<0x128>(v));  // row_ror:8
    const float a = MZ_OP(rl(v, 0), rl(v, 16)), b = MZ_OP(rl(v, 32), rl(v, 48));
    return MZ_OP(a, b);
#undef MZ_OP
}

// Per-board reduction of per-column partials. MODE 0: (sum, sumsq) -> (mean, rstd); MODE 1: (min, max).
// Two levels, one barrier: (1) inside each wave a masked butterfly per board that the wave's 64 columns
// touch (at most 2 boards for 9x9, more for small boards), lane b keeping board b's value; (2) lanes < NB
// publish wpart[parity][wave][board] (zeros / neutral where the wave holds none of the board), barrier,
// and every lane adds the four wave partials of its own boards.  wpart alternates between two buffers so
// the next reduction's writes cannot overtake this one's reads.
template <int MODE>
__device__ __forceinline__ void board_reduce(Ctx& c, const Cols& k, float (&a)[TC], float (&b)[TC], float n_inv) {
    const float na = MODE == 0 ? 0.f : INFINITY, nb = MODE == 0 ? 0.f : -INFINITY;
    // board range of this wave (wave-uniform): columns 64*wave .. 64*wave+63
    const int bmin = (64 * c.wave) / c.HW;
    int bmax = (64 * c.wave + 63) / c.HW;
    if (bmax > c.NB - 1) bmax = c.NB - 1;
    float ma = na, mb = nb;
    for (int bd = bmin; bd <= bmax; bd++) {
        float x = na, y = nb;
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            const bool in = k.board[ct] == bd;
            if (MODE == 0) {
                x += in ? a[ct] : 0.f;
                y += in ? b[ct] : 0.f;
            } else {
                x = fminf(x, in ? a[ct] : INFINITY);
                y = fmaxf(y, in ? b[ct] : -INFINITY);
            }
        }
        if (MODE == 0) {
            x = wave_red<0>(x);
            y = wave_red<0>(y);
        } else {
            x = wave_red<1>(x);
            y = wave_red<2>(y);
        }
        if (c.lane == bd) {
            ma = x;
            mb = y;
        }
    }
    float* wp = c.wpart + c.parity * (4 * MAXNB * 2);
    c.parity ^= 1;
    if (c.lane < c.NB) {
        wp[(c.wave * MAXNB + c.lane) * 2] = ma;
        wp[(c.wave * MAXNB + c.lane) * 2 + 1] = mb;
    }
    lds_barrier();
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        const int bd = k.board[ct] < MAXNB ? k.board[ct] : 0;
        float x = na, y = nb;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const float u = wp[(w * MAXNB + bd) * 2], v = wp[(w * MAXNB + bd) * 2 + 1];
            if (MODE == 0) {
                x += u;
                y += v;
            } else {
                x = fminf(x, u);
                y = fmaxf(y, v);
            }
        }
        if (MODE == 0) {
            const float mean = x * n_inv;
            const float var = fmaxf(y * n_inv - mean * mean, 0.f);
            a[ct] = k.valid[ct] ? mean : 0.f;
            b[ct] = k.valid[ct] ? 1.0f / sqrtf(var + 1e-5f) : 0.f;   // padded columns: neutral statistics
        } else {
            a[ct] = k.valid[ct] ? x : 0.f;
            b[ct] = k.valid[ct] ? y : 1.f;
        }
    }
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;

// LayerNorm statistics of NRT row tiles (all channels) x own columns -> per-column (mean, rstd).
template <int NRT>
__device__ __forceinline__ void ln_stats(Ctx& c, const Cols& k, const f32x16 (&v)[NRT][TC], int nch,
                                         float (&mean)[TC], float (&rstd)[TC]) {
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        f32x2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < NRT; rt++)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2_t x = {v[rt][ct][i], v[rt][ct][i + 1]};
                s2 += x;
                q2 = x * x + q2;
            }
        mean[ct] = k.valid[ct] ? s2.x + s2.y : 0.f;
        rstd[ct] = k.valid[ct] ? q2.x + q2.y : 0.f;
    }
    board_reduce<0>(c, k, mean, rstd, 1.0f / (float)(nch * c.HW));
}

// gamma/beta of row tile rt for both column tiles: 4 x 16-byte loads per column
// (uniform base + 32-bit per-lane offset + immediate).
__device__ __forceinline__ void gb_load(const u32x4* gb, int cn4, Ctx& c, const Cols& k, int rt,
                                        u32x4 (&g)[TC][4]) {
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        const unsigned off = (unsigned)(k.p[ct] * cn4 + c.h) * 16u;
#pragma unroll
        for (int qd = 0; qd < 4; qd++)
            g[ct][qd] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(gb) + off + (8 * rt + 2 * qd) * 16);
    }
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

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;

// Packed variant for the hot blocks: o[j] = bf16x2 of relu(LN(v)) rows (2j, 2j+1).  Per element: half a
// v_pk_fma_f32 (normalise), one v_fma_mix_f32 (fp16 gamma/beta), half a v_cvt_pk_bf16_f32 and half a
// v_pk_max_i16 (ReLU on the packed bf16 pair: a negative bf16 is a negative int16).
__device__ __forceinline__ void ln_apply_pk(const f32x16& v, float rstd, float nmr, const u32x4 (&g)[4],
                                            unsigned (&o)[8]) {
    const f32x2 a2 = {rstd, rstd}, n2 = {nmr, nmr};
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        f16x8 hv = __builtin_bit_cast(f16x8, g[qd]);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            f32x2 x = {v[4 * qd + e], v[4 * qd + e + 1]};
            f32x2 t = x * a2 + n2;
            // two scalar v_fma_mix_f32 (f32 * f16 + f16, no conversions) feeding one v_cvt_pk_bf16_f32; the
            // asm also keeps the SLP vectoriser from turning the pair into v_pk_fma_f32 + 4 x v_cvt_f32_f16
            const float y0 = fmaf(t.x, (float)hv[e], (float)hv[4 + e]);
            const float y1 = fmaf(t.y, (float)hv[e + 1], (float)hv[4 + e + 1]);
            unsigned pk;
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(y0), "v"(y1));
            s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), (s16x2){0, 0});
            o[2 * qd + e / 2] = __builtin_bit_cast(unsigned, r);
        }
    }
}
__device__ __forceinline__ bf16x8 frag_of(const unsigned (&o)[8], int s) {
    u32x4 r = {o[4 * s], o[4 * s + 1], o[4 * s + 2], o[4 * s + 3]};
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ void act_store_pk(char* act, int rs, int col, int rt, int h, const unsigned (&o)[8]) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        u32x2 w = {o[2 * qd], o[2 * qd + 1]};
        *reinterpret_cast<u32x2*>(act + col * rs + (32 * rt + 8 * qd + 4 * h) * 2) = w;
    }
}

// Write the 16 rows of (row tile rt, column tile) as bf16 into the [column][channel] LDS image.
__device__ __forceinline__ void act_store(char* act, int rs, int col, int rt, int h, const float (&y)[16]) {
#pragma unroll
    for (int qd = 0; qd < 4; qd++)
        *reinterpret_cast<bf16x4*>(act + col * rs + (32 * rt + 8 * qd + 4 * h) * 2) = pack4(&y[4 * qd]);
}

#if defined(MZ_ABL_NOMFMA)   // diagnostic builds only (tools/ablate.sh): keep operands alive, drop the matrix op
__device__ __forceinline__ f32x16 mfma_stub(bf16x8 a, bf16x8 b, f32x16 c) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
}
#define MFMA(a, b, c) mfma_stub((a), (b), (c))
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif

// out[ro][ct] += W * relu(LN(in)) with the B operand taken from the accumulator registers `in`.
// Fragment order in the stream: (rt_in, s, ro).
template <int NRI, int NRO>
__device__ __forceinline__ void conv1x1_regfed_core(Ctx& c, const Cols& k, Ring& ring, int& fi, const u32x4* gb,
                                                    const f32x16 (&in)[NRI][TC], f32x16 (&out)[NRO][TC],
                                                    const float (&mean)[TC], const float (&rstd)[TC]) {
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
template <int NRI, int NRO>
__device__ __forceinline__ void conv1x1_regfed(Ctx& c, const Cols& k, Ring& ring, int& fi, const u32x4* gb,
                                               const f32x16 (&in)[NRI][TC], f32x16 (&out)[NRO][TC]) {
    float mean[TC], rstd[TC];
    ln_stats<NRI>(c, k, in, NRI * 32, mean, rstd);
    conv1x1_regfed_core<NRI, NRO>(c, k, ring, fi, gb, in, out, mean, rstd);
}

__device__ __forceinline__ int nb_row(Ctx& c, const Cols& k, int ct, int t) {
    return ((k.nbmask[ct] >> t) & 1u) ? k.col[ct] + (t / 3 - 1) * c.Wd + (t % 3 - 1) : ZROW;
}

// LN + ReLU of `in` (NRI row tiles) into the LDS image, then out += conv3x3(image).
// Fragment order: (tap, ks, ro).
template <int NRI, int NRO>
__device__ __forceinline__ void conv3x3_ldsfed(Ctx& c, const Cols& k, Ring& ring, int& fi, const u32x4* gb,
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
    for (int t = 0; t < 9; t++) {
        int row[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) row[ct] = nb_row(c, k, ct, t);
#pragma unroll
        for (int ks = 0; ks < NRI * 2; ks++) {
            bf16x8 b[TC];
#pragma unroll
            for (int ct = 0; ct < TC; ct++)
                b[ct] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(act + row[ct] * RS + (16 * ks + 8 * c.h) * 2));
#pragma unroll
            for (int ro = 0; ro < NRO; ro++) {
                bf16x8 a = ring.get(fi++);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) out[ro][ct] = MFMA(a, b[ct], out[ro][ct]);
            }
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

// Two-deep register ring of gamma/beta chunks (one chunk = the 8 x 16-byte loads of one LayerNorm row
// tile for both column tiles).  Chunk j+2 is requested the moment chunk j has been consumed and pinned
// there, so that by the time it is needed at least a row tile's worth of weight fragments have been
// requested after it and the wait for it is a counted vmcnt, never a drain of the weight ring.
// Chunk order of a bottleneck block: LN1 rt 0..NRC-1, LN2 rt 0..NRB-1, LN3 .., LN4 ..
template <int C, int CB>
struct GRing {
    static constexpr int NRC = C / 32, NRB = CB / 32, NCH = NRC + 3 * NRB;
    u32x4 g[2][TC][4];
    unsigned lane_off;  // (wave * TC * 4 * 64 + lane) * 16 bytes
    __device__ __forceinline__ void init(Ctx& c, const Cols&) {
        lane_off = (unsigned)(c.wave * TC * 4 * 64 + c.lane) * 16u;
    }
    // chunk j (0 <= j < 2*NCH; j >= NCH addresses the NEXT block) into slot j % 2.
    // Memory layout of a bottleneck block's gamma/beta (pack_gb_cols): per LayerNorm, per row tile,
    // [wave][ct][qd][lane] x 16 bytes - every wave-load is one contiguous KiB.
    __device__ __forceinline__ void load(int j, const char* gbb, const char* gbn) {
        const char* base = j >= NCH ? gbn : gbb;
        const int jj = j >= NCH ? j - NCH : j;
        const int chunk_bytes = 4 * TC * 4 * 64 * 16;  // one row tile: 4 waves x TC x 4 qd x 1 KiB
#pragma unroll
        for (int ct = 0; ct < TC; ct++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                g[j % 2][ct][qd] = *reinterpret_cast<const u32x4*>(base + (size_t)jj * chunk_bytes + lane_off +
                                                                   (ct * 4 + qd) * 1024);
        __builtin_amdgcn_sched_barrier(0x38F);
    }
};

// BottleneckResidualBlock: x + c4(relu(LN(c3(relu(LN(c2(relu(LN(c1(relu(LN(x))))))))))))
// On entry the gamma/beta ring holds chunks 0 and 1 of this block; on exit chunks 0 and 1 of the next.
template <int C, int CB>
__device__ __forceinline__ void bottleneck(Ctx& c, const Cols& k, Ring& ring, GRing<C, CB>& gr, const char* gbb,
                                           const char* gbn, f32x16 (&X)[C / 32][TC]
#if defined(MZ_STAMPS)
                                           , unsigned (&st_acc)[12], unsigned long long& st_last
#endif
                                           ) {
    constexpr int NRC = C / 32, NRB = CB / 32;
    constexpr int RS = CB * 2 + 16;
    char* act = c.smem + c.L.act_off;
    int fi = 0, gj = 0;
    float mean[TC], rstd[TC];
    f32x16 a1[NRB][TC], a2[NRB][TC];

    // ---- LN1 + c1 (C -> CB), B operand straight from X ----
    STAMP(0);
    ln_stats<NRC>(c, k, X, C, mean, rstd);
    STAMP(1);
    zero_acc<NRB>(a1);
#pragma unroll
    for (int rt = 0; rt < NRC; rt++) {
        unsigned o[TC][8];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) ln_apply_pk(X[rt][ct], rstd[ct], -mean[ct] * rstd[ct], gr.g[gj % 2][ct], o[ct]);
        gr.load(gj + 2, gbb, gbn);
        gj++;
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int ro = 0; ro < NRB; ro++) {
                bf16x8 a = ring.get(fi++);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) a1[ro][ct] = MFMA(a, frag_of(o[ct], s), a1[ro][ct]);
            }
    }
    // ---- LN2 -> image -> c2 (3x3), LN3 -> image -> c3 (3x3) ----
#pragma unroll
    for (int layer = 0; layer < 2; layer++) {
        f32x16 (&in)[NRB][TC] = layer == 0 ? a1 : a2;
        f32x16 (&out)[NRB][TC] = layer == 0 ? a2 : a1;
        STAMP(2);
        ln_stats<NRB>(c, k, in, CB, mean, rstd);
        STAMP(3);
#pragma unroll
        for (int rt = 0; rt < NRB; rt++) {
#pragma unroll
            for (int ct = 0; ct < TC; ct++) {
                // padded columns: gamma = beta = 0 in the packed table and (mean, rstd) = (0, 0) => zeros
                unsigned o[8];
                ln_apply_pk(in[rt][ct], rstd[ct], -mean[ct] * rstd[ct], gr.g[gj % 2][ct], o);
                act_store_pk(act, RS, k.col[ct], rt, c.h, o);
            }
            gr.load(gj + 2, gbb, gbn);
            gj++;
        }
        STAMP(4);
        lds_barrier();
        STAMP(5);
        zero_acc<NRB>(out);
        // B fragments are read one k-step ahead of the MFMAs that use them (LDS latency hidden behind
        // the 4 MFMAs of the current k-step); the DS reads are pinned above the MFMAs.
        constexpr int KS = NRB * 2, NK = 9 * KS;
        bf16x8 bcur[TC], bnxt[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++)
            bcur[ct] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(act + nb_row(c, k, ct, 0) * RS + (8 * c.h) * 2));
#pragma unroll
        for (int kk = 0; kk < NK; kk++) {
            if (kk + 1 < NK) {
                const int t = (kk + 1) / KS, ks = (kk + 1) % KS;
#pragma unroll
                for (int ct = 0; ct < TC; ct++)
                    bnxt[ct] = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const u32x4*>(act + nb_row(c, k, ct, t) * RS + (16 * ks + 8 * c.h) * 2));
                __builtin_amdgcn_sched_barrier(0x7F);  // everything but DS instructions may cross
            }
#pragma unroll
            for (int ro = 0; ro < NRB; ro++) {
                bf16x8 a = ring.get(fi++);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) out[ro][ct] = MFMA(a, bcur[ct], out[ro][ct]);
            }
#pragma unroll
            for (int ct = 0; ct < TC; ct++) bcur[ct] = bnxt[ct];
        }
    }
    // ---- LN4 + c4 (CB -> C) accumulated into X ----
    STAMP(6);
    ln_stats<NRB>(c, k, a1, CB, mean, rstd);
    STAMP(7);
#pragma unroll
    for (int rt = 0; rt < NRB; rt++) {
        unsigned o[TC][8];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) ln_apply_pk(a1[rt][ct], rstd[ct], -mean[ct] * rstd[ct], gr.g[gj % 2][ct], o[ct]);
        gr.load(gj + 2, gbb, gbn);
        gj++;
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int ro = 0; ro < NRC; ro++) {
                bf16x8 a = ring.get(fi++);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) X[ro][ct] = MFMA(a, frag_of(o[ct], s), X[ro][ct]);
            }
    }
    STAMP(8);
    ring.advance(fi);  // 4*NRC*NRB + 36*NRB*NRB fragments: a multiple of RING for the supported shapes
}

// ---- shared prologue: per-board meta data and per-lane column info -----------------------------
template <int IN>
__device__ __forceinline__ bool setup(const NetDims& d, const NetIO& io, int NB, Ctx& c, Cols& k) {
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * NB;
    if (tid < MAXNB) {
        const int g = g0 + tid;
        const bool ok = tid < NB && g < io.B && (!io.active || io.active[g]);
        c.meta[tid] = (ok && IN == IN_HIDDEN) ? io.src_slot[g] : 0;
        c.meta[MAXNB + tid] = ok ? io.dst_slot[g] : 0;
        c.meta[2 * MAXNB + tid] = ok ? 1 : 0;
        c.meta[3 * MAXNB + tid] = (ok && IN == IN_HIDDEN) ? io.action_cells[io.action[g]] : -1;
    }
    lds_barrier();
    int any = 0;
    for (int b = 0; b < NB; b++) any |= c.meta[2 * MAXNB + b];
    if (!any) return false;
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        const int col = 64 * c.wave + 32 * ct + c.r;
        k.col[ct] = col;
        k.valid[ct] = col < NB * d.HW;
        k.board[ct] = k.valid[ct] ? col / d.HW : MAXNB;
        k.p[ct] = k.valid[ct] ? col - k.board[ct] * d.HW : 0;
        k.live[ct] = k.valid[ct] && c.meta[2 * MAXNB + k.board[ct]];
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int y = k.p[ct] / d.Wd + t / 3 - 1, x = k.p[ct] % d.Wd + t % 3 - 1;
            if (k.valid[ct] && y >= 0 && y < d.H && x >= 0 && x < d.Wd) m |= 1u << t;
        }
        k.nbmask[ct] = m;
    }
    return true;
}

__device__ __forceinline__ void ctx_init(Ctx& c, char* smem, const LdsPlan& L, const NetDims& d, int NB) {
    const int tid = threadIdx.x;
    c.smem = smem;
    c.L = L;
    c.HW = d.HW;
    c.NB = NB;
    c.Wd = d.Wd;
    c.lane = tid & 63;
    c.wave = tid >> 6;
    c.h = c.lane >> 5;
    c.r = c.lane & 31;
    c.wpart = reinterpret_cast<float*>(smem + L.colsum_off);
    c.parity = 0;
    c.bstat = reinterpret_cast<float*>(smem + L.bstat_off);
    c.meta = reinterpret_cast<int*>(smem + L.meta_off);
}

// X <-> [slot][C][HW] fp32 in the hidden-state pool (slot index from meta row `which`).
template <int NRC>
__device__ __forceinline__ void x_load(Ctx& c, const Cols& k, const float* pool, int which, int C,
                                       f32x16 (&X)[NRC][TC]) {
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        // unconditional loads (dead columns read slot meta[..][0], which is always a valid slot) + select:
        // a per-load branch would serialise 256 dependent round trips
        const int bd = k.valid[ct] ? k.board[ct] : 0;
        const float* src = pool + (long long)c.meta[which * MAXNB + bd] * C * c.HW + k.p[ct];
        const bool keep = k.live[ct];
#pragma unroll
        for (int rt = 0; rt < NRC; rt++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int ch = 32 * rt + (i & 3) + 8 * (i >> 2) + 4 * c.h;
                const float v = src[ch * c.HW];
                X[rt][ct][i] = keep ? v : 0.f;  // select, not multiply: dead slots may hold NaN bit patterns
            }
    }
}
template <int NRC>
__device__ __forceinline__ void x_store(Ctx& c, const Cols& k, float* pool, int C, const f32x16 (&X)[NRC][TC]) {
#pragma unroll
    for (int ct = 0; ct < TC; ct++) {
        if (!k.live[ct]) continue;
        float* dst = pool + (long long)c.meta[MAXNB + k.board[ct]] * C * c.HW + k.p[ct];
#pragma unroll
        for (int rt = 0; rt < NRC; rt++)
#pragma unroll
            for (int i = 0; i < 16; i++) dst[(long long)(32 * rt + (i & 3) + 8 * (i >> 2) + 4 * c.h) * c.HW] = X[rt][ct][i];
    }
}

// ---- [conv0] + bottleneck blocks b0..b1-1 + [rescale, heads] -----------------------------------
template <int C, int CB, int IN, int OUT>
__global__ __launch_bounds__(NT, 1) void k_seg(NetDims d, TowerDev tw, HeadDev hd, NetIO io, LdsPlan L, int NB, int b0,
                                               int b1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NRC = C / 32;
    constexpr int CIN0 = IN == IN_OBS ? 32 : C + 16;  // input channels of conv0, padded to 16
    constexpr int KS0 = CIN0 / 16;
    constexpr int F0 = (9 * KS0 * NRC + RING - 1) / RING * RING;  // conv0 fragments in the stream (padded)
    constexpr int RS0 = CIN0 * 2 + 16;
    constexpr int RSB = CB * 2 + 16;
    const int tid = threadIdx.x;
    Ctx c;
    ctx_init(c, smem, L, d, NB);
    Cols k;
    if (!setup<IN>(d, io, NB, c, k)) return;
    const int HW = d.HW, A = d.A;
    const int g0 = blockIdx.x * NB;
    char* act = smem + L.act_off;

    Ring ring;
    f32x16 X[NRC][TC];
    STAMP_DECL
    if (IN == IN_X) {
        ring.init(tw.stream + (long long)tw.foff[b0] * 64, c.lane);
        x_load<NRC>(c, k, io.pool, 1, C, X);
    } else {
        ring.init(tw.stream, c.lane);
        // ---------------- input image + conv0 ----------------
        for (int i = tid; i < RS0 / 4; i += NT) reinterpret_cast<int*>(act + ZROW * RS0)[i] = 0;
        if (IN == IN_OBS) {
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
            x_load<NRC>(c, k, io.pool, 0, C, X);   // StartResidualBlock: X starts as the parent state s
#pragma unroll
            for (int ct = 0; ct < TC; ct++) {
#pragma unroll
                for (int rt = 0; rt < NRC; rt++) {
                    float y[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) y[i] = X[rt][ct][i];
                    act_store(act, RS0, k.col[ct], rt, c.h, y);
                }
                // ConcatInputsBlock: the one-hot action planes as channels C .. C+a-1 (padded to 16)
                if (c.h == ct) {
                    float y[16];
                    const int cell = c.meta[3 * MAXNB + (k.valid[ct] ? k.board[ct] : 0)];
#pragma unroll
                    for (int i = 0; i < 16; i++) y[i] = (k.live[ct] && i < d.a && cell == i * HW + k.p[ct]) ? 1.f : 0.f;
                    *reinterpret_cast<bf16x8*>(act + k.col[ct] * RS0 + C * 2) = pack8(&y[0]);
                    *reinterpret_cast<bf16x8*>(act + k.col[ct] * RS0 + (C + 8) * 2) = pack8(&y[8]);
                }
            }
        }
        lds_barrier();
        {
            // conv0, fully unrolled: B fragments read one k-step ahead, fragment order (tap, ks, ro),
            // padded once at the end to a multiple of RING
            constexpr int NK = 9 * KS0;
            bf16x8 bcur[TC], bnxt[TC];
#pragma unroll
            for (int ct = 0; ct < TC; ct++)
                bcur[ct] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(act + nb_row(c, k, ct, 0) * RS0 + (8 * c.h) * 2));
            int fi = 0;
#pragma unroll
            for (int kk = 0; kk < NK; kk++) {
                if (kk + 1 < NK) {
                    const int t = (kk + 1) / KS0, ks = (kk + 1) % KS0;
#pragma unroll
                    for (int ct = 0; ct < TC; ct++)
                        bnxt[ct] = __builtin_bit_cast(
                            bf16x8, *reinterpret_cast<const u32x4*>(act + nb_row(c, k, ct, t) * RS0 + (16 * ks + 8 * c.h) * 2));
                    __builtin_amdgcn_sched_barrier(0x7F);
                }
#pragma unroll
                for (int ro = 0; ro < NRC; ro++) {
                    bf16x8 a = ring.get(fi++);
#pragma unroll
                    for (int ct = 0; ct < TC; ct++) X[ro][ct] = MFMA(a, bcur[ct], X[ro][ct]);
                }
#pragma unroll
                for (int ct = 0; ct < TC; ct++) bcur[ct] = bnxt[ct];
            }
#pragma unroll
            for (int i = NK * NRC; i < F0; i++) (void)ring.get(i);  // padding fragments of the stream
            ring.advance(F0);
        }
        lds_barrier();
    }
    // zero row of the tower's act image
    for (int i = tid; i < RSB / 4; i += NT) reinterpret_cast<int*>(act + ZROW * RSB)[i] = 0;
    lds_barrier();

    if (b0 < b1) {
        GRing<C, CB> gr;
        gr.init(c, k);
        const char* gbase = reinterpret_cast<const char*>(tw.gb);
        gr.load(0, gbase + tw.gboff[b0] * 16, nullptr);
        gr.load(1, gbase + tw.gboff[b0] * 16, nullptr);
        for (int blk = b0; blk < b1; blk++) {
            // the chunks requested past the end of the last block are never used: point them at valid memory
            const int nxt = (blk + 1 < b1) ? blk + 1 : blk;
#if defined(MZ_STAMPS)
            bottleneck<C, CB>(c, k, ring, gr, gbase + tw.gboff[blk] * 16, gbase + tw.gboff[nxt] * 16, X, st_acc, st_last);
#else
            bottleneck<C, CB>(c, k, ring, gr, gbase + tw.gboff[blk] * 16, gbase + tw.gboff[nxt] * 16, X);
#endif
        }
    }
#if defined(MZ_STAMPS)
    if (blockIdx.x == 0 && tid == 0)
        printf("k_seg<%d,%d> blocks %d..%d stamps(cycles, summed over blocks): pre=%u ln1stats=%u c1=%u lnstats23=%u apply23=%u bar=%u conv33=%u ln4stats=%u c4=%u\n",
               IN, OUT, b0, b1, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5], st_acc[6], st_acc[7], st_acc[8]);
#endif

    if (OUT == OUT_X) {
        x_store<NRC>(c, k, io.pool, C, X);
        return;
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
#pragma unroll
            for (int rt = 0; rt < NRC; rt++)
#pragma unroll
                for (int i = 0; i < 16; i++) X[rt][ct][i] = (X[rt][ct][i] - mn[ct]) / dd;
        }
        x_store<NRC>(c, k, io.pool, C, X);
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
    float hp0[TC], hp1[TC], hvv[TC];
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
        hp0[ct] = p0 + __shfl_xor(p0, 32);
        hp1[ct] = p1 + __shfl_xor(p1, 32);
        hvv[ct] = v + __shfl_xor(v, 32);
    }
    {
        // policy LN over (2, HW), value LN over (1, HW)
        float s[TC], q[TC], vs[TC], vq[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            const bool own = k.valid[ct] && c.h == 0;
            s[ct] = own ? hp0[ct] + hp1[ct] : 0.f;
            q[ct] = own ? hp0[ct] * hp0[ct] + hp1[ct] * hp1[ct] : 0.f;
            vs[ct] = own ? hvv[ct] : 0.f;
            vq[ct] = own ? hvv[ct] * hvv[ct] : 0.f;
        }
        board_reduce<0>(c, k, s, q, 1.0f / (float)(2 * HW));
        board_reduce<0>(c, k, vs, vq, 1.0f / (float)HW);
#pragma unroll
        for (int ct = 0; ct < TC; ct++) {
            if (k.valid[ct] && c.h == 0) {
                const int bd = k.board[ct], p = k.p[ct];
                float t0 = (hp0[ct] - s[ct]) * q[ct] * hd.plng[p] + hd.plnb[p];
                float t1 = (hp1[ct] - s[ct]) * q[ct] * hd.plng[HW + p] + hd.plnb[HW + p];
                pfeat[bd * 2 * HW + p] = fmaxf(t0, 0.f);
                pfeat[bd * 2 * HW + HW + p] = fmaxf(t1, 0.f);
                float t = (hvv[ct] - vs[ct]) * vq[ct] * hd.vlng[p] + hd.vlnb[p];
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
        if (!c.meta[2 * MAXNB + bd]) continue;
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

// ---- BroadcastResidualBlock: x + c2(relu(LN(bcast(relu(LN(c1(relu(LN(x))))))))) ------------------
template <int C>
__global__ __launch_bounds__(NT, 1) void k_bcast(NetDims d, TowerDev tw, NetIO io, LdsPlan L, int NB, int blk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NRC = C / 32;
    constexpr int NH = NRC / 2;
    constexpr int RS = C * 2 + 16;
    Ctx c;
    ctx_init(c, smem, L, d, NB);
    Cols k;
    if (!setup<IN_X>(d, io, NB, c, k)) return;
    const u32x4* gb = tw.gb + tw.gboff[blk];
    const u32x4* wbf = tw.wbf + tw.wboff[blk];
    const float* bb = tw.bb + tw.bboff[blk];
    const long long s1 = (long long)c.HW * (C / 4);
    const int KP = L.kp, QT = L.qt, TBS = L.tbs;
    char* tb = smem + L.tb_off;
    char* act = smem + L.act_off;
    Ring ring;
    ring.init(tw.stream + (long long)tw.foff[blk] * 64, c.lane);
    f32x16 X[NRC][TC];
    x_load<NRC>(c, k, io.pool, 1, C, X);
    int fi = 0;
    {
        // c1 (C -> C) is computed in two halves of C/2 output channels, TWICE: once for the LN2 statistics
        // and once more to normalise (keeping all C output rows next to X costs 128 more accumulators).
        // The stream holds c1's fragments twice, in this order.
        float m1[TC], r1[TC];
        ln_stats<NRC>(c, k, X, C, m1, r1);
        float mean[TC], rstd[TC];
#pragma unroll
        for (int ct = 0; ct < TC; ct++) mean[ct] = rstd[ct] = 0.f;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x16 ah[NH][TC];
            zero_acc<NH>(ah);
            conv1x1_regfed_core<NRC, NH>(c, k, ring, fi, gb, X, ah, m1, r1);
#pragma unroll
            for (int ct = 0; ct < TC; ct++)
                if (k.valid[ct]) {
#pragma unroll
                    for (int rt = 0; rt < NH; rt++)
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const float x = ah[rt][ct][i];
                            mean[ct] += x;
                            rstd[ct] = fmaf(x, x, rstd[ct]);
                        }
                }
        }
        board_reduce<0>(c, k, mean, rstd, 1.0f / (float)(C * c.HW));
        for (int i = threadIdx.x; i < C * NB * (KP - c.HW); i += NT) {  // zero the K padding of every board
            const int padw = KP - c.HW;
            const int ch = i / (NB * padw), rem = i - ch * (NB * padw);
            const int bd = rem / padw, pp = rem - bd * padw;
            *reinterpret_cast<__bf16*>(tb + ch * TBS + (bd * KP + c.HW + pp) * 2) = (__bf16)0.f;
        }
        // LN2 + ReLU -> bf16 -> transposed image TB[channel][board*KP + cell]
#pragma unroll
        for (int half = 0; half < 2; half++) {
            f32x16 ah[NH][TC];
            zero_acc<NH>(ah);
            conv1x1_regfed_core<NRC, NH>(c, k, ring, fi, gb, X, ah, m1, r1);
#pragma unroll
            for (int rt = 0; rt < NH; rt++) {
                u32x4 g[TC][4];
                gb_load(gb + s1, C / 4, c, k, half * NH + rt, g);
#pragma unroll
                for (int ct = 0; ct < TC; ct++) {
                    float y[16];
                    ln_apply(ah[rt][ct], mean[ct], rstd[ct], g[ct], y);
                    if (k.valid[ct]) {
                        const int cbase = k.board[ct] * KP + k.p[ct];
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int ch = 32 * (half * NH + rt) + (i & 3) + 8 * (i >> 2) + 4 * c.h;
                            *reinterpret_cast<__bf16*>(tb + ch * TBS + cbase * 2) = (__bf16)y[i];
                        }
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
        for (int bd = 0; bd < NB; bd++) {
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
            if ((int)threadIdx.x < NB) {
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
    x_store<NRC>(c, k, io.pool, C, X);
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

// gamma/beta [Cn][HW] for the bottleneck kernels: per row tile, [wave][ct][qd][lane]{g0..3, b0..3} fp16, where
// lane (r, h) of (wave, ct) owns column 64*wave + 32*ct + r = (board, cell p) and channels 32rt + 8qd + 4h + e.
void pack_gb_cols(std::vector<unsigned short>& out, const float* g, const float* b, int Cn, int HW, int NB) {
    for (int rt = 0; rt < Cn / 32; rt++)
        for (int w = 0; w < 4; w++)
            for (int ct = 0; ct < TC; ct++)
                for (int qd = 0; qd < 4; qd++)
                    for (int l = 0; l < 64; l++) {
                        const int col = 64 * w + 32 * ct + (l & 31), h = l >> 5;
                        const bool valid = col < NB * HW;
                        const int p = valid ? col % HW : 0;
                        const int ch = 32 * rt + 8 * qd + 4 * h;
                        for (int e = 0; e < 4; e++) out.push_back(valid ? f2h(g[(size_t)(ch + e) * HW + p]) : 0);
                        for (int e = 0; e < 4; e++) out.push_back(valid ? f2h(b[(size_t)(ch + e) * HW + p]) : 0);
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
    LdsPlan Lseg{}, Lbc{};
    void* bufs[16] = {nullptr};
    int nbufs = 0;
    TowerDev tw[2]{};
    HeadDev hd{};
    bool packed = false;
};

namespace {

bool shape_supported(const NetDims& d) { return d.C == 128 && d.Cb == 64 && d.O <= 32 && d.a <= 16 && d.HW <= 81 + 15; }

void plan_tail(LdsPlan& L, int end) {
    end = (end + 15) / 16 * 16;
    L.colsum_off = end;
    end += COLS * 2 * 4;
    L.bstat_off = end;
    end += (2 * (MAXNB + 1) + 4 * MAXNB * 2) * 4;
    L.meta_off = end;
    end += 4 * MAXNB * 4;
    L.total = (end + 15) / 16 * 16;
}

// k_seg: act image (conv0 input image / bottleneck image / head scratch share offset 0)
LdsPlan make_plan_seg(const NetDims& d) {
    LdsPlan L{};
    const int rs0 = (d.C + 16) * 2 + 16, rsb = d.Cb * 2 + 16;
    int end = (COLS + 1) * (rs0 > rsb ? rs0 : rsb);
    const int head_need = (MAXNB * (3 * d.HW + d.A + d.C) + 3 * d.C) * 4;
    if (head_need > end) end = head_need;
    L.act_off = 0;
    L.head_off = 0;
    plan_tail(L, end);
    return L;
}
// k_bcast: act image with C channels + transposed image TB
LdsPlan make_plan_bcast(const NetDims& d, int NB) {
    LdsPlan L{};
    const int rsc = d.C * 2 + 16;
    L.kp = (d.HW + 15) / 16 * 16;
    L.qt = (d.HW + 31) / 32;
    L.tbs = (NB * L.kp + 8) * 2;
    L.act_off = 0;
    L.tb_off = ((COLS + 1) * rsc + 15) / 16 * 16;
    plan_tail(L, L.tb_off + d.C * L.tbs);
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
    n->Lseg = make_plan_seg(d);
    n->Lbc = make_plan_bcast(d, n->NB);
    if (n->Lseg.total > 160 * 1024 || n->Lbc.total > 160 * 1024) {
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
        pack_ldsfed(fw, w0, C, cin0[t], 9, cin0_pad[t], cin0_pad[t] / 16);
        fw.pad_to(RING);
        for (int i = 0; i < d.R; i++) {
            const bool bc = d.bcast_every_n > 0 && (i % d.bcast_every_n == d.bcast_every_n - 1);
            T.dev.type[i] = bc ? 1 : 0;
            T.dev.gboff[i] = (long long)(T.gb.size() / 8);
            T.dev.foff[i] = (int)(T.stream.size() / 512);
            if (!bc) {
                const float *g1 = take((size_t)C * HW), *b1 = take((size_t)C * HW), *c1 = take((size_t)Cb * C);
                const float *g2 = take((size_t)Cb * HW), *b2 = take((size_t)Cb * HW), *c2 = take((size_t)Cb * Cb * 9);
                const float *g3 = take((size_t)Cb * HW), *b3 = take((size_t)Cb * HW), *c3 = take((size_t)Cb * Cb * 9);
                const float *g4 = take((size_t)Cb * HW), *b4 = take((size_t)Cb * HW), *c4 = take((size_t)C * Cb);
                pack_regfed(fw, c1, Cb, C);
                pack_ldsfed(fw, c2, Cb, Cb, 9, Cb, Cb / 16);
                pack_ldsfed(fw, c3, Cb, Cb, 9, Cb, Cb / 16);
                pack_regfed(fw, c4, C, Cb);
                pack_gb_cols(T.gb, g1, b1, C, HW, n->NB);
                pack_gb_cols(T.gb, g2, b2, Cb, HW, n->NB);
                pack_gb_cols(T.gb, g3, b3, Cb, HW, n->NB);
                pack_gb_cols(T.gb, g4, b4, Cb, HW, n->NB);
            } else {
                const float *g1 = take((size_t)C * HW), *b1 = take((size_t)C * HW), *c1 = take((size_t)C * C);
                const float *g2 = take((size_t)C * HW), *b2 = take((size_t)C * HW);
                const float *wb = take((size_t)HW * HW), *bbv = take((size_t)HW);
                const float *g3 = take((size_t)C * HW), *b3 = take((size_t)C * HW), *c2 = take((size_t)C * C);
                for (int rep = 0; rep < 2; rep++)  // statistics pass, then normalising pass
                    for (int half = 0; half < 2; half++) pack_regfed(fw, c1 + (size_t)half * (C / 2) * C, C / 2, C);
                pack_ldsfed(fw, c2, C, C, 1, C, C / 16);
                pack_gb(T.gb, g1, b1, C, HW);
                pack_gb(T.gb, g2, b2, C, HW);
                pack_gb(T.gb, g3, b3, C, HW);
                // B fragments of the broadcast linear: lane (q = 32qt + r, h) holds Wb[q][16ks + 8h + j]
                const int KP = n->Lbc.kp, QT = n->Lbc.qt;
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
        if (upload(n, th[t].stream, &p, (size_t)RING * 1024 * 2, err)) return -1;  // the ring prefetches past the end
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

namespace {

template <int IN, int OUT>
void launch_seg(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int b0,
                int b1) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg<128, 64, IN, OUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_seg<128, 64, IN, OUT>), dim3(grid), dim3(NT), n->Lseg.total, st, d, tw, n->hd, io, n->Lseg,
                       n->NB, b0, b1);
}

}  // namespace

int mfma_forward(MfmaNet* n, hipStream_t st, bool initial, const NetDims& d, const NetIO& io, std::string* err) {
    if (!n || !n->packed) {
        if (err) *err = "mfma weights not packed";
        return -1;
    }
    const int grid = (io.B + n->NB - 1) / n->NB;
    const TowerDev& tw = n->tw[initial ? 0 : 1];
    // runs of bottleneck blocks separated by broadcast blocks
    int b = 0;
    bool first = true;
    while (true) {
        int e = b;
        while (e < d.R && tw.type[e] == 0) e++;
        const bool last = e == d.R;
        if (first) {
            if (initial) {
                if (last) launch_seg<IN_OBS, OUT_FINAL>(n, st, grid, d, tw, io, b, e);
                else launch_seg<IN_OBS, OUT_X>(n, st, grid, d, tw, io, b, e);
            } else {
                if (last) launch_seg<IN_HIDDEN, OUT_FINAL>(n, st, grid, d, tw, io, b, e);
                else launch_seg<IN_HIDDEN, OUT_X>(n, st, grid, d, tw, io, b, e);
            }
        } else {
            if (last) launch_seg<IN_X, OUT_FINAL>(n, st, grid, d, tw, io, b, e);
            else launch_seg<IN_X, OUT_X>(n, st, grid, d, tw, io, b, e);
        }
        first = false;
        if (last) break;
        static bool bc_attr = false;
        if (!bc_attr) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bcast<128>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            bc_attr = true;
        }
        hipLaunchKernelGGL((k_bcast<128>), dim3(grid), dim3(NT), n->Lbc.total, st, d, tw, io, n->Lbc, n->NB, e);
        b = e + 1;
    }
    return 0;
}

}  // namespace mz
