// bf16 MFMA forward of initialInference (f o h) and recurrentInference (f o g) for gfx950.
//
// One 512-thread workgroup (8 wavefronts = 2 per SIMD, <= 256 VGPRs each) carries a tile of NB whole
// boards through a run of layers in one launch.  Every conv is the transposed GEMM
//     Y[out-channel][column] = W[out-channel][k] * Act[k][column],   column = (board, cell),
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//   * The waves form a WR x WC grid (WR*WC = 8): wave (wr, wc) owns the 32 columns 32*wc .. 32*wc+31 and
//     the 1/WR share of the channels; the fp32 residual stream X (4 row tiles = 64 registers) and the
//     bottleneck activations (2 row tiles) stay in that wave's registers for the whole run, and the
//     closing 1x1 conv of a block accumulates straight into X.  numChannels 128 uses WR = 1 (256 columns
//     per workgroup), numChannels 256 uses WR = 2 (128 columns): the same register footprint.
//   * Two waves per SIMD is the point of the layout: LayerNorm is VALU work (statistics, normalise,
//     pack) between the matrix phases; one wave's VALU issues beside the other's MFMAs, a single
//     wave per SIMD issues a VALU instruction only every 4 cycles, and with <= 256 registers the
//     accumulators are ordinary VGPRs (no v_accvgpr moves around every VALU use).
//   * Every conv reads its B operand (activations after LayerNorm + ReLU, bf16) from a [column][channel]
//     image in LDS: 1x1 convs from the wave's own columns, 3x3 convs through 9 shifted neighbour rows
//     (a zero row for off-board cells) - the implicit GEMM, never materialised.
//   * Weights are packed on the host into 1 KiB MFMA A-fragments in exact execution order, one linear
//     stream per tower.  The workgroup streams them global -> LDS with global_load_lds_dwordx4 (no
//     registers) into a three-slot ring, two stages ahead; a slot holds one stage (<= 24 fragments) that
//     all 8 waves read with ds_read_b128.  One barrier per stage publishes the slot that was requested
//     two stages ago and frees the one just computed for the next prefetch.
//   * On single-row wave grids the 1x1 convs of a bottleneck block skip the image: the LayerNorm output, packed in
//     accumulator order, IS the B operand (ln_to_regs / conv_regs; A fragments packed in the matching k order).
//   * The prediction heads: their three 1x1 convs are taken off X in registers by the tower's last launch (3 floats per
//     cell to HeadDev::hfeat); the rest (tiny fp32 latency chains) runs in its own kernel, one small workgroup per four
//     samples at high occupancy.
//   * LayerNorm over (C,H,W) of a sample = per-lane partial sums -> masked DPP butterfly per board ->
//     per-wave partials in LDS -> one barrier -> per-board mean / rstd.  ReLU, residual add, min-max
//     rescale, softmax and the value scaling are fused too.
// Launches per tower (MfmaNet::fuse, MZ_FUSE): a tower with one broadcast block (Go) is ONE kernel, k_tower = conv0 +
// bottleneck blocks + broadcast block + bottleneck blocks + rescale + head convs, X in registers from end to end; towers
// with several broadcast blocks run k_seg -> k_bseg (broadcast block + the blocks behind it) -> ..., X travelling through
// the sample's own destination hidden-state slot (overwritten at the end anyway); towers with none one k_seg.  k_heads
// follows.  HBM traffic per sample on the one-kernel path: parent hidden state in, child hidden state out,
// logits/policy/value out.
//
// Reference math: see net_fp32.hip / oracle/net.py (same citations).
// This file is compiled twice (Makefile): -DMZ_F16=0 -> bf16 operands (8-bit significand, fp32's exponent range), and
// -DMZ_F16=1 -> fp16 operands (11-bit significand: 8x smaller rounding error at the same matrix rate; the operands here
// are LayerNorm outputs, [0, 1] hidden states, 0/1 planes and O(1) weights, far inside fp16's range, and gfx950's MFMA
// keeps fp16 subnormals).  Everything else - tiling, LDS layout, fp32 accumulation, fp16 gamma/beta - is shared.
#include "net_mfma.h"

#include <math.h>

#ifndef MZ_F16
#define MZ_F16 0
#endif
#if MZ_F16
#define MZ_IMPL impl_f16
#define MfmaNet MfmaNetF16
#define mfma_create mfma_create_f16
#define mfma_destroy mfma_destroy_f16
#define mfma_pack_weights mfma_pack_weights_f16
#define mfma_forward mfma_forward_f16
#define mfma_tile_boards mfma_tile_boards_f16
#else
#define MZ_IMPL impl_bf16
#define MfmaNet MfmaNetBf16
#define mfma_create mfma_create_bf16
#define mfma_destroy mfma_destroy_bf16
#define mfma_pack_weights mfma_pack_weights_bf16
#define mfma_forward mfma_forward_bf16
#define mfma_tile_boards mfma_tile_boards_bf16
#endif

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mz {

namespace MZ_IMPL {

#if MZ_F16
typedef _Float16 op_t;   // MFMA operand element
#define MZ_CVT_PK "v_cvt_pk_f16_f32"
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#else
typedef __bf16 op_t;
#define MZ_CVT_PK "v_cvt_pk_bf16_f32"
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#endif
typedef __attribute__((ext_vector_type(8))) op_t bf16x8;   // (the names predate the fp16 variant: 8 / 4 operand elements)
typedef __attribute__((ext_vector_type(4))) op_t bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;

constexpr int MAXNB = 32;    // boards per workgroup upper bound (256 / 9 = 28)
// Zero rows that off-board 3x3 neighbours read.  One shared row would put every off-board lane of a ds_read_b128 on the
// banks of some on-board lane of its 16-lane group (the 16 image rows of a group tile all 64 banks exactly): +1 LDS cycle
// per group on most taps (PMC: SQ_LDS_BANK_CONFLICT was a sixth of SQ_LDS_IDX_ACTIVE, 58 % less with this).  Instead
// there are 16 zero rows at the image's row stride, starting at the image's phase (mod 256 B): a lane reads zero row
// (row index mod 16), which lies on the very banks its on-board read would have used (16 * RS is a multiple of 256 B
// for every RS = 2C + 16).  Served that way: the bottleneck images (row stride 2 Cb + 16: the 3x3 convs c2 / c3, two
// thirds of all MFMAs); the wider images of conv0 read zero row 0 only.  LdsPlan::zbytes = ZROWS * (2 Cb + 16).
constexpr int ZROWS = 16;

#if defined(MZ_STAMPS)  // diagnostic build: per-phase cycle counts of workgroup 0 (waves 0 and 7), printed by the kernel
#define STAMP(c, i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); (c).st_acc[i] += (unsigned)(t_ - (c).st_last); (c).st_last = t_; } while (0)
#define STAMP_PRINT(c, name) do { if (blockIdx.x == 0 && ((c).wave == 0 || (c).wave == 2 || (c).wave == 4 || (c).wave == 7) && (c).lane == 0) \
    printf("%s w%d: stats %u norm %u acquire %u mma %u xload %u xstore %u rescale %u head %u other %u bcast %u | hconv %u hln %u hfc %u | sbar %u scomb %u\n", name, (c).wave, \
           (c).st_acc[0], (c).st_acc[1], (c).st_acc[2], (c).st_acc[3], (c).st_acc[4], (c).st_acc[5], (c).st_acc[6], (c).st_acc[7], (c).st_acc[8], (c).st_acc[9], (c).st_acc[10], (c).st_acc[11], (c).st_acc[12], (c).st_acc[13], (c).st_acc[14]); } while (0)
#else
#define STAMP(c, i)
#define STAMP_PRINT(c, name)
#endif
enum { ST_STATS = 0, ST_NORM, ST_ACQ, ST_MMA, ST_XLOAD, ST_XSTORE, ST_RESCALE, ST_HEAD, ST_OTHER, ST_BCAST, ST_HCONV, ST_HLN, ST_HFC,
       ST_SBAR, ST_SCOMB };

enum { IN_OBS = 0, IN_HIDDEN = 1, IN_X = 2 };
enum { OUT_X = 0, OUT_FINAL = 1, OUT_REGS = 2 };   // OUT_REGS: X stays in registers (a fused kernel goes on)

struct TowerDev {
    const u32x4* stream;   // A fragments in execution order
    const u32x4* gb;       // LayerNorm gamma/beta, fp16: per LN [rt][qd][h][HW+1]{g0..3,b0..3}
    const u32x4* wbf;      // broadcast-linear B fragments (bf16)
    const float* bb;       // broadcast-linear bias, padded to QT*32
    const float* wact;     // dynamics: conv0's weights on the action planes, fp32 [a][9][C] (see seg_front)
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
    float* hfeat;       // [max batch][3][HW]: outputs of the three head convs, written by the tower's last launch
};
// Byte offsets into the workgroup's one dynamic LDS array.
struct LdsPlan {
    int img_off, zrow_off, zbytes, zstride, ring_off, tb_off, stat_off, meta_off, head_off, total;
    int img0_off;   // the input image of conv0 (its rows are 32 B wider); = img_off except in the one-launch tower plans
    int tbs;  // TB row stride in bytes (broadcast kernel)
    int kp, qt;
};

// Workgroup geometry: NW waves as a WR x WC grid, each wave owning NC column tiles of 32 columns; ring slots of SM
// fragments (KiB); WPE = waves per SIMD the register budget is cut for.
template <int C_, int CB_, int WR_, int NW_, int SM_, int WPE_, int NC_ = 1>
struct Geo {
    static constexpr int C = C_, CB = CB_, WR = WR_, NW = NW_, SM = SM_, WPE = WPE_, NC = NC_;
    static constexpr int NT = 64 * NW;
    static constexpr int WC = NW / WR, COLS = 32 * WC * NC;
    static_assert(NW % WR == 0 && NW % 2 == 0, "wave grid");
    static constexpr int NRC = C / 32, NRB = CB / 32;       // row tiles of the whole layer
    static constexpr int NRCW = NRC / WR, NRBW = NRB / WR;  // row tiles one wave owns
    static_assert(NRC % WR == 0 && NRB % WR == 0 && NRBW >= 1, "channel split");
};

// Stage split of a conv with KK k-steps of NROT fragments: equal stages of SK k-steps, <= SM fragments.
constexpr int stage_ksteps(int KK, int NROT, int SM) {
    const int skmax = SM / NROT;
    const int nst = (KK + skmax - 1) / skmax;
    return (KK + nst - 1) / nst;
}
constexpr int num_stages(int KK, int NROT, int SM) {
    const int sk = stage_ksteps(KK, NROT, SM);
    return (KK + sk - 1) / sk;
}
// fragments of stage j of that conv (0 past its last stage)
constexpr int stage_frags(int KK, int NROT, int SM, int j) {
    const int sk = stage_ksteps(KK, NROT, SM);
    const int rem = KK - j * sk;
    return rem <= 0 ? 0 : (rem < sk ? rem : sk) * NROT;
}
constexpr int NSLOT = 3;   // ring slots: the stage being computed + two in flight

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8 pack8(const float* y) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = (op_t)y[i];
    return r;
}
__device__ __forceinline__ bf16x4 pack4(const float* y) {
    bf16x4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = (op_t)y[i];
    return r;
}

struct Col {
    int col;          // column index in the workgroup
    int p;            // cell within the board (0 for padded columns)
    int pz;           // p, or HW for padded columns (the all-zero gamma/beta entry)
    int board;        // board within the tile (MAXNB for padded columns -> neutral statistics)
    bool valid;       // col < NB*HW
    bool live;        // valid and the board is an active sample
    unsigned nbmask;  // bit t set: 3x3 neighbour t of this cell is on the board
};

struct Ctx {
    char* smem;
    LdsPlan L;
    int HW, NB, Wd, lane, wave, wc, wr, h, r;
    float* wpart;    // [2][MAXNB][8 waves][2] per-wave board partials (double buffered)
    int parity;
    int* meta;       // [4][MAXNB]: src slot, dst slot, live, action cell
#if defined(MZ_STAMPS)
    unsigned st_acc[16];
    unsigned long long st_last;
#endif
};

// Three-slot LDS ring over the linear A-fragment stream, filled by LDS-DMA two stages ahead.  Protocol (all waves):
//   issue(S0); issue(S1);  acquire() -> issue(S2) -> compute S0;  acquire() -> issue(S3) -> compute S1; ...
// acquire() waits for this wave's own DMA (vmcnt(0): the stage due now was requested two stages ago, the next
// one a whole stage ago), then the workgroup barrier makes every wave's part visible and proves that all waves
// have left the slot the next issue() overwrites (the one computed before the barrier).
template <int NW, int SM>
struct WRing {
    const char* gptr;  // next fragment to load (wave-uniform)
    char* lds;         // ring base
    unsigned ldsoff;   // the same as an LDS byte address (what M0 takes)
    int ld_idx, rd_idx, wave;
    unsigned lane16;
    __device__ __forceinline__ void init(const u32x4* stream, char* ring, int wave_, int lane) {
        gptr = reinterpret_cast<const char*>(stream);
        lds = ring;
        ldsoff = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)ring;
        ld_idx = rd_idx = 0;
        wave = wave_;
        lane16 = (unsigned)lane * 16u;
    }
    __device__ __forceinline__ void issue(int nf) {
        if (nf == 0) return;   // nothing follows: the slot stays with the next real issue()
        // The DMA is issued from inline asm on purpose: with __builtin_amdgcn_global_load_lds in flight hipcc's
        // s_waitcnt pass turns every LDS-read wait of the MFMA loops into lgkmcnt(0), which undoes their software
        // pipelining.  M0 = LDS byte address of the fragment (wave-uniform), written in the same statement.
        const unsigned dst = ldsoff + ld_idx * (SM * 1024);
#pragma unroll
        for (int i = 0; i < (SM + NW - 1) / NW; i++) {
            const int f = wave + NW * i;
            if (f < nf) {
                const unsigned m0v = dst + f * 1024;
                const char* gp = gptr + (size_t)f * 1024;   // scalar base + the lane's 32-bit offset: no vector address arithmetic
                asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(lane16), "s"(m0v), "s"(gp) : "memory");
            }
        }
        gptr += (size_t)nf * 1024;
        ld_idx = ld_idx == NSLOT - 1 ? 0 : ld_idx + 1;
    }
    __device__ __forceinline__ const char* acquire() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const char* p = lds + rd_idx * (SM * 1024);
        rd_idx = rd_idx == NSLOT - 1 ? 0 : rd_idx + 1;
        return p;
    }
};

// Wave-wide reductions on the DPP path (quad_perm / row_ror inside rows of 16 lanes, then v_readlane of the
// four row totals).
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

// Per-board reduction of per-lane partials. MODE 0: (sum, sumsq) -> (mean, rstd); MODE 1: (min, max).
// Two levels, one barrier: (1) inside each wave a masked butterfly per board that the wave's 32 columns
// touch (at most 2 boards for boards of >= 32 cells, more for small boards), lane b keeping board b's
// value; (2) lanes < NB publish wpart[parity][board][wave] (neutral where the wave holds none of the
// board), barrier, and every lane combines the eight wave partials of its own board.  wpart alternates
// between two buffers so the next reduction's writes cannot overtake this one's reads.
// VM: the barrier also waits for this wave's vector-memory operations (LDS-DMA of a gamma/beta table that the other waves
// are about to read).
template <int MODE, int NW, int NC, bool VM = false>
__device__ __forceinline__ void board_reduce(Ctx& c, const Col (&k)[NC], float (&a)[NC], float (&b)[NC], float n_inv) {
    const float na = MODE == 0 ? 0.f : INFINITY, nb = MODE == 0 ? 0.f : -INFINITY;
    const int bmin = (32 * NC * c.wc) / c.HW;
    int bmax = (32 * NC * c.wc + 32 * NC - 1) / c.HW;
    if (bmax > c.NB - 1) bmax = c.NB - 1;
    float ma = na, mb = nb;
    for (int bd = bmin; bd <= bmax; bd++) {
        float x = na, y = nb;
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            const bool in = k[ct].board == bd;
            if (MODE == 0) {
                x += in ? a[ct] : 0.f;
                y += in ? b[ct] : 0.f;
            } else {
                x = fminf(x, in ? a[ct] : na);
                y = fmaxf(y, in ? b[ct] : nb);
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
    float* wp = c.wpart + c.parity * (MAXNB * NW * 2);
    c.parity ^= 1;
    if (c.lane < c.NB) *reinterpret_cast<f32x2*>(wp + (c.lane * NW + c.wave) * 2) = (f32x2){ma, mb};
    STAMP(c, ST_STATS);
    if (VM) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else lds_barrier();
    STAMP(c, ST_SBAR);
#pragma unroll
    for (int ct = 0; ct < NC; ct++) {
        const int bd = k[ct].board < MAXNB ? k[ct].board : 0;
        float x = na, y = nb;
#pragma unroll
        for (int w4 = 0; w4 < NW / 2; w4++) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(wp + (bd * NW + 2 * w4) * 2);
            const float u0 = t[0], v0 = t[1], u1 = t[2], v1 = t[3];
            if (MODE == 0) {
                x += u0 + u1;
                y += v0 + v1;
            } else {
                x = fminf(x, fminf(u0, u1));
                y = fmaxf(y, fmaxf(v0, v1));
            }
        }
        if (MODE == 0) {
            const float mean = x * n_inv;
            const float var = fmaxf(y * n_inv - mean * mean, 0.f);
            a[ct] = k[ct].valid ? mean : 0.f;
            b[ct] = k[ct].valid ? __builtin_amdgcn_rsqf(var + 1e-5f) : 0.f;   // padded columns: neutral statistics
        } else {
            a[ct] = k[ct].valid ? x : 0.f;
            b[ct] = k[ct].valid ? y : 1.f;
        }
    }
}

// LayerNorm statistics of this wave's NR row tiles x own column -> per-column (mean, rstd) of the board.
template <int NR, int NW, int NC, bool VM = false>
__device__ __forceinline__ void ln_stats(Ctx& c, const Col (&k)[NC], const f32x16 (&v)[NR][NC], int nch, float (&mean)[NC],
                                         float (&rstd)[NC]) {
#pragma unroll
    for (int ct = 0; ct < NC; ct++) {
        f32x2 s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < NR; rt++)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 x = {v[rt][ct][i], v[rt][ct][i + 1]};
                s2 += x;
                q2 = x * x + q2;
            }
        mean[ct] = k[ct].valid ? s2.x + s2.y : 0.f;
        rstd[ct] = k[ct].valid ? q2.x + q2.y : 0.f;
    }
    board_reduce<0, NW, NC, VM>(c, k, mean, rstd, 1.0f / (float)(nch * c.HW));
}

// gamma/beta of global row tile rt for this lane's (cell, channels): 4 x 16-byte loads.
// Table layout per LayerNorm: [rt][qd][h][HW+1] x {g0..3, b0..3} fp16, entry HW is all zero.
__device__ __forceinline__ void gb_load(const u32x4* gb, Ctx& c, const Col& k, int rt, u32x4 (&g)[4]) {
    const unsigned stride = (unsigned)(c.HW + 1) * 32u;  // bytes between qd
    const unsigned off = (unsigned)(c.h * (c.HW + 1) + k.pz) * 16u + (unsigned)(rt * 4) * stride;
#pragma unroll
    for (int qd = 0; qd < 4; qd++)
        g[qd] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(gb) + off + qd * stride);
    __builtin_amdgcn_sched_barrier(0x38F);  // keep the loads here (everything but VMEM may cross)
}

// The same entries from a copy of the table in LDS (gtab_issue).
__device__ __forceinline__ void gb_load_lds(const char* gtab, Ctx& c, const Col& k, int rt, u32x4 (&g)[4]) {
    const unsigned stride = (unsigned)(c.HW + 1) * 32u;
    const unsigned off = (unsigned)(c.h * (c.HW + 1) + k.pz) * 16u + (unsigned)(rt * 4) * stride;
#pragma unroll
    for (int qd = 0; qd < 4; qd++) g[qd] = *reinterpret_cast<const u32x4*>(gtab + off + qd * stride);
}
// One LayerNorm's gamma/beta table (bytes, contiguous) global -> LDS by LDS-DMA in 1 KiB pieces dealt to the waves: the three
// boards of a workgroup then share one fetch of it, and the lanes read it with ds_read_b128 while the LDS is idle anyway.
// The copy runs under a conv; the LayerNorm's statistics barrier (board_reduce<.., VM>) publishes it.
template <int NW>
__device__ __forceinline__ void gtab_issue(Ctx& c, const u32x4* table, int bytes, char* gtab) {
    const unsigned dst = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)gtab;
    const int npieces = (bytes + 1023) >> 10;
    for (int f = c.wave; f < npieces; f += NW) {
        const unsigned m0v = dst + f * 1024;
        const char* gp = reinterpret_cast<const char*>(table) + (size_t)f * 1024;
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(c.lane * 16), "s"(m0v), "s"(gp) : "memory");
    }
}

// y = relu((v - mean) * rstd * gamma + beta) for the 16 accumulator rows of one row tile.
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

// Packed variant: o[j] = 16-bit pair of relu(LN(v)) rows (2j, 2j+1).
// bf16 operands: per element half a v_pk_fma_f32 (normalise), one v_fma_mix_f32 (fp16 gamma/beta), half a v_cvt_pk_bf16_f32
// and half a v_pk_max_i16 (ReLU on the packed pair: a negative bf16 is a negative int16) = 2.5 instructions.
// fp16 operands: half a v_pk_fma_f32 and one v_fma_mixlo/mixhi_f16 with the clamp modifier = 1.5 instructions: the mix-FMA
// rounds straight to the fp16 half of the result register, and its clamp to [0, 1] IS the ReLU because the table holds
// gamma / S and beta / S (S a power of two >= the largest value the LayerNorm can put out, ln_scale() on the host) and the conv
// that reads these activations has its weights multiplied by S: the same products bit for bit (power-of-two scaling), except
// that activations below S * 2^-14 are fp16 subnormals (absolute error <= S * 2^-25).
__device__ __forceinline__ void ln_apply_pk(const f32x16& v, float rstd, float nmr, const u32x4 (&g)[4], unsigned (&o)[8]) {
    const f32x2 a2 = {rstd, rstd}, n2 = {nmr, nmr};
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
#if MZ_F16
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            f32x2 x = {v[4 * qd + e], v[4 * qd + e + 1]};
            f32x2 t = x * a2 + n2;
            unsigned pk;
            asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp" : "=v"(pk) : "v"(t.x), "v"(g[qd][e / 2]), "v"(g[qd][2 + e / 2]));
            asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1] clamp" : "+v"(pk) : "v"(t.y), "v"(g[qd][e / 2]), "v"(g[qd][2 + e / 2]));
            o[2 * qd + e / 2] = pk;
        }
#else
        f16x8 hv = __builtin_bit_cast(f16x8, g[qd]);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            f32x2 x = {v[4 * qd + e], v[4 * qd + e + 1]};
            f32x2 t = x * a2 + n2;
            const float y0 = fmaf(t.x, (float)hv[e], (float)hv[4 + e]);
            const float y1 = fmaf(t.y, (float)hv[e + 1], (float)hv[4 + e + 1]);
            unsigned pk;
            asm(MZ_CVT_PK " %0, %1, %2" : "=v"(pk) : "v"(y0), "v"(y1));
            s16x2 r = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), (s16x2){0, 0});
            o[2 * qd + e / 2] = __builtin_bit_cast(unsigned, r);
        }
#endif
    }
}
// 16 rows of global row tile rt, as packed bf16 pairs, into the [column][channel] image (row stride rs)
__device__ __forceinline__ void img_store_pk(char* img, int rs, int col, int rt, int h, const unsigned (&o)[8]) {
#pragma unroll
    for (int qd = 0; qd < 4; qd++) {
        u32x2 w = {o[2 * qd], o[2 * qd + 1]};
        *reinterpret_cast<u32x2*>(img + col * rs + (32 * rt + 8 * qd + 4 * h) * 2) = w;
    }
}
__device__ __forceinline__ void img_store(char* img, int rs, int col, int rt, int h, const float (&y)[16]) {
#pragma unroll
    for (int qd = 0; qd < 4; qd++)
        *reinterpret_cast<bf16x4*>(img + col * rs + (32 * rt + 8 * qd + 4 * h) * 2) = pack4(&y[4 * qd]);
}


// relu(LN(v)) of this wave's NR row tiles -> image rows of its own columns.  gb = this LayerNorm's table.
// gtab != nullptr: the table was copied to LDS under the conv in front (gtab_issue) and is read from there.
template <int NR, int NW, int NC, bool LDSG = false>
__device__ __forceinline__ void ln_to_image(Ctx& c, const Col (&k)[NC], const u32x4* gb, const f32x16 (&v)[NR][NC], int nch,
                                            int rs, const char* gtab = nullptr) {
    char* img = c.smem + c.L.img_off;
    u32x4 g[NR][NC][4];
    // the first two row tiles are requested before the statistics round trip, the rest right after it
    if (!LDSG) {
#pragma unroll
        for (int rt = 0; rt < (NR < 2 ? NR : 2); rt++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++) gb_load(gb, c, k[ct], c.wr * NR + rt, g[rt][ct]);
    }
    float mean[NC], rstd[NC];
    STAMP(c, ST_OTHER);
    ln_stats<NR, NW, NC, LDSG>(c, k, v, nch, mean, rstd);
    STAMP(c, ST_SCOMB);
#pragma unroll
    for (int rt = (LDSG ? 0 : 2); rt < NR; rt++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            if (LDSG) gb_load_lds(gtab, c, k[ct], c.wr * NR + rt, g[rt][ct]);
            else gb_load(gb, c, k[ct], c.wr * NR + rt, g[rt][ct]);
        }
#pragma unroll
    for (int rt = 0; rt < NR; rt++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            unsigned o[8];
            ln_apply_pk(v[rt][ct], rstd[ct], -mean[ct] * rstd[ct], g[rt][ct], o);   // padded columns: gamma = beta = 0 -> zeros
            img_store_pk(img, rs, k[ct].col, c.wr * NR + rt, c.h, o);
        }
    STAMP(c, ST_NORM);
}

// acc[ro][ct] += conv(image) for this wave's NROW output row tiles (of NROT in the layer) x NC column tiles.
// Stream order of the layer's fragments: (tap, ks, ro) with natural k.  The first two stages of the stream from
// here on must already be in flight; after1 / after2 = the two stages that follow this conv's last (0: nothing).
// LDS reads per MFMA = (NROW + NC) / (NROW * NC): the A fragments are shared by the wave's column tiles.
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// hook: called once, right after the first stage's barrier and DMA issue (every wave has left the LayerNorm in front of
// this conv by then, and loads requested here do not sit in front of that barrier's vmcnt(0))
template <class G, int KS, int TAPS, int NROT, int NROW, int RS, class Hook = NoHook>
__device__ __forceinline__ void conv_lds(Ctx& c, const Col (&k)[G::NC], WRing<G::NW, G::SM>& ring,
                                         f32x16 (&acc)[NROW][G::NC], int after1, int after2, Hook hook = Hook()) {
    constexpr int NC = G::NC;
    constexpr int KK = TAPS * KS;
    constexpr int SK = stage_ksteps(KK, NROT, G::SM);
    constexpr int NST = num_stages(KK, NROT, G::SM);
    const char* img = c.smem + c.L.img_off;
    const char* zrow = c.smem + c.L.zrow_off;
    const char* slot = nullptr;
    const int frag0 = (c.wr * NROW) * 1024 + c.lane * 16;
    struct Ops {
        bf16x8 b[NC], a[NROW];
    };
    // operands of k-step kk: B from the image (own columns, or the tap's neighbour rows), A from ring slot sl
    auto load_ops = [&](int kk, const char* sl, Ops& o) {
        const int t = kk / KS, ks = kk % KS;
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            const char* row;
            if (TAPS == 9) {
                const int d = (t / 3 - 1) * c.Wd + (t % 3 - 1);
                const int v = k[ct].col + d;
                row = ((k[ct].nbmask >> t) & 1u) ? img + v * RS
                                                 : zrow + (RS == G::CB * 2 + 16 ? ((v + 16 * ZROWS) & (ZROWS - 1)) * c.L.zstride : 0);
            } else {
                row = img + k[ct].col * RS;
            }
            o.b[ct] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(row + (16 * ks + 8 * c.h) * 2));
        }
#pragma unroll
        for (int ro = 0; ro < NROW; ro++)
            o.a[ro] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sl + ((kk % SK) * NROT + ro) * 1024));
    };
    // Software pipeline: the operands of k-step kk+1 are requested before the MFMAs of kk issue (inside a stage;
    // a new stage starts with its barrier), so LDS latency hides behind the matrix pipe instead of in front of it.
    // Measured no better (k_seg 2 % slower): two k-steps ahead with three operand sets, and running the look-ahead
    // across the stage barriers - the loops are bound by LDS throughput (1.5 ds_read_b128 per MFMA), not latency.
    Ops ocur, onxt;
#pragma unroll
    for (int kk = 0; kk < KK; kk++) {
        if (kk % SK == 0) {
            STAMP(c, kk == 0 ? ST_OTHER : ST_MMA);
            slot = ring.acquire() + frag0;
            STAMP(c, ST_ACQ);
            const int j = kk / SK + 2;   // the stage requested now
            ring.issue(j < NST ? stage_frags(KK, NROT, G::SM, j) : (j == NST ? after1 : after2));
            if (kk == 0) hook();
            load_ops(kk, slot, ocur);
        }
        const bool more = kk + 1 < KK && (kk + 1) % SK != 0;
        if (more) load_ops(kk + 1, slot, onxt);
        __builtin_amdgcn_sched_barrier(0);   // the scheduler would otherwise sink these reads below the MFMAs
#pragma unroll
        for (int ro = 0; ro < NROW; ro++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++) acc[ro][ct] = MFMA(ocur.a[ro], ocur.b[ct], acc[ro][ct]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) ocur = onxt;
    }
    STAMP(c, ST_MMA);
}

// relu(LN(v)) of this wave's NR row tiles kept in registers as MFMA B operands (WR = 1: the wave holds every
// channel of its columns).  A lane's 16 accumulator rows of row tile rt are the channels 32rt + 8qd + 4h + e; packed
// in that order they are the B operands of two k-steps whose k index runs k = 8h + j <-> channel
// 32rt + 16s + 4h + (j & 3) + 8(j >> 2) (s = k-step parity) - the order pack_conv_kperm gives the A fragments.
// No LDS image, no ds_write, no B reads for the 1x1 convs.
template <int NR, int NW, int NC, bool LDSG = false>
__device__ __forceinline__ void ln_to_regs(Ctx& c, const Col (&k)[NC], const u32x4* gb, const f32x16 (&v)[NR][NC], int nch,
                                           u32x4 (&bop)[2 * NR][NC], const char* gtab = nullptr) {
    u32x4 g[NR][NC][4];
    if (!LDSG) {
#pragma unroll
        for (int rt = 0; rt < (NR < 2 ? NR : 2); rt++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++) gb_load(gb, c, k[ct], rt, g[rt][ct]);
    }
    float mean[NC], rstd[NC];
    STAMP(c, ST_OTHER);
    ln_stats<NR, NW, NC, LDSG>(c, k, v, nch, mean, rstd);
    STAMP(c, ST_SCOMB);
#pragma unroll
    for (int rt = (LDSG ? 0 : 2); rt < NR; rt++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            if (LDSG) gb_load_lds(gtab, c, k[ct], rt, g[rt][ct]);
            else gb_load(gb, c, k[ct], rt, g[rt][ct]);
        }
#pragma unroll
    for (int rt = 0; rt < NR; rt++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            unsigned o[8];
            ln_apply_pk(v[rt][ct], rstd[ct], -mean[ct] * rstd[ct], g[rt][ct], o);   // padded columns: gamma = beta = 0 -> zeros
            bop[2 * rt][ct] = (u32x4){o[0], o[1], o[2], o[3]};
            bop[2 * rt + 1][ct] = (u32x4){o[4], o[5], o[6], o[7]};
        }
    STAMP(c, ST_NORM);
}

// acc[ro][ct] += W * bop for a 1x1 conv whose B operands are in registers (KS k-steps, fragments (ks, ro) in k-permuted
// order); same ring protocol as conv_lds.
template <class G, int KS, int NROT, class Hook = NoHook>
__device__ __forceinline__ void conv_regs(Ctx& c, WRing<G::NW, G::SM>& ring, const u32x4 (&bop)[KS][G::NC],
                                          f32x16 (&acc)[NROT][G::NC], int after1, int after2, Hook hook = Hook()) {
    constexpr int NC = G::NC;
    constexpr int SK = stage_ksteps(KS, NROT, G::SM);
    constexpr int NST = num_stages(KS, NROT, G::SM);
    const char* slot = nullptr;
    auto load_a = [&](int kk, bf16x8(&a)[NROT]) {
#pragma unroll
        for (int ro = 0; ro < NROT; ro++)
            a[ro] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(slot + ((kk % SK) * NROT + ro) * 1024));
    };
    bf16x8 acur[NROT], anxt[NROT];
#pragma unroll
    for (int kk = 0; kk < KS; kk++) {
        if (kk % SK == 0) {
            STAMP(c, kk == 0 ? ST_OTHER : ST_MMA);
            slot = ring.acquire() + c.lane * 16;
            STAMP(c, ST_ACQ);
            const int j = kk / SK + 2;
            ring.issue(j < NST ? stage_frags(KS, NROT, G::SM, j) : (j == NST ? after1 : after2));
            if (kk == 0) hook();
            load_a(kk, acur);
        }
        const bool more = kk + 1 < KS && (kk + 1) % SK != 0;
        if (more) load_a(kk + 1, anxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ro = 0; ro < NROT; ro++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++)
                acc[ro][ct] = MFMA(acur[ro], __builtin_bit_cast(bf16x8, bop[kk][ct]), acc[ro][ct]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int ro = 0; ro < NROT; ro++) acur[ro] = anxt[ro];
        }
    }
    STAMP(c, ST_MMA);
}

template <int NR, int NC>
__device__ __forceinline__ void zero_acc(f32x16 (&a)[NR][NC]) {
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
#pragma unroll
            for (int i = 0; i < 16; i++) a[r][ct][i] = 0.f;
}

// BottleneckResidualBlock: x + c4(relu(LN(c3(relu(LN(c2(relu(LN(c1(relu(LN(x))))))))))))
// Stage sizes of a block's four convs, and the two stages that open a block.
template <class G>
struct BlockStages {
    static constexpr int KSC = G::C / 16, KSB = G::CB / 16, SM = G::SM;
    static constexpr int c1(int j) { return stage_frags(KSC, G::NRB, SM, j); }
    static constexpr int c2(int j) { return stage_frags(9 * KSB, G::NRB, SM, j); }
    static constexpr int c4(int j) { return stage_frags(KSB, G::NRC, SM, j); }
    static constexpr int n1 = num_stages(KSC, G::NRB, SM), n4 = num_stages(KSB, G::NRC, SM);
    static_assert(num_stages(9 * KSB, G::NRB, SM) >= 2, "3x3 convs span at least two stages");
    static constexpr int open0 = c1(0), open1 = n1 > 1 ? c1(1) : c2(0);
};

// On entry the first two stages of the block are in flight; on exit those of the next block (if has_next).
template <class G>
__device__ __forceinline__ void bottleneck(Ctx& c, const Col (&k)[G::NC], WRing<G::NW, G::SM>& ring, const u32x4* gb,
                                           f32x16 (&X)[G::NRCW][G::NC], bool has_next) {
    constexpr int C = G::C, CB = G::CB;
    using BS = BlockStages<G>;
    const int nx0 = has_next ? BS::open0 : 0, nx1 = has_next ? BS::open1 : 0;
    constexpr int RSC = C * 2 + 16, RSB = CB * 2 + 16;
    constexpr int KSC = C / 16, KSB = CB / 16;
    const long long e = c.HW + 1;                   // table entries per (rt, qd, h)
    const u32x4* gb1 = gb;
    const u32x4* gb2 = gb1 + G::NRC * 8 * e;
    const u32x4* gb3 = gb2 + G::NRB * 8 * e;
    const u32x4* gb4 = gb3 + G::NRB * 8 * e;
    f32x16 a1[G::NRBW][G::NC];
    // Single-row wave grids: the tables of LN2, LN3, LN4 are copied to LDS under the conv in front of each, behind the
    // bottleneck image (which the register-fed 1x1 convs leave to the 3x3 convs alone).
    constexpr bool LG = G::WR == 1;
    char* gtab = c.smem + c.L.img_off + (G::COLS * RSB + 1023) / 1024 * 1024;
    const int gbytes = G::NRB * 8 * (int)e * 16;
    auto cp2 = [&]() { if (LG) gtab_issue<G::NW>(c, gb2, gbytes, gtab); };
    auto cp3 = [&]() { if (LG) gtab_issue<G::NW>(c, gb3, gbytes, gtab); };
    auto cp4 = [&]() { if (LG) gtab_issue<G::NW>(c, gb4, gbytes, gtab); };

    // LN1 -> c1 (1x1, C -> CB): B operands from registers when the wave holds all channels, else through the image
    if constexpr (G::WR == 1) {
        u32x4 bop[KSC][G::NC];
        ln_to_regs<G::NRC, G::NW, G::NC>(c, k, gb1, X, C, bop);
        zero_acc(a1);
        conv_regs<G, KSC, G::NRB>(c, ring, bop, a1, BS::c2(0), BS::c2(1), cp2);
    } else {
        ln_to_image<G::NRCW, G::NW, G::NC>(c, k, gb1, X, C, RSC);
        zero_acc(a1);
        conv_lds<G, KSC, 1, G::NRB, G::NRBW, RSC>(c, k, ring, a1, BS::c2(0), BS::c2(1));
    }
    // LN2 -> image -> c2 (3x3), LN3 -> image -> c3 (3x3); the image is consumed before a1 is rewritten
#pragma unroll
    for (int layer = 0; layer < 2; layer++) {
        ln_to_image<G::NRBW, G::NW, G::NC, LG>(c, k, layer == 0 ? gb2 : gb3, a1, CB, RSB, gtab);
        zero_acc(a1);
        if (layer == 0)
            conv_lds<G, KSB, 9, G::NRB, G::NRBW, RSB>(c, k, ring, a1, BS::c2(0), BS::c2(1), cp3);
        else
            conv_lds<G, KSB, 9, G::NRB, G::NRBW, RSB>(c, k, ring, a1, BS::c4(0), BS::n4 > 1 ? BS::c4(1) : nx0, cp4);
    }
    // LN4 -> c4 (1x1, CB -> C) accumulated into X
    if constexpr (G::WR == 1) {
        u32x4 bop[KSB][G::NC];
        ln_to_regs<G::NRB, G::NW, G::NC, LG>(c, k, gb4, a1, CB, bop, gtab);
        conv_regs<G, KSB, G::NRC>(c, ring, bop, X, nx0, nx1);
    } else {
        ln_to_image<G::NRBW, G::NW, G::NC>(c, k, gb4, a1, CB, RSB);
        conv_lds<G, KSB, 1, G::NRC, G::NRCW, RSB>(c, k, ring, X, nx0, nx1);
    }
}

// ---- shared prologue: per-board meta data and per-lane column info -----------------------------
template <int IN, int NC>
__device__ __forceinline__ bool setup(const NetDims& d, const NetIO& io, int NB, Ctx& c, Col (&kk)[NC]) {
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
    // the zero rows that off-board 3x3 neighbours read
    for (int i = tid; i < c.L.zbytes / 16; i += blockDim.x) reinterpret_cast<u32x4*>(c.smem + c.L.zrow_off)[i] = (u32x4){0, 0, 0, 0};
    lds_barrier();
    int any = 0;
    for (int b = 0; b < NB; b++) any |= c.meta[2 * MAXNB + b];
    if (!any) return false;
#pragma unroll
    for (int ct = 0; ct < NC; ct++) {
    Col& k = kk[ct];
    const int col = 32 * (NC * c.wc + ct) + c.r;
    k.col = col;
    k.valid = col < NB * d.HW;
    k.board = k.valid ? col / d.HW : MAXNB;
    k.p = k.valid ? col - k.board * d.HW : 0;
    k.pz = k.valid ? k.p : d.HW;
    k.live = k.valid && c.meta[2 * MAXNB + (k.valid ? k.board : 0)];
    unsigned m = 0;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int y = k.p / d.Wd + t / 3 - 1, x = k.p % d.Wd + t % 3 - 1;
        if (k.valid && y >= 0 && y < d.H && x >= 0 && x < d.Wd) m |= 1u << t;
    }
    k.nbmask = m;
    }
    return true;
}

template <class G>
__device__ __forceinline__ void ctx_init(Ctx& c, char* smem, const LdsPlan& L, const NetDims& d, int NB) {
    const int tid = threadIdx.x;
    c.smem = smem;
    c.L = L;
    c.HW = d.HW;
    c.NB = NB;
    c.Wd = d.Wd;
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.wc = c.wave % G::WC;
    c.wr = c.wave / G::WC;
    c.h = c.lane >> 5;
    c.r = c.lane & 31;
    c.wpart = reinterpret_cast<float*>(smem + L.stat_off);
    c.parity = 0;
    c.meta = reinterpret_cast<int*>(smem + L.meta_off);
#if defined(MZ_STAMPS)
    for (int i = 0; i < 16; i++) c.st_acc[i] = 0;
    c.st_last = __builtin_amdgcn_s_memtime();
#endif
}

// X <-> the hidden-state pool (slot index from meta row `which`); this wave's row tiles.  On this path a slot is
// laid out [C/4][HW][4] fp32 (channel quads interleaved): a lane's four consecutive accumulator registers (channels
// 32rt + 8qd + 4h + 0..3 of its cell) are one 16-byte access and a wave's accesses are contiguous 512-byte runs -
// 16 instead of 64 memory instructions per lane.  mz_read_hidden un-permutes (api.cpp).
template <int NR, int NC>
__device__ __forceinline__ void x_load(Ctx& c, const Col (&kk)[NC], const float* pool, int which, int C, f32x16 (&X)[NR][NC]) {
    // unconditional loads (dead columns read slot meta[..][0], which is always a valid slot) + select:
    // a per-load branch would serialise dependent round trips
#pragma unroll
    for (int ct = 0; ct < NC; ct++) {
        const Col& k = kk[ct];
        const int bd = k.valid ? k.board : 0;
        const float* src = pool + (long long)c.meta[which * MAXNB + bd] * C * c.HW + (k.p + c.h * c.HW) * 4;
        int hw = c.HW;
        asm volatile("" : "+s"(hw));   // opaque per call: keeps the compiler from holding (and spilling) the row offsets
        const bool keep = k.live;
#pragma unroll
        for (int rt = 0; rt < NR; rt++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(src + (8 * (c.wr * NR + rt) + 2 * qd) * hw * 4);
#pragma unroll
                for (int e = 0; e < 4; e++) X[rt][ct][4 * qd + e] = keep ? v[e] : 0.f;  // select: dead slots may hold NaN bit patterns
            }
    }
}
template <int NR, int NC>
__device__ __forceinline__ void x_store(Ctx& c, const Col (&kk)[NC], float* pool, int C, const f32x16 (&X)[NR][NC]) {
#pragma unroll
    for (int ct = 0; ct < NC; ct++) {
        const Col& k = kk[ct];
        if (!k.live) continue;
        float* dst = pool + (long long)c.meta[MAXNB + k.board] * C * c.HW + (k.p + c.h * c.HW) * 4;
        int hw = c.HW;
        asm volatile("" : "+s"(hw));
#pragma unroll
        for (int rt = 0; rt < NR; rt++)
#pragma unroll
            for (int qd = 0; qd < 4; qd++)
                *reinterpret_cast<f32x4*>(dst + (8 * (c.wr * NR + rt) + 2 * qd) * hw * 4) =
                    (f32x4){X[rt][ct][4 * qd], X[rt][ct][4 * qd + 1], X[rt][ct][4 * qd + 2], X[rt][ct][4 * qd + 3]};
    }
}

// ---- bottleneck blocks b0..b1-1, then the X store or [rescale + hidden-state store + head convs] ----------
template <class G, int OUT>
__device__ __forceinline__ void seg_rest(Ctx& c, const Col (&k)[G::NC], const NetDims& d, const TowerDev& tw, const HeadDev& hd,
                                         const NetIO& io, const LdsPlan& L, char* smem, WRing<G::NW, G::SM>& ring,
                                         f32x16 (&X)[G::NRCW][G::NC], int NB, int b0, int b1) {
    constexpr int C = G::C, NRCW = G::NRCW, NC = G::NC;
    const int HW = d.HW;
    const int g0 = blockIdx.x * NB;
    {
        const u32x4* gbase = tw.gb;
        for (int blk = b0; blk < b1; blk++)
            bottleneck<G>(c, k, ring, gbase + tw.gboff[blk], X, blk + 1 < b1);
    }
    if (OUT == OUT_REGS) return;

    if (OUT == OUT_X) {
        STAMP(c, ST_OTHER);
        x_store<NRCW, NC>(c, k, io.pool, C, X);
        STAMP(c, ST_XSTORE);
        STAMP_PRINT(c, "seg");
        return;
    }
    STAMP(c, ST_OTHER);
    // ---------------- RescaleBlockExt + hidden-state store ----------------
    {
        float mn[NC], mx[NC];
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            mn[ct] = INFINITY;
            mx[ct] = -INFINITY;
#pragma unroll
            for (int rt = 0; rt < NRCW; rt++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    mn[ct] = fminf(mn[ct], X[rt][ct][i]);
                    mx[ct] = fmaxf(mx[ct], X[rt][ct][i]);
                }
            if (!k[ct].valid) {
                mn[ct] = INFINITY;
                mx[ct] = -INFINITY;
            }
        }
        board_reduce<1, G::NW, NC>(c, k, mn, mx, 0.f);
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            const float dd = fmaxf(mx[ct] - mn[ct], 1e-5f);
#pragma unroll
            for (int rt = 0; rt < NRCW; rt++)
#pragma unroll
                for (int i = 0; i < 16; i++) X[rt][ct][i] = (X[rt][ct][i] - mn[ct]) / dd;
        }
        STAMP(c, ST_RESCALE);
        x_store<NRCW, NC>(c, k, io.pool, C, X);
        STAMP(c, ST_XSTORE);
    }
    // ---------------- the 1x1 convs that open the prediction heads (C -> policy 0, policy 1, value) -------------
    // Taken here off the registers, so that k_heads reads 3 values per cell instead of the whole hidden state again.
    // A lane holds 16 * NRCW channels of its cell: per-lane partial dot products -> LDS (the image is dead) -> the
    // 2 * WR partials of a cell are summed in a fixed order by the lane (wr = 0, h = 0) of that column.
    {
        float* hpart = reinterpret_cast<float*>(smem + L.img_off);   // [2 * WR][3][COLS]
        float part[NC][3];
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
#pragma unroll
            for (int o = 0; o < 3; o++) part[ct][o] = 0.f;
#pragma unroll
        for (int rt = 0; rt < NRCW; rt++) {
            f32x4 w[3][4];
#pragma unroll
            for (int o = 0; o < 3; o++)
#pragma unroll
                for (int qd = 0; qd < 4; qd++)
                    w[o][qd] = *reinterpret_cast<const f32x4*>(hd.hconv + o * C + 32 * (c.wr * NRCW + rt) + 8 * qd + 4 * c.h);
#pragma unroll
            for (int ct = 0; ct < NC; ct++)
#pragma unroll
                for (int o = 0; o < 3; o++)
#pragma unroll
                    for (int i = 0; i < 16; i++) part[ct][o] = fmaf(X[rt][ct][i], w[o][i >> 2][i & 3], part[ct][o]);
        }
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
#pragma unroll
            for (int o = 0; o < 3; o++) hpart[((c.wr * 2 + c.h) * 3 + o) * G::COLS + k[ct].col] = part[ct][o];
        lds_barrier();
        if (c.wr == 0 && c.h == 0) {
#pragma unroll
            for (int ct = 0; ct < NC; ct++) {
                if (!k[ct].live) continue;
                float* dst = hd.hfeat + ((long long)(io.g_base + g0 + k[ct].board) * 3) * HW + k[ct].p;
#pragma unroll
                for (int o = 0; o < 3; o++) {
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < 2 * G::WR; j++) sum += hpart[(j * 3 + o) * G::COLS + k[ct].col];
                    dst[o * HW] = sum;
                }
            }
        }
    }
    STAMP_PRINT(c, "seg+rescale");
}

// ---- [conv0] + bottleneck blocks b0..b1-1 + [rescale, heads] -----------------------------------
// ---- front of a tower launch: the ring's first two stages, then X from the pool (IN_X) or the input image + conv0 ----
template <class G, int IN>
__device__ __forceinline__ void seg_front(Ctx& c, const Col (&k)[G::NC], const NetDims& d, const TowerDev& tw, const NetIO& io,
                                          const LdsPlan& L, char* smem, WRing<G::NW, G::SM>& ring, f32x16 (&X)[G::NRCW][G::NC],
                                          int NB, int b0, int b1) {
    constexpr int C = G::C, NRCW = G::NRCW, NC = G::NC;
    // input channels of conv0 on the matrix cores, padded to 16: the observation planes, or the C channels of the parent
    // state - the dynamics net's one-hot action planes are added afterwards (below)
    constexpr int CIN0 = IN == IN_OBS ? 32 : C;
    constexpr int KS0 = CIN0 / 16;
    constexpr int RS0 = CIN0 * 2 + 16;
    using BS = BlockStages<G>;
    const int HW = d.HW;
    const int g0 = blockIdx.x * NB;
    char* img = smem + L.img0_off;

    if (IN == IN_X) {
        ring.init(tw.stream + (long long)tw.foff[b0] * 64, smem + L.ring_off, c.wave, c.lane);
        if (b0 < b1) {
            ring.issue(BS::open0);
            ring.issue(BS::open1);
        }
        x_load<NRCW, NC>(c, k, io.pool, 1, C, X);
        STAMP(c, ST_XLOAD);
    } else {
        ring.init(tw.stream, smem + L.ring_off, c.wave, c.lane);
        static_assert(num_stages(9 * KS0, G::NRC, G::SM) >= 2, "conv0 spans at least two stages");
        ring.issue(stage_frags(9 * KS0, G::NRC, G::SM, 0));
        ring.issue(stage_frags(9 * KS0, G::NRC, G::SM, 1));
        // ---------------- input image + conv0 ----------------
        if (IN == IN_OBS) {
            zero_acc(X);
            if (c.wr == 0) {
#pragma unroll
                for (int ct = 0; ct < NC; ct++) {
                    const int g = g0 + (k[ct].valid ? k[ct].board : 0);
                    float y[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int ch = 16 * c.h + i;
                        y[i] = (k[ct].live && ch < d.O) ? io.obs[((long long)g * d.O + ch) * HW + k[ct].p] : 0.f;
                    }
                    *reinterpret_cast<bf16x8*>(img + k[ct].col * RS0 + (16 * c.h) * 2) = pack8(&y[0]);
                    *reinterpret_cast<bf16x8*>(img + k[ct].col * RS0 + (16 * c.h + 8) * 2) = pack8(&y[8]);
                }
            }
        } else {
            x_load<NRCW, NC>(c, k, io.pool, 0, C, X);   // StartResidualBlock: X starts as the parent state s
#pragma unroll
            for (int ct = 0; ct < NC; ct++) {
#pragma unroll
                for (int rt = 0; rt < NRCW; rt++) {
                    float y[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) y[i] = X[rt][ct][i];
                    img_store(img, RS0, k[ct].col, c.wr * NRCW + rt, c.h, y);
                }
            }
        }
        c.L.img_off = L.img0_off;   // conv0 reads its own image; everything behind it the block images at img_off
        conv_lds<G, KS0, 9, G::NRC, NRCW, RS0>(c, k, ring, X, b0 < b1 ? BS::open0 : 0, b0 < b1 ? BS::open1 : 0);
        c.L.img_off = L.img_off;
        if (IN == IN_HIDDEN) {
            // ConcatInputsBlock + conv0 on the action planes: plane pl is one-hot at cell q (action cell = pl * HW + q, -1 =
            // none), so its share of conv0 is W[o][C + pl][t] at the one cell p with p + d(t) = q, per tap: a rank-1 update
            // in exact fp32 instead of a ninth k-step per tap on the matrix cores (11 % of conv0's MFMAs and weight bytes).
#pragma unroll
            for (int ct = 0; ct < NC; ct++) {
                const int cell = c.meta[3 * MAXNB + (k[ct].valid ? k[ct].board : 0)];
                if (k[ct].live && cell >= 0) {
                    const int pl = cell / HW, q = cell - pl * HW;
                    int tstar = -1;
#pragma unroll
                    for (int t = 0; t < 9; t++)
                        if (((k[ct].nbmask >> t) & 1u) && k[ct].p + (t / 3 - 1) * c.Wd + (t % 3 - 1) == q) tstar = t;
                    if (tstar >= 0 && pl < d.a) {
                        const float* wa = tw.wact + ((size_t)pl * 9 + tstar) * C;
#pragma unroll
                        for (int rt = 0; rt < NRCW; rt++)
#pragma unroll
                            for (int qd = 0; qd < 4; qd++) {
                                const f32x4 w = *reinterpret_cast<const f32x4*>(wa + 32 * (c.wr * NRCW + rt) + 8 * qd + 4 * c.h);
#pragma unroll
                                for (int e = 0; e < 4; e++) X[rt][ct][4 * qd + e] += w[e];
                            }
                    }
                }
            }
        }
    }

}

template <class G, int IN, int OUT>
__global__ __launch_bounds__(G::NT, G::WPE) void k_seg(NetDims d, TowerDev tw, HeadDev hd, NetIO io, LdsPlan L, int NB, int b0,
                                            int b1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NRCW = G::NRCW, NC = G::NC;
    Ctx c;
    ctx_init<G>(c, smem, L, d, NB);
    Col k[NC];
    if (!setup<IN, NC>(d, io, NB, c, k)) return;
    WRing<G::NW, G::SM> ring;
    f32x16 X[NRCW][NC];
    seg_front<G, IN>(c, k, d, tw, io, L, smem, ring, X, NB, b0, b1);
    seg_rest<G, OUT>(c, k, d, tw, hd, io, L, smem, ring, X, NB, b0, b1);
}

// ---- PredictionBlock (B/c_mainfunctions/PredictionBlock.java:46-84) ------------------------------
// One 256-thread workgroup per HS = 4 samples, many workgroups per CU: the heads are small latency chains
// (1x1 convs -> LayerNorm -> linears -> softmax / tanh), so they run at high occupancy instead of serialising the
// big workgroups of the tower kernels.  The three 1x1 convs are taken by the tower's last launch off its registers
// (HeadDev::hfeat).  Wave s owns sample s for the LayerNorms and the outputs (no workgroup barrier there); the two linears are shared
// work: an item = one output of one K chunk, its weights loaded once and applied to all four samples.
// All fp32: the per-board LayerNorm over a single HW-value channel amplifies rounding of whatever feeds it.
constexpr int HT = 256;      // threads
constexpr int HS = 4;        // samples per workgroup (= waves)
constexpr int HKC = 32;      // K chunk of the two linears
constexpr int HPMAX = 128;   // cells per sample upper bound (HW <= 96)
__global__ __launch_bounds__(HT) void k_heads(NetDims d, HeadDev hd, NetIO io) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = d.HW, C = d.C, A = d.A;
    extern __shared__ __attribute__((aligned(16))) char hsm[];
    const int K1 = 2 * HW, n1 = (K1 + HKC - 1) / HKC, n2 = (HW + HKC - 1) / HKC;
    const int items1 = n1 * A, items = items1 + n2 * C;
    float* feat = reinterpret_cast<float*>(hsm);  // [HS][3*HPMAX]: policy features [2*HW], then value features [HW]
    float* lgt = feat + HS * 3 * HPMAX;           // [HS][A]
    float* hv = lgt + HS * A;                     // [HS][C]
    float* part = hv + HS * C;                    // [HS][items] K-chunk partials: [n1][A] then [n2][C]
    const int g = blockIdx.x * HS + wave;         // this wave's sample
    const bool live = g < io.B && (!io.active || io.active[g]);

    // weights of this thread's first linear item: requested before anything else
    struct Item {
        const float* w;
        int foff, nout, kn;
    };
    auto item = [&](int it) {
        const bool pol = it < items1;
        const int j = pol ? it : it - items1;
        const int nout = pol ? A : C, K = pol ? K1 : HW;
        const int ck = j / nout, o = j - ck * nout;
        Item r;
        r.w = (pol ? hd.pfcw : hd.vfc1w) + (long long)(ck * HKC) * nout + o;
        r.foff = (pol ? 0 : K1) + ck * HKC;
        r.nout = nout;
        r.kn = K - ck * HKC;
        return r;
    };
    float wv[HKC];
    if (tid < items) {
        const Item q = item(tid);
#pragma unroll
        for (int i = 0; i < HKC; i++) wv[i] = i < q.kn ? q.w[(long long)i * q.nout] : 0.f;
    }
    // 1x1 convs C -> {policy 0, policy 1, value} and their LayerNorms, wave-local: lane l holds the cells l, l+64
    {
        float a00 = 0.f, a01 = 0.f, a02 = 0.f, a10 = 0.f, a11 = 0.f, a12 = 0.f;
        const bool v0 = lane < HW, v1 = lane + 64 < HW;
        if (live) {
            const float* hf = hd.hfeat + (long long)(io.g_base + g) * 3 * HW;   // written by k_seg<.., OUT_FINAL>
            if (v0) {
                a00 = hf[lane];
                a01 = hf[HW + lane];
                a02 = hf[2 * HW + lane];
            }
            if (v1) {
                a10 = hf[lane + 64];
                a11 = hf[HW + lane + 64];
                a12 = hf[2 * HW + lane + 64];
            }
        }
        // LayerNorm over (2, HW) for the policy features and over (1, HW) for the value features
        float s = (v0 ? a00 + a01 : 0.f) + (v1 ? a10 + a11 : 0.f);
        float q = (v0 ? a00 * a00 + a01 * a01 : 0.f) + (v1 ? a10 * a10 + a11 * a11 : 0.f);
        float vs = (v0 ? a02 : 0.f) + (v1 ? a12 : 0.f), vq = (v0 ? a02 * a02 : 0.f) + (v1 ? a12 * a12 : 0.f);
        s = wave_red<0>(s);
        q = wave_red<0>(q);
        vs = wave_red<0>(vs);
        vq = wave_red<0>(vq);
        const float pm = s / (float)(2 * HW), pr = __builtin_amdgcn_rsqf(fmaxf(q / (float)(2 * HW) - pm * pm, 0.f) + 1e-5f);
        const float vm = vs / (float)HW, vr = __builtin_amdgcn_rsqf(fmaxf(vq / (float)HW - vm * vm, 0.f) + 1e-5f);
        float* f = feat + wave * 3 * HPMAX;
        if (v0) {
            const int p = lane;
            f[p] = fmaxf((a00 - pm) * pr * hd.plng[p] + hd.plnb[p], 0.f);
            f[HW + p] = fmaxf((a01 - pm) * pr * hd.plng[HW + p] + hd.plnb[HW + p], 0.f);
            f[K1 + p] = fmaxf((a02 - vm) * vr * hd.vlng[p] + hd.vlnb[p], 0.f);
        }
        if (v1) {
            const int p = lane + 64;
            f[p] = fmaxf((a10 - pm) * pr * hd.plng[p] + hd.plnb[p], 0.f);
            f[HW + p] = fmaxf((a11 - pm) * pr * hd.plng[HW + p] + hd.plnb[HW + p], 0.f);
            f[K1 + p] = fmaxf((a12 - vm) * vr * hd.vlng[p] + hd.vlnb[p], 0.f);
        }
    }
    __syncthreads();
    // the two linears, split over K into chunks of HKC; chunk partials are summed in a fixed order
    for (int it = tid; it < items; it += HT) {
        const Item q = item(it);
        if (it != tid) {
#pragma unroll
            for (int i = 0; i < HKC; i++) wv[i] = i < q.kn ? q.w[(long long)i * q.nout] : 0.f;
        }
#pragma unroll
        for (int sm = 0; sm < HS; sm++) {
            const float* f = feat + sm * 3 * HPMAX + q.foff;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < HKC; i++) acc = fmaf(i < q.kn ? f[i] : 0.f, wv[i], acc);
            part[sm * items + it] = acc;
        }
    }
    __syncthreads();
    // outputs of sample `wave`
    const float* pt = part + wave * items;
    float* lg = lgt + wave * A;
    float* hvw = hv + wave * C;
    for (int a = lane; a < A; a += 64) {
        float acc = hd.pfcb[a];
        for (int ck = 0; ck < n1; ck++) acc += pt[ck * A + a];
        lg[a] = acc;
    }
    float pv = 0.f;
    for (int ch = lane; ch < C; ch += 64) {
        float acc = hd.vfc1b[ch];
        for (int ck = 0; ck < n2; ck++) acc += pt[items1 + ck * C + ch];
        pv = fmaf(fmaxf(acc, 0.f), hd.vfc2w[ch], pv);
    }
    (void)hvw;
    if (!live) return;
    float mx = -INFINITY;
    for (int a = lane; a < A; a += 64) mx = fmaxf(mx, lg[a]);   // same lane wrote lg[a]
    mx = wave_red<2>(mx);
    float se = 0.f;
    for (int a = lane; a < A; a += 64) se += expf(lg[a] - mx);
    se = wave_red<0>(se);
    for (int a = lane; a < A; a += 64) {
        const float l = lg[a];
        io.logits[(long long)g * A + a] = l;
        io.policy[(long long)g * A + a] = expf(l - mx) / se;
    }
    pv = wave_red<0>(pv);
    if (lane == 0) {
        float v = pv + hd.vfc2b[0];
        if (d.two_players) v = tanhf(v);
        io.value[g] = d.value_scale * (double)v;
    }
}
size_t heads_lds_bytes(const NetDims& d) {
    const int n1 = (2 * d.HW + HKC - 1) / HKC, n2 = (d.HW + HKC - 1) / HKC;
    return (size_t)HS * (3 * HPMAX + d.A + d.C + n1 * d.A + n2 * d.C) * 4;
}

// ---- BroadcastResidualBlock: x + c2(relu(LN(bcast(relu(LN(c1(relu(LN(x))))))))) ------------------
// LDS: [image, C channels][region 2]; region 2 is the weight ring while c1 and c2 run and the
// transposed image TB[channel][board*KP + cell] while the broadcast linear runs.
// KSB = k-steps of the broadcast linear (cells padded to 16*KSB); its weight fragments stay in registers.
// Body of the broadcast block: takes X from the destination slots (XREGS: from the registers of the blocks in front of
// it in the same kernel, parking it in the destination slots meanwhile), leaves x + block(x) in X.  nx0 / nx1 = the two
// ring stages that follow the block's last conv (the next bottleneck block's opening stages when the kernel goes on).
template <class G, int KSB, bool XREGS = false>
__device__ __forceinline__ void bcast_body(Ctx& c, const Col (&k)[G::NC], const TowerDev& tw, const NetIO& io, const LdsPlan& L,
                                           char* smem, WRing<G::NW, G::SM>& ring, f32x16 (&X)[G::NRCW][G::NC], int NB, int blk,
                                           int nx0, int nx1) {
    constexpr int C = G::C, NRC = G::NRC, NRCW = G::NRCW, NC = G::NC;
    constexpr int QT = (KSB + 1) / 2, KP = 16 * KSB;
    constexpr int RS = C * 2 + 16;
    constexpr int KSC = C / 16;
    constexpr int FS0 = stage_frags(KSC, NRC, G::SM, 0), FS1 = stage_frags(KSC, NRC, G::SM, 1);
    const long long e = c.HW + 1;
    const u32x4* gb1 = tw.gb + tw.gboff[blk];
    const u32x4* gb2 = gb1 + NRC * 8 * e;
    const u32x4* gb3 = gb2 + NRC * 8 * e;
    const u32x4* wbf = tw.wbf + tw.wboff[blk];
    const float* bb = tw.bb + tw.bboff[blk];
    const int TBS = L.tbs;
    char* tb = smem + L.tb_off;
    char* img = smem + L.img_off;
    if (XREGS) lds_barrier();   // every wave has left the ring slots of the blocks in front before they are refilled
    ring.init(tw.stream + (long long)tw.foff[blk] * 64, smem + L.ring_off, c.wave, c.lane);
    ring.issue(FS0);
    ring.issue(FS1);
    if (!XREGS) x_load<NRCW, NC>(c, k, io.pool, 1, C, X);   // XREGS: X simply stays in its registers through the block
    {
        // LN1 -> image -> c1 (1x1, C -> C); this wave's share of the output rows stays in registers
        f32x16 a[NRCW][NC];
        ln_to_image<NRCW, G::NW, NC>(c, k, gb1, X, C, RS);
        zero_acc(a);
        conv_lds<G, KSC, 1, NRC, NRCW, RS>(c, k, ring, a, 0, 0);
        STAMP(c, ST_OTHER);
        // LN2 + ReLU -> bf16 -> transposed image TB (the statistics barrier proves that every wave has
        // finished c1, so the ring under TB is dead)
        u32x4 g[NC][NRCW][4];   // requested before the statistics round trip
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
#pragma unroll
            for (int rt = 0; rt < NRCW; rt++) gb_load(gb2, c, k[ct], c.wr * NRCW + rt, g[ct][rt]);
        float mean[NC], rstd[NC];
        ln_stats<NRCW, G::NW, NC>(c, k, a, C, mean, rstd);
        for (int i = threadIdx.x; i < C * NB * (KP - c.HW); i += G::NT) {  // zero the K padding of every board
            const int padw = KP - c.HW;
            const int ch = i / (NB * padw), rem = i - ch * (NB * padw);
            const int bd = rem / padw, pp = rem - bd * padw;
            *reinterpret_cast<op_t*>(tb + ch * TBS + (bd * KP + c.HW + pp) * 2) = (op_t)0.f;
        }
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
#pragma unroll
            for (int rt = 0; rt < NRCW; rt++) {
                float y[16];
                ln_apply(a[rt][ct], mean[ct], rstd[ct], g[ct][rt], y);
                if (k[ct].valid) {
                    const int cbase = k[ct].board * KP + k[ct].p;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int ch = 32 * (c.wr * NRCW + rt) + (i & 3) + 8 * (i >> 2) + 4 * c.h;
                        *reinterpret_cast<op_t*>(tb + ch * TBS + cbase * 2) = (op_t)y[i];
                    }
                }
            }
    }
    // broadcast-linear B fragments and bias (requested here, once the c1 tiles are dead): lane (q = 32qt + r, h) holds Wb[q][16ks + 8h + j]
    u32x4 wreg[KSB][QT];
    float bias[QT];
#pragma unroll
    for (int ks = 0; ks < KSB; ks++)
#pragma unroll
        for (int qt = 0; qt < QT; qt++) wreg[ks][qt] = wbf[(ks * QT + qt) * 64 + c.lane];
#pragma unroll
    for (int qt = 0; qt < QT; qt++) bias[qt] = bb[32 * qt + c.r];
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    STAMP(c, ST_HCONV);   // (diagnostic slots reused: hconv = LN2 + transpose, hln = pass 0, hfc = statistics)
    // BroadcastBlock: out[ch][q] = sum_p in[ch][p] * Wb[q][p] + b[q], per board.  Work items (board, row tile)
    // are dealt round-robin to the waves; an item covers all QT q tiles, one after the other.
    // Pass 0 accumulates the LN3 statistics, pass 1 recomputes (36 MFMAs, cheaper than holding the tiles) and
    // writes relu(LN3(.)) to the image; its gamma/beta loads are requested before the recompute.
    float* ipart = c.wpart + 2 * MAXNB * G::NW * 2;   // [items][2]
    float* bstat = ipart + 2 * 256;                   // [MAXNB][2]
    const int nitems = NB * NRC;
    // separate launch: the residual stream for c2 is requested here (it was not held in registers through c1 and the
    // transpose), so its round trip runs under the two passes
    if (!XREGS) x_load<NRCW, NC>(c, k, io.pool, 1, C, X);
    // gamma/beta of LN3 for (row tile rtw, cells q): 4 x 16-byte loads, requested one (item, q tile) ahead in pass 1
    auto g3_load = [&](int it, int qt, u32x4(&g)[4]) {
        if (it >= nitems) return;
        const int rtw = it % NRC;
        const int q = 32 * qt + c.r;
        const unsigned stride = (unsigned)(c.HW + 1) * 32u;
        const unsigned off = (unsigned)(c.h * (c.HW + 1) + (q < c.HW ? q : c.HW)) * 16u + (unsigned)(rtw * 4) * stride;
#pragma unroll
        for (int qd = 0; qd < 4; qd++)
            g[qd] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(gb3) + off + qd * stride);
        __builtin_amdgcn_sched_barrier(0x38F);
    };
    for (int pass = 0; pass < 2; pass++) {
        u32x4 gn[4];
        if (pass == 1) g3_load(c.wave, 0, gn);
        for (int it = c.wave; it < nitems; it += G::NW) {
            const int bd = it / NRC, rtw = it - bd * NRC;
            const char* arow = tb + (32 * rtw + c.r) * TBS + (bd * KP + 8 * c.h) * 2;
            float ps = 0.f, pq = 0.f;
#pragma unroll
            for (int qt = 0; qt < QT; qt++) {
                const int q = 32 * qt + c.r;
                const bool qv = q < c.HW;
                u32x4 g[4];
                if (pass == 1) {
#pragma unroll
                    for (int qd = 0; qd < 4; qd++) g[qd] = gn[qd];
                    if (qt + 1 < QT) g3_load(it, qt + 1, gn);
                    else g3_load(it + G::NW, 0, gn);
                }
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KSB; ks++) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(arow + 32 * ks));
                    acc = MFMA(a, __builtin_bit_cast(bf16x8, wreg[ks][qt]), acc);
                }
                if (pass == 0) {
                    if (qv) {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const float x = acc[i] + bias[qt];
                            ps += x;
                            pq = fmaf(x, x, pq);
                        }
                    }
                } else if (qv) {
                    const float bmean = bstat[2 * bd], brstd = bstat[2 * bd + 1];
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] += bias[qt];
                    float y[16];
                    ln_apply(acc, bmean, brstd, g, y);
                    img_store(img, RS, bd * c.HW + q, rtw, c.h, y);
                }
            }
            if (pass == 0) {
                ps = wave_red<0>(ps);
                pq = wave_red<0>(pq);
                if (c.lane == 0) {
                    ipart[2 * it] = ps;
                    ipart[2 * it + 1] = pq;
                }
            }
        }
        if (pass == 0) {
            STAMP(c, ST_HLN);
            lds_barrier();
            if ((int)threadIdx.x < NB) {
                const int bd = threadIdx.x;
                float s = 0.f, q = 0.f;
                for (int j = 0; j < NRC; j++) {
                    s += ipart[2 * (bd * NRC + j)];
                    q += ipart[2 * (bd * NRC + j) + 1];
                }
                const float ninv = 1.0f / (float)(C * c.HW);
                const float mean = s * ninv;
                const float var = fmaxf(q * ninv - mean * mean, 0.f);
                bstat[2 * bd] = mean;
                bstat[2 * bd + 1] = 1.0f / sqrtf(var + 1e-5f);
            }
            lds_barrier();
            STAMP(c, ST_HFC);
        }
    }
    // padded columns of the image (beyond NB*HW) feed only padded outputs, but must be finite
#pragma unroll
    for (int ct = 0; ct < NC; ct++)
        if (!k[ct].valid) {
#pragma unroll
            for (int rt = 0; rt < NRCW; rt++) {
                const unsigned o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                img_store_pk(img, RS, k[ct].col, c.wr * NRCW + rt, c.h, o);
            }
        }
    lds_barrier();   // image complete, TB dead: region 2 is the ring again
    STAMP(c, ST_BCAST);
    // c2: 1x1 conv C -> C on the image (own columns), accumulated into X
    ring.issue(FS0);
    ring.issue(FS1);
    conv_lds<G, KSC, 1, NRC, NRCW, RS>(c, k, ring, X, nx0, nx1);
}

template <class G, int KSB>
__global__ __launch_bounds__(G::NT, G::WPE) void k_bcast(NetDims d, TowerDev tw, NetIO io, LdsPlan L, int NB, int blk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int C = G::C, NRCW = G::NRCW, NC = G::NC;
    Ctx c;
    ctx_init<G>(c, smem, L, d, NB);
    Col k[NC];
    if (!setup<IN_X, NC>(d, io, NB, c, k)) return;
    WRing<G::NW, G::SM> ring;
    f32x16 X[NRCW][NC];
    bcast_body<G, KSB>(c, k, tw, io, L, smem, ring, X, NB, blk, 0, 0);
    x_store<NRCW, NC>(c, k, io.pool, C, X);
    STAMP(c, ST_XSTORE);
    STAMP_PRINT(c, "bcast");
}

// Broadcast block blk fused with the run of bottleneck blocks b0..b1-1 behind it (b0 = blk + 1) and that run's tail:
// one X round trip through HBM and one launch boundary less than k_bcast + k_seg<IN_X>.  Runs on the broadcast
// kernel's LDS plan (its image holds C channels, enough for the bottleneck images).
// The whole tower in one launch when it holds one broadcast block (Go): conv0 + blocks 0..blk-1 + broadcast block blk +
// blocks blk+1..R-1 + tail.  X never travels between launches; it is parked in the destination slot only across the
// middle of the broadcast block.
template <class G, int IN, int KSB, int OUT>
__global__ __launch_bounds__(G::NT, G::WPE) void k_tower(NetDims d, TowerDev tw, HeadDev hd, NetIO io, LdsPlan L, int NB, int blk,
                                              int b1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NRCW = G::NRCW, NC = G::NC;
    using BS = BlockStages<G>;
    Ctx c;
    ctx_init<G>(c, smem, L, d, NB);
    Col k[NC];
    if (!setup<IN, NC>(d, io, NB, c, k)) return;
    WRing<G::NW, G::SM> ring;
    f32x16 X[NRCW][NC];
    seg_front<G, IN>(c, k, d, tw, io, L, smem, ring, X, NB, 0, blk);
    seg_rest<G, OUT_REGS>(c, k, d, tw, hd, io, L, smem, ring, X, NB, 0, blk);
    const int b0 = blk + 1;
    bcast_body<G, KSB, true>(c, k, tw, io, L, smem, ring, X, NB, blk, b0 < b1 ? BS::open0 : 0, b0 < b1 ? BS::open1 : 0);
    seg_rest<G, OUT>(c, k, d, tw, hd, io, L, smem, ring, X, NB, b0, b1);
}

// The same for a tower with two broadcast blocks (blk and blk2; PegSolitair: 16 blocks, every eighth a broadcast block).
template <class G, int IN, int KSB, int OUT>
__global__ __launch_bounds__(G::NT, G::WPE) void k_tower2(NetDims d, TowerDev tw, HeadDev hd, NetIO io, LdsPlan L, int NB, int blk,
                                               int blk2, int b1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NRCW = G::NRCW, NC = G::NC;
    using BS = BlockStages<G>;
    Ctx c;
    ctx_init<G>(c, smem, L, d, NB);
    Col k[NC];
    if (!setup<IN, NC>(d, io, NB, c, k)) return;
    WRing<G::NW, G::SM> ring;
    f32x16 X[NRCW][NC];
    seg_front<G, IN>(c, k, d, tw, io, L, smem, ring, X, NB, 0, blk);
    seg_rest<G, OUT_REGS>(c, k, d, tw, hd, io, L, smem, ring, X, NB, 0, blk);
    const int b0 = blk + 1;
    bcast_body<G, KSB, true>(c, k, tw, io, L, smem, ring, X, NB, blk, b0 < blk2 ? BS::open0 : 0, b0 < blk2 ? BS::open1 : 0);
    seg_rest<G, OUT_REGS>(c, k, d, tw, hd, io, L, smem, ring, X, NB, b0, blk2);
    const int b2 = blk2 + 1;
    bcast_body<G, KSB, true>(c, k, tw, io, L, smem, ring, X, NB, blk2, b2 < b1 ? BS::open0 : 0, b2 < b1 ? BS::open1 : 0);
    seg_rest<G, OUT>(c, k, d, tw, hd, io, L, smem, ring, X, NB, b2, b1);
}

template <class G, int KSB, int OUT>
__global__ __launch_bounds__(G::NT, G::WPE) void k_bseg(NetDims d, TowerDev tw, HeadDev hd, NetIO io, LdsPlan L, int NB, int blk,
                                             int b0, int b1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NRCW = G::NRCW, NC = G::NC;
    using BS = BlockStages<G>;
    Ctx c;
    ctx_init<G>(c, smem, L, d, NB);
    Col k[NC];
    if (!setup<IN_X, NC>(d, io, NB, c, k)) return;
    WRing<G::NW, G::SM> ring;
    f32x16 X[NRCW][NC];
    bcast_body<G, KSB>(c, k, tw, io, L, smem, ring, X, NB, blk, b0 < b1 ? BS::open0 : 0, b0 < b1 ? BS::open1 : 0);
    seg_rest<G, OUT>(c, k, d, tw, hd, io, L, smem, ring, X, NB, b0, b1);
}

// ------------------------------- host side: packing -------------------------------------------
inline unsigned short f2bf(float f) {
    unsigned int u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline unsigned short f2h(float f);
// fp32 -> MFMA operand element (round to nearest even)
inline unsigned short f2op(float f) { return MZ_F16 ? f2h(f) : f2bf(f); }
inline unsigned short f2h(float f) {
    _Float16 h = (_Float16)f;
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
}

struct FragWriter {
    std::vector<unsigned short>& out;
    // w(row, k) -> weight for MFMA row (0..31) and k index (0..15) of this fragment
    template <typename F>
    void frag(F w) {
        size_t base = out.size();
        out.resize(base + 512);
        for (int l = 0; l < 64; l++)
            for (int j = 0; j < 8; j++) out[base + l * 8 + j] = f2op(w(l & 31, 8 * (l >> 5) + j));
    }
};

// conv W[Cout][Cin][taps]: fragment order (tap, ks, ro) with natural k; ks runs to cin_pad/16
// (cin_use < Cin: only the first cin_use input channels of the tensor are packed)
void pack_conv(FragWriter& fw, const float* W, int Cout, int Cin, int taps, int cin_pad, int cin_use = -1, float ws = 1.f) {
    if (cin_use < 0) cin_use = Cin;
    for (int t = 0; t < taps; t++)
        for (int ks = 0; ks < cin_pad / 16; ks++)
            for (int ro = 0; ro < Cout / 32; ro++)
                fw.frag([&](int row, int kp) {
                    const int ci = 16 * ks + kp;
                    if (ci >= cin_use) return 0.f;
                    return ws * W[((size_t)(32 * ro + row) * Cin + ci) * taps + t];
                });
}
// 1x1 conv whose B operands come from registers (ln_to_regs): k-step ks, k index kp <-> input channel
// 32(ks >> 1) + 16(ks & 1) + 4(kp >> 3) + (kp & 3) + 8((kp & 7) >> 2)
void pack_conv_kperm(FragWriter& fw, const float* W, int Cout, int Cin, float ws = 1.f) {
    for (int ks = 0; ks < Cin / 16; ks++)
        for (int ro = 0; ro < Cout / 32; ro++)
            fw.frag([&](int row, int kp) {
                const int ci = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * (kp >> 3) + (kp & 3) + 8 * ((kp & 7) >> 2);
                return ws * W[(size_t)(32 * ro + row) * Cin + ci];
            });
}
// gamma/beta [Cn][HW] -> fp16 [rt][qd][h][HW+1]{g0..3, b0..3}: lane (cell p, half h) of row tile rt reads
// channels 32rt + 8qd + 4h + e; entry HW is all zero (padded columns)
void pack_gb(std::vector<unsigned short>& out, const float* g, const float* b, int Cn, int HW, float inv_s = 1.f) {
    for (int rt = 0; rt < Cn / 32; rt++)
        for (int qd = 0; qd < 4; qd++)
            for (int h = 0; h < 2; h++)
                for (int p = 0; p <= HW; p++) {
                    const int ch = 32 * rt + 8 * qd + 4 * h;
                    for (int e = 0; e < 4; e++) out.push_back(p < HW ? f2h(inv_s * g[(size_t)(ch + e) * HW + p]) : 0);
                    for (int e = 0; e < 4; e++) out.push_back(p < HW ? f2h(inv_s * b[(size_t)(ch + e) * HW + p]) : 0);
                }
}

// Scale S of a LayerNorm whose output goes through ln_apply_pk (see there): the power of two >= 256 that bounds
// |gamma| * sqrt(N - 1) + |beta| (no LayerNorm output of N elements exceeds sqrt(N - 1) in magnitude), so the clamp of the
// fp16 mix-FMA never cuts a value.  1 for bf16 operands (ReLU by integer max there).  0: the affine is too large for the
// fp16 path (S beyond 2^13 would push typical activations into the subnormals), or the consumer's weights would overflow.
float ln_scale(const float* g, const float* b, int Cn, int HW, const float* W, size_t wcount) {
    if (!MZ_F16) return 1.f;
    float gm = 0.f, bm = 0.f, wm = 0.f;
    for (size_t i = 0; i < (size_t)Cn * HW; i++) {
        gm = std::max(gm, std::fabs(g[i]));
        bm = std::max(bm, std::fabs(b[i]));
    }
    for (size_t i = 0; i < wcount; i++) wm = std::max(wm, std::fabs(W[i]));
    const float bound = gm * std::sqrt((float)(Cn * HW)) + bm;
    float S = 256.f;
    while (S < bound) S *= 2.f;
    if (!(S <= 8192.f) || !(wm * S < 60000.f)) return 0.f;
    return S;
}

struct TowerHost {
    std::vector<unsigned short> stream, gb, wbf;
    std::vector<float> bb, wact;
    TowerDev dev{};
};

}  // namespace MZ_IMPL
using namespace MZ_IMPL;

struct MfmaNet {
    NetDims d{};
    int geom = 0, NB = 0, WR = 1, NW = 8, SM = 32, NC = 1, COLS = 256;
    LdsPlan Lseg{}, Lbc{}, Ltw{};   // k_seg, k_bcast / k_bseg, k_tower
    void* bufs[16] = {nullptr};
    int nbufs = 0;
    int fuse = 2;             // launch chain: 0 k_seg -> k_bcast -> k_seg, 1 k_seg -> k_bseg, 2 k_tower where instantiated
    float* hfeat = nullptr;   // [max_batch][3][HW]
    int max_batch = 0;
    TowerDev tw[2]{};
    HeadDev hd{};
    bool packed = false;
};

namespace MZ_IMPL {

bool shape_supported(const NetDims& d) {
    const bool c128 = d.C == 128 && d.Cb == 64, c256 = d.C == 256 && d.Cb == 128;
    return (c128 || c256) && d.O <= 32 && d.a <= 16 && d.HW <= 96 && d.HW >= 4;
}

int align16(int x) { return (x + 15) / 16 * 16; }

void plan_tail(LdsPlan& L, int end, int NW, int Cb) {
    end = align16(end);
    while ((end - L.img_off) % 256) end += 16;   // zero rows start at the image's bank phase
    L.zrow_off = end;
    const int rest = (2 * MAXNB * NW * 2 + 2 * 256 + 2 * MAXNB) * 4 + 4 * MAXNB * 4 + 16;
    L.zstride = 2 * Cb + 16;
    L.zbytes = ZROWS * L.zstride;
    if (L.zbytes < 1024) L.zbytes = 1024;        // conv0 reads up to (C + 16) * 2 + 16 bytes of zero row 0
    if (end + L.zbytes + rest > 160 * 1024) {    // no room for 16 rows (Go 5x5's broadcast plan): one shared zero row
        L.zstride = 0;
        L.zbytes = 1024;
    }
    end += L.zbytes;
    L.stat_off = end;
    end += (2 * MAXNB * NW * 2 + 2 * 256 + 2 * MAXNB) * 4;  // wpart x2, ipart, bstat
    L.meta_off = end;
    end += 4 * MAXNB * 4;
    L.total = align16(end);
}

// k_seg: [ring][image]
LdsPlan make_plan_seg(const NetDims& d, int COLS, int NW, int SM) {
    LdsPlan L{};
    const int rs0 = d.C * 2 + 16;
    // the ring comes first: LDS-DMA destinations stay below 64 KiB
    L.ring_off = 0;
    L.img_off = NSLOT * SM * 1024;
    L.img0_off = L.img_off;
    int end = L.img_off + align16(COLS * rs0);
    L.head_off = 0;
    plan_tail(L, end, NW, d.Cb);
    return L;
}
// k_bcast: [region 2 = ring / TB][image, C channels].  with_conv0 (the one-launch towers): conv0's wider input image lies
// right behind the ring, over the tail of the TB region and the block image - both are dead while conv0 runs.
LdsPlan make_plan_bcast(const NetDims& d, int COLS, int NB, int NW, int SM, bool with_conv0 = false) {
    LdsPlan L{};
    const int rsc = d.C * 2 + 16, rs0 = d.C * 2 + 16;
    L.kp = (d.HW + 15) / 16 * 16;
    L.qt = (d.HW + 31) / 32;
    L.tbs = (NB * L.kp + 8) * 2;
    const int ring = NSLOT * SM * 1024, tbb = align16(d.C * L.tbs);
    L.ring_off = 0;
    L.tb_off = 0;
    L.img_off = ring > tbb ? ring : tbb;
    L.img0_off = with_conv0 ? ring : L.img_off;
    int end = L.img_off + align16(COLS * rsc);
    if (with_conv0 && L.img0_off + align16(COLS * rs0) > end) end = L.img0_off + align16(COLS * rs0);
    plan_tail(L, end, NW, d.Cb);
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

}  // namespace MZ_IMPL
using namespace MZ_IMPL;

MfmaNet* mfma_create(const NetDims& d, int max_batch, std::string* err) {
    if (!shape_supported(d)) {
        if (err)
            *err = "the MFMA paths (bf16 / fp16 operands) support numChannels/numBottleneckChannels 128/64 and 256/128 on boards of 4..96 "
                   "cells; use MZ_PRECISION_FP32 for other shapes";
        return nullptr;
    }
    MfmaNet* n = new MfmaNet();
    n->d = d;
    // Workgroup geometry (see Geo): numChannels 256 -> 2x4 waves on 128 columns; numChannels 128 -> 1x8 waves on 256
    // columns (measured best on Go 9x9: 1.31 ms per recurrentInference of 4096 boards against 1.55 / 1.58 / 1.42 ms
    // for the 2x3, 2x4 and 2x8 grids, which trade column packing for a second workgroup per CU).
    // MZ_GEOM=1 (numChannels 128 only) selects 4 waves x 2 column tiles instead: one wave per SIMD with the whole
    // register file, 1.0 instead of 1.5 LDS reads per MFMA in the 3x3 convs.  Measured 15 % slower on Go 9x9
    // (5.70 against 4.96 ms at 16384 boards): the matrix loops gain, but a lone wave per SIMD runs the LayerNorm
    // phases (dependent VALU chains, LDS and gamma/beta round trips) with nothing to overlap them.
    struct GeomDesc { int WR, NW, SM, NC; };
    static const GeomDesc kGeom[3] = {{1, 8, 24, 1}, {2, 8, 24, 1}, {1, 4, 24, 2}};
    n->geom = d.C == 256 ? 1 : 0;
    if (const char* e = std::getenv("MZ_GEOM"))
        if (std::atoi(e) == 1 && d.C == 128) n->geom = 2;
    n->WR = kGeom[n->geom].WR;
    n->NW = kGeom[n->geom].NW;
    n->SM = kGeom[n->geom].SM;
    n->NC = kGeom[n->geom].NC;
    n->COLS = 32 * (n->NW / n->WR) * n->NC;
    n->NB = n->COLS / d.HW;
    if (n->NB > MAXNB) n->NB = MAXNB;
    if (n->NB < 1) {
        if (err) *err = "board does not fit the workgroup tile";
        delete n;
        return nullptr;
    }
    n->Lseg = make_plan_seg(d, n->COLS, n->NW, n->SM);
    n->Lbc = make_plan_bcast(d, n->COLS, n->NB, n->NW, n->SM);
    n->Ltw = make_plan_bcast(d, n->COLS, n->NB, n->NW, n->SM, true);
    const int bc_items = n->NB * (d.C / 32) * n->Lbc.qt;
    const bool has_bcast = d.bcast_every_n > 0 && d.R >= d.bcast_every_n;
    if (n->Lseg.total > 160 * 1024 || (has_bcast && (n->Lbc.total > 160 * 1024 || bc_items > 256))) {
        if (err) *err = "LDS plan does not fit 160 KiB";
        delete n;
        return nullptr;
    }
    // MZ_FUSE (read when the context is created): the launch chain, see forward_t; MZ_DEBUG_STOP needs the unfused one
    if (const char* e = std::getenv("MZ_FUSE")) n->fuse = std::atoi(e);
    if (std::getenv("MZ_DEBUG_STOP")) n->fuse = 0;
    n->max_batch = max_batch > 0 ? max_batch : 1;
    if (hipMalloc(reinterpret_cast<void**>(&n->hfeat), (size_t)n->max_batch * 3 * d.HW * sizeof(float)) != hipSuccess) {
        if (err) *err = "hipMalloc failed in mfma_create";
        delete n;
        return nullptr;
    }
    return n;
}

void mfma_destroy(MfmaNet* n) {
    if (!n) return;
    for (int i = 0; i < n->nbufs; i++) (void)hipFree(n->bufs[i]);
    if (n->hfeat) (void)hipFree(n->hfeat);
    delete n;
}

namespace MZ_IMPL {
int scale_error(std::string* err) {
    if (err)
        *err = "a LayerNorm's gamma / beta (or the weights of the conv behind it) are too large for the fp16-operand path's scaled "
               "activations; use MZ_PRECISION_BF16 or MZ_PRECISION_FP32 for this network";
    return -1;
}
}  // namespace MZ_IMPL

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
    const int cin0_pad[2] = {32, C};   // dynamics: the C hidden-state channels; the action planes go through TowerDev::wact
    for (int t = 0; t < 2; t++) {
        TowerHost& T = th[t];
        FragWriter fw{T.stream};
        const float* w0 = take((size_t)C * cin0[t] * 9);
        pack_conv(fw, w0, C, cin0[t], 9, cin0_pad[t], t == 0 ? cin0[t] : C);
        if (t == 1) {
            T.wact.assign((size_t)d.a * 9 * C, 0.f);
            for (int pl = 0; pl < d.a; pl++)
                for (int tap = 0; tap < 9; tap++)
                    for (int o = 0; o < C; o++)
                        T.wact[((size_t)pl * 9 + tap) * C + o] = w0[((size_t)o * cin0[t] + C + pl) * 9 + tap];
        }
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
                const float s1 = ln_scale(g1, b1, C, HW, c1, (size_t)Cb * C), s2 = ln_scale(g2, b2, Cb, HW, c2, (size_t)Cb * Cb * 9);
                const float s3 = ln_scale(g3, b3, Cb, HW, c3, (size_t)Cb * Cb * 9), s4 = ln_scale(g4, b4, Cb, HW, c4, (size_t)C * Cb);
                if (s1 == 0.f || s2 == 0.f || s3 == 0.f || s4 == 0.f) return scale_error(err);
                if (n->WR == 1) pack_conv_kperm(fw, c1, Cb, C, s1);
                else pack_conv(fw, c1, Cb, C, 1, C, -1, s1);
                pack_conv(fw, c2, Cb, Cb, 9, Cb, -1, s2);
                pack_conv(fw, c3, Cb, Cb, 9, Cb, -1, s3);
                if (n->WR == 1) pack_conv_kperm(fw, c4, C, Cb, s4);
                else pack_conv(fw, c4, C, Cb, 1, Cb, -1, s4);
                pack_gb(T.gb, g1, b1, C, HW, 1.f / s1);
                pack_gb(T.gb, g2, b2, Cb, HW, 1.f / s2);
                pack_gb(T.gb, g3, b3, Cb, HW, 1.f / s3);
                pack_gb(T.gb, g4, b4, Cb, HW, 1.f / s4);
            } else {
                const float *g1 = take((size_t)C * HW), *b1 = take((size_t)C * HW), *c1 = take((size_t)C * C);
                const float *g2 = take((size_t)C * HW), *b2 = take((size_t)C * HW);
                const float *wb = take((size_t)HW * HW), *bbv = take((size_t)HW);
                const float *g3 = take((size_t)C * HW), *b3 = take((size_t)C * HW), *c2 = take((size_t)C * C);
                // LN1 goes through ln_apply_pk (scaled, c1 takes the scale); LN2 and LN3 through ln_apply (plain ReLU)
                const float s1 = ln_scale(g1, b1, C, HW, c1, (size_t)C * C);
                if (s1 == 0.f) return scale_error(err);
                pack_conv(fw, c1, C, C, 1, C, -1, s1);
                pack_conv(fw, c2, C, C, 1, C);
                pack_gb(T.gb, g1, b1, C, HW, 1.f / s1);
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
                                T.wbf.push_back((q < HW && p < HW) ? f2op(wb[(size_t)q * HW + p]) : 0);
                            }
            }
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
        if (upload(n, th[t].stream, &p, 1024, err)) return -1;
        th[t].dev.stream = (const u32x4*)p;
        if (upload(n, th[t].gb, &p, 2048, err)) return -1;   // gtab_issue copies whole KiB pieces
        th[t].dev.gb = (const u32x4*)p;
        if (upload(n, th[t].wbf, &p, 64, err)) return -1;
        th[t].dev.wbf = (const u32x4*)p;
        if (th[t].bb.empty()) th[t].bb.assign(128, 0.f);
        if (upload(n, th[t].bb, &p, 64, err)) return -1;
        th[t].dev.bb = (const float*)p;
        th[t].dev.wact = nullptr;
        if (!th[t].wact.empty()) {
            if (upload(n, th[t].wact, &p, 64, err)) return -1;
            th[t].dev.wact = (const float*)p;
        }
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
                    hb + o_vlnb, hb + o_vfc1w, hb + o_vfc1b, hb + o_vfc2w, hb + o_vfc2b, n->hfeat};
    n->packed = true;
    return 0;
}

namespace MZ_IMPL {

template <class G, int IN, int OUT>
void launch_seg(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int b0,
                int b1) {
    lds_optin(reinterpret_cast<const void*>(&k_seg<G, IN, OUT>));
    hipLaunchKernelGGL((k_seg<G, IN, OUT>), dim3(grid), dim3(G::NT), n->Lseg.total, st, d, tw, n->hd, io, n->Lseg, n->NB,
                       b0, b1);
}

template <class G, int KSB>
void launch_bcast(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int blk) {
    lds_optin(reinterpret_cast<const void*>(&k_bcast<G, KSB>));
    hipLaunchKernelGGL((k_bcast<G, KSB>), dim3(grid), dim3(G::NT), n->Lbc.total, st, d, tw, io, n->Lbc, n->NB, blk);
}

template <class G, int KSB, int OUT>
void launch_bseg(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int blk, int b0,
                 int b1) {
    lds_optin(reinterpret_cast<const void*>(&k_bseg<G, KSB, OUT>));
    hipLaunchKernelGGL((k_bseg<G, KSB, OUT>), dim3(grid), dim3(G::NT), n->Lbc.total, st, d, tw, n->hd, io, n->Lbc, n->NB,
                       blk, b0, b1);
}
template <class G, int IN, int KSB>
void launch_tower(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int blk) {
    lds_optin(reinterpret_cast<const void*>(&k_tower<G, IN, KSB, OUT_FINAL>));
    hipLaunchKernelGGL((k_tower<G, IN, KSB, OUT_FINAL>), dim3(grid), dim3(G::NT), n->Ltw.total, st, d, tw, n->hd, io, n->Ltw,
                       n->NB, blk, d.R);
}
// one launch for towers with a single broadcast block on the board sizes of the reference's games (k-step counts 2, 4, 6
// of the broadcast linear: 25-, 49- and 81-cell boards) and the two-waves-per-SIMD geometries; false: not this shape
template <class G, int IN>
bool try_launch_tower(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int blk) {
    if constexpr (G::NC != 1) {
        return false;
    } else {
        if (n->Ltw.total > 160 * 1024) return false;
        switch (n->Lbc.kp / 16) {
            case 2: launch_tower<G, IN, 2>(n, st, grid, d, tw, io, blk); return true;
            case 4: launch_tower<G, IN, 4>(n, st, grid, d, tw, io, blk); return true;
            case 6: launch_tower<G, IN, 6>(n, st, grid, d, tw, io, blk); return true;
            default: return false;
        }
    }
}
// two broadcast blocks in one launch: instantiated for numChannels 128 on 49-cell boards (PegSolitair)
template <class G, int IN>
bool try_launch_tower2(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int blk,
                       int blk2) {
    if constexpr (G::NC != 1 || G::C != 128) {
        return false;
    } else {
        if (n->Ltw.total > 160 * 1024 || n->Lbc.kp / 16 != 4) return false;
        lds_optin(reinterpret_cast<const void*>(&k_tower2<G, IN, 4, OUT_FINAL>));
        hipLaunchKernelGGL((k_tower2<G, IN, 4, OUT_FINAL>), dim3(grid), dim3(G::NT), n->Ltw.total, st, d, tw, n->hd, io, n->Ltw,
                           n->NB, blk, blk2, d.R);
        return true;
    }
}
template <class G, int OUT>
void launch_bseg_k(MfmaNet* n, hipStream_t st, int grid, const NetDims& d, const TowerDev& tw, const NetIO& io, int blk, int b0,
                   int b1) {
    switch (n->Lbc.kp / 16) {
        case 1: launch_bseg<G, 1, OUT>(n, st, grid, d, tw, io, blk, b0, b1); break;
        case 2: launch_bseg<G, 2, OUT>(n, st, grid, d, tw, io, blk, b0, b1); break;
        case 3: launch_bseg<G, 3, OUT>(n, st, grid, d, tw, io, blk, b0, b1); break;
        case 4: launch_bseg<G, 4, OUT>(n, st, grid, d, tw, io, blk, b0, b1); break;
        case 5: launch_bseg<G, 5, OUT>(n, st, grid, d, tw, io, blk, b0, b1); break;
        default: launch_bseg<G, 6, OUT>(n, st, grid, d, tw, io, blk, b0, b1); break;
    }
}

template <class G>
int forward_t(MfmaNet* n, hipStream_t st, bool initial, const NetDims& d, const NetIO& io) {
    const int grid = (io.B + n->NB - 1) / n->NB;
    const TowerDev& tw = n->tw[initial ? 0 : 1];
#if defined(MZ_FAST_BUILD)   // experiment builds (tools/fastlib.sh): only the one-launch Go 9x9 tower is instantiated
    {
        int e = 0;
        while (e < d.R && tw.type[e] == 0) e++;
        if (initial) launch_tower<G, IN_OBS, 6>(n, st, grid, d, tw, io, e);
        else launch_tower<G, IN_HIDDEN, 6>(n, st, grid, d, tw, io, e);
        hipLaunchKernelGGL(k_heads, dim3((io.B + HS - 1) / HS), dim3(HT), heads_lds_bytes(d), st, d, n->hd, io);
        return 0;
    }
#else
    // MZ_FUSE=0: every broadcast block as its own launch (k_bcast); MZ_FUSE=1: fused with the blocks behind it (k_bseg);
    // default: that, and the whole tower as one launch (k_tower) where it holds a single broadcast block
    const bool fuse = n->fuse != 0;
    if (fuse) {
        int e = 0;
        while (e < d.R && tw.type[e] == 0) e++;
        bool last = e == d.R;
        if (!last) {   // exactly one broadcast block: the whole tower as one launch
            int e2 = e + 1;
            while (e2 < d.R && tw.type[e2] == 0) e2++;
            const bool tower = n->fuse >= 2;
            if (tower && e2 == d.R &&
                (initial ? try_launch_tower<G, IN_OBS>(n, st, grid, d, tw, io, e)
                         : try_launch_tower<G, IN_HIDDEN>(n, st, grid, d, tw, io, e))) {
                hipLaunchKernelGGL(k_heads, dim3((io.B + HS - 1) / HS), dim3(HT), heads_lds_bytes(d), st, d, n->hd, io);
                return 0;
            }
            if (tower && e2 < d.R) {   // exactly two broadcast blocks
                int e3 = e2 + 1;
                while (e3 < d.R && tw.type[e3] == 0) e3++;
                if (e3 == d.R && (initial ? try_launch_tower2<G, IN_OBS>(n, st, grid, d, tw, io, e, e2)
                                          : try_launch_tower2<G, IN_HIDDEN>(n, st, grid, d, tw, io, e, e2))) {
                    hipLaunchKernelGGL(k_heads, dim3((io.B + HS - 1) / HS), dim3(HT), heads_lds_bytes(d), st, d, n->hd, io);
                    return 0;
                }
            }
        }
        if (initial) {
            if (last) launch_seg<G, IN_OBS, OUT_FINAL>(n, st, grid, d, tw, io, 0, e);
            else launch_seg<G, IN_OBS, OUT_X>(n, st, grid, d, tw, io, 0, e);
        } else {
            if (last) launch_seg<G, IN_HIDDEN, OUT_FINAL>(n, st, grid, d, tw, io, 0, e);
            else launch_seg<G, IN_HIDDEN, OUT_X>(n, st, grid, d, tw, io, 0, e);
        }
        while (!last) {
            const int blk = e, b0 = e + 1;
            e = b0;
            while (e < d.R && tw.type[e] == 0) e++;
            last = e == d.R;
            if (last) launch_bseg_k<G, OUT_FINAL>(n, st, grid, d, tw, io, blk, b0, e);
            else launch_bseg_k<G, OUT_X>(n, st, grid, d, tw, io, blk, b0, e);
        }
        hipLaunchKernelGGL(k_heads, dim3((io.B + HS - 1) / HS), dim3(HT), heads_lds_bytes(d), st, d, n->hd, io);
        return 0;
    }
    // runs of bottleneck blocks separated by broadcast blocks
    int b = 0;
    bool first = true;
    // diagnostic: MZ_DEBUG_STOP=n leaves the raw residual stream after n blocks in the destination slot
    static const int dbg_stop = std::getenv("MZ_DEBUG_STOP") ? std::atoi(std::getenv("MZ_DEBUG_STOP")) : -1;
    while (true) {
        int e = b;
        while (e < d.R && tw.type[e] == 0) e++;
        bool last = e == d.R;
        if (dbg_stop >= 0 && dbg_stop <= e) {
            e = dbg_stop;
            last = false;
        }
        if (first) {
            if (initial) {
                if (last) launch_seg<G, IN_OBS, OUT_FINAL>(n, st, grid, d, tw, io, b, e);
                else launch_seg<G, IN_OBS, OUT_X>(n, st, grid, d, tw, io, b, e);
            } else {
                if (last) launch_seg<G, IN_HIDDEN, OUT_FINAL>(n, st, grid, d, tw, io, b, e);
                else launch_seg<G, IN_HIDDEN, OUT_X>(n, st, grid, d, tw, io, b, e);
            }
        } else {
            if (last) launch_seg<G, IN_X, OUT_FINAL>(n, st, grid, d, tw, io, b, e);
            else launch_seg<G, IN_X, OUT_X>(n, st, grid, d, tw, io, b, e);
        }
        if (last) hipLaunchKernelGGL(k_heads, dim3((io.B + HS - 1) / HS), dim3(HT), heads_lds_bytes(d), st, d, n->hd, io);
        first = false;
        if (last || (dbg_stop >= 0 && dbg_stop == e)) break;
        switch (n->Lbc.kp / 16) {
            case 1: launch_bcast<G, 1>(n, st, grid, d, tw, io, e); break;
            case 2: launch_bcast<G, 2>(n, st, grid, d, tw, io, e); break;
            case 3: launch_bcast<G, 3>(n, st, grid, d, tw, io, e); break;
            case 4: launch_bcast<G, 4>(n, st, grid, d, tw, io, e); break;
            case 5: launch_bcast<G, 5>(n, st, grid, d, tw, io, e); break;
            default: launch_bcast<G, 6>(n, st, grid, d, tw, io, e); break;
        }
        b = e + 1;
    }
    return 0;
#endif
}

}  // namespace MZ_IMPL
using namespace MZ_IMPL;

int mfma_tile_boards(const MfmaNet* n) { return n ? n->NB : 1; }

int mfma_forward(MfmaNet* n, hipStream_t st, bool initial, const NetDims& d, const NetIO& io, std::string* err) {
    if (!n || !n->packed) {
        if (err) *err = "mfma weights not packed";
        return -1;
    }
    if (io.g_base < 0 || io.B + io.g_base > n->max_batch) {
        if (err) *err = "batch exceeds the context's inference capacity";
        return -1;
    }
    int rc;
#if defined(MZ_FAST_BUILD)
    rc = forward_t<Geo<128, 64, 1, 8, 24, 2>>(n, st, initial, d, io);
#else
    switch (n->geom) {
        case 0: rc = forward_t<Geo<128, 64, 1, 8, 24, 2>>(n, st, initial, d, io); break;
        case 1: rc = forward_t<Geo<256, 128, 2, 8, 24, 2>>(n, st, initial, d, io); break;
        default: rc = forward_t<Geo<128, 64, 1, 4, 24, 1, 2>>(n, st, initial, d, io); break;
    }
#endif
    if (hipError_t e = take_optin_error()) {
        if (err) *err = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e);
        return -1;
    }
    return rc;
}

}  // namespace mz
