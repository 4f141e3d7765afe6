// Device-side description of the representation / dynamics / prediction nets and the launchers of
// the inference kernels (fp32 exact path and bf16 MFMA path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mz {

constexpr int kMaxResiduals = 24;

struct NetDims {
    int H, Wd, HW, O, a, C, Cb, R, A, two_players;
    int bcast_every_n;
    double value_scale;
};

// Offsets (in floats) into the packed fp32 weight buffer used by the fp32 kernel.
struct BlockOff {
    int type;  // 0 = bottleneck, 1 = broadcast
    long long ln1g, ln1b, c1, ln2g, ln2b, c2, ln3g, ln3b, c3, ln4g, ln4b, c4, bw, bb;
};
struct TowerOff {
    long long conv0;     // [Cin][9][CoutPad], or A-operand units (fp32_conv_on_mfma)
    int cin0;            // input channels of conv0 (O or C+a)
    BlockOff blocks[kMaxResiduals];
};
struct HeadOff {
    long long pconv, plng, plnb, pfcw, pfcb;            // pconv [C][4], pfcw [2HW][A]
    long long vconv, vlng, vlnb, vfc1w, vfc1b, vfc2w, vfc2b;  // vfc1w [HW][C]
};
struct NetProgramF32 {
    TowerOff rep, dyn;
    HeadOff head;
    int c_pad, cb_pad;  // channel counts rounded up to a multiple of 4
};

// Batch I/O of one inference launch.  Hidden states are fp32 slots of C*HW floats in `pool`: Q16 order (above; [C][HW] for toy shapes) on the fp32 path,
// [C/4][HW][4] (channel quads interleaved) on the MFMA path; mz_read_hidden always returns [C][H][W].
struct NetIO {
    const float* obs;         // initial: [B][O][HW]
    const int* src_slot;      // recurrent: [B] (global slot)
    const int* dst_slot;      // [B] (global slot)
    const int* action;        // recurrent: [B]
    const int* active;        // [B] or nullptr (all active)
    const int* action_cells;  // [A]
    float* pool;
    float* logits;            // [B][A]
    float* policy;            // [B][A]
    double* value;            // [B]
    int B;
    int g_base;               // position of sample 0 in the context's per-sample scratch (two half batches in flight at once)
};

// Opt a kernel in to > 64 KiB of dynamic LDS.  hipFuncSetAttribute applies to the function on the CURRENT device, so
// the "already done" state is kept per (function, device) - one process may hold contexts on several GPUs - and under a
// lock.  A failure is remembered for take_optin_error(), which the forward entry points report.
void lds_optin(const void* fn);
hipError_t take_optin_error();

// The fp32 kernel keeps the activations of a network whose channel counts are multiples of 16 in the "Q16" order
//     index(c, p) = (((c / 16) * 4 + c % 4) * HW + p) * 4 + (c / 4) % 4        [c/16][c%4][cell][(c/4)%4]
// so that the B operands of four consecutive k-steps of v_mfma_f32_16x16x4_f32 (lane (cell, kq) of k-step 4G + s takes channel
// 16G + 4s + kq) are ONE 16-byte LDS read; hidden-state slots of the fp32 path are stored in that order too (mz_read_hidden
// un-permutes), and the LayerNorm gamma / beta tables are packed in it.  In LDS and in the gamma / beta tables a plane has
// HW + 1 cells, the last one a zero cell (what the convs read for off-board taps); the slots have planes of HW cells.  Other networks (toy shapes of the tests) stay in
// [c][cell] and run every conv on the vector ALU.
__host__ __device__ inline bool fp32_q16(int C, int Cb, int HW) { return (C & 15) == 0 && (Cb & 15) == 0 && HW <= 96; }
__host__ __device__ inline int fp32_q16_index(int c, int p, int HW) { return ((((c >> 4) << 2) + (c & 3)) * HW + p) * 4 + ((c >> 2) & 3); }
// true: k_net_fp32 runs this conv on v_mfma_f32_16x16x4_f32 and expects its weights in A-operand units (api.cpp:pack_conv)
__host__ __device__ inline bool fp32_conv_on_mfma(bool q16, int Cout) { return q16 && (Cout & 15) == 0; }
size_t fp32_lds_bytes(const NetDims& d);
hipError_t launch_net_fp32(hipStream_t st, bool initial, const NetDims& d, const NetProgramF32* prog_dev,
                           const float* wbuf, const NetIO& io);

}  // namespace mz
