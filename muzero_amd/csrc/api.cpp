// C ABI of libmuzero_hip.so (include/muzero_hip.h): context, weight packing, the per-move driver that
// chains the tree kernels and the inference kernels on one HIP stream, and the inference-level calls.
#include "../../include/muzero_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is opened with dlopen on first use

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "net.h"
#include "net_mfma.h"
#include "tree.h"

using namespace mz;

// Scratch of mz_search_moves_injected_noalloc (include/muzero_hip.h): trees, per-game state and step buffers of one
// call, carved by carve_noalloc().  One copy per GPU, at its own address on each: host code never passes this name to
// a kernel or a memset - only the per-device address hipGetSymbolAddress returned (g_scratch_ptr), resolved in
// mz_create with that device current.
static __device__ __align__(256) unsigned char scratch[MZ_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local std::string g_create_error;

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};

// Device `device` must be current.  Returns hipSuccess and leaves the address in g_scratch_ptr[device].
hipError_t resolve_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[device]) return hipSuccess;
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    return hipSuccess;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            return ctx_fail(ctx, std::string(#expr) + ": " + hipGetErrorString(e_));              \
        }                                                                                          \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct EventPair {
    hipEvent_t a, b;
    int kind;
};

}  // namespace

struct mz_ctx {
    mz_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    // second stream: mz_search_moves runs the simulation steps of the two halves of the batch on two streams, so that
    // the tree kernels and the head kernel of one half run under the tower of the other (and the ragged last round of
    // workgroups of one tower under the first round of the next)
    hipStream_t stream_b = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool split_streams = true;
    std::string error;
    NetDims nd{};
    TreeParams tp{};
    TreeStore ts{};
    int N = 0;  // node slots per game
    std::vector<void*> allocs;

    // weights
    size_t n_params = 0;
    float* w_master = nullptr;      // canonical blob (device)
    float* w_f32 = nullptr;         // packed for the fp32 kernel
    size_t w_f32_count = 0;
    NetProgramF32 prog_host{};
    NetProgramF32* prog_dev = nullptr;
    MfmaAny* mfma = nullptr;        // packed bf16 / fp16 weights + scratch of the MFMA paths
    bool weights_loaded = false;
    int* action_cells = nullptr;    // [A]
    int* order_dev = nullptr;       // [A]
    int* inv_order_dev = nullptr;

    // hidden-state pool: [B*N + pool_slots][C*HW]
    float* pool = nullptr;
    long long pool_total = 0;

    // per-step buffers
    int *net_src = nullptr, *net_dst = nullptr, *net_action = nullptr, *net_active = nullptr;
    float *net_logits = nullptr, *net_policy = nullptr;
    double* net_value = nullptr;
    unsigned long long* stats = nullptr;
    int* root_slots = nullptr;      // [max_games] = g*N

    // staging (device copies of host inputs / outputs)
    std::vector<DevBuf> stage;
    // recording
    bool recording = false;
    Recording rec{};
    DevBuf rec_buf[6];
    int rec_B = 0;                  // games of the last recorded call
    int last_B = 0;                 // games of the last mz_search_moves* call (mz_get_gumbel)
    unsigned long long gumbel_seed = 0, gumbel_move = 0;   // device Gumbel stream (mz_seed_gumbel)
    void* comm = nullptr;           // ncclComm_t (mz_comm_init)
    int comm_rank = 0, comm_world = 1;
    double* comm_buf = nullptr;     // device staging of mz_gather_stats
    size_t comm_buf_count = 0;
    // device outputs
    DeviceMoveOutputs dout{};
    // timing
    bool timing = false;
    std::vector<EventPair> events;
    size_t events_used = 0;
    double t_ms[MZ_T_COUNT] = {0};
    int64_t t_launches[MZ_T_COUNT] = {0};
    int64_t last_sims = 0;
    double last_mean_path = 0;
};

namespace {

int ctx_fail(mz_ctx* ctx, const std::string& msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return -1;
}

// Device (or managed) memory that device `dev` can address as its own: -1 = any device.
bool is_device_ptr_on(const void* p, int dev) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear
        return false;
    }
    if (attr.type == hipMemoryTypeManaged) return true;
    return attr.type == hipMemoryTypeDevice && (dev < 0 || attr.device == dev);
}

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

template <typename T>
int dev_alloc(mz_ctx* ctx, T** out, size_t count) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return ctx_fail(ctx, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
    ctx->allocs.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return 0;
}

// Staging slot i: returns a device pointer holding `bytes` of `src` (host or device; nullptr stays nullptr).
int stage_in(mz_ctx* ctx, size_t slot, const void* src, size_t bytes, const void** out) {
    if (!src) {
        *out = nullptr;
        return 0;
    }
    if (is_device_ptr(src)) {
        *out = src;
        return 0;
    }
    if (ctx->stage.size() <= slot) ctx->stage.resize(slot + 1);
    DevBuf& b = ctx->stage[slot];
    if (b.bytes < bytes) {
        if (b.p) (void)hipFree(b.p);
        HIP_TRY(hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    *out = b.p;
    return 0;
}

// Output staging: returns a device buffer to write into; copy_out() brings it to the caller afterwards.
struct OutSlot {
    void* user;
    void* dev;
    size_t bytes;
    bool copy;
};
int stage_out(mz_ctx* ctx, size_t slot, void* user, size_t bytes, OutSlot* o) {
    o->user = user;
    o->bytes = bytes;
    o->copy = false;
    o->dev = nullptr;
    if (!user) return 0;
    if (is_device_ptr(user)) {
        o->dev = user;
        return 0;
    }
    if (ctx->stage.size() <= slot) ctx->stage.resize(slot + 1);
    DevBuf& b = ctx->stage[slot];
    if (b.bytes < bytes) {
        if (b.p) (void)hipFree(b.p);
        HIP_TRY(hipMalloc(&b.p, bytes));
        b.bytes = bytes;
    }
    o->dev = b.p;
    o->copy = true;
    return 0;
}
int copy_out(mz_ctx* ctx, const OutSlot& o) {
    if (o.copy) HIP_TRY(hipMemcpyAsync(o.user, o.dev, o.bytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

// java.util.HashMap iteration order of Action keys 0..A-1 inserted ascending:
// hash = 59 + index, buckets ascending, insertion order inside a bucket, T = 16*2^k with A <= 0.75*T.
// (A bin reaches 9 entries only if A > 8*T, impossible once A <= 0.75*T.. so no treeify-resize here.)
std::vector<int> hashmap_order_all(int A) {
    int T = 16;
    while (A > (T * 3) / 4) T *= 2;
    std::vector<std::pair<int, int>> k;
    for (int i = 0; i < A; i++) k.push_back({(59 + i) & (T - 1), i});
    std::stable_sort(k.begin(), k.end());
    std::vector<int> o;
    for (auto& p : k) o.push_back(p.second);
    return o;
}

struct TimerScope {
    mz_ctx* ctx;
    int kind;
    hipStream_t st;
    size_t idx = (size_t)-1;
    TimerScope(mz_ctx* c, int k, hipStream_t s = nullptr) : ctx(c), kind(k), st(s ? s : c->stream) {
        if (!ctx->timing) return;
        if (ctx->events_used == ctx->events.size()) {
            EventPair ep;
            (void)hipEventCreate(&ep.a);
            (void)hipEventCreate(&ep.b);
            ctx->events.push_back(ep);
        }
        idx = ctx->events_used++;
        ctx->events[idx].kind = kind;
        (void)hipEventRecord(ctx->events[idx].a, st);
    }
    ~TimerScope() {
        if (idx != (size_t)-1) (void)hipEventRecord(ctx->events[idx].b, st);
    }
};

void timing_begin(mz_ctx* ctx) {
    ctx->events_used = 0;
    for (int i = 0; i < MZ_T_COUNT; i++) {
        ctx->t_ms[i] = 0;
        ctx->t_launches[i] = 0;
    }
}
void timing_end(mz_ctx* ctx) {
    if (!ctx->timing) return;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream_b) (void)hipStreamSynchronize(ctx->stream_b);
    for (size_t i = 0; i < ctx->events_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->events[i].a, ctx->events[i].b) == hipSuccess) {
            ctx->t_ms[ctx->events[i].kind] += ms;
            ctx->t_launches[ctx->events[i].kind] += 1;
        }
    }
}

// ---- RCCL, bound at run time ---------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy already in the process (e.g. the one a host framework ships) is reused by its SONAME
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("dlopen(librccl.so.1) failed: ") + dlerror();
            return;
        }
#define MZ_SYM(f)                                                        \
    r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.handle, "nccl" #f));   \
    if (!r.f) r.error = "librccl lacks nccl" #f;
        MZ_SYM(GetUniqueId)
        MZ_SYM(CommInitRank)
        MZ_SYM(CommDestroy)
        MZ_SYM(Broadcast)
        MZ_SYM(AllGather)
        MZ_SYM(GetErrorString)
#undef MZ_SYM
    });
    return &r;
}
#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) return ctx_fail(ctx, std::string(#expr) + ": " + rccl()->GetErrorString(r_)); \
    } while (0)

// ---- weight packing ---------------------------------------------------------------------------
struct BlobWalker {
    const float* base;
    size_t off = 0;
    const float* take(size_t n) {
        const float* p = base + off;
        off += n;
        return p;
    }
};

size_t count_params(const NetDims& d) {
    const size_t C = d.C, Cb = d.Cb, HW = d.HW, A = d.A;
    size_t tower = 0;
    for (int i = 0; i < d.R; i++) {
        bool bc = d.bcast_every_n > 0 && (i % d.bcast_every_n == d.bcast_every_n - 1);
        if (bc) tower += 6 * C * HW + 2 * C * C + HW * HW + HW;
        else tower += 2 * C * HW + 6 * Cb * HW + 2 * C * Cb + 18 * Cb * Cb;
    }
    size_t head = 2 * C + 4 * HW + A * 2 * HW + A + C + 2 * HW + C * HW + C + C + 1;
    return C * d.O * 9 + C * (C + d.a) * 9 + 2 * tower + head;
}

inline int pad4(int x) { return (x + 3) & ~3; }

// [Cout][Cin][taps] -> [Cin][taps][CoutPad] for the vector-ALU conv of net_fp32.hip, or - where that kernel takes the conv to
// the matrix cores (fp32_conv_on_mfma: whole 16-row tiles of output channels, boards of <= 96 cells) - the A operands of
// v_mfma_f32_16x16x4_f32 in 4 KiB units (tap, row tile of 16 output channels, 64 input channels): unit[s4][lane][e] =
// w[16 mt + lane % 16][4 (16 c16 + 4 s4 + e) + lane / 16][tap], zero beyond Cin, so that a lane's operands of 16 k-steps are
// four 16-byte loads and a wave's load is 1 KiB contiguous.
void pack_conv(std::vector<float>& dst, long long& off_out, const float* w, int Cout, int Cin, int taps, bool q16 = false) {
    if (fp32_conv_on_mfma(q16, Cout)) {
        while (dst.size() % 4) dst.push_back(0.f);
        const int nmt = Cout / 16, nc16 = (Cin + 63) / 64;
        off_out = (long long)dst.size();
        dst.resize(dst.size() + (size_t)taps * nmt * nc16 * 1024, 0.f);
        float* o = dst.data() + off_out;
        for (int t = 0; t < taps; t++)
            for (int mt = 0; mt < nmt; mt++)
                for (int c16 = 0; c16 < nc16; c16++) {
                    float* u = o + ((size_t)(t * nmt + mt) * nc16 + c16) * 1024;
                    for (int s = 0; s < 16; s++)
                        for (int lane = 0; lane < 64; lane++) {
                            const int ci = 4 * (16 * c16 + s) + lane / 16, co = 16 * mt + lane % 16;
                            if (ci < Cin) u[(s / 4) * 256 + lane * 4 + (s % 4)] = w[((size_t)co * Cin + ci) * taps + t];
                        }
                }
        return;
    }
    const int cp = pad4(Cout);
    off_out = (long long)dst.size();
    dst.resize(dst.size() + (size_t)Cin * taps * cp, 0.f);
    float* o = dst.data() + off_out;
    for (int co = 0; co < Cout; co++)
        for (int ci = 0; ci < Cin; ci++)
            for (int t = 0; t < taps; t++) o[((size_t)ci * taps + t) * cp + co] = w[((size_t)co * Cin + ci) * taps + t];
}
// LayerNorm gamma / beta [Cn][HW] in the order the fp32 kernel holds the activations in LDS (Q16 with planes of HW + 1 cells,
// net.h: gamma = beta = 0 at the zero cell, which therefore stays zero) - elementwise operands
void pack_affine(std::vector<float>& dst, long long& off_out, const float* w, int Cn, int HW, bool q16) {
    while (dst.size() % 4) dst.push_back(0.f);
    off_out = (long long)dst.size();
    const int hwp = q16 ? HW + 1 : HW;
    dst.resize(dst.size() + (size_t)Cn * hwp, 0.f);
    float* o = dst.data() + off_out;
    for (int c = 0; c < Cn; c++)
        for (int p = 0; p < HW; p++) o[q16 ? fp32_q16_index(c, p, hwp) : c * HW + p] = w[(size_t)c * HW + p];
}
void pack_copy(std::vector<float>& dst, long long& off_out, const float* w, size_t n) {
    // keep every tensor 16-byte aligned for float4 loads
    while (dst.size() % 4) dst.push_back(0.f);
    off_out = (long long)dst.size();
    dst.insert(dst.end(), w, w + n);
}
// [rows][cols] -> [cols][rows]
void pack_transpose(std::vector<float>& dst, long long& off_out, const float* w, int rows, int cols) {
    while (dst.size() % 4) dst.push_back(0.f);
    off_out = (long long)dst.size();
    dst.resize(dst.size() + (size_t)rows * cols);
    float* o = dst.data() + off_out;
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) o[(size_t)c * rows + r] = w[(size_t)r * cols + c];
}

void pack_tower_f32(const NetDims& d, BlobWalker& bw, std::vector<float>& dst, TowerOff& t, int cin0) {
    const int C = d.C, Cb = d.Cb, HW = d.HW;
    const bool q16 = fp32_q16(C, Cb, HW);
    t.cin0 = cin0;
    while (dst.size() % 4) dst.push_back(0.f);
    pack_conv(dst, t.conv0, bw.take((size_t)C * cin0 * 9), C, cin0, 9, q16);
    for (int i = 0; i < d.R; i++) {
        BlockOff& b = t.blocks[i];
        std::memset(&b, 0, sizeof(b));
        bool bc = d.bcast_every_n > 0 && (i % d.bcast_every_n == d.bcast_every_n - 1);
        b.type = bc ? 1 : 0;
        if (!bc) {
            pack_affine(dst, b.ln1g, bw.take((size_t)C * HW), C, HW, q16);
            pack_affine(dst, b.ln1b, bw.take((size_t)C * HW), C, HW, q16);
            while (dst.size() % 4) dst.push_back(0.f);
            pack_conv(dst, b.c1, bw.take((size_t)Cb * C), Cb, C, 1, q16);
            pack_affine(dst, b.ln2g, bw.take((size_t)Cb * HW), Cb, HW, q16);
            pack_affine(dst, b.ln2b, bw.take((size_t)Cb * HW), Cb, HW, q16);
            while (dst.size() % 4) dst.push_back(0.f);
            pack_conv(dst, b.c2, bw.take((size_t)Cb * Cb * 9), Cb, Cb, 9, q16);
            pack_affine(dst, b.ln3g, bw.take((size_t)Cb * HW), Cb, HW, q16);
            pack_affine(dst, b.ln3b, bw.take((size_t)Cb * HW), Cb, HW, q16);
            while (dst.size() % 4) dst.push_back(0.f);
            pack_conv(dst, b.c3, bw.take((size_t)Cb * Cb * 9), Cb, Cb, 9, q16);
            pack_affine(dst, b.ln4g, bw.take((size_t)Cb * HW), Cb, HW, q16);
            pack_affine(dst, b.ln4b, bw.take((size_t)Cb * HW), Cb, HW, q16);
            while (dst.size() % 4) dst.push_back(0.f);
            pack_conv(dst, b.c4, bw.take((size_t)C * Cb), C, Cb, 1, q16);
        } else {
            pack_affine(dst, b.ln1g, bw.take((size_t)C * HW), C, HW, q16);
            pack_affine(dst, b.ln1b, bw.take((size_t)C * HW), C, HW, q16);
            while (dst.size() % 4) dst.push_back(0.f);
            pack_conv(dst, b.c1, bw.take((size_t)C * C), C, C, 1, q16);
            pack_affine(dst, b.ln2g, bw.take((size_t)C * HW), C, HW, q16);
            pack_affine(dst, b.ln2b, bw.take((size_t)C * HW), C, HW, q16);
            pack_transpose(dst, b.bw, bw.take((size_t)HW * HW), HW, HW);  // W[q][p] -> Wt[p][q]
            pack_copy(dst, b.bb, bw.take((size_t)HW), (size_t)HW);
            pack_affine(dst, b.ln3g, bw.take((size_t)C * HW), C, HW, q16);
            pack_affine(dst, b.ln3b, bw.take((size_t)C * HW), C, HW, q16);
            while (dst.size() % 4) dst.push_back(0.f);
            pack_conv(dst, b.c2, bw.take((size_t)C * C), C, C, 1, q16);
        }
    }
}

int pack_and_upload(mz_ctx* ctx, const std::vector<float>& blob) {
    const NetDims& d = ctx->nd;
    std::vector<float> dst;
    dst.reserve(blob.size() + 4096);
    BlobWalker bw{blob.data()};
    NetProgramF32& pr = ctx->prog_host;
    std::memset(&pr, 0, sizeof(pr));
    pr.c_pad = pad4(d.C);
    pr.cb_pad = pad4(d.Cb);
    pack_tower_f32(d, bw, dst, pr.rep, d.O);
    pack_tower_f32(d, bw, dst, pr.dyn, d.C + d.a);
    HeadOff& h = pr.head;
    const int C = d.C, HW = d.HW, A = d.A;
    while (dst.size() % 4) dst.push_back(0.f);
    pack_conv(dst, h.pconv, bw.take((size_t)2 * C), 2, C, 1);
    pack_copy(dst, h.plng, bw.take((size_t)2 * HW), (size_t)2 * HW);
    pack_copy(dst, h.plnb, bw.take((size_t)2 * HW), (size_t)2 * HW);
    pack_transpose(dst, h.pfcw, bw.take((size_t)A * 2 * HW), A, 2 * HW);  // [A][2HW] -> [2HW][A]
    pack_copy(dst, h.pfcb, bw.take((size_t)A), (size_t)A);
    while (dst.size() % 4) dst.push_back(0.f);
    pack_conv(dst, h.vconv, bw.take((size_t)C), 1, C, 1);
    pack_copy(dst, h.vlng, bw.take((size_t)HW), (size_t)HW);
    pack_copy(dst, h.vlnb, bw.take((size_t)HW), (size_t)HW);
    pack_transpose(dst, h.vfc1w, bw.take((size_t)C * HW), C, HW);  // [C][HW] -> [HW][C]
    pack_copy(dst, h.vfc1b, bw.take((size_t)C), (size_t)C);
    pack_copy(dst, h.vfc2w, bw.take((size_t)C), (size_t)C);
    pack_copy(dst, h.vfc2b, bw.take((size_t)1), (size_t)1);
    if (bw.off != blob.size()) return ctx_fail(ctx, "weight blob walk mismatch");

    if (ctx->w_f32_count < dst.size()) {
        if (ctx->w_f32) (void)hipFree(ctx->w_f32);
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, dst.size() * sizeof(float)));
        ctx->w_f32 = (float*)p;
        ctx->w_f32_count = dst.size();
    }
    HIP_TRY(hipMemcpy(ctx->w_f32, dst.data(), dst.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->prog_dev, &pr, sizeof(pr), hipMemcpyHostToDevice));
    if (ctx->cfg.precision != MZ_PRECISION_FP32) {
        std::string err;
        if (mfma_pack_weights(ctx->mfma, ctx->nd, blob.data(), blob.size(), &err) != 0) return ctx_fail(ctx, err);
    }
    ctx->weights_loaded = true;
    return 0;
}

int run_net(mz_ctx* ctx, bool initial, const NetIO& io, hipStream_t st = nullptr) {
    if (!st) st = ctx->stream;
    TimerScope ts(ctx, initial ? MZ_T_INITIAL : MZ_T_RECURRENT, st);
    if (ctx->cfg.precision != MZ_PRECISION_FP32) {
        std::string err;
        if (mfma_forward(ctx->mfma, st, initial, ctx->nd, io, &err) != 0) return ctx_fail(ctx, err);
    } else {
        if (hipError_t e = launch_net_fp32(st, initial, ctx->nd, ctx->prog_dev, ctx->w_f32, io))
            return ctx_fail(ctx, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    }
    return 0;
}

int check_B(mz_ctx* ctx, int B) {
    if (!ctx) return -1;
    if (B < 1 || B > ctx->cfg.max_games) return ctx_fail(ctx, "B out of range [1, max_games]");
    return 0;
}

struct StagedOutputs {
    OutSlot s[14];
};

int stage_move_outputs(mz_ctx* ctx, int B, const mz_move_outputs* out, StagedOutputs& so, DeviceMoveOutputs& d) {
    const size_t A = ctx->nd.A;
    static const mz_move_outputs none{};
    if (!out) out = &none;
    size_t base = 16;
    if (stage_out(ctx, base + 0, out->action, B * sizeof(int32_t), &so.s[0])) return -1;
    if (stage_out(ctx, base + 1, out->selected_action, B * sizeof(int32_t), &so.s[1])) return -1;
    if (stage_out(ctx, base + 2, out->visit_counts, B * A * sizeof(int32_t), &so.s[2])) return -1;
    if (stage_out(ctx, base + 3, out->policy_target, B * A * sizeof(float), &so.s[3])) return -1;
    if (stage_out(ctx, base + 4, out->root_value_target, B * sizeof(float), &so.s[4])) return -1;
    if (stage_out(ctx, base + 5, out->v_mix, B * sizeof(float), &so.s[5])) return -1;
    if (stage_out(ctx, base + 6, out->root_value_inference, B * sizeof(float), &so.s[6])) return -1;
    if (stage_out(ctx, base + 7, out->entropy, B * sizeof(float), &so.s[7])) return -1;
    if (stage_out(ctx, base + 8, out->legal_action_max_entropy, B * sizeof(float), &so.s[8])) return -1;
    if (stage_out(ctx, base + 9, out->root_entropy_value_target, B * sizeof(float), &so.s[9])) return -1;
    if (stage_out(ctx, base + 10, out->playout_policy, B * A * sizeof(float), &so.s[10])) return -1;
    if (stage_out(ctx, base + 11, out->playout_appended, B * sizeof(uint8_t), &so.s[11])) return -1;
    if (stage_out(ctx, base + 12, out->sims_run, B * sizeof(int32_t), &so.s[12])) return -1;
    if (stage_out(ctx, base + 13, out->status, B * sizeof(int32_t), &so.s[13])) return -1;
    d.action = (int*)so.s[0].dev;
    d.selected_action = (int*)so.s[1].dev;
    d.visit_counts = (int*)so.s[2].dev;
    d.policy_target = (float*)so.s[3].dev;
    d.root_value_target = (float*)so.s[4].dev;
    d.v_mix = (float*)so.s[5].dev;
    d.root_value_inference = (float*)so.s[6].dev;
    d.entropy = (float*)so.s[7].dev;
    d.legal_action_max_entropy = (float*)so.s[8].dev;
    d.root_entropy_value_target = (float*)so.s[9].dev;
    d.playout_policy = (float*)so.s[10].dev;
    d.playout_appended = (unsigned char*)so.s[11].dev;
    d.sims_run = (int*)so.s[12].dev;
    d.status = (int*)so.s[13].dev;
    return 0;
}

int ensure_recording(mz_ctx* ctx) {
    if (!ctx->recording) {
        ctx->rec = Recording{};
        return 0;
    }
    const size_t B = ctx->cfg.max_games, A = ctx->nd.A, S = ctx->cfg.max_simulations;
    const size_t need[6] = {B * A * 4, B * A * 4, B * 8, B * S * A * 4, B * S * A * 4, B * S * 8};
    for (int i = 0; i < 6; i++) {
        if (ctx->rec_buf[i].bytes < need[i]) {
            if (ctx->rec_buf[i].p) (void)hipFree(ctx->rec_buf[i].p);
            HIP_TRY(hipMalloc(&ctx->rec_buf[i].p, need[i]));
            ctx->rec_buf[i].bytes = need[i];
        }
        HIP_TRY(hipMemsetAsync(ctx->rec_buf[i].p, 0, need[i], ctx->stream));
    }
    ctx->rec.init_logits = (float*)ctx->rec_buf[0].p;
    ctx->rec.init_policy = (float*)ctx->rec_buf[1].p;
    ctx->rec.init_value = (double*)ctx->rec_buf[2].p;
    ctx->rec.sim_logits = (float*)ctx->rec_buf[3].p;
    ctx->rec.sim_policy = (float*)ctx->rec_buf[4].p;
    ctx->rec.sim_value = (double*)ctx->rec_buf[5].p;
    ctx->rec.S = (int)S;
    return 0;
}

// Shared body of mz_search_moves / mz_search_moves_injected.
int search_impl(mz_ctx* ctx, int B, const float* obs, const uint8_t* legal, const uint8_t* to_play,
                const int32_t* num_simulations, const double* gumbel, const double* dirichlet,
                const uint8_t* exploration_time, const double* u, const mz_move_outputs* out, bool injected,
                const float* init_logits, const float* init_policy, const double* init_value, const float* sim_logits,
                const float* sim_policy, const double* sim_value, int S) {
    if (check_B(ctx, B)) return -1;
    if (!legal) return ctx_fail(ctx, "legal must not be NULL");
    if (!injected && !ctx->weights_loaded) return ctx_fail(ctx, "weights not loaded");
    if (!injected && !obs) return ctx_fail(ctx, "obs must not be NULL");
    {
        // Functions.draw needs its uniform: with u == NULL the draw would silently return index 0, legal or not
        const mz_search_config& sc = ctx->cfg.search;
        const bool draws = sc.hybrid ? (!sc.gumbel_action_selection || exploration_time != nullptr)
                                     : !sc.gumbel_action_selection;
        if (draws && !u) return ctx_fail(ctx, "u must not be NULL: the configured selectAction mode draws an action");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t A = ctx->nd.A;
    const NetDims& d = ctx->nd;
    timing_begin(ctx);

    const void *d_obs, *d_legal, *d_toplay, *d_nsims, *d_gumbel, *d_dir, *d_expl, *d_u;
    if (stage_in(ctx, 0, obs, injected ? 0 : (size_t)B * d.O * d.HW * sizeof(float), &d_obs)) return -1;
    if (stage_in(ctx, 1, legal, B * A, &d_legal)) return -1;
    if (stage_in(ctx, 2, to_play, B, &d_toplay)) return -1;
    if (stage_in(ctx, 3, num_simulations, B * sizeof(int32_t), &d_nsims)) return -1;
    if (stage_in(ctx, 4, gumbel, B * A * sizeof(double), &d_gumbel)) return -1;
    if (stage_in(ctx, 5, dirichlet, B * A * sizeof(double), &d_dir)) return -1;
    if (stage_in(ctx, 6, exploration_time, B, &d_expl)) return -1;
    if (stage_in(ctx, 7, u, B * sizeof(double), &d_u)) return -1;

    int steps = ctx->cfg.search.num_simulations;
    if (num_simulations) {
        if (is_device_ptr(num_simulations)) steps = ctx->cfg.max_simulations;
        else {
            steps = 0;
            for (int g = 0; g < B; g++) {
                if (num_simulations[g] < 1 || num_simulations[g] > ctx->cfg.max_simulations)
                    return ctx_fail(ctx, "num_simulations[g] out of range [1, max_simulations]");
                steps = std::max(steps, (int)num_simulations[g]);
            }
        }
    }

    TreeParams P = ctx->tp;
    P.B = B;
    P.device_gumbel = (!gumbel && ctx->cfg.search.device_gumbel) ? 1 : 0;
    P.gumbel_seed = ctx->gumbel_seed;
    P.gumbel_move = ctx->gumbel_move;
    if (P.device_gumbel) ctx->gumbel_move++;
    ctx->last_B = B;
    if (ctx->recording) ctx->rec_B = B;
    StagedOutputs so;
    DeviceMoveOutputs dout;
    if (stage_move_outputs(ctx, B, out, so, dout)) return -1;
    if (ensure_recording(ctx)) return -1;
    HIP_TRY(hipMemsetAsync(ctx->stats, 0, 2 * sizeof(unsigned long long), ctx->stream));

    NetOut init_out{}, step_out{};
    const void *d_il = nullptr, *d_ip = nullptr, *d_iv = nullptr, *d_sl = nullptr, *d_sp = nullptr, *d_sv = nullptr;
    if (injected) {
        if (S < steps) return ctx_fail(ctx, "S (recorded simulations) smaller than the simulation budget");
        if (stage_in(ctx, 8, init_logits, B * A * sizeof(float), &d_il)) return -1;
        if (stage_in(ctx, 9, init_policy, B * A * sizeof(float), &d_ip)) return -1;
        if (stage_in(ctx, 10, init_value, B * sizeof(double), &d_iv)) return -1;
        if (stage_in(ctx, 11, sim_logits, (size_t)B * S * A * sizeof(float), &d_sl)) return -1;
        if (stage_in(ctx, 12, sim_policy, (size_t)B * S * A * sizeof(float), &d_sp)) return -1;
        if (stage_in(ctx, 13, sim_value, (size_t)B * S * sizeof(double), &d_sv)) return -1;
        init_out = NetOut{(const float*)d_il, (const float*)d_ip, (const double*)d_iv, 1, 0};
        step_out = NetOut{(const float*)d_sl, (const float*)d_sp, (const double*)d_sv, S, 1};
    } else {
        // initialInference for all B roots -> hidden slot g*N + 0
        NetIO io{};
        io.obs = (const float*)d_obs;
        io.B = B;
        io.pool = ctx->pool;
        io.logits = ctx->net_logits;
        io.policy = ctx->net_policy;
        io.value = ctx->net_value;
        io.action_cells = ctx->action_cells;
        io.dst_slot = ctx->root_slots;  // g*N: slot 0 of every game
        if (run_net(ctx, true, io)) return -1;
        init_out = NetOut{ctx->net_logits, ctx->net_policy, ctx->net_value, 1, 0};
        step_out = init_out;
    }

    {
        TimerScope ts(ctx, MZ_T_TREE_ROOT);
        launch_root(ctx->stream, P, ctx->ts, (const uint8_t*)d_legal, (const uint8_t*)d_toplay, (const int*)d_nsims,
                    ctx->cfg.search.num_simulations, init_out, (const double*)d_gumbel, (const double*)d_dir, ctx->rec);
    }
    // Two half batches on two streams (real networks only; the cut is a multiple of the boards per workgroup tile, so
    // every game keeps its place in its tile and results are those of the undivided batch bit for bit).  Games are
    // independent units: the halves share nothing but the read-only weights.
    int cut = B;
    if (!injected && ctx->split_streams && ctx->stream_b) {
        const int tile = ctx->mfma ? mfma_tile_boards(ctx->mfma) : 1;
        const int half = (B / 2) / tile * tile;
        if (half >= 64 * tile) cut = half;   // small batches: one stream (nothing to hide, two launch chains to pay)
    }
    const int nhalves = cut < B ? 2 : 1;
    if (nhalves == 2) {
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream_b, ctx->ev_fork, 0));
    }
    for (int s = 0; s < steps; s++) {
        for (int h = 0; h < nhalves; h++) {
            hipStream_t st = h == 0 ? ctx->stream : ctx->stream_b;
            const int g0 = h == 0 ? 0 : cut, cnt = nhalves == 1 ? B : (h == 0 ? cut : B - cut);
            {
                TimerScope ts(ctx, MZ_T_TREE_SELECT, st);
                launch_select(st, P, ctx->ts, ctx->net_src, ctx->net_dst, ctx->net_action, ctx->net_active, g0, cnt);
            }
            if (!injected) {
                NetIO io{};
                io.B = cnt;
                io.g_base = g0;
                io.pool = ctx->pool;
                io.src_slot = ctx->net_src + g0;
                io.dst_slot = ctx->net_dst + g0;
                io.action = ctx->net_action + g0;
                io.active = ctx->net_active + g0;
                io.action_cells = ctx->action_cells;
                io.logits = ctx->net_logits + (size_t)g0 * A;
                io.policy = ctx->net_policy + (size_t)g0 * A;
                io.value = ctx->net_value + g0;
                if (run_net(ctx, false, io, st)) {
                    // leave no work of the second stream un-joined: later calls on ctx->stream reuse the same buffers
                    if (nhalves == 2) {
                        (void)hipStreamSynchronize(ctx->stream_b);
                        (void)hipStreamSynchronize(ctx->stream);
                    }
                    return -1;
                }
            }
            {
                TimerScope ts(ctx, MZ_T_TREE_BACKUP, st);
                launch_backup(st, P, ctx->ts, step_out, ctx->rec, g0, cnt);
            }
        }
    }
    if (nhalves == 2) {
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream_b));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    {
        TimerScope ts(ctx, MZ_T_TREE_FINISH);
        launch_finish(ctx->stream, P, ctx->ts, (const uint8_t*)d_expl, (const double*)d_u, dout, ctx->stats);
    }
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < 14; i++)
        if (copy_out(ctx, so.s[i])) return -1;
    unsigned long long st[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(st, ctx->stats, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->last_sims = (int64_t)st[0];
    ctx->last_mean_path = st[0] ? (double)st[1] / (double)st[0] : 0.0;
    timing_end(ctx);
    return 0;
}

// Layout of one call's scratch inside the static array: every array 256-byte aligned.  base == nullptr: sizes only.
struct NoAllocPlan {
    TreeStore T{};
    int *net_src = nullptr, *net_dst = nullptr, *net_action = nullptr, *net_active = nullptr;
    size_t gs_off = 0, gs_bytes = 0, total = 0;
};
NoAllocPlan carve_noalloc(unsigned char* base, size_t B, size_t N, size_t A) {
    NoAllocPlan p;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* r = base ? base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return r;
    };
    const size_t BNA = B * N * A, BA = B * A, BN = B * N;
    p.T.child_prior = (double*)take(BNA * 8);
    p.T.child_qsum = (double*)take(BNA * 8);
    p.T.child_eqsum = (double*)take(BNA * 8);
    p.T.child_visit = (int*)take(BNA * 4);
    p.T.child_slot = (int*)take(BNA * 4);
    p.T.root_logit = (double*)take(BA * 8);
    p.T.root_ipv = (double*)take(BA * 8);
    p.T.root_gumbel = (double*)take(BA * 8);
    p.T.root_action = (int*)take(BA * 4);
    p.T.node_value = (double*)take(BN * 8);
    p.T.node_vmix = (double*)take(BN * 8);
    p.T.node_iv = (double*)take(BN * 8);
    p.T.node_evmix = (double*)take(BN * 8);
    p.T.node_eiv = (double*)take(BN * 8);
    p.T.node_ereward = (double*)take(BN * 8);
    p.T.node_parent = (int*)take(BN * 4);
    p.T.node_pchild = (int*)take(BN * 4);
    p.T.node_toplay = (int*)take(BN * 4);
    p.gs_off = off;
    p.gs_bytes = B * sizeof(GameState);
    p.T.gs = (GameState*)take(p.gs_bytes);
    p.net_src = (int*)take(B * 4);
    p.net_dst = (int*)take(B * 4);
    p.net_action = (int*)take(B * 4);
    p.net_active = (int*)take(B * 4);
    p.total = off;
    return p;
}

}  // namespace

extern "C" {

const char* mz_version(void) { return "muzero_hip 0.1 (gfx950)"; }

const char* mz_last_error(const mz_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int mz_create(const mz_config* cfg, int device, mz_ctx** out) {
    if (!cfg || !out) return ctx_fail(nullptr, "null argument");
    *out = nullptr;
    const mz_net_config& n = cfg->net;
    if (n.action_space < 2 || n.action_space > kMaxActions) return ctx_fail(nullptr, "action_space must be in [2, 256]");
    if (cfg->search.initial_gumbel_m < 2 || cfg->search.initial_gumbel_m > kMaxM)
        return ctx_fail(nullptr, "initial_gumbel_m must be in [2, 64]");
    if (n.residuals < 1 || n.residuals > kMaxResiduals) return ctx_fail(nullptr, "residuals must be in [1, 24]");
    if (cfg->max_games < 1) return ctx_fail(nullptr, "max_games must be >= 1");
    if (cfg->search.num_simulations < 1 || cfg->max_simulations < cfg->search.num_simulations)
        return ctx_fail(nullptr, "need 1 <= num_simulations <= max_simulations");
    if (n.height < 1 || n.width < 1 || n.height * n.width > 32767 / 9) return ctx_fail(nullptr, "bad board size");
    if (n.channels < 1 || n.bottleneck < 1 || n.obs_planes < 1 || n.action_planes < 1)
        return ctx_fail(nullptr, "bad channel counts");
    if (cfg->precision != MZ_PRECISION_FP32 && cfg->precision != MZ_PRECISION_BF16 && cfg->precision != MZ_PRECISION_F16)
        return ctx_fail(nullptr, "unknown precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return ctx_fail(nullptr, "no HIP device: libmuzero_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return ctx_fail(nullptr, "device index out of range");

    mz_ctx* ctx = new mz_ctx();
    ctx->cfg = *cfg;
    ctx->device = device;
    auto fail = [&](const std::string& m) {
        g_create_error = m.empty() ? ctx->error : m;
        mz_destroy(ctx);
        return -1;
    };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipStreamCreate(&ctx->stream) != hipSuccess) return fail("hipStreamCreate failed");
    if (hipStreamCreate(&ctx->stream_b) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess)
        return fail("hipStreamCreate / hipEventCreate failed");
    if (const char* e = std::getenv("MZ_STREAMS")) ctx->split_streams = std::atoi(e) != 1;   // MZ_STREAMS=1: A/B runs
    if (hipError_t e = resolve_scratch(device))
        return fail(std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));

    NetDims& d = ctx->nd;
    d.H = n.height;
    d.Wd = n.width;
    d.HW = n.height * n.width;
    d.O = n.obs_planes;
    d.a = n.action_planes;
    d.C = n.channels;
    d.Cb = n.bottleneck;
    d.R = n.residuals;
    d.A = n.action_space;
    d.two_players = n.two_players;
    d.bcast_every_n = n.broadcast_every_n;
    d.value_scale = n.value_scale;
    if (fp32_lds_bytes(d) > 160 * 1024 && cfg->precision == MZ_PRECISION_FP32)
        return fail("board x channels too large for the LDS-resident fp32 kernel");

    const int N = cfg->max_simulations + 1;
    ctx->N = N;
    const size_t B = cfg->max_games, A = d.A;
    TreeParams& P = ctx->tp;
    P.B = (int)B;
    P.A = (int)A;
    P.N = N;
    P.m0 = cfg->search.initial_gumbel_m;
    P.c_visit = cfg->search.c_visit;
    P.two_players = n.two_players;
    P.known_bounds = cfg->search.known_bounds;
    P.hybrid = cfg->search.hybrid;
    P.gumbel_sel = cfg->search.gumbel_action_selection;
    P.c_scale = cfg->search.c_scale;
    P.discount = cfg->search.discount;
    P.bmin = cfg->search.bound_min;
    P.bmax = cfg->search.bound_max;
    P.frac = cfg->search.root_exploration_fraction;
    P.temp_root = cfg->search.temperature_root;
    P.device_gumbel = 0;
    P.gumbel_seed = P.gumbel_move = 0;
    P.g0 = 0;
    ctx->gumbel_seed = cfg->search.gumbel_seed;

    TreeStore& T = ctx->ts;
    const size_t BNA = B * N * A, BA = B * A, BN = B * N;
    if (dev_alloc(ctx, &T.child_prior, BNA) || dev_alloc(ctx, &T.child_qsum, BNA) ||
        dev_alloc(ctx, &T.child_eqsum, BNA) || dev_alloc(ctx, &T.child_visit, BNA) ||
        dev_alloc(ctx, &T.child_slot, BNA) || dev_alloc(ctx, &T.root_logit, BA) || dev_alloc(ctx, &T.root_ipv, BA) ||
        dev_alloc(ctx, &T.root_gumbel, BA) || dev_alloc(ctx, &T.root_action, BA) || dev_alloc(ctx, &T.node_value, BN) ||
        dev_alloc(ctx, &T.node_vmix, BN) || dev_alloc(ctx, &T.node_iv, BN) || dev_alloc(ctx, &T.node_evmix, BN) ||
        dev_alloc(ctx, &T.node_eiv, BN) || dev_alloc(ctx, &T.node_ereward, BN) || dev_alloc(ctx, &T.node_parent, BN) ||
        dev_alloc(ctx, &T.node_pchild, BN) || dev_alloc(ctx, &T.node_toplay, BN) || dev_alloc(ctx, &T.gs, B))
        return fail("");
    if (hipMemset(T.gs, 0, B * sizeof(GameState)) != hipSuccess) return fail("hipMemset failed");

    std::vector<int> order = hashmap_order_all((int)A), inv(A);
    for (size_t j = 0; j < A; j++) inv[order[j]] = (int)j;
    if (dev_alloc(ctx, &ctx->order_dev, A) || dev_alloc(ctx, &ctx->inv_order_dev, A) ||
        dev_alloc(ctx, &ctx->action_cells, A))
        return fail("");
    if (hipMemcpy(ctx->order_dev, order.data(), A * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->inv_order_dev, inv.data(), A * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail("hipMemcpy failed");
    T.order = ctx->order_dev;
    T.inv_order = ctx->inv_order_dev;
    std::vector<int> cells(A);
    for (size_t i = 0; i < A; i++) cells[i] = ((int)i < d.a * d.HW) ? (int)i : -1;
    if (hipMemcpy(ctx->action_cells, cells.data(), A * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail("hipMemcpy failed");

    ctx->pool_total = (long long)BN + std::max(0, cfg->pool_slots);
    if (dev_alloc(ctx, &ctx->pool, (size_t)ctx->pool_total * d.C * d.HW)) return fail("");
    const size_t Bio = std::max<size_t>(B, (size_t)std::max(0, cfg->pool_slots));
    if (dev_alloc(ctx, &ctx->net_src, Bio) || dev_alloc(ctx, &ctx->net_dst, Bio) || dev_alloc(ctx, &ctx->net_action, Bio) ||
        dev_alloc(ctx, &ctx->net_active, Bio) ||
        dev_alloc(ctx, &ctx->net_logits, Bio * A) || dev_alloc(ctx, &ctx->net_policy, Bio * A) ||
        dev_alloc(ctx, &ctx->net_value, Bio) || dev_alloc(ctx, &ctx->stats, 2))
        return fail("");

    {
        std::vector<int> roots(B);
        for (size_t g = 0; g < B; g++) roots[g] = (int)(g * N);
        if (dev_alloc(ctx, &ctx->root_slots, B)) return fail("");
        if (hipMemcpy(ctx->root_slots, roots.data(), B * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return fail("hipMemcpy failed");
    }
    ctx->n_params = count_params(d);
    if (dev_alloc(ctx, &ctx->w_master, ctx->n_params) || dev_alloc(ctx, &ctx->prog_dev, 1)) return fail("");
    if (cfg->precision != MZ_PRECISION_FP32) {
        std::string err;
        ctx->mfma = mfma_create(cfg->precision == MZ_PRECISION_F16, d, (int)Bio, &err);
        if (!ctx->mfma) return fail(err);
    }
    *out = ctx;
    return 0;
}

void mz_destroy(mz_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream_b) (void)hipStreamSynchronize(ctx->stream_b);
    for (void* p : ctx->allocs) (void)hipFree(p);
    for (auto& b : ctx->stage)
        if (b.p) (void)hipFree(b.p);
    for (auto& b : ctx->rec_buf)
        if (b.p) (void)hipFree(b.p);
    if (ctx->w_f32) (void)hipFree(ctx->w_f32);
    if (ctx->mfma) mfma_destroy(ctx->mfma);
    if (ctx->comm) (void)rccl()->CommDestroy((ncclComm_t)ctx->comm);
    if (ctx->comm_buf) (void)hipFree(ctx->comm_buf);
    for (auto& e : ctx->events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream_b) (void)hipStreamDestroy(ctx->stream_b);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mz_set_action_cells(mz_ctx* ctx, const int32_t* cells) {
    if (!ctx || !cells) return ctx_fail(ctx, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int lim = ctx->nd.a * ctx->nd.HW;
    for (int i = 0; i < ctx->nd.A; i++)
        if (cells[i] < -1 || cells[i] >= lim) return ctx_fail(ctx, "action cell out of range");
    HIP_TRY(hipMemcpy(ctx->action_cells, cells, ctx->nd.A * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

size_t mz_num_params(const mz_ctx* ctx) { return ctx ? ctx->n_params : 0; }

int mz_load_weights(mz_ctx* ctx, const float* blob, size_t count) {
    if (!ctx || !blob) return ctx_fail(ctx, "null argument");
    if (count != ctx->n_params)
        return ctx_fail(ctx, "weight blob has " + std::to_string(count) + " floats, expected " +
                                 std::to_string(ctx->n_params));
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<float> host(count);
    if (is_device_ptr(blob)) {
        HIP_TRY(hipMemcpy(host.data(), blob, count * sizeof(float), hipMemcpyDeviceToHost));
        if (blob != ctx->w_master)
            HIP_TRY(hipMemcpy(ctx->w_master, blob, count * sizeof(float), hipMemcpyDeviceToDevice));
    } else {
        std::memcpy(host.data(), blob, count * sizeof(float));
        HIP_TRY(hipMemcpy(ctx->w_master, blob, count * sizeof(float), hipMemcpyHostToDevice));
    }
    return pack_and_upload(ctx, host);
}

int mz_weights_device_ptr(mz_ctx* ctx, void** ptr, size_t* bytes) {
    if (!ctx || !ptr || !bytes) return ctx_fail(ctx, "null argument");
    *ptr = ctx->w_master;
    *bytes = ctx->n_params * sizeof(float);
    return 0;
}

int mz_commit_weights(mz_ctx* ctx) {
    if (!ctx) return -1;
    return mz_load_weights(ctx, ctx->w_master, ctx->n_params);
}

int mz_comm_unique_id(void* id) {
    if (!id) return ctx_fail(nullptr, "null argument");
    Rccl* R = rccl();
    if (!R->error.empty()) return ctx_fail(nullptr, R->error);
    static_assert(sizeof(ncclUniqueId) == MZ_COMM_ID_BYTES, "id size");
    ncclUniqueId u;
    ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) return ctx_fail(nullptr, std::string("ncclGetUniqueId: ") + R->GetErrorString(r));
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

int mz_comm_init(mz_ctx* ctx, const void* id, int rank, int world) {
    if (!ctx || !id) return ctx_fail(ctx, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return ctx_fail(ctx, "need 0 <= rank < world");
    if (ctx->comm) return ctx_fail(ctx, "communicator already initialised");
    Rccl* R = rccl();
    if (!R->error.empty()) return ctx_fail(ctx, R->error);
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    NCCL_TRY(R->CommInitRank(&comm, world, u, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return 0;
}

int mz_comm_destroy(mz_ctx* ctx) {
    if (!ctx) return -1;
    if (!ctx->comm) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ncclResult_t r = rccl()->CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    if (ctx->comm_buf) (void)hipFree(ctx->comm_buf);
    ctx->comm_buf = nullptr;
    ctx->comm_buf_count = 0;
    return r == ncclSuccess ? 0 : ctx_fail(ctx, std::string("ncclCommDestroy: ") + rccl()->GetErrorString(r));
}

int mz_broadcast_weights(mz_ctx* ctx, int root) {
    if (!ctx) return -1;
    if (!ctx->comm) return ctx_fail(ctx, "mz_comm_init has not been called");
    if (root < 0 || root >= ctx->comm_world) return ctx_fail(ctx, "root out of range");
    if (ctx->comm_rank == root && !ctx->weights_loaded) return ctx_fail(ctx, "the root rank has no weights loaded");
    HIP_TRY(hipSetDevice(ctx->device));
    // in place on the fp32 master blob, on the context's stream (ordered behind any search still queued there)
    NCCL_TRY(rccl()->Broadcast(ctx->w_master, ctx->w_master, ctx->n_params, ncclFloat32, root, (ncclComm_t)ctx->comm,
                               ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // every rank repacks, the root too: a root that wrote the master blob through mz_weights_device_ptr without
    // mz_commit_weights would otherwise broadcast weights it does not itself run
    return mz_commit_weights(ctx);
}

int mz_gather_stats(mz_ctx* ctx, const double* in, int n, double* out) {
    if (!ctx || !in || !out) return ctx_fail(ctx, "null argument");
    if (n < 1) return ctx_fail(ctx, "n must be >= 1");
    if (!ctx->comm) return ctx_fail(ctx, "mz_comm_init has not been called");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t need = (size_t)n * (ctx->comm_world + 1);
    if (ctx->comm_buf_count < need) {
        if (ctx->comm_buf) (void)hipFree(ctx->comm_buf);
        ctx->comm_buf = nullptr;
        ctx->comm_buf_count = 0;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, need * sizeof(double)));
        ctx->comm_buf = (double*)p;
        ctx->comm_buf_count = need;
    }
    double* send = ctx->comm_buf;
    double* recv = ctx->comm_buf + n;
    HIP_TRY(hipMemcpyAsync(send, in, (size_t)n * sizeof(double), hipMemcpyDefault, ctx->stream));
    NCCL_TRY(rccl()->AllGather(send, recv, (size_t)n, ncclFloat64, (ncclComm_t)ctx->comm, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, recv, (size_t)n * ctx->comm_world * sizeof(double), hipMemcpyDefault, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int mz_comm_barrier(mz_ctx* ctx) {
    if (!ctx) return -1;
    double one = 1.0;
    std::vector<double> all((size_t)std::max(ctx->comm_world, 1));
    return mz_gather_stats(ctx, &one, 1, all.data());
}

int mz_initial_inference(mz_ctx* ctx, int B, const float* obs, const int32_t* dst_slot, float* logits, float* policy,
                         double* value) {
    if (!ctx) return -1;
    if (B < 1 || B > std::max(ctx->cfg.max_games, ctx->cfg.pool_slots)) return ctx_fail(ctx, "B out of range");
    if (!obs || !dst_slot) return ctx_fail(ctx, "null argument");
    if (!ctx->weights_loaded) return ctx_fail(ctx, "weights not loaded");
    if (is_device_ptr(dst_slot)) return ctx_fail(ctx, "slot arrays must be host pointers");
    HIP_TRY(hipSetDevice(ctx->device));
    timing_begin(ctx);
    const NetDims& d = ctx->nd;
    const size_t A = d.A;
    std::vector<int> dst(B);
    const long long base = (long long)ctx->cfg.max_games * ctx->N;
    for (int g = 0; g < B; g++) {
        if (dst_slot[g] < 0 || dst_slot[g] >= ctx->cfg.pool_slots) return ctx_fail(ctx, "dst_slot out of range");
        dst[g] = (int)(base + dst_slot[g]);
    }
    const void* d_obs;
    if (stage_in(ctx, 0, obs, (size_t)B * d.O * d.HW * sizeof(float), &d_obs)) return -1;
    HIP_TRY(hipMemcpyAsync(ctx->net_dst, dst.data(), B * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    NetIO io{};
    io.obs = (const float*)d_obs;
    io.B = B;
    io.pool = ctx->pool;
    io.dst_slot = ctx->net_dst;
    io.action_cells = ctx->action_cells;
    io.logits = ctx->net_logits;
    io.policy = ctx->net_policy;
    io.value = ctx->net_value;
    if (run_net(ctx, true, io)) return -1;
    HIP_TRY(hipGetLastError());
    if (logits) HIP_TRY(hipMemcpyAsync(logits, ctx->net_logits, B * A * sizeof(float), hipMemcpyDefault, ctx->stream));
    if (policy) HIP_TRY(hipMemcpyAsync(policy, ctx->net_policy, B * A * sizeof(float), hipMemcpyDefault, ctx->stream));
    if (value) HIP_TRY(hipMemcpyAsync(value, ctx->net_value, B * sizeof(double), hipMemcpyDefault, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    timing_end(ctx);
    return 0;
}

int mz_recurrent_inference(mz_ctx* ctx, int B, const int32_t* src_slot, const int32_t* action, const int32_t* dst_slot,
                           float* logits, float* policy, double* value) {
    if (!ctx) return -1;
    if (B < 1 || B > std::max(ctx->cfg.max_games, ctx->cfg.pool_slots)) return ctx_fail(ctx, "B out of range");
    if (!src_slot || !action || !dst_slot) return ctx_fail(ctx, "null argument");
    if (!ctx->weights_loaded) return ctx_fail(ctx, "weights not loaded");
    if (is_device_ptr(src_slot) || is_device_ptr(dst_slot) || is_device_ptr(action))
        return ctx_fail(ctx, "slot/action arrays must be host pointers");
    HIP_TRY(hipSetDevice(ctx->device));
    timing_begin(ctx);
    const size_t A = ctx->nd.A;
    const long long base = (long long)ctx->cfg.max_games * ctx->N;
    std::vector<int> src(B), dst(B), act(B);
    for (int g = 0; g < B; g++) {
        if (src_slot[g] < 0 || src_slot[g] >= ctx->cfg.pool_slots || dst_slot[g] < 0 ||
            dst_slot[g] >= ctx->cfg.pool_slots)
            return ctx_fail(ctx, "slot out of range");
        if (action[g] < 0 || action[g] >= (int)A) return ctx_fail(ctx, "action out of range");
        src[g] = (int)(base + src_slot[g]);
        dst[g] = (int)(base + dst_slot[g]);
        act[g] = action[g];
    }
    {
        // one launch reads every parent and writes every child: a destination that is another sample's source (or
        // destination) would race
        std::vector<int> writer(std::max(ctx->cfg.pool_slots, 1), -1);
        for (int g = 0; g < B; g++) {
            if (writer[dst_slot[g]] >= 0) return ctx_fail(ctx, "dst_slot values must be distinct");
            writer[dst_slot[g]] = g;
        }
        for (int g = 0; g < B; g++)
            if (writer[src_slot[g]] >= 0 && writer[src_slot[g]] != g)
                return ctx_fail(ctx, "dst_slot of one sample equals src_slot of another");
    }
    HIP_TRY(hipMemcpyAsync(ctx->net_src, src.data(), B * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->net_dst, dst.data(), B * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->net_action, act.data(), B * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    NetIO io{};
    io.B = B;
    io.pool = ctx->pool;
    io.src_slot = ctx->net_src;
    io.dst_slot = ctx->net_dst;
    io.action = ctx->net_action;
    io.action_cells = ctx->action_cells;
    io.logits = ctx->net_logits;
    io.policy = ctx->net_policy;
    io.value = ctx->net_value;
    if (run_net(ctx, false, io)) return -1;
    HIP_TRY(hipGetLastError());
    if (logits) HIP_TRY(hipMemcpyAsync(logits, ctx->net_logits, B * A * sizeof(float), hipMemcpyDefault, ctx->stream));
    if (policy) HIP_TRY(hipMemcpyAsync(policy, ctx->net_policy, B * A * sizeof(float), hipMemcpyDefault, ctx->stream));
    if (value) HIP_TRY(hipMemcpyAsync(value, ctx->net_value, B * sizeof(double), hipMemcpyDefault, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // src/dst/act vectors go out of scope
    timing_end(ctx);
    return 0;
}

int mz_read_hidden(mz_ctx* ctx, int32_t slot, float* out) {
    if (!ctx || !out) return ctx_fail(ctx, "null argument");
    if (slot < 0 || slot >= ctx->cfg.pool_slots) return ctx_fail(ctx, "slot out of range");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->nd.C * ctx->nd.HW;
    const long long base = (long long)ctx->cfg.max_games * ctx->N;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->cfg.precision != MZ_PRECISION_FP32) {
        // the MFMA paths keep a slot as [C/4][HW][4] (channel quads interleaved, net_mfma.hip x_load / x_store);
        // the caller gets the reference's [C][H][W]
        const int HW = ctx->nd.HW;
        std::vector<float> q(n), r(n);
        HIP_TRY(hipMemcpy(q.data(), ctx->pool + (size_t)(base + slot) * n, n * sizeof(float), hipMemcpyDefault));
        for (int ch = 0; ch < ctx->nd.C; ch++)
            for (int p = 0; p < HW; p++) r[(size_t)ch * HW + p] = q[((size_t)(ch / 4) * HW + p) * 4 + ch % 4];
        HIP_TRY(hipMemcpy(out, r.data(), n * sizeof(float), hipMemcpyDefault));
        return 0;
    }
    if (fp32_q16(ctx->nd.C, ctx->nd.Cb, ctx->nd.HW)) {   // the fp32 kernel's Q16 order (net.h)
        const int HW = ctx->nd.HW;
        std::vector<float> q(n), r(n);
        HIP_TRY(hipMemcpy(q.data(), ctx->pool + (size_t)(base + slot) * n, n * sizeof(float), hipMemcpyDefault));
        for (int ch = 0; ch < ctx->nd.C; ch++)
            for (int p = 0; p < HW; p++) r[(size_t)ch * HW + p] = q[fp32_q16_index(ch, p, HW)];
        HIP_TRY(hipMemcpy(out, r.data(), n * sizeof(float), hipMemcpyDefault));
        return 0;
    }
    HIP_TRY(hipMemcpy(out, ctx->pool + (size_t)(base + slot) * n, n * sizeof(float), hipMemcpyDefault));
    return 0;
}

int mz_search_moves(mz_ctx* ctx, int B, const float* obs, const uint8_t* legal, const uint8_t* to_play,
                    const int32_t* num_simulations, const double* gumbel, const double* dirichlet,
                    const uint8_t* exploration_time, const double* u, const mz_move_outputs* out) {
    return search_impl(ctx, B, obs, legal, to_play, num_simulations, gumbel, dirichlet, exploration_time, u, out, false,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
}

int mz_search_moves_injected(mz_ctx* ctx, int B, const uint8_t* legal, const uint8_t* to_play,
                             const int32_t* num_simulations, const float* init_logits, const float* init_policy,
                             const double* init_value, const float* sim_logits, const float* sim_policy,
                             const double* sim_value, int S, const double* gumbel, const double* dirichlet,
                             const uint8_t* exploration_time, const double* u, const mz_move_outputs* out) {
    if (!init_logits || !init_policy || !init_value || !sim_logits || !sim_policy || !sim_value)
        return ctx_fail(ctx, "injected network outputs must not be NULL");
    return search_impl(ctx, B, nullptr, legal, to_play, num_simulations, gumbel, dirichlet, exploration_time, u, out,
                       true, init_logits, init_policy, init_value, sim_logits, sim_policy, sim_value, S);
}

size_t mz_noalloc_scratch_bytes(const mz_ctx* ctx, int B) {
    if (!ctx || B < 1) return 0;
    return carve_noalloc(nullptr, (size_t)B, (size_t)ctx->N, (size_t)ctx->nd.A).total;
}

int mz_search_moves_injected_noalloc(mz_ctx* ctx, int B, const uint8_t* legal, const uint8_t* to_play,
                                     const int32_t* num_simulations, const float* init_logits,
                                     const float* init_policy, const double* init_value, const float* sim_logits,
                                     const float* sim_policy, const double* sim_value, int S, const double* gumbel,
                                     const double* dirichlet, const uint8_t* exploration_time, const double* u,
                                     const mz_move_outputs* out) {
    if (check_B(ctx, B)) return -1;
    if (!legal || !init_logits || !init_policy || !init_value || !sim_logits || !sim_policy || !sim_value)
        return ctx_fail(ctx, "legal and the injected network outputs must not be NULL");
    {
        const mz_search_config& sc = ctx->cfg.search;
        const bool draws = sc.hybrid ? (!sc.gumbel_action_selection || exploration_time != nullptr)
                                     : !sc.gumbel_action_selection;
        if (draws && !u) return ctx_fail(ctx, "u must not be NULL: the configured selectAction mode draws an action");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    unsigned char* base = (ctx->device >= 0 && ctx->device < kMaxDevices) ? g_scratch_ptr[ctx->device] : nullptr;
    if (!base) return ctx_fail(ctx, "scratch address of this device was not resolved");
    const size_t A = ctx->nd.A;
    NoAllocPlan pl = carve_noalloc(base, (size_t)B, (size_t)ctx->N, A);
    if (pl.total > (size_t)MZ_NOALLOC_SCRATCH_BYTES)
        return ctx_fail(ctx, "B x (max_simulations + 1) x action_space needs " + std::to_string(pl.total) +
                                 " bytes of scratch, the no-alloc entry point holds " +
                                 std::to_string((size_t)MZ_NOALLOC_SCRATCH_BYTES) + ": use mz_search_moves_injected");
    // every buffer must already be on the device: staging a host buffer would allocate
    static const mz_move_outputs none{};
    const mz_move_outputs& o = out ? *out : none;
    const void* ptrs[] = {legal, to_play, num_simulations, init_logits, init_policy, init_value, sim_logits, sim_policy,
                          sim_value, gumbel, dirichlet, exploration_time, u, o.action, o.selected_action, o.visit_counts,
                          o.policy_target, o.root_value_target, o.v_mix, o.root_value_inference, o.entropy,
                          o.legal_action_max_entropy, o.root_entropy_value_target, o.playout_policy, o.playout_appended,
                          o.sims_run, o.status};
    for (const void* p : ptrs)
        if (p && !is_device_ptr_on(p, ctx->device))
            return ctx_fail(ctx, "the no-alloc entry point takes pointers to memory of the context's own device only");
    const int steps = num_simulations ? ctx->cfg.max_simulations : ctx->cfg.search.num_simulations;
    if (S < steps) return ctx_fail(ctx, "S (recorded simulations) smaller than the simulation budget");

    TreeParams P = ctx->tp;
    P.B = B;
    P.device_gumbel = (!gumbel && ctx->cfg.search.device_gumbel) ? 1 : 0;
    P.gumbel_seed = ctx->gumbel_seed;
    P.gumbel_move = ctx->gumbel_move;
    if (P.device_gumbel) ctx->gumbel_move++;
    TreeStore T = pl.T;
    T.order = ctx->order_dev;          // read-only tables made by mz_create
    T.inv_order = ctx->inv_order_dev;
    // the array keeps the last call's contents: per-game state starts from zero, everything else is written before
    // it is read (k_root initialises a game's root, k_backup every node it expands)
    HIP_TRY(hipMemsetAsync(base + pl.gs_off, 0, pl.gs_bytes, ctx->stream));
    DeviceMoveOutputs d{};
    d.action = o.action;
    d.selected_action = o.selected_action;
    d.visit_counts = o.visit_counts;
    d.policy_target = o.policy_target;
    d.root_value_target = o.root_value_target;
    d.v_mix = o.v_mix;
    d.root_value_inference = o.root_value_inference;
    d.entropy = o.entropy;
    d.legal_action_max_entropy = o.legal_action_max_entropy;
    d.root_entropy_value_target = o.root_entropy_value_target;
    d.playout_policy = o.playout_policy;
    d.playout_appended = o.playout_appended;
    d.sims_run = o.sims_run;
    d.status = o.status;
    const NetOut init_out{init_logits, init_policy, init_value, 1, 0};
    const NetOut step_out{sim_logits, sim_policy, sim_value, S, 1};
    launch_root(ctx->stream, P, T, legal, to_play, num_simulations, ctx->cfg.search.num_simulations, init_out, gumbel,
                dirichlet, Recording{});
    for (int s = 0; s < steps; s++) {
        launch_select(ctx->stream, P, T, pl.net_src, pl.net_dst, pl.net_action, pl.net_active);
        launch_backup(ctx->stream, P, T, step_out, Recording{});
    }
    launch_finish(ctx->stream, P, T, exploration_time, u, d, nullptr);
    HIP_TRY(hipGetLastError());
    return 0;   // queued on the context's stream: mz_synchronize() before reading the outputs
}

int mz_set_recording(mz_ctx* ctx, int enable) {
    if (!ctx) return -1;
    ctx->recording = enable != 0;
    return 0;
}

int mz_get_recording(mz_ctx* ctx, float* init_logits, float* init_policy, double* init_value, float* sim_logits,
                     float* sim_policy, double* sim_value) {
    if (!ctx) return -1;
    if (!ctx->recording || !ctx->rec_buf[0].p) return ctx_fail(ctx, "nothing recorded");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    void* dst[6] = {init_logits, init_policy, init_value, sim_logits, sim_policy, sim_value};
    // game-major buffers: the first rec_B games are a prefix
    const size_t B = (size_t)ctx->rec_B, A = ctx->nd.A, S = ctx->cfg.max_simulations;
    const size_t bytes[6] = {B * A * 4, B * A * 4, B * 8, B * S * A * 4, B * S * A * 4, B * S * 8};
    for (int i = 0; i < 6; i++)
        if (dst[i] && bytes[i]) HIP_TRY(hipMemcpy(dst[i], ctx->rec_buf[i].p, bytes[i], hipMemcpyDefault));
    return 0;
}

int mz_get_recording_games(mz_ctx* ctx) { return (ctx && ctx->recording) ? ctx->rec_B : 0; }

int mz_get_gumbel(mz_ctx* ctx, double* gumbel) {
    if (!ctx || !gumbel) return ctx_fail(ctx, "null argument");
    const size_t B = (size_t)ctx->last_B, A = ctx->nd.A;
    if (!B) return ctx_fail(ctx, "no search has run");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<double> rg(B * A), out(B * A, 0.0);
    std::vector<int> ra(B * A);
    std::vector<GameState> gs(B);
    HIP_TRY(hipMemcpy(rg.data(), ctx->ts.root_gumbel, B * A * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ra.data(), ctx->ts.root_action, B * A * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(gs.data(), ctx->ts.gs, B * sizeof(GameState), hipMemcpyDeviceToHost));
    for (size_t g = 0; g < B; g++) {
        if (gs[g].L < 2 || gs[g].L > (int)A) continue;   // shortcut / failed games drew nothing
        for (int j = 0; j < gs[g].L; j++) out[g * A + ra[g * A + j]] = rg[g * A + j];
    }
    HIP_TRY(hipMemcpy(gumbel, out.data(), B * A * 8, hipMemcpyDefault));
    return 0;
}

int mz_seed_gumbel(mz_ctx* ctx, uint64_t seed, uint64_t move_counter) {
    if (!ctx) return -1;
    ctx->gumbel_seed = seed;
    ctx->gumbel_move = move_counter;
    return 0;
}

int mz_child_order(const uint8_t* legal, int A, int32_t* order) {
    if (!legal || !order || A < 1) return -1;
    // java.util.HashMap<Action, Node> filled in ascending action order (Node.java:295-304): bucket
    // (59 + index) & (T - 1), T doubling from 16 when size > 0.75 T or when a bin of a table < 64 reaches 9 entries
    // (treeifyBin resizes instead); iteration = buckets ascending, insertion order inside a bucket.
    int T = 16, thr = 12, size = 0;
    std::vector<int> cnt(64, 0);
    for (int a = 0; a < A && T < 64; a++) {
        if (!legal[a]) continue;
        const int c = ++cnt[(59 + a) & (T - 1)];
        size++;
        for (int pass = 0; pass < 2; pass++) {
            const bool grow = pass == 0 ? (c >= 9) : (size > thr);
            if (!grow) continue;
            T *= 2;
            thr *= 2;
            if (T < 64) {
                std::fill(cnt.begin(), cnt.end(), 0);
                for (int a2 = 0; a2 <= a; a2++)
                    if (legal[a2]) cnt[(59 + a2) & (T - 1)]++;
            }
        }
    }
    int L = 0;
    for (int a = 0; a < A; a++) L += legal[a] ? 1 : 0;
    while (L > thr) {
        T *= 2;
        thr *= 2;
    }
    std::vector<std::pair<int, int>> k;
    for (int a = 0; a < A; a++)
        if (legal[a]) k.push_back({(59 + a) & (T - 1), a});
    std::stable_sort(k.begin(), k.end());
    for (int j = 0; j < L; j++) order[j] = k[j].second;
    return L;
}

int mz_tree_num_nodes(mz_ctx* ctx, int game) {
    if (!ctx || game < 0 || game >= ctx->cfg.max_games) return -1;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    GameState gs;
    if (hipMemcpy(&gs, ctx->ts.gs + game, sizeof(gs), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return gs.L >= 2 ? gs.num_nodes : 0;
}

int mz_tree_node(mz_ctx* ctx, int game, int slot, int32_t* parent_slot, int32_t* action, int32_t* to_play,
                 int32_t* visit_count, double* scalars, int32_t* child_action, double* child_prior, double* child_logit,
                 int32_t* child_visit, double* child_qsum, int32_t* child_slot) {
    if (!ctx || game < 0 || game >= ctx->cfg.max_games) return -1;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    GameState gs;
    HIP_TRY(hipMemcpy(&gs, ctx->ts.gs + game, sizeof(gs), hipMemcpyDeviceToHost));
    if (gs.L < 2 || slot < 0 || slot >= gs.num_nodes) return ctx_fail(ctx, "no such node");
    const TreeStore& T = ctx->ts;
    const size_t A = ctx->nd.A;
    const size_t gn = (size_t)game * ctx->N + slot;
    const int nch = slot == 0 ? gs.L : (int)A;
    auto rd = [&](void* dst, const void* src, size_t bytes) {
        return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    };
    int par = -1, pch = 0, tp = 0;
    double sc[6];
    bool ok = rd(&par, T.node_parent + gn, 4) && rd(&pch, T.node_pchild + gn, 4) && rd(&tp, T.node_toplay + gn, 4) &&
              rd(&sc[0], T.node_value + gn, 8) && rd(&sc[1], T.node_vmix + gn, 8) && rd(&sc[2], T.node_iv + gn, 8) &&
              rd(&sc[3], T.node_evmix + gn, 8) && rd(&sc[4], T.node_eiv + gn, 8) && rd(&sc[5], T.node_ereward + gn, 8);
    if (!ok) return ctx_fail(ctx, "hipMemcpy failed");
    std::vector<int> order(A);
    if (slot == 0) ok = rd(order.data(), T.root_action + (size_t)game * A, A * 4);
    else ok = rd(order.data(), ctx->order_dev, A * 4);
    int vis = 0;
    double qsum = 0, eqsum = 0;
    int act = -1;
    if (slot == 0) {
        vis = gs.root_visit;
        qsum = gs.root_qsum;
        eqsum = gs.root_eqsum;
    } else {
        const size_t ci = ((size_t)game * ctx->N + par) * A + pch;
        ok = ok && rd(&vis, T.child_visit + ci, 4) && rd(&qsum, T.child_qsum + ci, 8) && rd(&eqsum, T.child_eqsum + ci, 8);
        if (par == 0) ok = ok && rd(&act, T.root_action + (size_t)game * A + pch, 4);
        else ok = ok && rd(&act, ctx->order_dev + pch, 4);
    }
    if (!ok) return ctx_fail(ctx, "hipMemcpy failed");
    if (parent_slot) *parent_slot = par;
    if (action) *action = act;
    if (to_play) *to_play = tp;
    if (visit_count) *visit_count = vis;
    if (scalars) {
        scalars[0] = sc[0];
        scalars[1] = sc[1];
        scalars[2] = sc[2];
        scalars[3] = qsum;
        scalars[4] = sc[3];
        scalars[5] = sc[4];
        scalars[6] = eqsum;
        scalars[7] = sc[5];
    }
    const size_t cb = gn * A;
    if (child_action) std::memcpy(child_action, order.data(), nch * 4);
    if (child_prior && !rd(child_prior, T.child_prior + cb, nch * 8)) return ctx_fail(ctx, "hipMemcpy failed");
    if (child_visit && !rd(child_visit, T.child_visit + cb, nch * 4)) return ctx_fail(ctx, "hipMemcpy failed");
    if (child_qsum && !rd(child_qsum, T.child_qsum + cb, nch * 8)) return ctx_fail(ctx, "hipMemcpy failed");
    if (child_slot && !rd(child_slot, T.child_slot + cb, nch * 4)) return ctx_fail(ctx, "hipMemcpy failed");
    if (child_logit) {
        if (slot == 0) {
            if (!rd(child_logit, T.root_logit + (size_t)game * A, nch * 8)) return ctx_fail(ctx, "hipMemcpy failed");
        } else {
            for (int i = 0; i < nch; i++) child_logit[i] = NAN;  // not kept below the root (never read by the search)
        }
    }
    return nch;
}

int mz_tree_minmax(mz_ctx* ctx, int game, double* minimum, double* maximum) {
    if (!ctx || game < 0 || game >= ctx->cfg.max_games) return -1;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    GameState gs;
    HIP_TRY(hipMemcpy(&gs, ctx->ts.gs + game, sizeof(gs), hipMemcpyDeviceToHost));
    if (minimum) *minimum = gs.mm_min;
    if (maximum) *maximum = gs.mm_max;
    return 0;
}

int mz_set_timing(mz_ctx* ctx, int enable) {
    if (!ctx) return -1;
    ctx->timing = enable != 0;
    return 0;
}

int mz_set_stream_split(mz_ctx* ctx, int enable) {
    if (!ctx) return -1;
    ctx->split_streams = enable != 0;
    return 0;
}

int mz_get_timing(mz_ctx* ctx, double* ms, int64_t* launches) {
    if (!ctx) return -1;
    for (int i = 0; i < MZ_T_COUNT; i++) {
        if (ms) ms[i] = ctx->t_ms[i];
        if (launches) launches[i] = ctx->t_launches[i];
    }
    return 0;
}

int mz_get_last_stats(mz_ctx* ctx, int64_t* simulations, double* mean_path_len) {
    if (!ctx) return -1;
    if (simulations) *simulations = ctx->last_sims;
    if (mean_path_len) *mean_path_len = ctx->last_mean_path;
    return 0;
}

int mz_synchronize(mz_ctx* ctx) {
    if (!ctx) return -1;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
