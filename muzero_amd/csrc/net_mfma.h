// MFMA implementation of the inference nets (net_mfma.hip), built twice: bf16 operands and fp16 operands.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "net.h"

namespace mz {

// blob: canonical fp32 weight blob (host pointer), see muzero_amd/netspec.py
#define MZ_DECLARE_MFMA(SUF, NET)                                                                                    \
    struct NET;                                                                                                      \
    NET* mfma_create_##SUF(const NetDims& d, int max_batch, std::string* err);                                       \
    void mfma_destroy_##SUF(NET* n);                                                                                 \
    int mfma_pack_weights_##SUF(NET* n, const NetDims& d, const float* blob, size_t count, std::string* err);        \
    int mfma_forward_##SUF(NET* n, hipStream_t st, bool initial, const NetDims& d, const NetIO& io, std::string* err);  \
    int mfma_tile_boards_##SUF(const NET* n);
MZ_DECLARE_MFMA(bf16, MfmaNetBf16)
MZ_DECLARE_MFMA(f16, MfmaNetF16)
#undef MZ_DECLARE_MFMA

// The variant a context runs, behind one handle.
struct MfmaAny {
    bool f16 = false;
    MfmaNetBf16* b = nullptr;
    MfmaNetF16* h = nullptr;
};
inline MfmaAny* mfma_create(bool f16, const NetDims& d, int max_batch, std::string* err) {
    MfmaAny* a = new MfmaAny();
    a->f16 = f16;
    if (f16) a->h = mfma_create_f16(d, max_batch, err);
    else a->b = mfma_create_bf16(d, max_batch, err);
    if (!a->h && !a->b) {
        delete a;
        return nullptr;
    }
    return a;
}
inline void mfma_destroy(MfmaAny* a) {
    if (!a) return;
    if (a->h) mfma_destroy_f16(a->h);
    if (a->b) mfma_destroy_bf16(a->b);
    delete a;
}
inline int mfma_pack_weights(MfmaAny* a, const NetDims& d, const float* blob, size_t count, std::string* err) {
    return a->f16 ? mfma_pack_weights_f16(a->h, d, blob, count, err) : mfma_pack_weights_bf16(a->b, d, blob, count, err);
}
// boards one workgroup tile carries: a batch cut at a multiple of it keeps every game's place in its tile
inline int mfma_tile_boards(const MfmaAny* a) { return a->f16 ? mfma_tile_boards_f16(a->h) : mfma_tile_boards_bf16(a->b); }
inline int mfma_forward(MfmaAny* a, hipStream_t st, bool initial, const NetDims& d, const NetIO& io, std::string* err) {
    return a->f16 ? mfma_forward_f16(a->h, st, initial, d, io, err) : mfma_forward_bf16(a->b, st, initial, d, io, err);
}

}  // namespace mz
