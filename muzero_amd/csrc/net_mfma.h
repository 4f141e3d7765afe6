// bf16 MFMA implementation of the inference nets (net_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "net.h"

namespace mz {

struct MfmaNet;

MfmaNet* mfma_create(const NetDims& d, int max_batch, std::string* err);
void mfma_destroy(MfmaNet* n);
// blob: canonical fp32 weight blob (host pointer), see muzero_amd/netspec.py
int mfma_pack_weights(MfmaNet* n, const NetDims& d, const float* blob, size_t count, std::string* err);
int mfma_forward(MfmaNet* n, hipStream_t st, bool initial, const NetDims& d, const NetIO& io, std::string* err);

}  // namespace mz
