#include "net_mfma.h"
namespace mz {
struct MfmaNet { int dummy; };
MfmaNet* mfma_create(const NetDims&, int, std::string* err) { if (err) *err = "bf16 MFMA path not built yet"; return nullptr; }
void mfma_destroy(MfmaNet* n) { delete n; }
int mfma_pack_weights(MfmaNet*, const NetDims&, const float*, size_t, std::string* err) { if (err) *err = "bf16 MFMA path not built yet"; return -1; }
int mfma_forward(MfmaNet*, hipStream_t, bool, const NetDims&, const NetIO&, std::string* err) { if (err) *err = "bf16 MFMA path not built yet"; return -1; }
}
