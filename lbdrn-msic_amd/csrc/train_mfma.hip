// Fused MFMA training step (placeholder until the cooperative kernel lands: reports "unsupported"
// so that lbdrn_train_epoch routes to the generic kernels).
#include "common.hpp"

namespace lbdrn {

bool mfma_train_supported(const lbdrn_geom&, const lbdrn_net&) { return false; }
size_t mfma_train_workspace(const lbdrn_geom&, const lbdrn_net&, int) { return 0; }
int mfma_train_epoch(const lbdrn_geom&, const lbdrn_net&, const uint16_t*, const uint16_t*,
                     const int64_t*, int64_t, int, float*, float*, float*, int64_t, double, float*,
                     void*, size_t, hipStream_t)
{
    set_error("fused MFMA train kernel not available");
    return LBDRN_E_UNSUPPORTED;
}

}  // namespace lbdrn
