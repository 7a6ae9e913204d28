// fftconv_pow2.hip -- register-resident power-of-two fast path (placeholder until the
// kernels land: reports "unsupported" so every plan takes the generic path).
#include "conv_plan.hpp"
namespace pfb {
bool pow2_supported(const pfb_conv_plan*) { return false; }
int pow2_prepare(pfb_conv_plan*) { return PFB_OK; }
int pow2_apply(pfb_conv_plan*, int, int, const void*, const void*, double, double, void*,
               const void*, hipStream_t) {
    set_error("pow2_apply: fast path not built");
    return PFB_ERR_UNSUPPORTED;
}
}  // namespace pfb
