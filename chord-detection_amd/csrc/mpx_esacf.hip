// placeholder until the ESACF kernels land
#include "mpx_internal.hpp"
namespace mpx {
int esacf_run(mpx_ctx* ctx, const float*, int64_t, const FrameDesc*, int64_t, int, const mpx_esacf_params*, int, int,
              double*, int, double*, hipStream_t) {
    return set_error(ctx, MPX_EUNSUPPORTED, "ESACF kernels not built yet");
}
}  // namespace mpx
