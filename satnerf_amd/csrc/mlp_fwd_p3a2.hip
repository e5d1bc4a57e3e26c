// one (mode, aux-size) instantiation of the fused forward kernel per translation unit (parallel builds)
#include "mlp_fwd3.inc"
namespace sr {
int launch_fwd_p3a2(const FwdParams& p, int save_fmt, hipStream_t st) { return launch_fwd3_any<2>(p, save_fmt, st); }
}  // namespace sr
