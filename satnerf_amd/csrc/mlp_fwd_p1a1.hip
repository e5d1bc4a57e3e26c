// one (mode, aux-size) instantiation of the fused forward kernel per translation unit (parallel builds)
#include "mlp_fwd.inc"
namespace sr {
int launch_fwd_p1a1(const FwdParams& p, bool save, hipStream_t st) { return launch_fwd<1, 1>(p, save, st); }
}  // namespace sr
