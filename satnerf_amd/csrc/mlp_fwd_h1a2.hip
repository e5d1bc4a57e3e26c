// the fp16-operand build of the fused forward kernel (SR_MODE_F16: single-pass fp16 MFMA, fp32 accumulate), aux size 2
#define SR_F16 1
#include "mlp_fwd2.inc"
namespace sr {
int launch_fwd_h1a2(const FwdParams& p, int save_fmt, hipStream_t st) { return launch_fwd_any<2>(p, save_fmt, st); }
}  // namespace sr
