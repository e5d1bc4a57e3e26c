// the 512-wide, fp16-operand build of the fused forward kernel (opt.py:50's default fc_units, SR_MODE_F16), aux size 2
#define SR_FEAT 512
#define SR_F16 1
#include "mlp_fwd512g.inc"
namespace sr {
int launch_fwd512_h1a2(const FwdParams& p, int save_fmt, hipStream_t st) { return launch_fwd512_any<2>(p, save_fmt, st); }
}  // namespace sr
