// the 512-wide build of the fused forward kernel (opt.py:50's default fc_units; inference, single-pass bf16), aux size 2
#define SR_FEAT 512
#include "mlp_fwd512g.inc"
namespace sr {
int launch_fwd512_p1a2(const FwdParams& p, int save_fmt, hipStream_t st) { return launch_fwd512_any<2>(p, save_fmt, st); }
long fwd512_stream_pieces_a2() { return FwdStream<2>::total_pieces(); }
}  // namespace sr
