// the 512-wide build of the fused backward (dX) kernel (opt.py:50's default fc_units): 4 waves x 512 VGPRs, 8-bit workspaces
#define SR_FEAT 512
#include "mlp_bwd.inc"
namespace sr {
int launch_bwd512(const BwdParams& p, int fmt, hipStream_t st) {
  if (fmt != SR_FMT8) {
    set_error("sr_satnerf_mlp_bwd: feat=512 trains on the 8-bit workspaces only");
    return 1;
  }
  return launch_bwd_fmt<SR_FMT8>(p, st);
}
long bwd512_stream_pieces() { return BwdStream::total_pieces(); }
int dpre8_units_512() { return kD8Units; }
int act8_units_512(int auxs) { return act8_units(auxs); }
}  // namespace sr
