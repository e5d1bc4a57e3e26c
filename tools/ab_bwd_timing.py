"""Shader-cycle stamps of the dX kernel (library built with -DSR_BWD_TIMING: tools/build_variant.sh bwdtime mlp_bwd.hip -DSR_BWD_TIMING):
per wave, cycles before the trunk and in the generated trunk (896 MFMAs), with the 100-MHz real-time ticks giving the clock."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops, data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = "cuda:0"
args = data.default_args(mlp_mode="bf16")
torch.manual_seed(0)
models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
tr = Trainer(models, args, use_graph=False)
n = 1024
rays, ts = data.synthetic_rays(n); rays, ts = rays.to(dev), ts.to(dev); tgt = torch.rand(n, 3, device=dev)
seen = {}
real = ops.satnerf_mlp_bwd
def spy(*a, **k):
    out = real(*a, **k)
    seen["d_t"] = out[1]
    return out
ops.satnerf_mlp_bwd = spy
for _ in range(20): tr._forward_backward(rays, ts, tgt); tr.state.zero_grad()
torch.cuda.synchronize()
d = seen["d_t"].view(-1)[: (n * 64 // 32) * 8].view(-1, 8).double().cpu()
pre, trunk, ticks, pre_ticks = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
st = [float(d[:, k].median()) for k in range(4, 8)]
print(f"stages (cycles per wave): prologue + d_head {st[0]:.0f} | bH (12 MFMAs, 12 epilogues) {st[1]:.0f} | bS3 + bS2 (64, 8) {st[2]:.0f} | bG2 + bDT (200, 8) {st[3]:.0f} | "
      f"bG1 (136, 8) {float(pre.median()) - sum(st):.0f}")
print(f"pre-trunk {pre.median():.0f} cycles ({pre_ticks.median() / 100:.1f} us), trunk {trunk.median():.0f} cycles = {trunk.median() / 896:.1f} per MFMA per wave "
      f"({trunk.median() / 896 / 2:.1f} per MFMA slot of the SIMD), {ticks.median() / 100:.1f} us at {trunk.median() / ticks.median() * 0.1:.2f} GHz")
