"""Parity-mode (bf16x3) forward MLP kernel time for one library build (SATRENDER_LIB); with a -DSR_CORE_TIMING build (tools/ab_core3.sh) and SR_CORE_TIMING=1 also the in-kernel cycle / clock stamps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops, data
from satnerf_amd.models import load_model
dev = "cuda:0"
args = data.default_args(mlp_mode="bf16x3")
torch.manual_seed(0)
m = load_model(args).to(dev); emb = torch.nn.Embedding(30, 4).to(dev)
rays, ts = data.synthetic_rays(1024); rays, ts = rays.to(dev), ts.to(dev)
hi, lo, l0 = m.packed("bf16x3")
z = ops.ray_sample(rays, torch.rand(1024, 64, device=dev), 64)
def run():  # returns (albedo, sigma, sun_v, beta)
    return ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, 65536, 64, 256, 4, "bf16x3", hi, lo, l0, acts=None, fmt=8)
for _ in range(10): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(os.path.basename(os.environ.get("SATRENDER_LIB", "default")), "bf16x3 mlp us", round(e0.elapsed_time(e1) / 50 * 1e3, 1))
if os.environ.get("SR_CORE_TIMING"):  # libraries built with -DSR_CORE_TIMING return shader cycles per wave in sigma / sun_v
    a, sg, sv, b = run()
    torch.cuda.synchronize()
    n = 65536 // 32
    c, p, rt = sg[:n].cpu(), sv[:n].cpu(), b[:n].cpu()
    print(f"  core: {rt.mean() / 100:.1f} us of the 100 MHz clock -> shader clock {c.mean() / rt.mean() * 0.1:.3f} GHz")
    print(f"  core cycles per wave: mean {c.mean():.0f} min {c.min():.0f} max {c.max():.0f}  per MFMA {c.mean() / 4218:.1f} (max {c.max() / 4218:.1f}) | prologue cycles: mean {p.mean():.0f} max {p.max():.0f}")
