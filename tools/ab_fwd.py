"""Forward MLP kernel time (inference and saving activations) for one library build (SATRENDER_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops, data
from satnerf_amd.models import load_model
dev = "cuda:0"
args = data.default_args(mlp_mode="bf16")
torch.manual_seed(0)
m = load_model(args).to(dev); emb = torch.nn.Embedding(30, 4).to(dev)
rays, ts = data.synthetic_rays(1024); rays, ts = rays.to(dev), ts.to(dev)
if os.environ.get("ZERO_WEIGHTS"):  # DVFS experiment: all-zero operands draw less power
    with torch.no_grad():
        for p_ in m.parameters(): p_.zero_()
hi, lo, l0 = m.packed("bf16")
z = ops.ray_sample(rays, torch.rand(1024, 64, device=dev), 64)
def run(acts=None):
    return ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, 65536, 64, 256, 4, "bf16", hi, lo, l0, acts=acts, fmt=8)
out = {}
for name, acts in (("inference", None), ("save8", ops.acts_workspace(65536, 256, dev, 8))):
    for _ in range(10): run(acts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run(acts)
    e1.record(); torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 50 * 1e3, 1)
print(os.path.basename(os.environ.get("SATRENDER_LIB", "default")), out)
if os.environ.get("SR_CORE_TIMING"):  # libraries built with -DSR_CORE_TIMING return shader cycles per wave in sigma / sun_v
    a, sg, sv, b = run(None)
    torch.cuda.synchronize()
    c, p = sg[:2048].cpu(), sv[:2048].cpu()
    rt = b[:2048].cpu()
    print(f"  core: {rt.mean() / 100:.1f} us of the 100 MHz clock -> shader clock {c.mean() / rt.mean() * 0.1:.3f} GHz")
    print(f"  core cycles per wave: mean {c.mean():.0f} min {c.min():.0f} max {c.max():.0f}  per MFMA {c.mean() / 1406:.1f} (max {c.max() / 1406:.1f}) | prologue cycles: mean {p.mean():.0f} max {p.max():.0f}")
