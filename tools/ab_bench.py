"""Time the bf16 fused-MLP kernel for one library build (SATRENDER_LIB); prints one line.  Used for A/B of variants."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data as O  # synthetic rays / default args (the oracle is test infrastructure)
from satnerf_amd import ops
from satnerf_amd.models import load_model
dev = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
args = O.default_args()
m = load_model(args).to(dev)
emb = torch.nn.Embedding(30, 4).to(dev)
rays, ts = O.synthetic_rays(n_rays); rays = rays.to(dev); ts = ts.to(dev)
hi, lo, l0 = m.packed(mode)
z = ops.ray_sample(rays, torch.rand(n_rays, 64, device=dev), 64)
run = lambda: ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, n_rays * 64, 64, 256, 4, mode, hi, lo, l0)
for _ in range(10): run()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 40)
print(f"{os.path.basename(os.environ.get('SATRENDER_LIB', 'default')):24s} {mode:7s} rays={n_rays} {best*1e3:8.1f} us  {n_rays/best*1e3/1e6:6.2f} Mrays/s  {n_rays*84410368/best/1e9:7.1f} TFLOP/s")
