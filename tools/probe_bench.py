"""Bisect bench.py's training leg against the plain Trainer loop (same box, same process)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from satnerf_amd import data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)

def plain(seed_after=None, cpu_rgb=False, n=300):
    args = data.default_args(mlp_mode="bf16")
    torch.manual_seed(0)
    models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
    rays, ts = data.synthetic_rays(1 << 20, seed=20240628)
    rgb = torch.rand(1 << 20, 3, generator=torch.Generator().manual_seed(7)).to(dev) if cpu_rgb else torch.rand(1 << 20, 3, device=dev)
    bank = data.RayBank(rays.to(dev), rgb, ts.to(dev), 1024, seed=11)
    if seed_after is not None: torch.manual_seed(seed_after)
    tr = Trainer(models, args, world_size=1)
    for _ in range(100): tr.step_from_bank(bank)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.step_from_bank(bank)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for rep in range(2):
    dt, _, _ = bench.measure("train", "bf16", 1024, 64, 300, 50, 1, 0, dev, want_kernels=False)
    print(f"bench.measure: {dt/300*1e6:.1f} us/step", flush=True)
    print(f"plain loop: {plain():.1f} us/step", flush=True)
    print(f"plain loop, manual_seed(1234) before the Trainer: {plain(seed_after=1234):.1f} us/step", flush=True)
    print(f"plain loop, CPU-generated targets: {plain(cpu_rgb=True):.1f} us/step", flush=True)
