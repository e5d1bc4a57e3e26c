"""Throughput of the layer-by-layer path (fc_units != 256): forward render_rays and one training step, 1024 rays x 64 samples."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data as O  # synthetic rays / default args (the oracle is test infrastructure)
from satnerf_amd import rendering
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = "cuda:0"
feat = int(sys.argv[1]) if len(sys.argv) > 1 else 512
what = sys.argv[2] if len(sys.argv) > 2 else "both"
args = O.default_args(fc_units=feat)
m = load_model(args).to(dev); emb = torch.nn.Embedding(30, 4).to(dev)
m.fused = False
rays, ts = O.synthetic_rays(1024); rays = rays.to(dev); ts = ts.to(dev)
if what in ("both", "fwd"):
    with torch.no_grad():
        for _ in range(3): rendering.render_rays({"coarse": m, "t": emb}, args, rays, ts)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): rendering.render_rays({"coarse": m, "t": emb}, args, rays, ts)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"layer path feat={feat}: forward {dt*1e3:.2f} ms -> {1024/dt/1e6:.3f} M rays/s")
if what in ("both", "train"):
    tr = Trainer({"coarse": m, "t": emb}, args)
    tgt = torch.rand(1024, 3, device=dev)
    for _ in range(3): tr.step(rays, ts, tgt)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): tr.step(rays, ts, tgt)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"layer path feat={feat}: train step {dt*1e3:.2f} ms -> {1024/dt/1e6:.3f} M rays/s")
