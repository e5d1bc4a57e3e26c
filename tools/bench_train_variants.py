"""Training-step time of Trainer variants (graph / eager, autograd / kernel-direct, saved-state formats) at 1024 rays x 64 samples."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from satnerf_amd import data as O  # synthetic rays / default args (the oracle is test infrastructure)
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = "cuda:0"
for kw in ({}, {"sc_lambda": 0.1}, {"n_importance": 64}):
    args = O.default_args(mlp_mode="bf16", **kw)
    models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
    if kw.get("n_importance"): models["fine"] = load_model(args).to(dev)
    tr = Trainer(models, args)
    rays, ts = O.synthetic_rays(1024); rays = rays.to(dev); ts = ts.to(dev); tgt = torch.rand(1024, 3, device=dev)
    for _ in range(5): tr.step(rays, ts, tgt)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(30): tr.step(rays, ts, tgt)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 30
    print(kw, "direct" if tr.direct else "autograd", f"{dt*1e3:.3f} ms/step -> {1024/dt/1e6:.2f} M rays/s")
