"""A few thousand kernel-direct training steps (colour + depth-supervision batches) on a learnable synthetic scene: loss curve and step time."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from satnerf_amd import data as O  # synthetic rays / default args (the oracle is test infrastructure)
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
from satnerf_amd.data import RayBank, DepthBank
dev = "cuda:0"
torch.manual_seed(0)
args = O.default_args(mlp_mode="bf16", ds_lambda=1000.0)
tr = Trainer({"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}, args)
rays, ts = O.synthetic_rays(1 << 16, seed=3)
# a learnable synthetic scene: colour depends smoothly on the ray origin / direction
rgb = (0.5 + 0.4 * torch.sin(3 * rays[:, 0:3] + rays[:, 3:6])).clamp(0, 1)
bank = RayBank(rays.to(dev), rgb.to(dev), ts.to(dev), 1024, seed=1)
d_rays, d_ts = O.synthetic_rays(1 << 14, seed=4)
depths = torch.stack([0.3 + 0.2 * torch.sin(d_rays[:, 0] * 2), torch.ones(1 << 14)], 1)
dbank = DepthBank(d_rays.to(dev), depths.to(dev), d_ts.to(dev), 1024, seed=2)
hist = []
for i in range(600):
    l = tr.step_from_bank(bank, dbank if i < 300 else None)
    if i % 50 == 0 or i == 599:
        hist.append(round(l.item(), 4))
print(hist)
assert all(h == h for h in hist) and hist[-1] < hist[6], hist
p = tr.state.params
print("params finite:", bool(torch.isfinite(p).all()), "max |p|", p.abs().max().item(), "adam step", tr.adam_state.item(), tr.n_steps)
