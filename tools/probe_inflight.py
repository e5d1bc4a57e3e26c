"""Training-step period of pure graph replays against the number of replays the host may keep in flight."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = torch.device("cuda:0")
args = data.default_args(mlp_mode="bf16")
torch.manual_seed(0)
models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
rays, ts = data.synthetic_rays(1 << 20)
bank = data.RayBank(rays.to(dev), torch.rand(1 << 20, 3, device=dev), ts.to(dev), 1024, seed=1)
tr = Trainer(models, args)
for _ in range(80): tr.step_from_bank(bank)
torch.cuda.synchronize()
for rep in range(2):
    for K in (0, 2, 4, 8, 32):
        evs = [torch.cuda.Event() for _ in range(max(K, 1))]
        torch.cuda.synchronize(); t0 = time.time()
        for i in range(400):
            if K:
                e = evs[i % K]
                if i >= K: e.synchronize()
            tr.step_from_bank(bank)
            if K: e.record()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 400
        print(f"in flight <= {K or 'unbounded'}: {dt*1e6:.1f} us/step", flush=True)
