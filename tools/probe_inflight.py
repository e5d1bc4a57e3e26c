"""Training-step period of the captured step under different host-side launch patterns (the GPU work is identical)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = torch.device("cuda:0")

def make(sampler):
    os.environ["SATNERF_GRAPH_SAMPLER"] = sampler
    args = data.default_args(mlp_mode="bf16")
    torch.manual_seed(0)
    models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
    rays, ts = data.synthetic_rays(1 << 20)
    bank = data.RayBank(rays.to(dev), torch.rand(1 << 20, 3, device=dev), ts.to(dev), 1024, seed=1)
    tr = Trainer(models, args)
    for _ in range(80): tr.step_from_bank(bank)
    torch.cuda.synchronize()
    return tr, bank

dummy = torch.zeros(64, device=dev)
def run(name, tr, bank, between=None, K=0, n=400):
    evs = [torch.cuda.Event() for _ in range(max(K, 1))]
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(n):
        if K:
            e = evs[i % K]
            if i >= K: e.synchronize()
        tr.step_from_bank(bank)
        if between: between()
        if K: e.record()
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    print(f"{name}: {dt*1e6:.1f} us/step", flush=True)

tr1, b1 = make("1")
tr0, b0 = make("0")
for rep in range(3):
    run("in-graph sampler", tr1, b1)
    run("in-graph sampler + eager dummy kernel between replays", tr1, b1, between=lambda: dummy.add_(1.0))
    run("in-graph sampler, <= 2 replays in flight", tr1, b1, K=2)
    run("eager gather + replay", tr0, b0)
