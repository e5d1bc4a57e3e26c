"""Period of back-to-back hipGraph replays of the forward step, with and without the eager bank gather in between."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data, rendering
from satnerf_amd.models import load_model
dev = torch.device("cuda:0")
args = data.default_args(mlp_mode="bf16")
torch.manual_seed(0)
models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
rays, ts = data.synthetic_rays(1 << 18)
bank = data.RayBank(rays.to(dev), torch.rand(1 << 18, 3, device=dev), ts.to(dev), 1024, seed=1)
g = rendering.GraphedRenderer(models, args, 1024, dev, kernel_rng=True)
for _ in range(60): g.render_next(bank)
for name, fn in (("gather + replay", lambda: g.render_next(bank)), ("replay only", g.replay)):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(300): fn()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 300
    print(f"{name}: {dt*1e6:.1f} us/step -> {1024/dt/1e6:.2f} M rays/s")
