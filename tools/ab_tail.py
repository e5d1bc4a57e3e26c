"""Stand-alone timing of the step's tail launch (sr_grad_tail_adam: split-K reduction + Adam + re-pack) for one library build (SATRENDER_LIB):
the arguments of a real captured step are recorded once, then the launch is repeated back to back between HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops, data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = "cuda:0"
args = data.default_args(mlp_mode="bf16")
if os.environ.get("AB_WIDTH"): args.fc_units = int(os.environ["AB_WIDTH"])
torch.manual_seed(0)
models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
tr = Trainer(models, args, use_graph=True, steps_per_epoch=1000)
n = 1024
rays, ts = data.synthetic_rays(n); rays, ts = rays.to(dev), ts.to(dev); tgt = torch.rand(n, 3, device=dev)
seen = {}
real = ops.grad_tail_adam
def spy(*a, **k):
    seen["a"], seen["k"] = a, k
    return real(*a, **k)
ops.grad_tail_adam = spy
for _ in range(3): tr.step(rays, ts, tgt, validate=False)
torch.cuda.synchronize()
ops.grad_tail_adam = real
a, k = seen["a"], seen["k"]
for variant in ("pack", "nopack"):
    kk = dict(k)
    if variant == "nopack": kk["pack"] = None
    for _ in range(20): real(*a, **kk)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): real(*a, **kk)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100)
    print(os.path.basename(os.environ.get("SATRENDER_LIB", "default")), variant, f"{best * 1e3:.1f} us per launch")
