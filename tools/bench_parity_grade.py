import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from satnerf_amd import data as O
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev="cuda:0"
rays, ts = O.synthetic_rays(1024); rays=rays.to(dev); ts=ts.to(dev); tgt=torch.rand(1024,3,device=dev)
for mode, fmt in (("bf16x3", 32), ("bf16x3", 16), ("bf16", 8)):
    args = O.default_args(mlp_mode=mode); args.bwd_fmt = fmt
    torch.manual_seed(0)
    tr = Trainer({"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30,4).to(dev)}, args, steps_per_epoch=1000)
    for _ in range(5): tr.step(rays, ts, tgt, validate=False)
    torch.cuda.synchronize(); t0=time.time()
    for _ in range(20): tr.step(rays, ts, tgt, validate=False)
    torch.cuda.synchronize(); dt=(time.time()-t0)/20
    print(f"mode {mode} bwd_fmt {fmt}: direct={tr.direct} {dt*1e3:.3f} ms per step -> {1024/dt/1e6:.3f} M rays/s")
