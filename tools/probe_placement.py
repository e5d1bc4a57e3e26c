"""Does the placement of the training workspaces change the step time?  SATNERF_WS_PAD shifts acts / dpre / partial by 1x / 2x / 3x pad bytes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = torch.device("cuda", 0)
rays, ts = data.synthetic_rays(1 << 20, seed=20240628)
rays, ts = rays.to(dev), ts.to(dev)
rgb = torch.rand(1 << 20, 3, device=dev)
for pad in (0, 4096, 65536, 256 * 1024 + 4096, 1 << 20, (1 << 20) + 12288, 3 * (1 << 19) + 256, 0, 4096 * 33):
    os.environ["SATNERF_WS_PAD"] = str(pad)
    args = data.default_args(mlp_mode="bf16")
    torch.manual_seed(0)
    models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
    bank = data.RayBank(rays, rgb, ts, 1024, seed=11)
    tr = Trainer(models, args)
    for _ in range(80): tr.step_from_bank(bank)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): tr.step_from_bank(bank)
    torch.cuda.synchronize()
    print(f"pad {pad:>8d}: {(time.perf_counter() - t0) / 300 * 1e6:.1f} us/step", flush=True)
    del tr
