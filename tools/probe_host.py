"""Where does a training step's wall time go: host enqueue intervals vs device time (HIP events around every replay)."""
import os, sys, time, statistics
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
for _ in range(100): tr.step_from_bank(bank)
torch.cuda.synchronize()
for rep in range(3):
    N = 300
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    host = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evs[0].record()
    for i in range(N):
        a = time.perf_counter()
        tr.step_from_bank(bank)
        evs[i + 1].record()
        host.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    dev_iv = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(N)]
    q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
    print(f"wall {t_all/N*1e6:.1f} us/step | host enqueue total {t_enq/N*1e6:.1f} us/step, per-step host median {statistics.median(host)*1e6:.1f} p90 {q(host,0.9)*1e6:.1f} max {max(host)*1e6:.1f}"
          f" | device interval median {statistics.median(dev_iv):.1f} p10 {q(dev_iv,0.1):.1f} p90 {q(dev_iv,0.9):.1f} max {max(dev_iv):.1f}", flush=True)
os.system("rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\\|power\\|mclk' | head -6")
