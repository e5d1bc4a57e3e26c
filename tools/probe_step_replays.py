"""What separates two replays of the captured training step?  rocprofv3's kernel trace shows the four kernels of a step abutting and 8.4 us
between the tail of one step and the forward of the next (profiles/r05_train_kernel_stats.csv's run).  This script times N replays of the same
captured step (a) back to back on one stream, (b) alternating between two streams chained by events (the command processor can fetch the next
graph while the previous one runs)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = "cuda:0"
args = data.default_args(mlp_mode="bf16")
torch.manual_seed(0)
models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
tr = Trainer(models, args, steps_per_epoch=1000)
n = 1024
rays, ts = data.synthetic_rays(1 << 18)
bank = data.RayBank(rays.to(dev), torch.rand(1 << 18, 3, device=dev), ts.to(dev), n, seed=3)
for _ in range(300): tr.step_from_bank(bank)
torch.cuda.synchronize()
g = tr._graph
N = 2000
def one_stream():
    for _ in range(N): g.replay()
def two_streams():
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    for s in ss: s.wait_stream(torch.cuda.current_stream())
    for i in range(N):
        s, o = ss[i & 1], ss[(i + 1) & 1]
        s.wait_stream(o)
        with torch.cuda.stream(s): g.replay()
    for s in ss: torch.cuda.current_stream().wait_stream(s)
# (c) K steps captured into ONE graph: the gap is paid once per K steps
from satnerf_amd import ops
graphs = {}
for K in (2, 4, 8):
    gk = torch.cuda.CUDAGraph()
    with ops.graph_capture(gk):
        for _ in range(K):
            tr._gather_from_banks()
            tr._forward_backward(*tr._static[:3])
    graphs[K] = gk
def multi(K):
    def fn():
        for _ in range(N // K): graphs[K].replay()
    return fn
for name, fn in (("one stream", one_stream), ("two streams", two_streams), ("2 steps per graph", multi(2)), ("4 steps per graph", multi(4)), ("8 steps per graph", multi(8)),
                 ("one stream", one_stream), ("4 steps per graph", multi(4))):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{name}: {dt / N * 1e3:.4f} ms per step")
