"""What do the small launches of the captured training step cost?  Times bench.py's training leg with some of them turned into no-ops
(results are wrong: timing experiment only).  SKIP = comma list of: adam, pack, tail, loss."""
import os, sys, time, gc, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from satnerf_amd import ops
from satnerf_amd.models import SatNeRF
skip = set(filter(None, os.environ.get("SKIP", "").split(",")))
if "adam" in skip:
    ops.adam_step_graph = lambda *a, **k: None
if "pack" in skip:
    orig = SatNeRF.repack
    state = {"n": 0}
    def repack(self, *a, **k):
        state["n"] += 1
        if state["n"] <= 2:
            return orig(self, *a, **k)
    SatNeRF.repack = repack
if "tail" in skip:
    ops.grad_tail = lambda *a, **k: None
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
for rep in range(2):
    dt, _, _ = bench.measure("train", "bf16", 1024, 64, 200, 50, 1, 0, dev, want_kernels=False)
    print(f"SKIP={sorted(skip)}: {dt / 200 * 1e3:.4f} ms/step")
    bench.release_leg()
