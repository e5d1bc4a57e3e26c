"""The launch floor of the step's collective on ONE GPU: a 1-rank RCCL ("nccl") process group all-reduces the flat fp32 gradient
(662,537 + 120 floats = 2.65 MB), timed with HIP events, eagerly and back to back with a kernel before it (as in the step: tail ->
all-reduce -> Adam).  A 1-rank all-reduce moves no bytes between GPUs: what it costs is RCCL's launch + its kernel's fixed work, the part of
the 8-GPU collective that does not depend on the links (the wire time of 2.65 MB over 7 xGMI links is ~3 us).  Used by DESIGN section 6."""
import os, sys, json, torch
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
flat = torch.randn(662537 + 120, device="cuda")
other = torch.randn(1 << 22, device="cuda")
for _ in range(20): dist.all_reduce(flat)
torch.cuda.synchronize()
def timed(fn, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_ar = timed(lambda: dist.all_reduce(flat))
t_k = timed(lambda: other.mul_(1.0))
t_both = timed(lambda: (other.mul_(1.0), dist.all_reduce(flat)))
# captured: the collective inside a hipGraph (how SATNERF_GRAPH_ALLREDUCE=1 runs it)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): dist.all_reduce(flat)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        other.mul_(1.0); dist.all_reduce(flat)
    t_graph = timed(g.replay)
except Exception as e:  # noqa: BLE001
    t_graph = None
print(json.dumps({"allreduce_us_1rank": round(t_ar, 2), "elementwise_16MB_us": round(t_k, 2), "kernel_then_allreduce_us": round(t_both, 2),
                  "graph_kernel_then_allreduce_us": None if t_graph is None else round(t_graph, 2), "bytes": flat.numel() * 4,
                  "rccl": ".".join(map(str, torch.cuda.nccl.version()))}))
dist.destroy_process_group()
