import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wk = len(sys.argv) > 1 and sys.argv[1] == "1"
for rep in range(2):
    dt, _, _ = bench.measure("train", "bf16", 1024, 64, 300, 50, 1, 0, dev, want_kernels=wk)
    print(f"bench.measure(want_kernels={wk}): {dt/300*1e6:.1f} us/step, host enqueue {bench.measure.host_enqueue_s/300*1e6:.1f} us/step", flush=True)
