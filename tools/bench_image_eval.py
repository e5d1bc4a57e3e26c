"""Whole-image evaluation throughput: render_image_outputs / batched_inference over H*W rays (1024 x 64 fused path)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data as O  # synthetic rays / default args (the oracle is test infrastructure)
from satnerf_amd import rendering
from satnerf_amd.models import load_model
dev = "cuda:0"
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
args = O.default_args(mlp_mode="bf16", chunk=chunk)
models = {"coarse": load_model(args).to(dev).eval(), "t": torch.nn.Embedding(30, 4).to(dev)}
rays, ts = O.synthetic_rays(side * side, seed=9)
rays, ts = rays.to(dev), ts.to(dev)
for name, fn in (("render_image_outputs", lambda: rendering.render_image_outputs(models, rays, ts, args)),
                 ("batched_inference", lambda: rendering.batched_inference(models, rays, ts, args))):
    fn(); torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.time(); out = fn(); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"{name}: {side}x{side} image, chunk {chunk}: {dt*1e3:.1f} ms -> {side*side/dt/1e6:.2f} M rays/s, peak memory {torch.cuda.max_memory_allocated()/2**20:.0f} MiB")
    del out
