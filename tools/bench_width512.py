"""fc_units=512 (opt.py:50; run_all.sh's sat-nerf width), 1024 rays x 64 samples: the fused forward kernel (bf16) against the
layer-by-layer path (3-pass bf16 GEMMs), eager render_rays and the hipGraph-replayed renderer; the fused-MLP kernel alone."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data, ops, rendering
from satnerf_amd.models import load_model
dev = "cuda:0"
rays, ts = data.synthetic_rays(1024); rays, ts = rays.to(dev), ts.to(dev)
for tau in (4, 16):
    for mode in ("bf16", "f16", "bf16x3"):
        args = data.default_args(fc_units=512, t_embbeding_tau=tau, mlp_mode=mode)
        torch.manual_seed(0)
        m = load_model(args).to(dev); emb = torch.nn.Embedding(30, tau).to(dev)
        models = {"coarse": m, "t": emb}
        with torch.no_grad():
            for _ in range(3): rendering.render_rays(models, args, rays, ts)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(20): rendering.render_rays(models, args, rays, ts)
            torch.cuda.synchronize(); dt = (time.time() - t0) / 20
        line = f"feat 512 tau {tau} {mode:7s} ({'fused kernel' if m.fused_forward(mode) else 'layer path'}): render_rays {dt*1e3:.3f} ms -> {1024/dt/1e6:.2f} M rays/s"
        if m.fused_forward(mode):
            g = rendering.GraphedRenderer(models, args, 1024, dev)
            for _ in range(5): g(rays, ts)
            torch.cuda.synchronize(); t0 = time.time()
            for _ in range(50): g(rays, ts)
            torch.cuda.synchronize(); dg = (time.time() - t0) / 50
            hi, lo, l0 = m.packed(mode)
            z = ops.ray_sample(rays, torch.rand(1024, 64, device=dev), 64)
            run = lambda: ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, 65536, 64, 512, tau, mode, hi, lo, l0)
            for _ in range(5): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            k = e0.elapsed_time(e1) / 30
            flop = 65536 * 5259264  # 5,259,264 FLOP per point at feat 512
            line += f" | graphed {dg*1e3:.3f} ms -> {1024/dg/1e6:.2f} M rays/s | MLP kernel {k*1e3:.1f} us = {flop/k/1e9:.0f} TFLOP/s ({flop/k/1e9/2500*100:.0f} % of bf16 peak)"
        print(line)

# ---- training at fc_units = 512: the kernel-direct graph-captured step (bf16 + 8-bit workspaces) against the autograd / layer path (bf16x3)
from satnerf_amd.train import Trainer  # noqa: E402

target = torch.rand(1024, 3, device=dev) * 0.2 + 0.4
for mode in ("bf16", "f16", "bf16x3"):
    args = data.default_args(fc_units=512, t_embbeding_tau=4, mlp_mode=mode)
    torch.manual_seed(0)
    tr = Trainer({"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}, args)
    nw, nt = (20, 100) if tr.direct else (3, 10)
    for _ in range(nw): tr.step(rays, ts, target)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(nt): tr.step(rays, ts, target)
    torch.cuda.synchronize(); dt = (time.time() - t0) / nt
    print(f"feat 512 training {mode:7s} ({'kernel-direct, graph' if tr.direct else 'autograd, layer path'}): {dt*1e3:.3f} ms/step -> {1024/dt/1e6:.3f} M rays/s")
