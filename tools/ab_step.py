"""Per-kernel HIP-event timings of one eager training step (bench.py's roofline leg) for one library build (SATRENDER_LIB)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import ops, data
from satnerf_amd.models import load_model
from satnerf_amd.train import Trainer
dev = "cuda:0"
args = data.default_args(mlp_mode=os.environ.get("AB_MODE", "bf16"))
if os.environ.get("AB_FMT"): args.bwd_fmt = int(os.environ["AB_FMT"])
if os.environ.get("AB_WIDTH"): args.fc_units = int(os.environ["AB_WIDTH"])  # 512: opt.py:50's default width
torch.manual_seed(0)
models = {"coarse": load_model(args).to(dev), "t": torch.nn.Embedding(30, 4).to(dev)}
if os.environ.get("ZERO_WEIGHTS"):  # DVFS experiment: zero operands draw far less power (lr = 0 keeps them zero)
    with torch.no_grad():
        for p_ in models["coarse"].parameters(): p_.zero_()
    models["coarse"].mark_weights_changed()
tr = Trainer(models, args, use_graph=False, lr=0.0 if os.environ.get("ZERO_WEIGHTS") else 5e-4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rays, ts = data.synthetic_rays(n); rays, ts = rays.to(dev), ts.to(dev); tgt = torch.rand(n, 3, device=dev)
dbg9 = None
if os.environ.get("AB_TIMING9"):  # wgrad9.hip built with -DSR_W9_TIMING: per workgroup (shader cycles, 100-MHz ticks, tiles)
    dbg9 = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
    os.environ["SR_W9_DBG"] = str(dbg9.data_ptr())
for _ in range(10): tr.step(rays, ts, tgt)
timer = ops.KernelTimer(); ops.kernel_timer = timer
for _ in range(30): tr.step(rays, ts, tgt)
torch.cuda.synchronize()
print(os.path.basename(os.environ.get("SATRENDER_LIB", "default")), {k: round(timer.mean_ms(k) * 1e3, 1) for k in ("mlp_fwd", "mlp_bwd", "wgrad")})

if dbg9 is not None:
    st = dbg9.cpu()[3 * 1024:7 * 1024].view(-1, 4).double()
    t1 = dbg9.cpu()[7 * 1024:].double()[:st.shape[0]]
    keep = st[:, 0] > 0
    if keep.any():
        k0_ = st[keep, 0].min()
        print(f"  tables fetched {float((t1[keep] - k0_).median()) / 100:.1f} us (max {float((t1[keep] - k0_).max()) / 100:.1f})")
    if os.environ.get("AB_PER_BLOCK"):  # slice-major numbering: workgroup i works on block i % 14 (the last 4: blocks 0..3)
        nb = 14
        idx = torch.arange(st.shape[0])
        blk = torch.where(idx < (st.shape[0] // nb) * nb, idx % nb, idx - (st.shape[0] // nb) * nb)
        for b in range(nb):
            sel = (blk == b) & keep
            if sel.any():
                print(f"    block {b:2d}: loop start {float((st[sel, 1] - k0_).mean()) / 100:5.1f}  loop end {float((st[sel, 2] - k0_).mean()) / 100:6.1f} "
                      f"(max {float((st[sel, 2] - k0_).max()) / 100:6.1f})  tiles {int(dbg9.cpu()[:3 * 1024].view(-1, 3)[sel][:, 2].float().mean())}")
    st = st[keep]
    if st.shape[0]:
        k0 = st[:, 0].min()
        f = lambda c: f"{float((st[:, c] - k0).median()) / 100:.1f} (max {float((st[:, c] - k0).max()) / 100:.1f})"
        print(f"  stamps (us after the first workgroup's entry): entry {f(0)}, loop start {f(1)}, loop end {f(2)}, epilogue done {f(3)}")
    d = dbg9.cpu()[:3 * 1024].view(-1, 3).double(); d = d[d[:, 2] > 0]
    cyc, ticks, nt = d[:, 0], d[:, 1], d[:, 2]
    print(f"  wgrad9 in situ, {d.shape[0]} workgroups: cycles/tile median {float((cyc / nt).median()):.0f}, slice loop {float(ticks.median()) / 100:.1f} us median / "
          f"{float(ticks.max()) / 100:.1f} us max, clock {float((cyc / ticks).median()) * 0.1:.2f} GHz")
