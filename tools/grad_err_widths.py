"""Per-parameter gradient error of the kernel-direct throughput-mode step against fp32 autograd through the oracle, widths 256 and 512."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import satnerf_oracle as O  # noqa: E402  (diagnostic tool: oracle as the checker)
from satnerf_amd.models import load_model  # noqa: E402
from satnerf_amd.train import Trainer  # noqa: E402

DEV = torch.device("cuda:0")


def maxnorm_rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


for feat, n in ((256, 96), (512, 96), (256, 1024), (512, 1024)):
    tau = 4
    args = O.default_args(fc_units=feat, t_embbeding_tau=tau, mlp_mode="bf16")
    params = O.procedural_satnerf_params(feat, tau, seed=61)
    embw = O.procedural_uniform((30, tau), 1.0, 62)
    m = load_model(args)
    m.load_state_dict(params)
    emb = torch.nn.Embedding(30, tau)
    emb.load_state_dict({"weight": embw})
    models = {"coarse": m.to(DEV), "t": emb.to(DEV)}
    rays, ts = O.synthetic_rays(n, seed=63)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(65))
    torch.manual_seed(66)
    u = torch.rand(n, 64, device=DEV).cpu()
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    lo = O.satnerf_loss(O.render_rays({"coarse": po, "t": eo}, args, rays, ts, O.ReplayRng([u, torch.zeros(n, 64)])), target)
    lo.backward()
    tr = Trainer(models, args, use_graph=False)
    torch.manual_seed(66)
    parts = tr._forward_backward(rays.to(DEV), ts.to(DEV), target.to(DEV))
    sd = dict(models["coarse"].named_parameters())
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[k].grad) for k in po if po[k].grad is not None}
    errs["embedding"] = maxnorm_rel(models["t"].weight.grad.cpu(), eo.grad)
    print(f"feat {feat} n {n} loss {parts.sum().item():.6f} vs {lo.item():.6f}")
    for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:8]:
        print(f"   {k:28s} {v:.2e}")
