import sys, os, torch
sys.path.insert(0, os.getcwd())
from oracle import satnerf_oracle as O
from satnerf_amd import rendering
from satnerf_amd.models import load_model
DEV = "cuda:0"
def maxnorm_rel(a, b): return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
for tau, n_samples, mode, n_rays in [(16, 64, "bf16", 48), (4, 50, "bf16", 37), (16, 64, "f16", 1)]:
    args = O.default_args(t_embbeding_tau=tau, n_samples=n_samples, mlp_mode=mode)
    params = O.procedural_satnerf_params(256, tau, seed=21)
    embw = O.procedural_uniform((30, tau), 1.0, 22)
    m = load_model(args); m.load_state_dict(params)
    emb = torch.nn.Embedding(30, tau); emb.load_state_dict({"weight": embw})
    models = {"coarse": m.to(DEV), "t": emb.to(DEV)}
    rays, ts = O.synthetic_rays(n_rays, seed=23)
    g = torch.Generator().manual_seed(24)
    u, nz = torch.rand(n_rays, n_samples, generator=g), torch.randn(n_rays, n_samples, generator=g)
    target = torch.rand(n_rays, 3, generator=g)
    loss_of = lambda r, t: ((r["rgb_coarse"] - t) ** 2).sum() + r["depth_coarse"].sum() + (r["weights_coarse"].unsqueeze(-1) * r["beta_coarse"]).sum()
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    loss_of(O.render_rays({"coarse": po, "t": eo}, O.default_args(t_embbeding_tau=tau, n_samples=n_samples), rays, ts, O.ReplayRng([u, nz])), target).backward()
    with rendering.replay_rng([u.to(DEV), nz.to(DEV)]):
        res = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    loss_of(res, target.to(DEV)).backward()
    sd = dict(models["coarse"].named_parameters())
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[k].grad) for k in po}
    errs["embedding"] = maxnorm_rel(models["t"].weight.grad.cpu(), eo.grad)
    print((tau, n_samples, mode, n_rays), {k: float(f"{v:.2e}") for k, v in sorted(errs.items(), key=lambda kv: -kv[1])})
