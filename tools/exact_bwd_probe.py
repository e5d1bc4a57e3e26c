import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import satnerf_oracle as O
from helpers import load_golden, golden_draws, maxnorm_rel
from test_hip_parity import build_models
from satnerf_amd import rendering
DEV = torch.device("cuda:0")
g = load_golden("backward")
args = O.default_args(mlp_mode="bf16x3")
models = build_models(args)
models["coarse"].train()
models["coarse"].fused = False
with rendering.replay_rng([x.to(DEV) for x in golden_draws(g)]):
    res = rendering.render_rays(models, args, g["rays"].to(DEV), g["ts"].to(DEV))
loss = res["rgb_coarse"].sum() + res["depth_coarse"].sum() + (res["weights_coarse"].unsqueeze(-1) * res["beta_coarse"]).sum()
loss.backward()
print("loss", loss.item(), float(g["loss"]))
sd = dict(models["coarse"].named_parameters())
for k, v in g.items():
    if k.startswith("grad_") and k != "grad_embedding":
        print(k, f"{maxnorm_rel(sd[k[5:]].grad.cpu(), v):.2e}")
print("emb", f"{maxnorm_rel(models['t'].weight.grad.cpu(), g['grad_embedding']):.2e}")
