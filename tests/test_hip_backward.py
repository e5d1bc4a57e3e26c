"""Backward parity of the HIP path: compositing backward (fp32, tight) and the fused-MLP backward (bf16 MFMA with fp32
accumulation -> mixed-precision tolerances) against autograd through the CPU oracle / the reference's golden gradients."""
import pytest
import torch

from oracle import satnerf_oracle as O
from tests.helpers import golden_draws, load_golden, maxnorm_rel
from tests.test_hip_parity import DEV, build_models

pytestmark = pytest.mark.gpu

GRAD_TOL = 3e-2  # bf16 operands in dX / dW GEMMs and bf16-rounded saved activations; fp32 accumulation


def test_composite_backward_matches_autograd():
    from satnerf_amd import ops

    g = torch.Generator().manual_seed(7)
    n, s = 37, 64
    z = torch.sort(torch.rand(n, s, generator=g), -1)[0]
    sigma = (torch.randn(n, s, generator=g) * 3).requires_grad_(True)
    noise = torch.randn(n, s, generator=g)
    albedo = torch.rand(n, s, 3, generator=g).requires_grad_(True)
    sun = torch.rand(n, s, generator=g).requires_grad_(True)
    sky = torch.rand(n, 3, generator=g).requires_grad_(True)
    gr, gd, gw, gt = torch.randn(n, 3, generator=g), torch.randn(n, generator=g), torch.randn(n, s, generator=g), torch.randn(n, s, generator=g)
    w, t = O.alpha_composite(z, sigma, noise * 0.3)
    depth = torch.sum(w * z, -1)
    irr = sun.unsqueeze(-1) + (1 - sun.unsqueeze(-1)) * sky.unsqueeze(1)
    rgb = torch.clamp(torch.sum(w.unsqueeze(-1) * albedo * irr, -2) * 1.7 - 0.1, 0, 1)  # scaled so the clamp bites on some rays
    # the kernel clamps the plain sum; emulate the same by differentiating the un-scaled clamp separately
    rgb = torch.clamp(torch.sum(w.unsqueeze(-1) * albedo * irr, -2), 0, 1)
    ((rgb * gr).sum() + (depth * gd).sum() + (w * gw).sum() + (t * gt).sum()).backward()
    d = lambda x: x.detach().to(DEV).contiguous()  # noqa: E731
    wg, tg, _, _ = ops.composite(d(z), d(sigma), d(noise), 0.3, d(albedo), d(sun), d(sky))
    d_sigma, d_albedo, d_sun, d_sky = ops.composite_bwd(d(z), d(sigma), d(noise), 0.3, d(albedo), d(sun), d(sky), wg, tg, d(gr), d(gd), d(gw), d(gt))
    assert maxnorm_rel(d_sigma.cpu(), sigma.grad) < 1e-4
    assert maxnorm_rel(d_albedo.cpu(), albedo.grad) < 1e-5
    assert maxnorm_rel(d_sun.cpu(), sun.grad) < 1e-5
    assert maxnorm_rel(d_sky.cpu(), sky.grad) < 1e-5


def _run_backward(models, args, rays, ts, draws, loss_of):
    from satnerf_amd import rendering

    for p in models["coarse"].parameters():
        p.grad = None
    models["t"].weight.grad = None
    with rendering.replay_rng([x.to(DEV) for x in draws]):
        res = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    loss = loss_of(res)
    loss.backward()
    torch.cuda.synchronize()
    return loss, res


@pytest.mark.parametrize("mode,fmt", [("bf16x3", 32), ("bf16x3", 16), ("bf16", 16), ("bf16", 8), ("f16", 8), ("f16", 16)])
def test_gradients_match_reference_golden(mode, fmt):
    """fmt = format of the saved training state: 16-bit (parity mode's default), the throughput mode's 8-bit workspaces, or 32 =
    the parity-grade backward (fp32 state + 3-pass GEMMs, layer by layer): the reference's own gradients to 2e-4."""
    g = load_golden("backward")
    args = O.default_args(mlp_mode=mode, bwd_fmt=fmt)
    models = build_models(args)
    models["coarse"].train()
    loss, _ = _run_backward(models, args, g["rays"], g["ts"], golden_draws(g),
                            lambda r: r["rgb_coarse"].sum() + r["depth_coarse"].sum() + (r["weights_coarse"].unsqueeze(-1) * r["beta_coarse"]).sum())
    assert abs(loss.item() - float(g["loss"])) < {"bf16x3": 1e-4, "f16": 2e-3, "bf16": 2e-2}[mode] * abs(float(g["loss"]))
    sd = dict(models["coarse"].named_parameters())
    errs = {}
    for k, v in g.items():
        if k.startswith("grad_") and k != "grad_embedding":
            errs[k[5:]] = maxnorm_rel(sd[k[5:]].grad.cpu(), v)
    errs["embedding"] = maxnorm_rel(models["t"].weight.grad.cpu(), g["grad_embedding"])
    print(mode, fmt, {k: f"{e:.1e}" for k, e in errs.items()})
    assert len(errs) == 14
    # per arithmetic (measured max over the 14 tensors: 7.0e-3 | 1.4e-2 | 2.1e-2): the backward GEMMs are single-pass bf16 in every
    # mode; the throughput mode adds bf16 forward activations, its 8-bit workspaces add the PHASE8 / MX8 rounding of the saved state
    tol = {("bf16x3", 32): 2e-4, ("bf16x3", 16): 1.2e-2, ("bf16", 16): 2.2e-2, ("bf16", 8): GRAD_TOL, ("f16", 8): GRAD_TOL, ("f16", 16): 1.2e-2}[(mode, fmt)]
    assert max(errs.values()) < tol, errs
    # every p.grad is a view of ONE flat buffer
    flat = models["coarse"].flat_grads()
    assert next(models["coarse"].parameters()).grad.data_ptr() == flat.data_ptr()


def test_loss_gradients_match_reference_golden():
    """SatNerfLoss (+ solar correction) + DepthLoss over three ragged ray chunks: the train-time twin of batched_inference."""
    from satnerf_amd import rendering
    from satnerf_amd.train import satnerf_loss

    g = load_golden("batched_losses")
    args = O.default_args(chunk=100, sc_lambda=0.05, mlp_mode="bf16x3")
    models = build_models(args)
    draws = [x.to(DEV) for x in golden_draws(g)]
    rays, ts = g["rays"].to(DEV), g["ts"].to(DEV)
    with rendering.replay_rng(draws):
        outs = [rendering.render_rays(models, args, rays[i:i + 100], ts[i:i + 100]) for i in range(0, 250, 100)]
    res = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    l_sat = satnerf_loss(res, g["target"].to(DEV), lambda_sc=0.05)
    l_d = (1000.0 / 3.0) * torch.mean(g["dweights"].to(DEV) * (res["depth_coarse"] - g["dtarget"].to(DEV)) ** 2)
    assert abs(l_sat.item() - float(g["loss_satnerf"])) < 1e-4 * abs(float(g["loss_satnerf"]))
    assert abs(l_d.item() - float(g["loss_depth"])) < 1e-4 * abs(float(g["loss_depth"]))
    (l_sat + l_d).backward()
    sd = dict(models["coarse"].named_parameters())
    errs = {"fc_net.6.weight": maxnorm_rel(sd["fc_net.6.weight"].grad.cpu(), g["grad_fc_net_6_weight"]),
            "beta_from_xyz.2.weight": maxnorm_rel(sd["beta_from_xyz.2.weight"].grad.cpu(), g["grad_beta_2_weight"]),
            "sun_v_net.0.bias": maxnorm_rel(sd["sun_v_net.0.bias"].grad.cpu(), g["grad_sun_v_0_bias"]),
            "embedding": maxnorm_rel(models["t"].weight.grad.cpu(), g["grad_embedding"])}
    print({k: f"{e:.1e}" for k, e in errs.items()})
    assert max(errs.values()) < GRAD_TOL, errs


def test_training_steps_reduce_the_loss_and_track_the_oracle():
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    model = load_model(args).to(DEV)
    emb = torch.nn.Embedding(30, 4).to(DEV)
    params0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    emb0 = emb.weight.detach().cpu().clone()
    tr = Trainer({"coarse": model, "t": emb}, args)
    rays, ts = O.synthetic_rays(512, seed=3)
    target = torch.rand(512, 3, generator=torch.Generator().manual_seed(4)) * 0.2 + 0.4
    losses = [tr.step(rays.to(DEV), ts.to(DEV), target.to(DEV)).item() for _ in range(30)]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0] - 0.05, losses
    # first-step loss equals the oracle's on the same init (different jitter draws -> loose)
    with torch.no_grad():
        res = O.render_rays({"coarse": params0, "t": emb0}, O.default_args(), rays, ts)
        l0 = O.satnerf_loss(res, target).item()
    assert abs(losses[0] - l0) < 0.05 * abs(l0) + 0.02, (losses[0], l0)


def test_direct_step_matches_autograd_path_and_graph_replay():
    """Trainer's kernel-direct step (fused loss, no autograd) == render_rays + autograd + torch loss, and its hipGraph replay
    reproduces the eager step."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer, satnerf_loss

    args = O.default_args(mlp_mode="bf16")
    rays, ts = O.synthetic_rays(256, seed=9)
    rays, ts = rays.to(DEV), ts.to(DEV)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(1)).to(DEV)

    def fresh():
        torch.manual_seed(0)
        m = load_model(args).to(DEV)
        e = torch.nn.Embedding(30, 4).to(DEV)
        return {"coarse": m, "t": e}

    # autograd path
    ma = fresh()
    tra = Trainer(ma, O.default_args(mlp_mode="bf16"), loss_fn=lambda r, t: satnerf_loss(r, t))
    assert not tra.direct
    torch.manual_seed(5)
    res = rendering.render_rays(ma, tra.args, rays, ts)
    la = satnerf_loss(res, target)
    la.backward()
    ga = tra.state.grads.clone()
    # direct path, eager
    md = fresh()
    trd = Trainer(md, O.default_args(mlp_mode="bf16"), use_graph=False)
    assert trd.direct
    torch.manual_seed(5)
    ld = trd._forward_backward(rays, ts, target)
    gd = trd.state.grads.clone()
    assert abs(la.item() - ld.sum().item()) < 1e-5 * abs(la.item())
    assert maxnorm_rel(gd.cpu(), ga.cpu()) < 1e-4
    # graph replay: three steps track three eager steps
    mg, me = fresh(), fresh()
    trg, tre = Trainer(mg, O.default_args(mlp_mode="bf16"), use_graph=True), Trainer(me, O.default_args(mlp_mode="bf16"), use_graph=False)
    lg, le = [], []
    for _ in range(3):
        lg.append(trg.step(rays, ts, target).item())
        le.append(tre.step(rays, ts, target).item())
    assert trg._graph is not None
    # different jitter draws per trainer (graph-safe generator offsets) -> statistically equal, not bitwise
    assert all(abs(a - b) < 0.05 * abs(b) + 0.02 for a, b in zip(lg, le)), (lg, le)
    assert maxnorm_rel(trg.state.params.cpu(), tre.state.params.cpu()) < 5e-2


@pytest.mark.parametrize("tau,n_samples,mode,n_rays", [(16, 64, "bf16x3", 48), (4, 128, "bf16x3", 48), (4, 50, "bf16x3", 48),
                                                       # the 8-bit state = the 4-wave weight-gradient kernel (csrc/wgrad9.hip): two aux fragments
                                                       # (tau 16: raw duties on two waves), a ragged point count (37 x 50 = 1,850 points:
                                                       # 58 tiles, the last one partial, most slices one tile or none), one ray
                                                       # -- on these tiny batches the 8-BIT STATE itself moves the small sun-visibility
                                                       # gradients by 4e-2 .. 1.2e-1 (the r02 kernel, SATNERF_WGRAD_V1=1, measures the same
                                                       # 5.4e-2 / 4.3e-2 / 1.2e-1): the gate is 1.5x that, what is tested is the kernel's
                                                       # handling of the shapes
                                                       (16, 64, "bf16", 48), (4, 50, "bf16", 37), (16, 64, "f16", 1)])
def test_gradients_vs_oracle_autograd_variants(tau, n_samples, mode, n_rays):
    """Two aux k-steps (tau=16) and sample counts that are not one 64-lane wave, against autograd through the oracle."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model

    args = O.default_args(t_embbeding_tau=tau, n_samples=n_samples, mlp_mode=mode)
    params = O.procedural_satnerf_params(256, tau, seed=21)
    embw = O.procedural_uniform((30, tau), 1.0, 22)
    m = load_model(args)
    m.load_state_dict(params)
    emb = torch.nn.Embedding(30, tau)
    emb.load_state_dict({"weight": embw})
    models = {"coarse": m.to(DEV), "t": emb.to(DEV)}
    rays, ts = O.synthetic_rays(n_rays, seed=23)
    g = torch.Generator().manual_seed(24)
    u, nz = torch.rand(n_rays, n_samples, generator=g), torch.randn(n_rays, n_samples, generator=g)
    target = torch.rand(n_rays, 3, generator=g)
    loss_of = lambda r, t: ((r["rgb_coarse"] - t) ** 2).sum() + r["depth_coarse"].sum() + (r["weights_coarse"].unsqueeze(-1) * r["beta_coarse"]).sum()  # noqa: E731
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    loss_of(O.render_rays({"coarse": po, "t": eo}, O.default_args(t_embbeding_tau=tau, n_samples=n_samples), rays, ts, O.ReplayRng([u, nz])), target).backward()
    with rendering.replay_rng([u.to(DEV), nz.to(DEV)]):
        res = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    loss_of(res, target.to(DEV)).backward()
    sd = dict(models["coarse"].named_parameters())
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[k].grad) for k in po}
    errs["embedding"] = maxnorm_rel(models["t"].weight.grad.cpu(), eo.grad)
    worst = max(errs, key=errs.get)
    print(tau, n_samples, mode, n_rays, "worst", worst, f"{errs[worst]:.1e}")
    if mode == "bf16x3":
        assert errs[worst] < GRAD_TOL, errs
    else:
        # per tensor: every tensor within 3.5e-2 (1.5 x the worst of them, rgb_from_xyzdir.2.weight at 2.4e-2) EXCEPT the small
        # sun-visibility gradients named here, which the 8-bit state moves on batches this small (measured 5.4e-2 / 4.4e-2 on the 48- / 37-ray
        # cases; 1.2e-1, 5.3e-2, 3.9e-2 on the one-ray case: tools/grad_errs_tiny.py); at the benched shape they are inside the common gate
        # (tests/test_hip_benched_shape.py)
        loose = {48: {"sun_v_net.2.weight": 8e-2}, 37: {"sun_v_net.2.weight": 6.5e-2},
                 1: {"sun_v_net.2.weight": 1.8e-1, "sun_v_net.4.weight": 8e-2, "sun_v_net.6.weight": 6e-2}}[n_rays]
        bad = {k: v for k, v in errs.items() if v >= loose.get(k, 3.5e-2)}
        assert not bad, (bad, errs)
    assert all(torch.isfinite(sd[k].grad).all() for k in po)


def test_trainer_direct_step_with_128_samples_uses_separate_kernels():
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(n_samples=128, mlp_mode="bf16")
    tr = Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, args)
    rays, ts = O.synthetic_rays(256, seed=3)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(4)) * 0.2 + 0.4
    losses = [tr.step(rays.to(DEV), ts.to(DEV), target.to(DEV)).item() for _ in range(20)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0]


def _depth_setup(n_color=128, n_depth=96, tau=4):
    from satnerf_amd.models import load_model

    args = O.default_args(mlp_mode="bf16x3", ds_lambda=1000.0)
    params = O.procedural_satnerf_params(256, tau, seed=31)
    embw = O.procedural_uniform((30, tau), 1.0, 32)
    m = load_model(args)
    m.load_state_dict(params)
    emb = torch.nn.Embedding(30, tau)
    emb.load_state_dict({"weight": embw})
    rays, ts = O.synthetic_rays(n_color, seed=33)
    d_rays, d_ts = O.synthetic_rays(n_depth, seed=34)
    g = torch.Generator().manual_seed(35)
    target = torch.rand(n_color, 3, generator=g)
    depths = torch.stack([0.2 + 0.5 * torch.rand(n_depth, generator=g), 0.5 + torch.rand(n_depth, generator=g)], 1)
    return args, params, embw, {"coarse": m.to(DEV), "t": emb.to(DEV)}, (rays, ts, target), (d_rays, d_ts, depths)


def test_depth_supervision_pass_matches_oracle_autograd():
    """Trainer._depth_pass (main.py:134-141 + metrics.DepthLoss) against autograd through the oracle on the same jitter."""
    from satnerf_amd.train import Trainer

    args, params, embw, models, _, (d_rays, d_ts, depths) = _depth_setup()
    tr = Trainer(models, args, use_graph=False)
    assert tr.direct
    tr.models["coarse"].repack("bf16x3", backward=True)
    torch.manual_seed(77)
    loss = tr._depth_pass(d_rays.to(DEV), d_ts.to(DEV), depths.to(DEV), 0.0)
    torch.manual_seed(77)
    u = torch.rand(d_rays.shape[0], args.n_samples, device=DEV).cpu()
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    res = O.render_rays({"coarse": po, "t": eo}, O.default_args(), d_rays, d_ts, O.ReplayRng([u, torch.zeros_like(u)]))
    lo = O.depth_loss(res, depths[:, 0], depths[:, 1], lambda_ds=1000.0)
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-4 * abs(lo.item())
    sd = dict(models["coarse"].named_parameters())
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[k].grad) for k in po if po[k].grad is not None and po[k].grad.abs().max() > 0}
    assert "sigma_from_xyz.0.weight" in errs and "fc_net.0.weight" in errs
    worst = max(errs, key=errs.get)
    print("depth pass worst", worst, f"{errs[worst]:.1e}")
    assert errs[worst] < GRAD_TOL, errs
    # heads that depth does not reach stay at zero gradient
    for k in ("rgb_from_xyzdir.0.weight", "beta_from_xyz.0.weight", "sun_v_net.0.weight"):
        assert sd[k].grad.abs().max().item() == 0.0


def test_depth_supervised_step_direct_matches_autograd_path_and_graph():
    from satnerf_amd import rendering
    from satnerf_amd.data import DepthBank, RayBank
    from satnerf_amd.train import Trainer, depth_loss, satnerf_loss

    args, params, embw, models, color, depth = _depth_setup()
    color = tuple(t.to(DEV) for t in color)
    depth = tuple(t.to(DEV) for t in depth)
    rays, ts, target = color
    # direct, eager: colour + depth gradients in one flat buffer
    tr = Trainer(models, args, use_graph=False)
    torch.manual_seed(5)
    parts = tr._forward_backward(rays, ts, target, depth=depth)
    g_direct = tr.state.grads.clone()
    # same draws through render_rays + autograd + the torch losses
    tr.state.zero_grad()
    torch.manual_seed(5)
    u1 = torch.rand(rays.shape[0], 64, device=DEV)
    u2 = torch.rand(depth[0].shape[0], 64, device=DEV)
    with rendering.replay_rng([u1, torch.zeros_like(u1)]):
        res = rendering.render_rays(models, args, rays, ts)
    with rendering.replay_rng([u2, torch.zeros_like(u2)]):
        res_d = rendering.render_rays(models, args, depth[0], depth[1])
    la = satnerf_loss(res, target) + depth_loss(res_d, depth[2][:, 0], depth[2][:, 1], 1000.0)
    la.backward()
    assert abs(parts.sum().item() - la.item()) < 1e-4 * abs(la.item())
    assert maxnorm_rel(g_direct.cpu(), tr.state.grads.cpu()) < 1e-4
    tr.state.zero_grad()
    # a depth batch without ds_lambda is refused
    with pytest.raises(ValueError):
        Trainer(models, O.default_args(mlp_mode="bf16x3"), use_graph=False).step(rays, ts, target, depth=depth)
    # graph-captured steps from the banks: the loss (dominated by the depth term) goes down
    trg = Trainer(models, args, use_graph=True)
    bank = RayBank(rays, target, ts, batch_size=64, seed=1)
    dbank = DepthBank(depth[0], depth[2], depth[1], batch_size=48, seed=2)
    losses = [trg.step_from_bank(bank, dbank).item() for _ in range(25)]
    assert trg._graph is not None and len(trg._static) == 6
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < 0.7 * losses[0], losses
    # past ds_drop the caller stops passing the depth batch: a second graph shape is captured transparently
    l_plain = trg.step_from_bank(bank).item()
    assert len(trg._static) == 3 and l_plain == l_plain


@pytest.mark.parametrize("feat,tau,sc", [(512, 16, 0.0), (128, 4, 0.1)])
def test_layer_path_gradients_vs_oracle_autograd(feat, tau, sc):
    """Widths outside the fused kernel train through autograd over the per-layer HIP Functions: gradients of every parameter
    and of the embedding against autograd through the oracle on the same draws (solar-correction pass included)."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import satnerf_loss

    args = O.default_args(fc_units=feat, t_embbeding_tau=tau, sc_lambda=sc)
    params = O.procedural_satnerf_params(feat, tau, seed=51)
    embw = O.procedural_uniform((30, tau), 1.0, 52)
    m = load_model(args)
    m.load_state_dict(params)
    assert not m.fused
    emb = torch.nn.Embedding(30, tau)
    emb.load_state_dict({"weight": embw})
    models = {"coarse": m.to(DEV), "t": emb.to(DEV)}
    n = 40
    rays, ts = O.synthetic_rays(n, seed=53)
    g = torch.Generator().manual_seed(54)
    draws = [torch.rand(n, 64, generator=g), torch.randn(n, 64, generator=g)] + ([torch.randn(n, 64, generator=g)] if sc > 0 else [])
    target = torch.rand(n, 3, generator=g)
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    lo = O.satnerf_loss(O.render_rays({"coarse": po, "t": eo}, args, rays, ts, O.ReplayRng(draws)), target, lambda_sc=sc)
    lo.backward()
    with rendering.replay_rng([d.to(DEV) for d in draws]):
        res = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    lh = satnerf_loss(res, target.to(DEV), lambda_sc=sc)
    lh.backward()
    assert abs(lh.item() - lo.item()) < 1e-4 * abs(lo.item())
    sd = dict(models["coarse"].named_parameters())
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[k].grad) for k in po}
    errs["embedding"] = maxnorm_rel(models["t"].weight.grad.cpu(), eo.grad)
    worst = max(errs, key=errs.get)
    print(feat, tau, "worst", worst, f"{errs[worst]:.1e}")
    assert errs[worst] < 1e-3, errs


@pytest.mark.parametrize("tau,sc,ds", [(4, 0.0, 0.0), (16, 0.1, 0.5)])
def test_width_512_fused_training_gradients_vs_oracle_autograd(tau, sc, ds):
    """opt.py:50's default fc_units = 512 (every sat-nerf line of run_all.sh) in the throughput arithmetic: the 512-wide builds of
    the fused forward / dX kernels + the 8-bit weight-gradient kernel.  The kernel-direct Trainer's gradients of every parameter
    and of the embedding (colour pass + solar correction + depth supervision) against autograd through the fp32 oracle on the
    same draws; tolerance = the one of the 256-wide throughput mode (single-pass bf16 MFMA + 8-bit saved state)."""
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    args = O.default_args(fc_units=512, t_embbeding_tau=tau, sc_lambda=sc, ds_lambda=ds, mlp_mode="bf16")
    params = O.procedural_satnerf_params(512, tau, seed=61)
    embw = O.procedural_uniform((30, tau), 1.0, 62)
    m = load_model(args)
    m.load_state_dict(params)
    assert not m.fused and m.fused_training("bf16", 8) and not m.fused_training("bf16x3", 16)
    emb = torch.nn.Embedding(30, tau)
    emb.load_state_dict({"weight": embw})
    models = {"coarse": m.to(DEV), "t": emb.to(DEV)}
    n, nd = 384, 128
    rays, ts = O.synthetic_rays(n, seed=63)
    d_rays, d_ts = O.synthetic_rays(nd, seed=64)
    g = torch.Generator().manual_seed(65)
    target = torch.rand(n, 3, generator=g)
    depths = torch.stack([0.3 + 0.4 * torch.rand(nd, generator=g), 0.5 + torch.rand(nd, generator=g)], 1)
    # the kernel-direct step draws its stratified jitter from torch's device generator: colour batch first, then the depth batch
    torch.manual_seed(66)
    u = torch.rand(n, 64, device=DEV).cpu()
    u_d = torch.rand(nd, 64, device=DEV).cpu()
    draws = [u, torch.zeros(n, 64)] + ([torch.zeros(n, 64)] if sc > 0 else [])
    d_draws = [u_d, torch.zeros(nd, 64)] + ([torch.zeros(nd, 64)] if sc > 0 else [])
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    mo = {"coarse": po, "t": eo}
    lo = O.satnerf_loss(O.render_rays(mo, args, rays, ts, O.ReplayRng(draws)), target, lambda_sc=sc)
    if ds > 0:
        lo = lo + O.depth_loss(O.render_rays(mo, args, d_rays, d_ts, O.ReplayRng(d_draws)), depths[:, 0], depths[:, 1], ds)
    lo.backward()

    tr = Trainer(models, args, use_graph=False)
    assert tr.direct
    torch.manual_seed(66)
    parts = tr._forward_backward(rays.to(DEV), ts.to(DEV), target.to(DEV),
                                 depth=(d_rays.to(DEV), d_ts.to(DEV), depths.to(DEV)) if ds > 0 else None)
    lh = parts.sum()
    assert abs(lh.item() - lo.item()) < 2e-2 * abs(lo.item()), (lh.item(), lo.item())
    sd = dict(models["coarse"].named_parameters())
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[k].grad) for k in po if po[k].grad is not None}
    errs["embedding"] = maxnorm_rel(models["t"].weight.grad.cpu(), eo.grad)
    worst = max(errs, key=errs.get)
    print(512, tau, "worst", worst, f"{errs[worst]:.1e}")
    assert errs[worst] < 5e-2, errs  # (the 256-wide step on the same batch: 3e-2, tools/grad_err_widths.py)


def test_trainer_width_512_direct_training_run():
    """512-wide kernel-direct steps (graph-captured) make the loss fall; the parity mode keeps the autograd path."""
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(fc_units=512, mlp_mode="bf16")
    tr = Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, args)
    assert tr.direct
    rays, ts = O.synthetic_rays(512, seed=3)
    target = torch.rand(512, 3, generator=torch.Generator().manual_seed(4)) * 0.2 + 0.4
    losses = [tr.step(rays.to(DEV), ts.to(DEV), target.to(DEV)).item() for _ in range(12)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


def test_trainer_runs_width_512_through_autograd_path():
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(fc_units=512)
    tr = Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, args)
    assert not tr.direct
    rays, ts = O.synthetic_rays(256, seed=3)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(4)) * 0.2 + 0.4
    losses = [tr.step(rays.to(DEV), ts.to(DEV), target.to(DEV)).item() for _ in range(12)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


def test_classic_nerf_gradients_and_training():
    """Classic nerf trains through autograd over the per-layer HIP Functions: gradients against autograd through the oracle."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer, nerf_loss

    args = O.default_args(model="nerf", n_importance=16)
    pc, pf = O.procedural_nerf_params(256, seed=61), O.procedural_nerf_params(256, seed=62)
    models = {}
    for typ, prm in (("coarse", pc), ("fine", pf)):
        m = load_model(args)
        m.load_state_dict(prm)
        models[typ] = m.to(DEV)
    n = 24
    rays, _ = O.synthetic_rays(n, seed=63, classic=True)
    g = torch.Generator().manual_seed(64)
    draws = [torch.rand(n, 64, generator=g), torch.randn(n, 64, generator=g), torch.rand(n, 16, generator=g), torch.randn(n, 80, generator=g)]
    target = torch.rand(n, 3, generator=g)
    po = {t: {k: v.clone().requires_grad_(True) for k, v in prm.items()} for t, prm in (("coarse", pc), ("fine", pf))}
    res_o = O.render_rays(po, args, rays, None, O.ReplayRng(draws))
    lo = ((res_o["rgb_coarse"] - target) ** 2).mean() + ((res_o["rgb_fine"] - target) ** 2).mean()
    lo.backward()
    with rendering.replay_rng([d.to(DEV) for d in draws]):
        res = rendering.render_rays(models, args, rays.to(DEV), None)
    lh = nerf_loss(res, target.to(DEV))
    lh.backward()
    assert abs(lh.item() - lo.item()) < 1e-4 * abs(lo.item())
    for typ in ("coarse", "fine"):
        sd = dict(models[typ].named_parameters())
        errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[typ][k].grad) for k in po[typ]}
        worst = max(errs, key=errs.get)
        print(typ, "worst", worst, f"{errs[worst]:.1e}")
        # ReLU gates: the 3-pass bf16 GEMM's ~1e-5 pre-activation error flips the gate of units sitting that close to zero
        # (a few hundred of 3 M unit-points here), each a discrete change of one point's contribution -- same bar as the
        # fused kernel's gradients
        assert errs[worst] < GRAD_TOL, errs
    tr = Trainer(models, args)
    assert not tr.direct
    losses = [tr.step(rays.to(DEV), None, target.to(DEV)).item() for _ in range(8)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


def test_in_kernel_jitter_is_uniform_reproducible_and_steps():
    """sr_ray_setup_rng: the Philox jitter lands every depth inside its stratum, is uniform, repeats for the same (seed, step)
    and changes with either; the sky head equals the plain launch."""
    from satnerf_amd import ops
    from satnerf_amd.models import load_model

    m = load_model(O.default_args()).to(DEV)
    sk = m.sky_color
    w = (sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
    rays, _ = O.synthetic_rays(2048, seed=81)
    rays = rays.to(DEV)
    step = torch.zeros(1, device=DEV)
    z0, sky0 = ops.ray_setup(rays, None, 64, *w, seed=123, step_counter=step)
    z0b, _ = ops.ray_setup(rays, None, 64, *w, seed=123, step_counter=step)
    assert torch.equal(z0, z0b)
    step += 1
    z1, _ = ops.ray_setup(rays, None, 64, *w, seed=123, step_counter=step)
    z2, _ = ops.ray_setup(rays, None, 64, *w, seed=124, step_counter=step)
    assert not torch.equal(z0, z1) and not torch.equal(z1, z2)
    lo, sky_ref = ops.ray_setup(rays, torch.zeros(2048, 64, device=DEV), 64, *w)
    hi, _ = ops.ray_setup(rays, torch.ones(2048, 64, device=DEV), 64, *w)
    assert torch.equal(sky0, sky_ref)
    u = ((z1 - lo) / (hi - lo)).flatten()
    assert (u >= 0).all() and (u < 1.0 + 1e-6).all()
    assert abs(u.mean().item() - 0.5) < 5e-3 and abs(u.var().item() - 1 / 12) < 2e-3
    hist = torch.histc(u.clamp(0, 1 - 1e-7), bins=16, min=0, max=1) / u.numel()
    assert (hist - 1 / 16).abs().max().item() < 4e-3
    # neighbouring samples and rays are uncorrelated
    uu = ((z1 - lo) / (hi - lo))
    c1 = torch.corrcoef(torch.stack([uu[:, :-1].flatten(), uu[:, 1:].flatten()]))[0, 1].abs().item()
    c2 = torch.corrcoef(torch.stack([uu[:-1].flatten(), uu[1:].flatten()]))[0, 1].abs().item()
    assert c1 < 2e-2 and c2 < 2e-2


def test_coarse_plus_fine_gradients_and_training():
    """BASELINE configs[2] shape (coarse + fine with importance sampling) through autograd over the fused kernels: both models'
    gradients against autograd through the oracle on the same draws; the trainer's step lowers the loss."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    args = O.default_args(n_importance=32, mlp_mode="bf16x3")
    pc, pf = O.procedural_satnerf_params(256, 4, seed=91), O.procedural_satnerf_params(256, 4, seed=92)
    embw = O.procedural_uniform((30, 4), 1.0, 93)
    models = {}
    for typ, prm in (("coarse", pc), ("fine", pf)):
        m = load_model(args)
        m.load_state_dict(prm)
        models[typ] = m.to(DEV)
    emb = torch.nn.Embedding(30, 4)
    emb.load_state_dict({"weight": embw})
    models["t"] = emb.to(DEV)
    n = 40
    rays, ts = O.synthetic_rays(n, seed=94)
    g = torch.Generator().manual_seed(95)
    draws = [torch.rand(n, 64, generator=g), torch.randn(n, 64, generator=g), torch.rand(n, 32, generator=g), torch.randn(n, 96, generator=g)]
    target = torch.rand(n, 3, generator=g)
    loss_of = lambda r, t: sum(((r[f"rgb_{k}"] - t) ** 2).mean() + (r[f"weights_{k}"].unsqueeze(-1) * r[f"beta_{k}"]).sum() * 1e-2  # noqa: E731
                               for k in ("coarse", "fine"))
    po = {t: {k: v.clone().requires_grad_(True) for k, v in prm.items()} for t, prm in (("coarse", pc), ("fine", pf))}
    eo = embw.clone().requires_grad_(True)
    lo = loss_of(O.render_rays({"coarse": po["coarse"], "fine": po["fine"], "t": eo}, O.default_args(n_importance=32), rays, ts, O.ReplayRng(draws)), target)
    lo.backward()
    with rendering.replay_rng([d.to(DEV) for d in draws]):
        res = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    assert res["weights_fine"].shape == (n, 96)
    lh = loss_of(res, target.to(DEV))
    lh.backward()
    assert abs(lh.item() - lo.item()) < 1e-4 * abs(lo.item())
    for typ in ("coarse", "fine"):
        sd = dict(models[typ].named_parameters())
        errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[typ][k].grad) for k in po[typ]}
        worst = max(errs, key=errs.get)
        print(typ, "worst", worst, f"{errs[worst]:.1e}")
        assert errs[worst] < GRAD_TOL, errs
    assert maxnorm_rel(models["t"].weight.grad.cpu(), eo.grad) < GRAD_TOL
    targs = O.default_args(n_importance=32, mlp_mode="bf16")
    with pytest.raises(NotImplementedError):  # the reference's SatNerfLoss cannot train sat-nerf + fine (metrics.py:22): no silent coarse-only loss
        Trainer(models, targs)
    tr = Trainer(models, targs, loss_fn=loss_of)
    assert not tr.direct and tr.state.params.numel() == 2 * 662537 + 120
    fine0 = models["fine"].flat_params().detach().clone()
    losses = [tr.step(rays.to(DEV), ts.to(DEV), target.to(DEV)).item() for _ in range(10)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    assert (models["fine"].flat_params().detach() - fine0).abs().max().item() > 0  # the fine model is trained too


def test_direct_step_with_solar_correction_matches_autograd_path():
    """sc_lambda > 0 on the kernel-direct path (second pass along the sun direction + sr_sc_loss) == render_rays under autograd +
    the torch SatNerfLoss with solar correction, same jitter; and the captured step trains."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer, satnerf_loss

    args = O.default_args(mlp_mode="bf16x3", sc_lambda=0.1)
    params = O.procedural_satnerf_params(256, 4, seed=101)
    m = load_model(args)
    m.load_state_dict(params)
    emb = torch.nn.Embedding(30, 4)
    emb.load_state_dict({"weight": O.procedural_uniform((30, 4), 1.0, 102)})
    models = {"coarse": m.to(DEV), "t": emb.to(DEV)}
    rays, ts = O.synthetic_rays(160, seed=103)
    rays, ts = rays.to(DEV), ts.to(DEV)
    target = torch.rand(160, 3, generator=torch.Generator().manual_seed(104)).to(DEV)
    tr = Trainer(models, args, use_graph=False)
    assert tr.direct
    torch.manual_seed(11)
    parts = tr._forward_backward(rays, ts, target)
    g_direct = tr.state.grads.clone()
    tr.state.zero_grad()
    torch.manual_seed(11)
    u = torch.rand(160, 64, device=DEV)
    with rendering.replay_rng([u, torch.zeros_like(u), torch.zeros_like(u)]):
        res = rendering.render_rays(models, args, rays, ts)
    la = satnerf_loss(res, target, lambda_sc=0.1)
    la.backward()
    assert abs(parts.sum().item() - la.item()) < 1e-4 * abs(la.item()), (parts.sum().item(), la.item())
    assert maxnorm_rel(g_direct.cpu(), tr.state.grads.cpu()) < 1e-4
    tr.state.zero_grad()
    # S = 128 (two 64-lane segments in sr_sc_loss) against the same torch formulation
    args128 = O.default_args(mlp_mode="bf16x3", sc_lambda=0.1, n_samples=128)
    tr128 = Trainer(models, args128, use_graph=False)
    torch.manual_seed(12)
    p128 = tr128._forward_backward(rays[:48], ts[:48], target[:48])
    g128 = tr128.state.grads.clone()
    tr128.state.zero_grad()
    torch.manual_seed(12)
    u = torch.rand(48, 128, device=DEV)
    with rendering.replay_rng([u, torch.zeros_like(u), torch.zeros_like(u)]):
        res = rendering.render_rays(models, args128, rays[:48], ts[:48])
    la = satnerf_loss(res, target[:48], lambda_sc=0.1)
    la.backward()
    assert abs(p128.sum().item() - la.item()) < 1e-4 * abs(la.item())
    assert maxnorm_rel(g128.cpu(), tr128.state.grads.cpu()) < 1e-4
    tr128.state.zero_grad()
    # captured steps
    trg = Trainer(models, O.default_args(mlp_mode="bf16", sc_lambda=0.1))
    losses = [trg.step(rays, ts, target).item() for _ in range(20)]
    assert trg._graph is not None and all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("feat,tau,n_rays", [(256, 4, 300), (256, 16, 41), (512, 4, 70), (512, 16, 9)])
def test_generated_dx_trunk_writes_the_same_bytes_as_the_compiler_scheduled_one(monkeypatch, feat, tau, n_rays):
    """csrc/gen/bwd_core.py's instruction stream (the seven trunk layers of the dX kernel, default) against the compiler-scheduled loop of
    csrc/mlp_bwd.inc (SATNERF_BWD_V1=1): the same arithmetic per value in the same order, so the dpre workspace -- MX8 bytes, scale bytes,
    the table of exponent maxima behind the last tile -- and d_t must be identical bit for bit.  tau 16 = two aux fragments (other phase units)."""
    from satnerf_amd import ops
    from satnerf_amd.models import load_model

    torch.manual_seed(0)
    s, mode = 64, "bf16"
    args = O.default_args(mlp_mode=mode, t_embbeding_tau=tau, fc_units=feat)
    model = load_model(args).to(DEV)
    emb = torch.nn.Embedding(30, tau).to(DEV)
    rays, ts = O.synthetic_rays(n_rays, seed=9)
    rays, ts = rays.to(DEV), ts.to(DEV)
    n = n_rays * s
    model.repack(mode, backward=True)
    hi, lo, l0 = model.packed(mode)
    bstream, _ = model.packed_backward()
    z = ops.ray_sample(rays, torch.rand(n_rays, s, device=DEV), s)
    acts = ops.acts_workspace(n, feat, DEV, 8)
    albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, n, s, feat, tau, mode, hi, lo, l0,
                                                 acts=acts, fmt=8)
    g = torch.Generator(device=DEV).manual_seed(3)
    ga, gs, gv, gb = (torch.randn(n, 3, device=DEV, generator=g) * 1e-3, torch.randn(n, device=DEV, generator=g) * 1e-3,
                      torch.randn(n, device=DEV, generator=g) * 1e-3, torch.randn(n, device=DEV, generator=g) * 1e-4)
    # (bytes no kernel writes -- the unused half of the head groups' scale slots -- must not be compared as garbage: zeroed workspaces)
    monkeypatch.setattr(ops, "_ws_empty", lambda n_, dtype, device, slot: torch.zeros(n_, dtype=dtype, device=device))
    outs = []
    for v1 in ("1", "0"):
        monkeypatch.setenv("SATNERF_BWD_V1", v1)
        dpre, d_t = ops.satnerf_mlp_bwd(feat, tau, n, bstream, acts, albedo, sigma, sun_v, beta, ga, gs, gv, gb, fmt=8)
        torch.cuda.synchronize()
        outs.append((dpre.clone(), d_t.clone()))
    tiles = (n + 31) // 32
    from satnerf_amd import packing

    per_tile = packing.dpre8_units(feat) * 512
    assert torch.equal(outs[0][1], outs[1][1])
    a, b = outs[0][0][:tiles * per_tile].view(tiles, -1), outs[1][0][:tiles * per_tile].view(tiles, -1)
    bad = (a != b).any(1).nonzero().view(-1)
    assert bad.numel() == 0, (bad[:8], (a != b).sum().item())
    ws_tiles = (tiles + 7) // 8 * 8
    ta, tb = outs[0][0][ws_tiles * per_tile:], outs[1][0][ws_tiles * per_tile:]
    assert torch.equal(ta[:(tiles + 3) // 4 * 8], tb[:(tiles + 3) // 4 * 8])
