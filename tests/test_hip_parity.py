"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.

Tolerances (max-norm relative): bf16x3 parity mode <= 1e-4 on rgb / depth / weights as north_star
states; integer-free fp32 stages (sampling, compositing, sky) much tighter.  The single-pass bf16 throughput mode
is measured and bounded per output at ~1.7x its measured error (1e-3 .. 3e-3; BF16_*_GATES below).
"""
import pytest
import torch

from oracle import satnerf_oracle as O
from tests.helpers import make_models, golden_cfg, golden_draws, load_golden, maxnorm_rel

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _lazy():
    from satnerf_amd import ops, rendering
    from satnerf_amd.models import load_model

    return ops, rendering, load_model


def build_models(args, seeds=(1, 2), emb_seed=7):
    _, _, load_model = _lazy()
    models = {}
    for typ, seed in (("coarse", seeds[0]),) + ((("fine", seeds[1]),) if args.n_importance > 0 else ()):
        m = load_model(args)
        m.load_state_dict(O.procedural_satnerf_params(args.fc_units, args.t_embbeding_tau, seed=seed))
        models[typ] = m.to(DEV).eval()
    emb = torch.nn.Embedding(args.t_embbeding_vocab, args.t_embbeding_tau)
    emb.load_state_dict({"weight": O.procedural_uniform((args.t_embbeding_vocab, args.t_embbeding_tau), 1.0, emb_seed)})
    models["t"] = emb.to(DEV)
    return models


def test_native_library_is_loaded():
    from satnerf_amd import _lib

    assert _lib.lib().sr_version() == 100
    maps = open("/proc/self/maps").read()
    assert "libsatrender.so" in maps


def test_ray_sample_bit_exact():
    ops, _, _ = _lazy()
    for s in (64, 128, 50, 2):
        rays, _ = O.synthetic_rays(77, seed=s)
        u = torch.rand(77, s)
        want = O.stratified_depths(rays, s, u)
        got = ops.ray_sample(rays.to(DEV), u.to(DEV), s).cpu()
        assert torch.equal(got, want), (s, (got - want).abs().max())


def test_sky_head():
    ops, _, _ = _lazy()
    p = O.procedural_satnerf_params(256, 4, seed=1)
    rays, _ = O.synthetic_rays(301, seed=3)
    sun = rays[:, 8:11]
    k = torch.relu(torch.nn.functional.linear(sun, p["sky_color.0.weight"], p["sky_color.0.bias"]))
    want = torch.sigmoid(torch.nn.functional.linear(k, p["sky_color.2.weight"], p["sky_color.2.bias"]))
    d = {k: v.to(DEV) for k, v in p.items()}
    got = ops.sky(rays.to(DEV)[:, 8:11], d["sky_color.0.weight"], d["sky_color.0.bias"], d["sky_color.2.weight"], d["sky_color.2.bias"]).cpu()
    assert maxnorm_rel(got, want) < 2e-6


def test_composite_golden_extreme_sigmas():
    ops, _, _ = _lazy()
    g = load_golden("composite_extreme")
    raw, z = g["raw"].to(DEV), g["z"].to(DEV)
    n, s = z.shape
    sky = raw[:, 0, 5:8].contiguous()  # per-ray sky; the golden's per-sample sky varies, so rebuild the expectation
    zc, noise = g["z"], g["noise"] * float(g["noise_std"])
    w_ref, t_ref = O.alpha_composite(zc, g["raw"][..., 3], noise)
    irr = g["raw"][..., 4:5] + (1 - g["raw"][..., 4:5]) * g["raw"][:, :1, 5:8]
    rgb_ref = torch.clamp(torch.sum(w_ref.unsqueeze(-1) * g["raw"][..., :3] * irr, -2), 0, 1)
    w, t, depth, rgb = ops.composite(z, raw[..., 3].contiguous(), g["noise"].to(DEV), float(g["noise_std"]), raw[..., :3].contiguous(),
                                     raw[..., 4].contiguous(), sky)
    assert maxnorm_rel(w.cpu(), g["out_weights"]) < 2e-6
    assert maxnorm_rel(t.cpu(), g["out_transparency"]) < 2e-6
    assert maxnorm_rel(depth.cpu(), g["out_depth"]) < 2e-6
    assert maxnorm_rel(rgb.cpu(), rgb_ref) < 2e-6
    # element-wise on the weights that matter; alpha = 1 - exp(-x) carries an absolute 1-ulp-of-1 (6e-8) uncertainty
    # between exp implementations, so the element-wise relative bound is only meaningful for w >~ 1e-3
    m = g["out_weights"] > 1e-3
    assert ((w.cpu() - g["out_weights"]).abs()[m] / g["out_weights"][m]).max() < 1e-4


@pytest.mark.parametrize("s,i", [(64, 64), (64, 48), (50, 37), (128, 128), (192, 192)])
def test_sample_pdf_merge_matches_oracle(s, i):
    ops, _, _ = _lazy()
    g = torch.Generator().manual_seed(s * 1000 + i)
    n = 45
    z = torch.sort(torch.rand(n, s, generator=g), -1)[0]
    w = torch.rand(n, s, generator=g) ** 4
    w[3] = 0.0
    w[4, 10:40] = 0.0
    u = torch.rand(n, i, generator=g)
    mid = 0.5 * (z[:, :-1] + z[:, 1:])
    z_new = O.importance_depths(mid, w[:, 1:-1], u)
    want = torch.sort(torch.cat([z, z_new], -1), -1)[0]
    got = ops.sample_pdf_merge(z.to(DEV), w.to(DEV), u.to(DEV)).cpu()
    assert got.shape == want.shape
    assert (got[:, 1:] >= got[:, :-1]).all()
    # z_new = bins + (u - cdf)/denom * width amplifies the fp32 rounding ORDER of the cdf (wave scan here, sequential
    # cumsum in torch) by 1/denom, denom >= eps = 1e-5: compare against an fp64 evaluation and require the HIP result to
    # be as close to it as the fp32 oracle is (x4 + 1e-6), and within 1e-4 of the oracle everywhere.
    z64 = O.importance_depths(mid.double(), w[:, 1:-1].double(), u.double())
    want64 = torch.sort(torch.cat([z.double(), z64], -1), -1)[0]
    err_hip, err_ref = (got.double() - want64).abs().max().item(), (want.double() - want64).abs().max().item()
    assert (got - want).abs().max() < 1e-4
    assert ((got - want).abs() > 2e-6).float().mean() < 2e-3
    assert err_hip <= 4 * err_ref + 1e-6, (err_hip, err_ref)


def test_sample_pdf_standalone_golden():
    _, rendering, _ = _lazy()
    g = load_golden("sample_pdf")
    with rendering.replay_rng([g["u"].to(DEV)]):
        z = rendering.sample_pdf(g["bins"].to(DEV), g["weights"].to(DEV), 48, det=False).cpu()
    zd = rendering.sample_pdf(g["bins"].to(DEV), g["weights"].to(DEV), 48, det=True).cpu()
    assert z.shape == g["z_rand"].shape
    # det=True (unreachable in the reference: perturb is hard-wired to 1, rendering.py:59) ends at u == 1.0 exactly, where the
    # result flips between the last two bins depending on whether the fp32 cdf tail rounds to 1.0000001 or 0.99999994 -- a
    # 1-ulp discontinuity of the algorithm itself (summation order), so that single endpoint is excluded
    for got, want in ((z, g["z_rand"]), (zd[:, :-1], g["z_det"][:, :-1])):
        assert (got - want).abs().max() < 1e-4
        assert ((got - want).abs() > 2e-6).float().mean() < 5e-3


# per-output gates of the single-pass bf16 arithmetic, ~1.7x what it measures against the reference (r04: albedo 2.1e-3, sigma 3.3e-3,
# sun 5.8e-4, beta 2.4e-3; rgb 1.2e-3 / depth 7.3e-4 / weights 1.1e-3 .. 1.5e-3 after compositing): a regression that costs half a digit fails
BF16_POINT_GATES = {"albedo": 4e-3, "sigma": 6e-3, "sun": 1.5e-3, "sky": 1e-6, "beta": 4e-3}
BF16_RENDER_GATES = {"rgb": 2.5e-3, "depth": 1.5e-3, "weights": 3e-3, "beta": 4e-3, "sun": 1.5e-3}


def _gate(errs, gates, what=""):
    bad = {k: (e, gates[k.split("_")[0]]) for k, e in errs.items() if e >= gates[k.split("_")[0]]}
    assert not bad, (what, bad, errs)


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-4), ("bf16", None)])
def test_mlp_forward_points_golden(mode, tol):
    _, _, load_model = _lazy()
    g = load_golden("mlp_forward")
    m = load_model(O.default_args())
    m.load_state_dict(O.procedural_satnerf_params(256, 4, seed=1))
    m = m.to(DEV)
    out = m(g["xyz"].to(DEV), input_sun_dir=g["sun"].to(DEV), input_t=g["t"].to(DEV), mlp_mode=mode).cpu()
    assert out.shape == (257, 9)
    errs = {name: maxnorm_rel(out[:, sl], g["out"][:, sl]) for name, sl in
            (("albedo", slice(0, 3)), ("sigma", slice(3, 4)), ("sun", slice(4, 5)), ("sky", slice(5, 8)), ("beta", slice(8, 9)))}
    print(mode, errs)
    if tol is None:
        _gate(errs, BF16_POINT_GATES, mode)
    else:
        assert max(errs.values()) < tol, errs
    sig = m(g["xyz"].to(DEV), input_sun_dir=g["sun"].to(DEV), input_t=g["t"].to(DEV), sigma_only=True, mlp_mode=mode).cpu()
    assert maxnorm_rel(sig, g["sigma_only"]) < (BF16_POINT_GATES["sigma"] if tol is None else tol)


RENDER_CASES = ["satnerf_coarse", "satnerf_sc", "satnerf_fine", "satnerf_noise", "satnerf_s128", "satnerf_s50_ragged"]


@pytest.mark.parametrize("name", RENDER_CASES)
def test_render_rays_golden_parity_mode(name):
    _, rendering, _ = _lazy()
    g = load_golden(name)
    args = golden_cfg(g)
    args.mlp_mode = "bf16x3"
    models = build_models(args)
    draws = [d.to(DEV) for d in golden_draws(g)]
    with torch.no_grad(), rendering.replay_rng(draws):
        res = rendering.render_rays(models, args, g["rays"].to(DEV), g["ts"].to(DEV))
    torch.cuda.synchronize()
    expected = {k[4:]: v for k, v in g.items() if k.startswith("out_")}
    assert set(res) == set(expected)
    worst = {}
    for k, v in expected.items():
        assert tuple(res[k].shape) == tuple(v.shape), k
        worst[k] = maxnorm_rel(res[k].cpu(), v)
    print(name, {k: f"{e:.1e}" for k, e in worst.items()})
    for k, e in worst.items():
        # north_star: rgb, depth, weights within 1e-4; the per-sample heads ride the same bound
        assert e < 1e-4, (k, e)


def test_render_rays_throughput_mode_is_close():
    _, rendering, _ = _lazy()
    g = load_golden("satnerf_coarse")
    args = golden_cfg(g)
    args.mlp_mode = "bf16"
    models = build_models(args)
    with torch.no_grad(), rendering.replay_rng([d.to(DEV) for d in golden_draws(g)]):
        res = rendering.render_rays(models, args, g["rays"].to(DEV), g["ts"].to(DEV))
    errs = {k: maxnorm_rel(res[k + "_coarse"].cpu(), g["out_" + k + "_coarse"]) for k in ("rgb", "depth", "weights", "beta")}
    print("bf16 mode", errs)
    _gate(errs, BF16_RENDER_GATES, "bf16 render_rays")


def test_batched_inference_ragged_chunks():
    _, rendering, _ = _lazy()
    g = load_golden("batched_losses")
    args = O.default_args(chunk=100, sc_lambda=0.05, mlp_mode="bf16x3")
    models = build_models(args)
    with rendering.replay_rng([d.to(DEV) for d in golden_draws(g)]):
        res = rendering.batched_inference(models, g["rays"].to(DEV), g["ts"].to(DEV), args)
    for k in ("rgb_coarse", "depth_coarse", "weights_coarse", "sun_sc_coarse"):
        assert tuple(res[k].shape) == tuple(g["out_" + k].shape)
        assert maxnorm_rel(res[k].cpu(), g["out_" + k]) < 1e-4, k


def test_large_batch_properties_full_size():
    """BASELINE config 2 size (1024 x 64): size-independent properties + a sampled oracle comparison."""
    _, rendering, _ = _lazy()
    args = O.default_args(mlp_mode="bf16x3")
    models = build_models(args)
    rays, ts = O.synthetic_rays(1024)
    torch.manual_seed(1)
    with torch.no_grad():
        res = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    w, t = res["weights_coarse"], res["transparency_coarse"]
    assert w.shape == (1024, 64) and torch.isfinite(w).all()
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-4).all()
    assert (t[:, 1:] <= t[:, :-1] + 1e-6).all() and (t[:, 0] == 1).all()
    assert (res["rgb_coarse"] >= 0).all() and (res["rgb_coarse"] <= 1).all()
    assert (res["sky_coarse"][:, 0] == res["sky_coarse"][:, 63]).all()
    # chunking is transparent: rendering two halves with the same draws gives the same rays
    u, nz = torch.rand(1024, 64, device=DEV), torch.randn(1024, 64, device=DEV)
    with torch.no_grad(), rendering.replay_rng([u, nz]):
        full = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    with torch.no_grad(), rendering.replay_rng([u[:500], nz[:500], u[500:], nz[500:]]):
        a = rendering.render_rays(models, args, rays[:500].to(DEV), ts[:500].to(DEV))
        b = rendering.render_rays(models, args, rays[500:].to(DEV), ts[500:].to(DEV))
    for k in ("rgb_coarse", "depth_coarse", "weights_coarse"):
        assert torch.equal(full[k], torch.cat([a[k], b[k]], 0)), k
    # sampled oracle comparison at the headline size: 48 rays spread over the batch, same draws, parity mode <= 1e-4
    pick = torch.arange(0, 1024, 21)[:48]
    mo = make_models(args)
    want = O.render_rays(mo, args, rays[pick], ts[pick], O.ReplayRng([u[pick].cpu(), nz[pick].cpu()]))
    errs = {k: maxnorm_rel(full[k][pick.to(DEV)].cpu(), want[k]) for k in ("rgb_coarse", "depth_coarse", "weights_coarse", "beta_coarse", "sun_coarse")}
    print("bf16x3 @1024x64", {k: f"{e:.1e}" for k, e in errs.items()})
    assert max(errs.values()) < 1e-4, errs
    # the throughput arithmetic (single-pass bf16) on the same batch: reported, and bounded at the level bf16 operands allow
    fast = O.default_args(mlp_mode="bf16")
    with torch.no_grad(), rendering.replay_rng([u, nz]):
        fres = rendering.render_rays(build_models(fast), fast, rays.to(DEV), ts.to(DEV))
    ferrs = {k: maxnorm_rel(fres[k][pick.to(DEV)].cpu(), want[k]) for k in ("rgb_coarse", "depth_coarse", "weights_coarse")}
    print("bf16   @1024x64", {k: f"{e:.1e}" for k, e in ferrs.items()})
    _gate(ferrs, BF16_RENDER_GATES, "bf16 @1024x64")


def test_width_512_golden_through_the_layer_path():
    """fc_units=512, tau=16 (opt.py's default width; run_all.sh trains Sat-NeRF with it): the reference's own render_rays
    output, through the layer-by-layer MFMA path (satnerf_amd.generic, csrc/linear.hip)."""
    _, rendering, _ = _lazy()
    g = load_golden("satnerf_feat512")
    args = golden_cfg(g)
    assert args.fc_units == 512
    models = build_models(args)
    assert not models["coarse"].fused
    with torch.no_grad(), rendering.replay_rng([d.to(DEV) for d in golden_draws(g)]):
        res = rendering.render_rays(models, args, g["rays"].to(DEV), g["ts"].to(DEV))
    expected = {k[4:]: v for k, v in g.items() if k.startswith("out_")}
    assert set(res) == set(expected)
    worst = {k: maxnorm_rel(res[k].cpu(), v) for k, v in expected.items()}
    print({k: f"{e:.1e}" for k, e in worst.items()})
    assert max(worst.values()) < 1e-4, worst
    # the reference-signature forward on points agrees too, and the image-output path accepts the model
    m = models["coarse"]
    n, s = g["rays"].shape[0], args.n_samples
    out = m(torch.rand(130, 3, device=DEV), input_sun_dir=torch.rand(130, 3, device=DEV), input_t=torch.rand(130, 16, device=DEV))
    assert out.shape == (130, 9) and torch.isfinite(out).all()
    with rendering.replay_rng([d.to(DEV) for d in golden_draws(g)]):
        img = rendering.render_image_outputs(models, g["rays"].to(DEV), g["ts"].to(DEV), args)
    assert maxnorm_rel(img["rgb"].cpu(), expected["rgb_coarse"]) < 1e-4 and maxnorm_rel(img["depth"].cpu(), expected["depth_coarse"]) < 1e-4


@pytest.mark.parametrize("k1,k2,n_out,act1,out_act,div2", [(3, 0, 64, None, None, 1), (200, 3, 136, "sin", None, 1), (96, 16, 1, None, "softplus", 8),
                                                           (130, 0, 3, "sin", "sigmoid_rgb", 1), (64, 5, 70, "relu", "sigmoid", 4)])
def test_linear_layer_kernels_vs_torch(k1, k2, n_out, act1, out_act, div2):
    """sr_linear_fwd / bwd_input / bwd_weight against fp64 torch on ragged sizes, two sources, per-ray rows, every activation."""
    ops, _, _ = _lazy()
    g = torch.Generator().manual_seed(k1 * 7 + n_out)
    p = 8 * 37  # not a multiple of the 128-row tile
    x1 = torch.randn(p, k1, generator=g)
    x2 = torch.randn(p // div2, k2, generator=g) if k2 else None
    w = torch.randn(n_out, k1 + k2, generator=g) / (k1 + k2) ** 0.5
    b = torch.randn(n_out, generator=g)
    gy = torch.randn(p, n_out, generator=g)
    w0 = 1.7

    def ref(x1, x2, w, b):
        a = torch.sin(w0 * x1) if act1 == "sin" else torch.relu(x1) if act1 == "relu" else x1
        if x2 is not None:
            a = torch.cat([a, torch.repeat_interleave(x2, div2, 0)], 1)
        pre = a @ w.T + b
        return {None: pre, "softplus": torch.nn.functional.softplus(pre), "sigmoid": torch.sigmoid(pre),
                "sigmoid_rgb": torch.sigmoid(pre) * 1.002 - 0.001}[out_act]

    d = [t.double().requires_grad_(True) if t is not None else None for t in (x1, x2, w, b)]
    y_ref = ref(*d)
    y_ref.backward(gy.double())
    dev = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    srcs = [(dev(x1), act1, w0, 1)] + ([(dev(x2), None, 1.0, div2)] if k2 else [])
    y = ops.linear_fwd(srcs, dev(w), dev(b), p, out_act)
    assert maxnorm_rel(y.cpu().double(), y_ref.detach()) < 2e-5  # bf16 hi/lo 3-pass drops lo*lo (~2^-17)
    yy = y if out_act else None
    d1 = ops.linear_bwd_input(dev(gy), yy, out_act, dev(w), 0, srcs[0], p)
    assert maxnorm_rel(d1.cpu().double(), d[0].grad) < 5e-5
    if k2:
        d2 = ops.linear_bwd_input(dev(gy), yy, out_act, dev(w), k1, srcs[1], p)
        assert maxnorm_rel(d2.view(p // div2, div2, k2).sum(1).cpu().double(), d[1].grad) < 5e-5
    dw, db = ops.linear_bwd_weight(dev(gy), yy, out_act, srcs, p, n_out)
    assert maxnorm_rel(dw.cpu().double(), d[2].grad) < 5e-5 and maxnorm_rel(db.cpu().double(), d[3].grad) < 5e-5


def _oracle_vs_hip(args, n_rays, seed, tol, sample=None, check=("rgb", "depth", "weights", "beta")):
    """Fresh random weights (not the golden ones), same draws on both sides; `sample` rays are compared against the oracle."""
    _, rendering, load_model = _lazy()
    tau = args.t_embbeding_tau
    models = {}
    params = {}
    for typ, sd in (("coarse", 11),) + ((("fine", 12),) if args.n_importance > 0 else ()):
        params[typ] = O.procedural_satnerf_params(args.fc_units, tau, seed=sd)
        m = load_model(args)
        m.load_state_dict(params[typ])
        models[typ] = m.to(DEV).eval()
    embw = O.procedural_uniform((args.t_embbeding_vocab, tau), 1.0, 13)
    emb = torch.nn.Embedding(args.t_embbeding_vocab, tau)
    emb.load_state_dict({"weight": embw})
    models["t"] = emb.to(DEV)
    rays, ts = O.synthetic_rays(n_rays, seed=seed)
    g = torch.Generator().manual_seed(seed)
    s, i = args.n_samples, args.n_importance
    draws = [torch.rand(n_rays, s, generator=g), torch.randn(n_rays, s, generator=g)]
    if i > 0:
        draws += [torch.rand(n_rays, i, generator=g), torch.randn(n_rays, s + i, generator=g)]
    with torch.no_grad(), rendering.replay_rng([d.to(DEV) for d in draws]):
        res = rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))
    torch.cuda.synchronize()
    idx = torch.arange(n_rays) if sample is None else torch.linspace(0, n_rays - 1, sample).long()
    oargs = O.default_args(**{k: v for k, v in vars(args).items() if k != "mlp_mode"})
    with torch.no_grad():
        want = O.render_rays({**params, "t": embw}, oargs, rays[idx], ts[idx], O.ReplayRng([d[idx] for d in draws]))
    errs = {}
    for typ in ("coarse",) + (("fine",) if i > 0 else ()):
        for k in check:
            errs[f"{k}_{typ}"] = maxnorm_rel(res[f"{k}_{typ}"].cpu()[idx], want[f"{k}_{typ}"])
    print({k: f"{e:.1e}" for k, e in errs.items()})
    assert max(errs.values()) < tol, errs
    return res


def test_tau16_two_aux_ksteps():
    """class-default t_embedding_dims=16 (models/satnerf.py:82) needs two aux k-steps in the weight stream."""
    _oracle_vs_hip(O.default_args(t_embbeding_tau=16, mlp_mode="bf16x3"), 77, 31, 1e-4)
    _oracle_vs_hip(O.default_args(t_embbeding_tau=16, mlp_mode="bf16"), 77, 31, 5e-3)


def test_config3_4096_rays_with_importance_sampling():
    """BASELINE configs[2]: 4096 rays x 64 samples + N_importance=64 (two models); 48 sampled rays against the oracle."""
    res = _oracle_vs_hip(O.default_args(n_importance=64, mlp_mode="bf16x3"), 4096, 32, 1e-4, sample=48, check=("rgb", "depth", "weights"))
    assert res["weights_fine"].shape == (4096, 128) and torch.isfinite(res["rgb_fine"]).all()
    t = res["transparency_fine"]
    assert (t[:, 1:] <= t[:, :-1] + 1e-6).all()


def test_config5_8192_rays_128_samples_chunked():
    """BASELINE configs[4] shape through batched_inference (chunk 5120 -> ragged second chunk); 32 sampled rays vs the oracle."""
    _, rendering, load_model = _lazy()
    args = O.default_args(n_samples=128, mlp_mode="bf16x3")
    params = O.procedural_satnerf_params(256, 4, seed=11)
    m = load_model(args)
    m.load_state_dict(params)
    embw = O.procedural_uniform((30, 4), 1.0, 13)
    emb = torch.nn.Embedding(30, 4)
    emb.load_state_dict({"weight": embw})
    models = {"coarse": m.to(DEV).eval(), "t": emb.to(DEV)}
    rays, ts = O.synthetic_rays(8192, seed=33)
    g = torch.Generator().manual_seed(33)
    u, nz = torch.rand(8192, 128, generator=g), torch.randn(8192, 128, generator=g)
    draws = [u[:5120], nz[:5120], u[5120:], nz[5120:]]  # per-chunk draw order
    with rendering.replay_rng([d.to(DEV) for d in draws]):
        res = rendering.batched_inference(models, rays.to(DEV), ts.to(DEV), args)
    assert res["rgb_coarse"].shape == (8192, 3) and res["weights_coarse"].shape == (8192, 128)
    idx = torch.linspace(0, 8191, 32).long()
    with torch.no_grad():
        want = O.render_rays({"coarse": params, "t": embw}, O.default_args(n_samples=128), rays[idx], ts[idx], O.ReplayRng([u[idx], nz[idx]]))
    for k in ("rgb_coarse", "depth_coarse", "weights_coarse"):
        assert maxnorm_rel(res[k].cpu()[idx], want[k]) < 1e-4, k


def test_odd_inputs_non_contiguous_rays_int32_ts_and_tiny_batches():
    _, rendering, load_model = _lazy()
    args = O.default_args(mlp_mode="bf16x3")
    models = build_models(args)
    rays, ts = O.synthetic_rays(5, seed=40)
    wide = torch.zeros(5, 16)
    wide[:, 2:13] = rays
    u, nz = torch.rand(5, 64), torch.randn(5, 64)
    with torch.no_grad():
        want = O.render_rays({"coarse": O.procedural_satnerf_params(256, 4, seed=1), "t": O.procedural_uniform((30, 4), 1.0, 7)}, O.default_args(),
                             rays, ts, O.ReplayRng([u, nz]))
        with rendering.replay_rng([u.to(DEV), nz.to(DEV)]):
            got = rendering.render_rays(models, args, wide.to(DEV)[:, 2:13], ts.to(DEV).int())  # strided view, int32 indices
        with rendering.replay_rng([u[:1].to(DEV), nz[:1].to(DEV)]):
            one = rendering.render_rays(models, args, rays[:1].to(DEV), ts[:1].to(DEV))  # a single ray: 64 points < one workgroup
    assert maxnorm_rel(got["rgb_coarse"].cpu(), want["rgb_coarse"]) < 1e-4
    assert maxnorm_rel(got["weights_coarse"].cpu(), want["weights_coarse"]) < 1e-4
    assert maxnorm_rel(one["weights_coarse"].cpu(), want["weights_coarse"][:1]) < 1e-4


def test_graphed_renderer_matches_eager_and_keeps_the_rng_stream():
    _, rendering, _ = _lazy()
    args = O.default_args(mlp_mode="bf16x3", n_importance=32)
    models = build_models(args)
    rays, ts = O.synthetic_rays(300, seed=50)
    rays, ts = rays.to(DEV), ts.to(DEV)
    gr = rendering.GraphedRenderer(models, args, 300, DEV)
    torch.manual_seed(3)
    a1 = {k: v.clone() for k, v in gr(rays, ts).items()}   # capture + first replay
    a2 = {k: v.clone() for k, v in gr(rays, ts).items()}   # second replay draws fresh jitter
    torch.manual_seed(3)
    with torch.no_grad():
        rendering.render_rays(models, args, rays, ts)       # stands in for the warm-up + capture passes' draws
    # statistically identical renders, different jitter
    assert set(a1) == set(a2) and a1["rgb_fine"].shape == (300, 3)
    assert not torch.equal(a1["weights_coarse"], a2["weights_coarse"])
    assert (a1["rgb_fine"] - a2["rgb_fine"]).abs().max() < 0.2
    assert torch.isfinite(a2["depth_fine"]).all()
    # batched_inference with use_graph: same keys / shapes as eager, ragged tail handled
    args2 = O.default_args(mlp_mode="bf16x3", chunk=128, use_graph=True)
    m2 = build_models(args2)
    out = rendering.batched_inference(m2, rays, ts, args2)
    assert out["rgb_coarse"].shape == (300, 3) and out["weights_coarse"].shape == (300, 64)
    assert torch.isfinite(out["rgb_coarse"]).all()


def test_graphed_renderer_kernel_rng_draws_fresh_uniform_jitter_and_matches_oracle():
    """GraphedRenderer(kernel_rng=True): no RNG launches in the graph -- the sampling kernel draws the stratified jitter itself
    (Philox keyed by the seed, stepping with a device counter the launch advances).  The jitter is recovered from the returned
    depths' weights path via ops.ray_setup on the same counter value: in [0,1), uniform, different per replay, reproducible per
    (seed, step); and the render equals the oracle's on those draws at the parity tolerance."""
    from satnerf_amd import ops, rendering

    args = O.default_args(mlp_mode="bf16x3")
    models = build_models(args)
    n = 512
    rays, ts = O.synthetic_rays(n, seed=71)
    rays_d, ts_d = rays.to(DEV), ts.to(DEV)
    gr = rendering.GraphedRenderer(models, args, n, DEV, kernel_rng=True, seed=99)
    outs = [{k: v.clone() for k, v in gr(rays_d, ts_d).items()} for _ in range(3)]
    step = int(gr._krng[1][0].item())
    assert step == 3 and int(gr._krng[1][3].item()) == 0  # 3 replays ticked it (the warm-up's tick is rolled back); arrival counter back at 0
    assert not torch.equal(outs[0]["weights_coarse"], outs[1]["weights_coarse"])
    # the draws of the LAST replay (counter value step - 1), reproduced by a stand-alone launch on the same (seed, step)
    sk = models["coarse"].sky_color
    ctr = torch.tensor([step - 1, 0, 0, 0], dtype=torch.float32, device=DEV)
    z, _ = ops.ray_setup(rays_d, None, 64, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data, seed=99, step_counter=ctr)
    z0, _ = ops.ray_setup(rays_d, torch.zeros(n, 64, device=DEV), 64, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
    z1, _ = ops.ray_setup(rays_d, torch.ones(n, 64, device=DEV), 64, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
    span = (z1 - z0)
    u = ((z - z0) / span.clamp_min(1e-12)).cpu()
    ok = (span > 1e-6).cpu()
    assert (u[ok] >= -1e-4).all() and (u[ok] < 1 + 1e-4).all()
    assert abs(float(u[ok].mean()) - 0.5) < 0.01 and abs(float(u[ok].var()) - 1 / 12) < 0.01
    assert int(ctr[0].item()) == step - 1  # tick=False: a plain launch leaves the counter alone
    # the render of that replay against the oracle on the recovered draws
    u = u.clamp(0, 1 - 1e-7).float()
    params = {k: v.detach().cpu() for k, v in models["coarse"].state_dict().items()}
    want = O.render_rays({"coarse": params, "t": models["t"].weight.detach().cpu()}, args, rays, ts, O.ReplayRng([u, torch.zeros(n, 64)]))
    got = outs[2]
    # u is recovered through a division, so depths match to ~1e-6 of the span rather than bit-exactly; the bar stays 1e-4
    for k in ("rgb_coarse", "depth_coarse", "weights_coarse"):
        assert maxnorm_rel(got[k].cpu(), want[k]) < 1e-4, k


@pytest.mark.parametrize("mode,feat,s,n_imp,sc,noise", [("bf16x3", 256, 64, 0, 0.0, 0.0), ("bf16", 256, 64, 64, 0.0, 0.3), ("f16", 256, 64, 64, 0.1, 0.0), ("bf16", 256, 128, 0, 0.1, 0.0),
                                                        ("bf16x3", 256, 32, 32, 0.0, 0.0), ("bf16", 512, 64, 0, 0.0, 0.0), ("bf16", 256, 16, 0, 0.0, 0.0)])
def test_one_launch_render_is_bit_identical_to_the_three_launches(mode, feat, s, n_imp, sc, noise):
    """sr_satnerf_render_fwd (sampling -> MLP -> sky head + compositing in ONE launch, per-point values handed over in LDS)
    against sr_ray_setup -> sr_satnerf_mlp_fwd -> sr_composite_fwd on the same draws: every result tensor identical, ragged ray
    counts (the last workgroup is partly empty), coarse + fine, solar correction, noise."""
    _, rendering, _ = _lazy()
    from satnerf_amd import ops

    tau = 16 if feat == 512 else 4
    args = O.default_args(mlp_mode=mode, fc_units=feat, t_embbeding_tau=tau, n_samples=s, n_importance=n_imp, sc_lambda=sc, noise_std=noise)
    models = build_models(args)
    assert ops.render_fused_ok(feat, mode, s)
    for n in (1, 37, 300):
        rays, ts = O.synthetic_rays(n, seed=80 + n)
        rays, ts = rays.to(DEV), ts.to(DEV)
        g = torch.Generator().manual_seed(5)
        draws = [torch.rand(n, s, generator=g), torch.randn(n, s, generator=g)] + ([torch.randn(n, s, generator=g)] if sc > 0 else [])
        if n_imp > 0:
            draws += [torch.rand(n, n_imp, generator=g), torch.randn(n, s + n_imp, generator=g)] + ([torch.randn(n, s + n_imp, generator=g)] if sc > 0 else [])
        draws = [d.to(DEV) for d in draws]
        with torch.no_grad(), rendering.replay_rng(draws), rendering.fused_render(True):
            one = rendering.render_rays(models, args, rays, ts)
        with torch.no_grad(), rendering.replay_rng(draws), rendering.fused_render(False):
            three = rendering.render_rays(models, args, rays, ts)
        assert set(one) == set(three)
        for k in one:
            assert one[k].shape == three[k].shape, k
            assert torch.equal(one[k], three[k]), (k, n, float((one[k] - three[k]).abs().max()))
    assert not ops.render_fused_ok(feat, mode, 50) and not ops.render_fused_ok(384, mode, 64)


@pytest.mark.parametrize("name", ["satnerf_coarse", "satnerf_sc", "satnerf_fine", "satnerf_noise", "satnerf_s128"])
def test_f16_mode_matches_reference_goldens_eight_times_closer_than_bf16(name):
    """mlp_mode='f16': the single-pass kernel with fp16 instead of bf16 MFMA operands (11 instead of 8 significand bits on the
    weights and the activations, same MFMA rate, fp32 accumulation, fp32 first layer).  Against the reference's own outputs:
    rgb / depth / weights within 3e-4 (measured <= 1.6e-4; the bf16 mode: ~1.1e-3), every result within 6e-4 (measured <= 3.6e-4)."""
    _, rendering, _ = _lazy()
    g = load_golden(name)
    args = golden_cfg(g)
    args.mlp_mode = "f16"
    models = build_models(args)
    draws = [d.to(DEV) for d in golden_draws(g)]
    with torch.no_grad(), rendering.replay_rng(draws):
        res = rendering.render_rays(models, args, g["rays"].to(DEV), g["ts"].to(DEV))
    expected = {k[4:]: v for k, v in g.items() if k.startswith("out_")}
    assert set(res) == set(expected)
    errs = {k: maxnorm_rel(res[k].cpu(), v) for k, v in expected.items()}
    print(name, "f16", {k: f"{e:.1e}" for k, e in errs.items()})
    typ = "fine" if args.n_importance > 0 else "coarse"
    assert max(errs[f"{k}_{typ}"] for k in ("rgb", "depth", "weights")) < 3e-4, errs
    assert max(errs.values()) < 6e-4, errs


def test_bank_walking_renderer_renders_chunk_after_chunk():
    """GraphedRenderer(bank=...): a replay is ONE kernel that takes its chunk of the resident ray bank from the device counter
    (sr_render_args.bank_chunks) and advances it.  (i) With explicit jitter the bank launch for chunk k equals the plain launch
    on rows k*n..(k+1)*n bit for bit; (ii) graph replays walk the chunks in order, wrap around, draw fresh jitter per visit and
    equal an eager launch with the counter set by hand."""
    from satnerf_amd import ops, rendering

    args = O.default_args(mlp_mode="bf16x3")
    models = build_models(args)
    m = models["coarse"]
    n, chunks = 96, 3
    rays, ts = O.synthetic_rays(n * chunks + 17, seed=91)  # 17 rows too many: the tail is not a full chunk and is never visited
    rays, ts = rays.to(DEV), ts.to(DEV)
    hi, lo, l0 = m.packed("bf16x3")
    sk = [m.sky_color[0].weight.data, m.sky_color[0].bias.data, m.sky_color[2].weight.data, m.sky_color[2].bias.data]
    emb = models["t"].weight.data
    u = torch.rand(n, 64, device=DEV)
    for k in (0, 2, 4):
        ctr = torch.tensor([k, 0, 0, 0], dtype=torch.float32, device=DEV)
        a = ops.render_fwd(rays[:n * chunks], ts[:n * chunks], emb, 64, 256, 4, "bf16x3", hi, lo, l0, *sk, u=u, step_counter=ctr, bank_chunks=chunks)
        c = k % chunks
        b = ops.render_fwd(rays[c * n:(c + 1) * n], ts[c * n:(c + 1) * n], emb, 64, 256, 4, "bf16x3", hi, lo, l0, *sk, u=u)
        for key in a:
            assert torch.equal(a[key], b[key]), (key, k)
        assert int(ctr[0].item()) == k  # no tick asked for
    gr = rendering.GraphedRenderer(models, args, n, DEV, seed=5, bank=(rays, ts))
    seen = []
    for _ in range(5):
        out = {k: v.clone() for k, v in gr.replay().items()}
        step = gr._launches - 1
        assert gr.last_chunk == step % chunks and int(gr._krng[1][0].item()) == step + 1
        ctr = torch.tensor([step, 0, 0, 0], dtype=torch.float32, device=DEV)
        want = ops.render_fwd(gr.rays, gr.ts, emb, 64, 256, 4, "bf16x3", hi, lo, l0, *sk, seed=5, step_counter=ctr, bank_chunks=chunks, want_z=False)
        assert torch.equal(out["rgb_coarse"], want["rgb"]) and torch.equal(out["weights_coarse"], want["weights"])
        seen.append((gr.last_chunk, out["depth_coarse"]))
    assert [c for c, _ in seen] == [0, 1, 2, 0, 1]  # the warm-up launch before the capture does not consume a chunk
    assert not torch.equal(seen[0][1], seen[3][1])  # same chunk, another step: fresh jitter
    with pytest.raises(RuntimeError):
        gr(rays[:n], ts[:n])
    # replay_chunks: 7 more chunks as 4 (one launch) + 3 single launches, every result equal to the hand-set-counter launch
    got = []
    for outs in gr.replay_chunks(7, group=4):
        got += [{k: v.clone() for k, v in o.items()} for o in outs]
    assert len(got) == 7 and gr._launches == 12 and int(gr._krng[1][0].item()) == 12
    for i, o in enumerate(got):
        ctr = torch.tensor([5 + i, 0, 0, 0], dtype=torch.float32, device=DEV)
        want = ops.render_fwd(gr.rays, gr.ts, emb, 64, 256, 4, "bf16x3", hi, lo, l0, *sk, seed=5, step_counter=ctr, bank_chunks=chunks, want_z=False)
        assert torch.equal(o["rgb_coarse"], want["rgb"]) and torch.equal(o["depth_coarse"], want["depth"]), i


def test_public_inference_signature_matches_render_rays():
    """rendering.inference(model, args, rays_xyz, z_vals, rays_d, sun_d, rays_t) -- the reference's models.satnerf.inference
    signature with explicit points and embedding vectors -- agrees with the fused ray path on the same depths and noise."""
    from satnerf_amd import ops, rendering

    g = load_golden("satnerf_noise")
    args = golden_cfg(g)
    args.mlp_mode = "bf16x3"
    models = build_models(args)
    rays, ts = g["rays"].to(DEV), g["ts"].to(DEV)
    draws = [d.to(DEV) for d in golden_draws(g)]
    with torch.no_grad(), rendering.replay_rng(draws):
        want = rendering.render_rays(models, args, rays, ts)
    z = ops.ray_sample(rays, draws[0], args.n_samples)
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    with rendering.replay_rng([draws[1]]):
        got = rendering.inference(models["coarse"], args, xyz, z, rays_d=rays[:, 3:6], sun_d=rays[:, 8:11], rays_t=models["t"](ts))
    assert set(got) == {"rgb", "depth", "weights", "transparency", "albedo", "sun", "sky", "beta"}
    for k in got:
        assert got[k].shape == want[f"{k}_coarse"].shape, k
        assert maxnorm_rel(got[k].cpu(), want[f"{k}_coarse"].cpu()) < 2e-5, k
    with pytest.raises(TypeError):
        rendering.inference(models["coarse"], args, xyz, z)


@pytest.mark.parametrize("name", ["satnerf_fine", "satnerf_noise", "satnerf_sc", "satnerf_s50_ragged"])
def test_render_image_outputs_match_reductions_of_the_reference_results(name):
    """render_image_outputs == save_nerf_output_to_images' reductions (eval_satnerf.py:106-146) of the reference's own
    render_rays outputs (golden), on the same draws -- coarse and coarse+fine, noise, the solar-correction draw order,
    and a chunk size that leaves a ragged last chunk."""
    from satnerf_amd import rendering

    g = load_golden(name)
    args = golden_cfg(g)
    args.mlp_mode = "bf16x3"
    models = build_models(args)
    typ = "fine" if args.n_importance > 0 else "coarse"
    want = O.image_outputs({k[4:]: v for k, v in g.items() if k.startswith("out_")}, typ)
    with rendering.replay_rng([d.to(DEV) for d in golden_draws(g)]):
        got = rendering.render_image_outputs(models, g["rays"].to(DEV), g["ts"].to(DEV), args)
    assert got["typ"] == typ
    for k, v in want.items():
        assert tuple(got[k].shape) == tuple(v.shape), (k, got[k].shape, v.shape)
        assert maxnorm_rel(got[k].cpu(), v) < 1e-4, (k, maxnorm_rel(got[k].cpu(), v))


def test_latlonalt_kernel_matches_reference_golden():
    import numpy as np

    from satnerf_amd import rendering

    g = load_golden("latlonalt")
    lat, lon, alt = rendering.latlonalt_from_depth(g["rays"].to(DEV), g["depth"].to(DEV), np.asarray(g["center"]), float(g["range"]))
    assert lat.dtype == torch.float64 and lat.shape == (200,)
    assert np.abs(lat.cpu().numpy() - np.asarray(g["lats"])).max() < 1e-11   # degrees (1e-11 deg ~ 1 micrometre)
    assert np.abs(lon.cpu().numpy() - np.asarray(g["lons"])).max() < 1e-11
    assert np.abs(alt.cpu().numpy() - np.asarray(g["alts"])).max() < 1e-7    # metres
    # a strided view of wider rows and a ragged size
    lat2, _, alt2 = rendering.latlonalt_from_depth(g["rays"].to(DEV)[:77], g["depth"].to(DEV)[:77], np.asarray(g["center"]), float(g["range"]))
    assert torch.equal(lat2, lat[:77]) and torch.equal(alt2, alt[:77])


def test_classic_nerf_golden_coarse_and_fine():
    """BASELINE configs[0] (classic nerf, (N,8) rays, coarse + fine with importance sampling): the reference's render_rays
    output through the layer-by-layer HIP path."""
    _, rendering, load_model = _lazy()
    g = load_golden("nerf_coarse_fine")
    args = golden_cfg(g)
    assert args.model == "nerf"
    models = {}
    for typ, seed in (("coarse", 1), ("fine", 2)):
        m = load_model(args)
        m.load_state_dict(O.procedural_nerf_params(args.fc_units, seed=seed))
        models[typ] = m.to(DEV).eval()
    with torch.no_grad(), rendering.replay_rng([d.to(DEV) for d in golden_draws(g)]):
        res = rendering.render_rays(models, args, g["rays"].to(DEV), None)
    expected = {k[4:]: v for k, v in g.items() if k.startswith("out_")}
    assert set(res) == set(expected)
    worst = {k: maxnorm_rel(res[k].cpu(), v) for k, v in expected.items()}
    print({k: f"{e:.1e}" for k, e in worst.items()})
    assert max(worst.values()) < 1e-4, worst
    out = models["coarse"](torch.rand(70, 3, device=DEV) * 4, input_dir=torch.rand(70, 3, device=DEV))
    assert out.shape == (70, 4) and torch.isfinite(out).all()
    assert models["coarse"](torch.rand(70, 3, device=DEV), sigma_only=True).shape == (70, 1)


def test_edge_cases_empty_and_ragged_inputs_of_the_newer_entry_points():
    """Zero rays, sample counts that are not a multiple of the 64-lane wave, one-ray batches and bad arguments for the entry
    points added after the first parity round (image outputs, lat/lon/alt, layer path, wgrad plan)."""
    import numpy as np

    ops, rendering, load_model = _lazy()
    # zero-size calls return empty results without launching
    z0 = torch.empty(0, 50, device=DEV)
    img = ops.composite_image(z0, z0, None, 0.0, torch.empty(0, 50, 3, device=DEV), z0, z0, torch.empty(0, 3, device=DEV))
    assert img.shape == (0, 13)
    lat, lon, alt = ops.latlonalt_from_depth(torch.empty(0, 11, device=DEV), torch.empty(0, device=DEV), np.zeros(3), 1.0)
    assert lat.shape == (0,) and lat.dtype == torch.float64
    assert ops.positional_map(torch.empty(0, 3, device=DEV), 10).shape == (0, 60)
    # S = 50 (ragged wave) image outputs equal the reductions of the full outputs; a single ray works
    args = O.default_args(n_samples=50, mlp_mode="bf16x3")
    models = build_models(args)
    rays, ts = O.synthetic_rays(33, seed=71)
    rays, ts = rays.to(DEV), ts.to(DEV)
    g = torch.Generator().manual_seed(72)
    draws = [torch.rand(33, 50, generator=g).to(DEV), torch.randn(33, 50, generator=g).to(DEV)]
    with torch.no_grad(), rendering.replay_rng(draws):
        full = rendering.render_rays(models, args, rays, ts)
    with rendering.replay_rng(draws):
        img = rendering.render_image_outputs(models, rays, ts, args)
    want = O.image_outputs({k: v.cpu() for k, v in full.items()}, "coarse")
    for k, v in want.items():
        assert maxnorm_rel(img[k].cpu(), v) < 1e-5, k
    with rendering.replay_rng([d[:1] for d in draws]):
        one = rendering.render_image_outputs(models, rays[:1], ts[:1], args)
    assert maxnorm_rel(one["rgb"].cpu(), img["rgb"][:1].cpu()) < 1e-6
    # argument errors are Python exceptions, not launches
    with pytest.raises(ValueError):
        ops.linear_fwd([(torch.zeros(4, 8, device=DEV), "tanh", 1.0, 1)], torch.zeros(3, 8, device=DEV), None, 4)  # unknown activation
    with pytest.raises((ValueError, KeyError)):
        ops.linear_fwd([(torch.zeros(4, 8, device=DEV), None, 1.0, 1)], torch.zeros(3, 9, device=DEV), None, 4)    # weight / source mismatch
    with pytest.raises(ValueError):
        ops.latlonalt_from_depth(rays, torch.zeros(5, device=DEV), np.zeros(3), 1.0)                               # depth count
    with pytest.raises(RuntimeError):
        rendering.render_image_outputs(models, rays.cpu(), ts.cpu(), args)                                         # no CPU path
    with pytest.raises(ValueError):
        load_model(O.default_args(model="bogus"))
    with pytest.raises(NotImplementedError):  # s-nerf has no working fine branch in the reference either (rendering.py:133)
        rendering.render_rays({"coarse": load_model(O.default_args(model="s-nerf")).to(DEV)}, O.default_args(model="s-nerf", n_importance=4), rays, None)


def test_width_512_fused_forward_kernel():
    """fc_units=512 (opt.py:50; what run_all.sh trains sat-nerf with), no-grad forward in the throughput arithmetic: the 512-wide
    build of the fused kernel (csrc/mlp_fwd512_*.hip) against the reference's own render_rays output (bounded like the 256 kernel
    in bf16) and against the layer-by-layer path, which reproduces that golden to 1e-4."""
    _, rendering, _ = _lazy()
    g = load_golden("satnerf_feat512")
    args = golden_cfg(g)
    args.mlp_mode = "bf16"
    models = build_models(args)
    m = models["coarse"]
    assert not m.fused and m.fused_forward("bf16") and m.fused_forward("f16") and not m.fused_forward("bf16x3")
    draws = [d.to(DEV) for d in golden_draws(g)]
    with torch.no_grad(), rendering.replay_rng(draws):
        res = rendering.render_rays(models, args, g["rays"].to(DEV), g["ts"].to(DEV))
    expected = {k[4:]: v for k, v in g.items() if k.startswith("out_")}
    assert set(res) == set(expected)
    errs = {k: maxnorm_rel(res[k].cpu(), v) for k, v in expected.items()}
    print("feat 512 fused bf16", {k: f"{e:.1e}" for k, e in errs.items()})
    assert max(errs[k] for k in ("rgb_coarse", "depth_coarse", "weights_coarse")) < 3e-3, errs  # measured 6.6e-4 / 3.9e-4 / 1.0e-3
    assert max(errs.values()) < 5e-3, errs                                                       # beta 2.4e-3, albedo 2.0e-3
    # the fp16-operand build of the same kernel: an order of magnitude closer
    args.mlp_mode = "f16"
    with torch.no_grad(), rendering.replay_rng(draws):
        res16 = rendering.render_rays(models, args, g["rays"].to(DEV), g["ts"].to(DEV))
    errs16 = {k: maxnorm_rel(res16[k].cpu(), v) for k, v in expected.items()}
    print("feat 512 fused f16", {k: f"{e:.1e}" for k, e in errs16.items()})
    assert max(errs16[k] for k in ("rgb_coarse", "depth_coarse", "weights_coarse")) < 3e-4 and max(errs16.values()) < 6e-4, errs16
    args.mlp_mode = "bf16"
    # per-point outputs through the reference-signature forward: fused bf16 vs the layer path (3-pass) on random points
    x, sun, t = torch.rand(777, 3, device=DEV) * 2 - 1, torch.nn.functional.normalize(torch.randn(777, 3, device=DEV), dim=1), torch.rand(777, 16, device=DEV)
    with torch.no_grad():
        fast = m(x, input_sun_dir=sun, input_t=t, mlp_mode="bf16")
        slow = m(x, input_sun_dir=sun, input_t=t, mlp_mode="bf16x3")
    assert fast.shape == slow.shape == (777, 9)
    assert maxnorm_rel(fast.cpu(), slow.cpu()) < 2e-2
    # the parity arithmetic has no 512-wide fused build (one bf16 plane fills the 512 registers), and the C ABI says so;
    # 16-bit workspaces do not exist at this width either
    from satnerf_amd import _lib, ops
    hi, lo, l0 = m.packed("bf16x3")
    with pytest.raises(_lib.SatRenderError):
        ops.satnerf_mlp(x, None, sun, None, t, None, 777, 1, 512, 16, "bf16x3", hi, lo, l0)
    with pytest.raises(ValueError):
        ops.acts_workspace(777, 512, x.device, 16)
    assert ops.acts_workspace(777, 512, x.device, 8).numel() > 0
