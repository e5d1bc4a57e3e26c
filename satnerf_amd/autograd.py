"""Training path of ``render_rays``: the same HIP kernels as the no-grad path plus their hand-written backward.

The reference lets autograd differentiate ~120 ATen ops per chunk (main.py:119-154 -> rendering.py -> models/satnerf.py);
here ONE ``torch.autograd.Function`` per inference pass wraps [fused MLP forward (saving activations) -> sky head ->
compositing] and its backward runs [compositing backward -> fused dX chain -> weight-gradient GEMMs -> sky / embedding
backward].  Gradients required: every model parameter and the embedding rows; none w.r.t. rays, depths or
xyz (``z_vals_`` from sample_pdf is detached in the reference, rendering.py:122-124).

Parameter gradients are ACCUMULATED IN PLACE into the model's flat gradient buffer (``SatNeRF.flat_grads()``, of which
every ``p.grad`` is a view) and into ``embedding.weight.grad`` -- one buffer for the optimizer and the data-parallel
all-reduce instead of 38 AccumulateGrad nodes.  The Function is attached to the graph through a 1-element trigger leaf.
"""
from __future__ import annotations

import torch

from . import ops



class _InferenceFn(torch.autograd.Function):
    """One models/satnerf.inference pass (MLP + compositing) with a hand-written backward."""

    @staticmethod
    def forward(ctx, trigger, model, emb_module, args, rays, z, ts, dir_cols, noise, mode):
        n, s = z.shape
        feat, tau = model.feat, model.t_embedding_dims
        hi, lo, l0 = model.packed(mode)
        emb = emb_module.weight.data
        from .train import _fmt_of

        fmt = _fmt_of(args)
        acts = ops.acts_workspace(n * s, feat, rays.device, fmt)
        albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, dir_cols[0]:dir_cols[1]], rays[:, 8:11], z, emb, ts, n * s, s, feat, tau,
                                                     mode, hi, lo, l0, acts=acts, fmt=fmt)
        sk = model.sky_color
        sky = ops.sky(rays[:, 8:11], sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
        noise_std = float(args.noise_std)
        use_noise = noise_std != 0
        weights, transparency, depth, rgb = ops.composite(z, sigma.view(n, s), noise if use_noise else None, noise_std, albedo.view(n, s, 3),
                                                         sun_v.view(n, s), sky)
        ctx.model, ctx.emb_module, ctx.noise_std, ctx.shape, ctx.fmt = model, emb_module, noise_std, (n, s), fmt
        ctx.save_for_backward(rays, z, ts, noise if use_noise else None, acts, albedo, sigma, sun_v, beta, sky, weights, transparency)
        return rgb, depth, weights, transparency, albedo.view(n, s, 3), sun_v.view(n, s, 1), sky, beta.view(n, s, 1)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_weights, g_transp, g_albedo, g_sun, g_sky, g_beta):
        rays, z, ts, noise, acts, albedo, sigma, sun_v, beta, sky, weights, transparency = ctx.saved_tensors
        model, emb_module, (n, s) = ctx.model, ctx.emb_module, ctx.shape
        feat, tau = model.feat, model.t_embedding_dims
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        d_sigma, d_albedo, d_sun, d_sky = ops.composite_bwd(z, sigma.view(n, s), noise, ctx.noise_std, albedo.view(n, s, 3), sun_v.view(n, s), sky,
                                                           weights, transparency, c(g_rgb), c(g_depth), c(g_weights), c(g_transp))
        # direct gradients on the per-sample outputs add to the ones routed through compositing
        if g_albedo is not None:
            d_albedo = d_albedo + g_albedo
        if g_sun is not None:
            d_sun = d_sun + g_sun.reshape(n, s)
        if g_sky is not None:
            d_sky = d_sky + g_sky
        g_beta_pt = None if g_beta is None else g_beta.reshape(n * s).contiguous().float()
        bstream, maps = model.packed_backward()
        dpre, d_t = ops.satnerf_mlp_bwd(feat, tau, n * s, bstream, acts, albedo, sigma, sun_v, beta, d_albedo.contiguous(), d_sigma, d_sun.contiguous(),
                                        g_beta_pt, fmt=ctx.fmt)
        grad_flat = model.flat_grads()
        ops.satnerf_wgrad(feat, tau, n * s, dpre, acts, maps["blocks"], maps["gidx"], maps["gscale"], grad_flat, accumulate=True, fmt=ctx.fmt,
                          loads=maps["loads8"])
        sk = model.sky_color
        ops.sky_bwd(rays[:, 8:11], sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sky, d_sky.contiguous(), sk[0].weight.grad, sk[0].bias.grad,
                    sk[2].weight.grad, sk[2].bias.grad)
        w = emb_module.weight
        if w.grad is None:
            w.grad = torch.zeros_like(w.data)
        ops.embedding_bwd(d_t, ts, n, s, tau, w.grad)
        return (None,) * 10


def render_rays_train(models, args, rays, ts, rng):
    """``render_rays`` with grad (the NeRF_pl.forward path, main.py:60-75): same draws, same dict, differentiable.
    The backward's saved state follows ``train._fmt_of(args)``: 8-bit workspaces by default in the 'bf16' / 'f16' modes (gradients
    ~2e-2 of the reference's), ``args.bwd_fmt = 16`` for the 16-bit state (~1e-2 / 7e-3), 'bf16x3' + ``bwd_fmt = 32`` for 2e-4."""
    from .rendering import _mode_of

    n_samples, n_importance = args.n_samples, args.n_importance
    rays = rays.contiguous().float()
    ts = ts.contiguous().long().view(-1)
    n, dev = rays.shape[0], rays.device
    emb_module = models["t"]
    if not hasattr(emb_module, "weight"):
        raise TypeError("training needs models['t'] to be an nn.Embedding")
    from .train import _fmt_of

    mode = _mode_of(args)
    z = ops.ray_sample(rays, rng.rand(n, n_samples, dev), n_samples)
    result = {}

    def run(typ, z_cur):
        model = models[typ]
        model.flat_grads()  # make sure every p.grad aliases the flat buffer before backward writes into it
        if not hasattr(model, "_trigger") or model._trigger.device != dev:
            model._trigger = torch.zeros(1, device=dev, requires_grad=True)
        noise = rng.randn(n, z_cur.shape[1], dev)
        keys = ("rgb", "depth", "weights", "transparency", "albedo", "sun", "sky", "beta")
        if not model.fused_training(mode, _fmt_of(args)):  # layer-by-layer path: every Linear is its own autograd Function (satnerf_amd.generic)
            from .generic import inference_pass

            res = inference_pass(model, args, rays, z_cur, ts, emb_module, (3, 6), noise)
            if args.sc_lambda > 0:
                sc = inference_pass(model, args, rays, z_cur, ts, emb_module, (8, 11), rng.randn(n, z_cur.shape[1], dev))
                res["weights_sc"], res["transparency_sc"], res["sun_sc"] = sc["weights"], sc["transparency"], sc["sun"]
            for k, v in res.items():
                result[f"{k}_{typ}"] = v
            return
        out = _InferenceFn.apply(model._trigger, model, emb_module, args, rays, z_cur, ts, (3, 6), noise, mode)
        res = dict(zip(keys, out))
        res["sky"] = res["sky"].unsqueeze(1).expand(n, z_cur.shape[1], 3)
        if args.sc_lambda > 0:
            noise_sc = rng.randn(n, z_cur.shape[1], dev)
            sc = dict(zip(keys, _InferenceFn.apply(model._trigger, model, emb_module, args, rays, z_cur, ts, (8, 11), noise_sc, mode)))
            res["weights_sc"], res["transparency_sc"], res["sun_sc"] = sc["weights"], sc["transparency"], sc["sun"]
        for k, v in res.items():
            result[f"{k}_{typ}"] = v

    run("coarse", z)
    if n_importance > 0:
        u = rng.rand(n, n_importance, dev)
        z_fine = ops.sample_pdf_merge(z, result["weights_coarse"].detach(), u)  # detached: rendering.py:122-124
        run("fine", z_fine)
    return result
