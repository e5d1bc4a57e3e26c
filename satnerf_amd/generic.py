"""Layer-by-layer Sat-NeRF for the widths / depths the fused kernel does not cover (``fc_units != 256``; opt.py:50 defaults to
512, which is what run_all.sh trains Sat-NeRF with).

Every ``nn.Linear`` of ``SatNeRF.forward`` (models/satnerf.py:156-208) is one launch of the tiled MFMA GEMM in
``csrc/linear.hip``: the previous layer's Siren is applied while its pre-activation is loaded, concatenations
(``[xyz | h]`` at the skip, ``[feats | sun_d]``, ``[feats | t]``) are two sources of one launch, per-ray inputs are read with
``row_div = n_samples`` instead of being ``repeat_interleave``d.  Each layer is a ``torch.autograd.Function`` whose backward
runs the input-gradient and weight-gradient GEMMs, so training works through autograd; compositing and the sky head have
their own Functions over the same kernels the fused path uses.  Same arithmetic class as the fused parity mode (bf16 hi/lo
3-pass, ~1e-6), several times slower than a fused kernel would be (every activation makes an HBM round trip).
"""
from __future__ import annotations

import torch

from . import ops


class _LinearFn(torch.autograd.Function):
    """y = out_act(Linear([act1(x1) | act2(x2)]));  cfg = (act1, w01, div1, act2, w02, div2, out_act, n_points)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, cfg):
        act1, w01, div1, act2, w02, div2, out_act, n_points = cfg
        srcs = [(x1, act1, w01, div1)] + ([(x2, act2, w02, div2)] if x2 is not None else [])
        y = ops.linear_fwd(srcs, weight, bias, n_points, out_act)
        ctx.cfg = cfg
        ctx.save_for_backward(x1, x2, weight, y if out_act else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x1, x2, weight, y = ctx.saved_tensors
        act1, w01, div1, act2, w02, div2, out_act, n_points = ctx.cfg
        gy = gy.contiguous().float()
        srcs = [(x1, act1, w01, div1)] + ([(x2, act2, w02, div2)] if x2 is not None else [])
        d1 = d2 = None
        if ctx.needs_input_grad[0]:
            d1 = _reduce_rows(ops.linear_bwd_input(gy, y, out_act, weight, 0, srcs[0], n_points), div1, x1)
        if x2 is not None and ctx.needs_input_grad[1]:
            d2 = _reduce_rows(ops.linear_bwd_input(gy, y, out_act, weight, x1.shape[1], srcs[1], n_points), div2, x2)
        dw, db = ops.linear_bwd_weight(gy, y, out_act, srcs, n_points, weight.shape[0])
        return d1, d2, dw, db, None


def _reduce_rows(d, div, x):
    """Per-point gradient of a per-ray source (row_div > 1): sum over the ray's samples (torch.repeat_interleave backward)."""
    if div == 1:
        return d
    n = d.shape[0] // div
    out = d.view(n, div, d.shape[1]).sum(1)
    if out.shape[0] != x.shape[0]:  # the source may carry more rows than this pass read
        out = torch.cat([out, out.new_zeros(x.shape[0] - out.shape[0], out.shape[1])], 0)
    return out


class _SkyFn(torch.autograd.Function):
    """sky_color head (models/satnerf.py:138-143) per ray: sr_sky_fwd / sr_sky_bwd."""

    @staticmethod
    def forward(ctx, sun, w1, b1, w2, b2):
        sky = ops.sky(sun, w1, b1, w2, b2)
        ctx.save_for_backward(sun, w1, b1, w2, sky)
        return sky

    @staticmethod
    def backward(ctx, g):
        sun, w1, b1, w2, sky = ctx.saved_tensors
        gw1, gb1, gw2, gb2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros(3, device=w1.device)
        ops.sky_bwd(sun, w1, b1, w2, sky, g.contiguous().float(), gw1, gb1, gw2, gb2)
        return None, gw1, gb1, gw2, gb2


class _CompositeFn(torch.autograd.Function):
    """Compositing (models/satnerf.py:52-70): sr_composite_fwd / sr_composite_bwd."""

    @staticmethod
    def forward(ctx, z, sigma, noise, noise_std, albedo, sun_v, sky, clamp_rgb=True):
        weights, transparency, depth, rgb = ops.composite(z, sigma, noise, noise_std, albedo, sun_v, sky, clamp_rgb=clamp_rgb)
        ctx.noise_std, ctx.clamp_rgb = noise_std, clamp_rgb
        ctx.save_for_backward(z, sigma, noise, albedo, sun_v, sky, weights, transparency)
        return rgb, depth, weights, transparency

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_w, g_t):
        z, sigma, noise, albedo, sun_v, sky, weights, transparency = ctx.saved_tensors
        c = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        d_sigma, d_albedo, d_sun, d_sky = ops.composite_bwd(z, sigma, noise, ctx.noise_std, albedo, sun_v, sky, weights, transparency, c(g_rgb),
                                                           c(g_depth), c(g_w), c(g_t), clamp_rgb=ctx.clamp_rgb)
        if sun_v is None:
            d_sun = d_sky = None
        return None, d_sigma, None, None, d_albedo, d_sun, d_sky, None


def satnerf_points(model, xyz, sun_rows, t_rows, row_div):
    """``SatNeRF.forward`` (models/satnerf.py:156-208) for P points: xyz (P,3) per point, sun_rows (R,3) / t_rows (R,tau) per ray
    (point p reads row p // row_div).  Returns albedo (P,3), sigma (P,), sun_v (P,), beta (P,) -- differentiable w.r.t. the
    model parameters and t_rows."""
    p = xyz.shape[0]

    def lin(x1, layer, act1=None, w01=1.0, x2=None, act2=None, w02=1.0, div1=1, div2=1, out=None):
        return _LinearFn.apply(x1, x2, layer.weight, layer.bias, (act1, float(w01), div1, act2, float(w02), div2, out, p))

    fc = model.fc_net
    pre = lin(xyz, fc[0])
    w0 = float(getattr(fc[1], "w0", 30.0))
    for i in range(1, model.layers):
        if i in model.skips:  # torch.cat([input_xyz, xyz_], -1): the raw coordinates come first (models/satnerf.py:177)
            pre = lin(xyz, fc[2 * i], x2=pre, act2="sin", w02=w0)
        else:
            pre = lin(pre, fc[2 * i], act1="sin", w01=w0)
        w0 = float(getattr(fc[2 * i + 1], "w0", 1.0))
    sigma = lin(pre, model.sigma_from_xyz[0], act1="sin", w01=w0, out="softplus")
    feats = lin(pre, model.feats_from_xyz, act1="sin", w01=w0)  # no non-linearity on the features (models/satnerf.py:118)
    rgb = lin(lin(feats, model.rgb_from_xyzdir[0]), model.rgb_from_xyzdir[2], act1="sin", out="sigmoid_rgb")
    sv = model.sun_v_net
    s = lin(feats, sv[0], x2=sun_rows, div2=row_div)
    s = lin(s, sv[2], act1="sin", w01=float(getattr(sv[1], "w0", 1.0)))
    s = lin(s, sv[4], act1="sin")
    sun_v = lin(s, sv[6], act1="sin", out="sigmoid")
    b = lin(feats, model.beta_from_xyz[0], x2=t_rows, div2=row_div)
    beta = lin(b, model.beta_from_xyz[2], act1="sin", out="softplus")
    return rgb, sigma.view(-1), sun_v.view(-1), beta.view(-1)


def nerf_points(model, xyz, dir_rows, row_div, sigma_only=False):
    """Classic ``NeRF.forward`` (models/nerf.py:184-227) for P points: positional maps, ReLU trunk with the encoded-xyz skip,
    colour head on [feats | map(dir)].  Returns rgb (P,3) (None if sigma_only), sigma (P,)."""
    p = xyz.shape[0]

    def lin(x1, layer, act1=None, x2=None, act2=None, div2=1, out=None):
        return _LinearFn.apply(x1, x2, layer.weight, layer.bias, (act1, 1.0, 1, act2, 1.0, div2, out, p))

    e = ops.positional_map(xyz, model.mapping_sizes[0])
    fc = model.fc_net
    pre = lin(e, fc[0])
    for i in range(1, model.layers):
        pre = lin(e, fc[2 * i], x2=pre, act2="relu") if i in model.skips else lin(pre, fc[2 * i], act1="relu")
    sigma = lin(pre, model.sigma_from_xyz[0], act1="relu", out="softplus").view(-1)
    if sigma_only:
        return None, sigma
    feats = lin(pre, model.feats_from_xyz, act1="relu")
    h = lin(feats, model.rgb_from_xyzdir[0], x2=ops.positional_map(dir_rows, model.mapping_sizes[1]), div2=row_div)
    return lin(h, model.rgb_from_xyzdir[2], act1="relu", out="sigmoid_rgb"), sigma


def nerf_inference_pass(model, args, rays, z, noise):
    """Classic ``models.nerf.inference`` (models/nerf.py:71-133): no sun / sky / clamp; rays are (N,8)."""
    n, s = z.shape
    rgbs, sigma = nerf_points(model, ops.points_along(rays, 3, z), rays[:, 3:6], s)
    noise_std = float(args.noise_std)
    rgb, depth, weights, transparency = _CompositeFn.apply(z, sigma.view(n, s), noise if noise_std != 0 else None, noise_std, rgbs.view(n, s, 3), None,
                                                          None, False)
    return {"rgb": rgb, "depth": depth, "weights": weights, "transparency": transparency}


def inference_pass(model, args, rays, z, ts, emb, dir_cols, noise):
    """One ``models.satnerf.inference`` pass (models/satnerf.py:4-79) through the layer-by-layer path: same dict as the fused
    ``rendering._inference``; differentiable when grad is enabled.  ``emb`` = nn.Embedding (training) or its weight tensor."""
    n, s = z.shape
    xyz = ops.points_along(rays, dir_cols[0], z)
    t_rows = emb(ts) if callable(emb) else emb[ts]
    sun = rays[:, 8:11]
    albedo, sigma, sun_v, beta = satnerf_points(model, xyz, sun, t_rows.contiguous().float(), s)
    sk = model.sky_color
    sky = _SkyFn.apply(sun, sk[0].weight, sk[0].bias, sk[2].weight, sk[2].bias)
    albedo, sigma, sun_v = albedo.view(n, s, 3), sigma.view(n, s), sun_v.view(n, s)
    noise_std = float(args.noise_std)
    rgb, depth, weights, transparency = _CompositeFn.apply(z, sigma, noise if noise_std != 0 else None, noise_std, albedo, sun_v, sky)
    return {"rgb": rgb, "depth": depth, "weights": weights, "transparency": transparency, "albedo": albedo, "sun": sun_v.unsqueeze(-1),
            "sky": sky.unsqueeze(1).expand(n, s, 3), "beta": beta.view(n, s, 1)}
