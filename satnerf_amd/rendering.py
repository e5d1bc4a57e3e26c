"""Drop-in for the reference's ``rendering.py`` + ``eval_satnerf.batched_inference`` on MI355X.

Same call signatures, argument meaning, result-dict keys/shapes and error behaviour as

    rendering.render_rays(models, args, rays, ts)              rendering.py:52-158
    rendering.sample_pdf(bins, weights, N_importance, det, eps) rendering.py:10-49
    eval_satnerf.batched_inference(models, rays, ts, args)      eval_satnerf.py:46-66

but every stage runs in hand-written HIP kernels behind the C ABI (include/satrender.h); this module is
orchestration only and has NO PyTorch/CPU fallback.

Random draws: the reference draws from torch's global generator in a fixed order per call
(rand_like (N,S) -> randn (N,S) [-> randn (N,S) if sc] [-> rand (N,I) -> randn (N,S+I) [-> randn]]);
``render_rays`` draws the same tensors in the same order on the rays' device, so seeding reproduces the
reference's stream on that device.  Tests inject captured draws with ``replay_rng``.
"""
from __future__ import annotations

import contextlib
import os
from collections import defaultdict

import torch

from . import ops

_MODE = os.environ.get("SATNERF_AMD_MODE", "bf16x3")


def set_default_mode(mode: str) -> None:
    """'bf16x3' (parity mode, ~2e-6 vs fp32), 'bf16' (single-pass throughput mode, ~1e-3) or 'f16' (single-pass fp16 operands:
    the throughput of 'bf16' at ~1.5e-4; width 256)."""
    global _MODE
    if mode not in ops.MODES:
        raise ValueError(f"mode must be one of {sorted(ops.MODES)}")
    _MODE = mode


def default_mode() -> str:
    return _MODE


# ------------------------------------------------------------------------------------------------ RNG hook
class _TorchRng:
    def rand(self, n, m, device):
        return torch.rand(n, m, device=device)

    def randn(self, n, m, device):
        return torch.randn(n, m, device=device)


class _ReplayRng:
    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def _next(self, n, m, device):
        d = self.draws[self.i]
        self.i += 1
        if tuple(d.shape) != (n, m):
            raise ValueError(f"replayed draw {self.i - 1} has shape {tuple(d.shape)}, expected {(n, m)}")
        return d.to(device=device, dtype=torch.float32).contiguous()

    rand = randn = _next


class _KernelRng(_TorchRng):
    """The stratified jitter of the fused path is drawn INSIDE the sampling kernel (Philox-4x32-10 keyed by ``seed``, counter =
    (ray, sample, step); ``counter`` = a 4-float device block whose [0] the launch advances): the same distribution as
    ``torch.rand`` but not torch's stream, so a captured forward needs no RNG launch and none of the generator-state
    bookkeeping torch adds to a graph.  The ``randn`` of models/satnerf.py:58 is skipped while ``noise_std == 0`` (it is
    multiplied by zero there); every other draw still comes from torch."""

    def __init__(self, seed, counter, bank_chunks=0):
        self.seed, self.counter, self.bank_chunks = int(seed), counter, int(bank_chunks)


_rng = _TorchRng()
_FUSED_RENDER = os.environ.get("SATNERF_FUSED_RENDER", "1") != "0"


@contextlib.contextmanager
def fused_render(enabled):
    """Switch the one-launch render pass (sr_satnerf_render_fwd) on / off; off = sampling, MLP and compositing as separate
    launches (the same arithmetic bit for bit -- tests compare the two)."""
    global _FUSED_RENDER
    prev, _FUSED_RENDER = _FUSED_RENDER, bool(enabled)
    try:
        yield
    finally:
        _FUSED_RENDER = prev


@contextlib.contextmanager
def kernel_rng(seed, counter, bank_chunks=0):
    """Draw ``render_rays``' stratified jitter inside the kernel (see ``_KernelRng``); ``counter`` = zeros(4) float32 on the GPU.
    ``bank_chunks`` > 0: the rays / ts passed to ``render_rays`` are a bank of that many equal chunks and each call renders the
    chunk the device counter points at (``GraphedRenderer(bank=...)``)."""
    global _rng
    prev, _rng = _rng, _KernelRng(seed, counter, bank_chunks)
    try:
        yield _rng
    finally:
        _rng = prev


@contextlib.contextmanager
def replay_rng(draws):
    """Feed ``render_rays`` a recorded list of random tensors instead of fresh draws (parity tests)."""
    global _rng
    prev, _rng = _rng, _ReplayRng(draws)
    try:
        yield _rng
    finally:
        _rng = prev


# ------------------------------------------------------------------------------------------- render_rays
def _mode_of(args):
    return getattr(args, "mlp_mode", None) or _MODE


def _inference(model, args, rays, z, ts, emb_weight, dir_cols, noise, sky=None):
    """models/satnerf.inference (models/satnerf.py:4-79) for the points rays[:, 0:3] + rays[:, dir_cols] * z."""
    n, s = z.shape
    mode = _mode_of(args)
    if not model.fused_forward(mode):  # widths / depths / modes outside the fused kernel: layer by layer (satnerf_amd.generic)
        from .generic import inference_pass

        return inference_pass(model, args, rays, z, ts, emb_weight, dir_cols, noise)
    hi, lo, l0 = model.packed(mode)
    albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, dir_cols[0]:dir_cols[1]], rays[:, 8:11], z, emb_weight, ts, n * s, s,
                                                 model.feat, model.t_embedding_dims, mode, hi, lo, l0)
    if sky is None:
        sk = model.sky_color
        sky = ops.sky(rays[:, 8:11], sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
    sigma, sun_v = sigma.view(n, s), sun_v.view(n, s)
    albedo = albedo.view(n, s, 3)
    use_noise = args.noise_std != 0
    weights, transparency, depth, rgb = ops.composite(z, sigma, noise if use_noise else None, args.noise_std, albedo, sun_v, sky)
    return {"rgb": rgb, "depth": depth, "weights": weights, "transparency": transparency, "albedo": albedo,
            "sun": sun_v.unsqueeze(-1), "sky": sky.unsqueeze(1).expand(n, s, 3), "beta": beta.view(n, s, 1)}


def inference(model, args, rays_xyz, z_vals, rays_d=None, sun_d=None, rays_t=None):
    """``models.satnerf.inference`` (models/satnerf.py:4-79) with its own signature: explicit sample positions ``rays_xyz``
    (N,S,3), depths ``z_vals`` (N,S), per-ray ``sun_d`` (N,3) and per-ray embedding VECTORS ``rays_t`` (N,tau); ``rays_d`` is
    accepted and unused, as in the reference's sat-nerf variant.  Compatibility entry point (no grad): ``render_rays`` does
    not go through it -- it hands rays to the fused kernel, which forms the points itself."""
    if sun_d is None or rays_t is None:
        raise TypeError("sat-nerf inference needs sun_d and rays_t (models/satnerf.py:199,204)")
    if not rays_xyz.is_cuda:
        raise RuntimeError("rays_xyz must be on the GPU: satnerf_amd has no CPU path")
    n, s = rays_xyz.shape[0], rays_xyz.shape[1]
    with torch.no_grad():
        out = model(rays_xyz.reshape(-1, 3), input_sun_dir=torch.repeat_interleave(sun_d, s, dim=0),
                    input_t=torch.repeat_interleave(rays_t, s, dim=0), mlp_mode=_mode_of(args)).view(n, s, 9)
        albedo, sigma, sun_v, sky, beta = out[..., 0:3].contiguous(), out[..., 3].contiguous(), out[..., 4].contiguous(), out[..., 5:8], out[..., 8:9]
        noise = _rng.randn(n, s, rays_xyz.device)  # models/satnerf.py:58 -- always drawn
        weights, transparency, depth, rgb = ops.composite(z_vals.contiguous().float(), sigma, noise if args.noise_std != 0 else None, args.noise_std,
                                                          albedo, sun_v, sky[:, 0].contiguous())
    return {"rgb": rgb, "depth": depth, "weights": weights, "transparency": transparency, "albedo": albedo, "sun": sun_v.unsqueeze(-1),
            "sky": sky, "beta": beta}


def validate_ts(ts, models):
    """``nn.Embedding`` raises IndexError on an out-of-range index (rendering.py:100); the fused kernels index the table
    directly, so the range is checked here with one fused min/max reduction and a device sync -- on EVERY call with a caller-owned
    tensor (a pointer / version key cannot identify a tensor's contents: the caching allocator hands a fresh per-step ``ts`` the
    previous one's address), never inside a hipGraph capture.  Objects that own their indices (``RayBank``, a ``GraphedRenderer``
    bank) are checked once by their owners.  Returns ``ts``."""
    emb = models.get("t") if isinstance(models, dict) else None
    if ts is None or emb is None or torch.cuda.is_current_stream_capturing():
        return ts
    vocab = (emb.weight if hasattr(emb, "weight") else emb).shape[0]
    if ts.numel():
        lo, hi = (int(v) for v in torch.aminmax(ts))
        if lo < 0 or hi >= vocab:
            raise IndexError(f"ts holds image indices in [{lo}, {hi}] but the embedding has {vocab} rows (t_embbeding_vocab)")
    return ts


def render_rays(models, args, rays, ts, _ts_validated=False):
    """Render a chunk of rays: stratified sampling -> fused Sat-NeRF MLP -> compositing [-> fine pass]."""
    n_samples, n_importance, variant = args.n_samples, args.n_importance, args.model
    if variant == "nerf":
        return _render_rays_nerf(models, args, rays)
    if variant == "s-nerf":
        return _render_rays_snerf(models, args, rays)
    if variant != "sat-nerf":
        raise ValueError(f"model {variant} is not valid")
    if ts is None:
        raise TypeError("sat-nerf needs per-ray image indices ts (rendering.py:100 would fail in torch.cat)")
    if not rays.is_cuda:
        raise RuntimeError("rays must be on the GPU: satnerf_amd has no CPU path")
    if not _ts_validated:
        validate_ts(ts, models)
    if torch.is_grad_enabled() and any(p.requires_grad for p in models["coarse"].parameters()):
        from .autograd import render_rays_train

        return render_rays_train(models, args, rays, ts, _rng)
    rays = rays.contiguous().float()
    ts = ts.contiguous().long().view(-1)
    n, dev = rays.shape[0], rays.device
    emb = models["t"].weight.data if hasattr(models["t"], "weight") else models["t"]
    emb = emb.contiguous().float()

    coarse = models["coarse"]
    mode = _mode_of(args)
    kernel = isinstance(_rng, _KernelRng)
    skip_noise = kernel and args.noise_std == 0
    use_noise = args.noise_std != 0
    result = {}

    def fused_ok(model, s):
        return _FUSED_RENDER and hasattr(model, "fused_forward") and model.fused_forward(mode) and ops.render_fused_ok(model.feat, mode, s)

    def sc_pass(typ, z_cur, res):  # solar correction: same depths along the sun direction (rendering.py:102-108)
        noise_sc = None if skip_noise else _rng.randn(n, z_cur.shape[1], dev)
        sc = _inference(models[typ], args, rays, z_cur, ts, emb, (8, 11), noise_sc)
        res["weights_sc"], res["transparency_sc"], res["sun_sc"] = sc["weights"], sc["transparency"], sc["sun"]

    def run_fused(typ, z_cur, u_cur):
        """sampling (coarse) / given depths (fine) -> MLP -> sky head + compositing in ONE launch (sr_satnerf_render_fwd)."""
        model = models[typ]
        s = n_samples if z_cur is None else z_cur.shape[1]
        noise = None if skip_noise else _rng.randn(n, s, dev)  # models/satnerf.py:58 -- always drawn
        hi, lo, l0 = model.packed(mode)
        sk = model.sky_color
        need_z = n_importance > 0 or args.sc_lambda > 0
        draw = z_cur is None and u_cur is None
        o = ops.render_fwd(rays, ts, emb, s, model.feat, model.t_embedding_dims, mode, hi, lo, l0, sk[0].weight.data, sk[0].bias.data,
                           sk[2].weight.data, sk[2].bias.data, z=z_cur, u=u_cur, noise=noise if use_noise else None, noise_std=args.noise_std,
                           seed=_rng.seed if draw else 0, step_counter=_rng.counter if draw else None, tick=draw, want_z=need_z,
                           bank_chunks=chunks)
        res = {"rgb": o["rgb"], "depth": o["depth"], "weights": o["weights"], "transparency": o["transparency"], "albedo": o["albedo"],
               "sun": o["sun_v"].unsqueeze(-1), "sky": o["sky"].unsqueeze(1).expand(n, s, 3), "beta": o["beta"].unsqueeze(-1)}
        if args.sc_lambda > 0:
            sc_pass(typ, o["z"], res)
        for k, v in res.items():
            result[f"{k}_{typ}"] = v
        return o["z"]

    def run(typ, z_cur, sky=None):
        noise = None if skip_noise else _rng.randn(n, z_cur.shape[1], dev)  # models/satnerf.py:58 -- always drawn
        res = _inference(models[typ], args, rays, z_cur, ts, emb, (3, 6), noise, sky)
        if args.sc_lambda > 0:
            sc_pass(typ, z_cur, res)
        for k, v in res.items():
            result[f"{k}_{typ}"] = v

    chunks = _rng.bank_chunks if kernel else 0
    if chunks:  # rays / ts = a resident bank; the launch picks its chunk from the device counter
        if not fused_ok(coarse, n_samples) or n_importance > 0 or args.sc_lambda > 0 or use_noise:
            raise NotImplementedError("bank-walking renders need the one-launch coarse pass (no fine model / solar correction / noise)")
        n = n // chunks
    if fused_ok(coarse, n_samples):
        z = run_fused("coarse", None, None if kernel else _rng.rand(n, n_samples, dev))
    elif hasattr(coarse, "fused_forward") and coarse.fused_forward(mode):  # stratified depths (rendering.py:62-78, perturb = 1) + the coarse sky head in one launch
        sk = coarse.sky_color
        if kernel:
            z, sky = ops.ray_setup(rays, None, n_samples, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data,
                                   seed=_rng.seed, step_counter=_rng.counter, tick=True)
        else:
            z, sky = ops.ray_setup(rays, _rng.rand(n, n_samples, dev), n_samples, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data,
                                   sk[2].bias.data)
        run("coarse", z, sky)
    else:
        z = ops.ray_sample(rays, _rng.rand(n, n_samples, dev), n_samples)
        run("coarse", z)
    if n_importance > 0:  # rendering.py:118-156
        u = _rng.rand(n, n_importance, dev)
        z_fine = ops.sample_pdf_merge(z, result["weights_coarse"], u)
        if fused_ok(models["fine"], z_fine.shape[1]):
            run_fused("fine", z_fine, None)
        else:
            run("fine", z_fine)
    return result


def _as_satnerf(models, args):
    """s-nerf on the Sat-NeRF kernels: (models with the 1-row zero embedding, args with model = 'sat-nerf')."""
    import copy

    if args.n_importance > 0:
        raise NotImplementedError("s-nerf with n_importance > 0 fails in the reference itself (rendering.py:133 reads an undefined name)")
    coarse = models["coarse"]
    if not hasattr(coarse, "dummy_embedding"):
        raise TypeError("args.model == 's-nerf' needs models['coarse'] = satnerf_amd.models.ShadowNeRF (load_model)")
    args2 = copy.copy(args)
    args2.model = "sat-nerf"
    return {"coarse": coarse, "t": coarse.dummy_embedding()}, args2


def _render_rays_snerf(models, args, rays):
    """``render_rays`` for s-nerf (rendering.py:85-96, models/snerf.py:4-75): ``ts`` is unused, the result has no ``beta``."""
    models2, args2 = _as_satnerf(models, args)
    ts0 = torch.zeros(rays.shape[0], dtype=torch.int64, device=rays.device)
    res = render_rays(models2, args2, rays, ts0, _ts_validated=True)
    return {k: v for k, v in res.items() if not k.startswith("beta_")}


def _render_rays_nerf(models, args, rays):
    """``render_rays`` for the classic nerf (rendering.py:126-128,141-153; BASELINE configs[0]): rays are (N,8), no ts; every
    layer runs through the MFMA GEMM of the layer-by-layer path and is differentiable when grad is enabled."""
    from .generic import nerf_inference_pass

    if not rays.is_cuda:
        raise RuntimeError("rays must be on the GPU: satnerf_amd has no CPU path")
    rays = rays.contiguous().float()
    n, dev, s = rays.shape[0], rays.device, args.n_samples
    grad = torch.is_grad_enabled() and any(p.requires_grad for p in models["coarse"].parameters())
    result = {}
    with contextlib.nullcontext() if grad else torch.no_grad():
        z = ops.ray_sample(rays, _rng.rand(n, s, dev), s)
        for k, v in nerf_inference_pass(models["coarse"], args, rays, z, _rng.randn(n, s, dev)).items():
            result[f"{k}_coarse"] = v
        if args.n_importance > 0:
            z_fine = ops.sample_pdf_merge(z, result["weights_coarse"].detach(), _rng.rand(n, args.n_importance, dev))
            for k, v in nerf_inference_pass(models["fine"], args, rays, z_fine, _rng.randn(n, z_fine.shape[1], dev)).items():
                result[f"{k}_fine"] = v
    return result


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5):
    """``rendering.sample_pdf`` (rendering.py:10-49): draw ``N_importance`` depths per ray from the piecewise-constant pdf
    ``weights`` (N, nb-1) over ``bins`` (N, nb).  ``det`` uses u = linspace(0,1,N_importance) instead of a uniform draw.
    (``render_rays`` itself uses the fused resample+merge kernel.)"""
    if not bins.is_cuda:
        raise RuntimeError("bins must be on the GPU: satnerf_amd has no CPU path")
    n = bins.shape[0]
    if det:
        u = torch.linspace(0, 1, N_importance, device=bins.device).expand(n, N_importance).contiguous()
    else:
        u = _rng.rand(n, N_importance, bins.device)
    return ops.sample_pdf(bins.contiguous().float(), weights.contiguous().float(), u, eps)


@torch.no_grad()
def render_image_outputs(models, rays, ts, args):
    """Whole-image evaluation: ``batched_inference`` + the per-pixel reductions of
    ``eval_satnerf.save_nerf_output_to_images`` (eval_satnerf.py:106-146) fused into the compositing kernel.

    Returns {"typ", "rgb" (N,3), "depth" (N,), "acc" (N,), "sun" (N,1), "albedo" (N,3), "beta" (N,1), "sky" (N,3)} for the finest
    model present: rgb/depth are ``results[f"rgb_{typ}"]`` / ``results[f"depth_{typ}"]`` and the others
    ``sum(weights.unsqueeze(-1) * results[key], -2)`` -- 52 B/ray leave the GPU kernels instead of 2,576 B (x2 with a fine model).
    Same random-draw order per chunk as ``render_rays``."""
    if args.model != "sat-nerf":
        raise NotImplementedError(f"model {args.model}: only sat-nerf is built on the HIP path (SURVEY.md section 8)")
    if ts is None:
        raise TypeError("sat-nerf needs per-ray image indices ts")
    if not rays.is_cuda:
        raise RuntimeError("rays must be on the GPU: satnerf_amd has no CPU path")
    n_total, s, n_imp = rays.shape[0], args.n_samples, args.n_importance
    validate_ts(ts, models)
    typ = "fine" if n_imp > 0 else "coarse"
    emb = (models["t"].weight.data if hasattr(models["t"], "weight") else models["t"]).contiguous().float()
    image = torch.empty(n_total, 13, dtype=torch.float32, device=rays.device)
    mode = _mode_of(args)
    for i in range(0, n_total, args.chunk):
        r = rays[i:i + args.chunk].contiguous().float()
        t = ts[i:i + args.chunk].contiguous().long().view(-1)
        n, dev = r.shape[0], r.device
        z = ops.ray_sample(r, _rng.rand(n, s, dev), s)

        def heads(model, z_cur):
            sk = model.sky_color
            sky = ops.sky(r[:, 8:11], sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
            k = z_cur.shape[1]
            if model.fused_forward(mode):
                hi, lo, l0 = model.packed(mode)
                albedo, sigma, sun_v, beta = ops.satnerf_mlp(r[:, 0:3], r[:, 3:6], r[:, 8:11], z_cur, emb, t, n * k, k, model.feat,
                                                             model.t_embedding_dims, mode, hi, lo, l0)
            else:
                from .generic import satnerf_points

                albedo, sigma, sun_v, beta = satnerf_points(model, ops.points_along(r, 3, z_cur), r[:, 8:11], emb[t], k)
            noise = _rng.randn(n, k, dev)  # models/satnerf.py:58 -- always drawn
            return (z_cur, sigma.view(n, k), noise if args.noise_std != 0 else None, args.noise_std, albedo.view(n, k, 3), sun_v.view(n, k)), beta.view(n, k), sky

        comp, beta, sky = heads(models["coarse"], z)
        if args.sc_lambda > 0:
            _rng.randn(n, s, dev)  # the solar-correction pass of render_rays draws here; its outputs are not image outputs
        if n_imp > 0:
            weights = ops.composite(*comp, sky)[0]
            z_fine = ops.sample_pdf_merge(z, weights, _rng.rand(n, n_imp, dev))
            comp, beta, sky = heads(models["fine"], z_fine)
            if args.sc_lambda > 0:
                _rng.randn(n, s + n_imp, dev)  # ... and so does the fine pass's solar-correction pass (draw 6 of SURVEY.md 8c)
        image[i:i + n] = ops.composite_image(*comp, beta, sky)
    out = {k: image[:, a:b] for k, (a, b) in ops.IMAGE_COLUMNS.items()}
    out["depth"], out["acc"] = out["depth"][:, 0], out["acc"][:, 0]
    out["typ"] = typ
    return out


def latlonalt_from_depth(rays, depth, center, scene_range):
    """``SatelliteDataset.get_latlonalt_from_nerf_prediction`` (datasets/satellite.py:246-275) on the GPU in fp64:
    ``center`` (3,) / ``scene_range`` are the dataset's ECEF normalisation (datasets/satellite.py:225-226).  Returns
    (lats, lons, alts) as fp64 device tensors (the reference returns numpy arrays: ``.cpu().numpy()`` them)."""
    if not rays.is_cuda:
        raise RuntimeError("rays must be on the GPU: satnerf_amd has no CPU path")
    return ops.latlonalt_from_depth(rays.float(), depth.to(rays.device), center, scene_range)


def batched_inference_sharded(models, rays, ts, args, render_fn=None):
    """``batched_inference`` of one image over the ranks of an initialised ``torch.distributed`` group ("split an
    image's rays in contiguous row blocks, concatenate"): rank r renders rows ``shard_rays(N, r, W)`` with the ordinary chunk loop
    and every rank receives the full per-key tensors (one all-gather per key; ragged shares are padded to the largest).  Models are
    replicated, rays are independent, so the result equals the single-rank call given the same per-ray draws.  ``render_fn``
    (default ``batched_inference``) exists so the gather logic can be exercised without a GPU."""
    import torch.distributed as dist

    from .train import shard_rays

    world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
    render_fn = render_fn or batched_inference
    n = rays.shape[0]
    lo, hi = shard_rays(n, rank, world)
    local = render_fn(models, rays[lo:hi], None if ts is None else ts[lo:hi], args)
    if world == 1:
        return local
    spans = [shard_rays(n, r, world) for r in range(world)]
    most = max(b - a for a, b in spans)
    out = {}
    for k in sorted(local):  # same key order on every rank
        v = local[k]
        if v is None:
            out[k] = None
            continue
        pad = torch.zeros((most,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[:v.shape[0]] = v
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out[k] = torch.cat([p[:b - a] for p, (a, b) in zip(parts, spans)], 0)
    return out


class GraphedRenderer:
    """``render_rays`` (no grad) for a FIXED chunk shape replayed from one hipGraph.

    A chunk is ~8 launches of 5-400 us; issued eagerly from Python they leave the GPU idle about half the time at 1024 rays
    (profiles/r01_bench_forward.json).  The captured graph contains the same launches (including the RNG draws, which advance
    torch's generator exactly as eager calls would), so results and the random stream are unchanged.  Outputs are views of
    static buffers: they are overwritten by the next call -- clone what must survive.  Weights may change between calls
    (their streams are re-packed before the replay when they did) but not be re-allocated.

    ``kernel_rng=True`` (``seed``): the stratified jitter is drawn inside the sampling kernel instead (``_KernelRng``: same
    distribution, its own counter-based stream) -- the graph then holds no RNG launches and no generator bookkeeping.
    """

    def __init__(self, models, args, n_rays, device, kernel_rng=False, seed=0, bank=None):
        self.models, self.args, self.n = models, args, n_rays
        self._krng = (int(seed), torch.zeros(4, dtype=torch.float32, device=device)) if (kernel_rng or bank is not None) else None
        self._chunks, self._launches = 0, 0
        self.max_inflight = int(os.environ.get("SATNERF_RENDER_MAX_INFLIGHT", "0"))
        if bank is not None:
            # ``bank`` = (rays (M,11), ts (M,)) resident on the GPU: every replay renders the NEXT n_rays rows (wrapping around), as
            # eval_satnerf.batched_inference walks an image chunk by chunk -- the kernel takes its chunk from the device counter,
            # so a step is one graph replay of one kernel: no gather launch, no host work (sr_render_args.bank_chunks)
            b_rays, b_ts = bank
            self._chunks = b_rays.shape[0] // n_rays
            if self._chunks < 1 or not b_rays.is_cuda:
                raise ValueError("bank must hold at least n_rays rows on the GPU")
            self.rays = b_rays[:self._chunks * n_rays].contiguous().float()
            self.ts = b_ts.reshape(-1)[:self._chunks * n_rays].contiguous().long()
            validate_ts(self.ts, models)
            self._rgbs, self.graph, self.out, self._packed_for = None, None, None, {}
            return
        self.rays = torch.zeros(n_rays, 11, device=device)
        self.ts = torch.zeros(n_rays, dtype=torch.int64, device=device)
        self._rgbs = torch.zeros(n_rays, 3, device=device)  # gather target for a ray bank's colours (unused by rendering)
        self.graph, self.out = None, None
        self._packed_for = {}

    def _refresh_weights(self):
        """Weight streams are re-packed (eagerly, into the fixed buffers the graph reads) only when the parameters changed."""
        mode = _mode_of(self.args)
        for typ in ("coarse", "fine"):
            m = self.models.get(typ)
            if m is None or not (hasattr(m, "fused_forward") and m.fused_forward(mode)):
                continue
            stamp = (m.weights_version(), m.flat_params().data_ptr(), mode)
            if self._packed_for.get(typ) != stamp:
                m.repack(mode)
                self._packed_for[typ] = (m.weights_version(), m.flat_params().data_ptr(), mode)

    @property
    def last_chunk(self):
        """Bank mode: index of the chunk the most recent replay rendered (rows last_chunk * n .. + n of the bank)."""
        return (self._launches - 1) % self._chunks if self._chunks else None

    def _run(self):
        if self._krng is not None:
            with kernel_rng(*self._krng, bank_chunks=self._chunks):
                return render_rays(self.models, self.args, self.rays, self.ts, _ts_validated=bool(self._chunks))
        return render_rays(self.models, self.args, self.rays, self.ts)

    @torch.no_grad()
    def replay(self):
        """Render the rays currently in ``self.rays`` / ``self.ts`` (fill them in place, e.g. ``RayBank.next_batch(out=...)``)."""
        self._refresh_weights()
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                keep = self._krng[1].clone() if self._krng is not None else None
                self._run()  # warm-up outside capture: lazy initialisation (LDS attributes, index maps)
                if keep is not None:  # the warm-up ticked the device counter: put it back, so the first replay renders chunk 0 / draws step 0
                    self._krng[1].copy_(keep)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with ops.graph_capture(self.graph):
                self.out = self._run()
        self._pace()
        self.graph.replay()
        self._launches += 1
        return self.out

    def _pace(self):
        """Optional bound on the replays queued ahead of the device (``max_inflight``; 0 = unbounded, the default: the one-kernel
        forward step does not show the slow submission mode of ``train.Trainer._pace`` and an event per step costs it 5 %)."""
        k = self.max_inflight // 8
        if k <= 0 or self._launches % 8:
            return
        ring = self.__dict__.get("_pace_ring")
        if ring is None:
            ring = self._pace_ring = [torch.cuda.Event() for _ in range(k)]
            for ev in ring:
                ev.record()
        ev = ring[(self._launches // 8) % k]
        ev.synchronize()
        ev.record()

    @torch.no_grad()
    def replay_chunks(self, k, group=4):
        """Bank mode: render the next ``k`` chunks, ``group`` consecutive chunks per graph launch (the remainder one by one) --
        the launch gap between replays (~8 us on MI355X against a ~90 us kernel) is paid once per group; this is how a whole
        image is walked.  Generator: yields, per launch, the list of that launch's result dicts (each launch of a size reuses
        that size's static output buffers: consume or clone before advancing)."""
        if not self._chunks:
            raise RuntimeError("replay_chunks needs GraphedRenderer(bank=...)")
        self._refresh_weights()
        if self.graph is None:
            self.replay()
            k -= 1
            yield [self.out]
        multi = self.__dict__.setdefault("_multi", {})
        while k > 0:
            g = group if k >= group else 1
            if g == 1:
                yield [self.replay()]
            else:
                if g not in multi:
                    graph = torch.cuda.CUDAGraph()
                    torch.cuda.synchronize()
                    with ops.graph_capture(graph):
                        outs = [self._run() for _ in range(g)]
                    multi[g] = (graph, outs)
                multi[g][0].replay()
                self._launches += g
                yield multi[g][1]
            k -= g

    def render_next(self, bank):
        """Gather the bank's next batch straight into the static inputs and render it."""
        if self._chunks:
            raise RuntimeError("this renderer walks its own bank: call replay()")
        if bank.batch_size != self.n:
            raise ValueError(f"GraphedRenderer was built for {self.n} rays, the bank serves {bank.batch_size}")
        bank.next_batch(out=(self.rays, self.ts, self._rgbs))
        return self.replay()

    @torch.no_grad()
    def __call__(self, rays, ts):
        if self._chunks:
            raise RuntimeError("this renderer walks its own bank: call replay()")
        if rays.shape[0] != self.n:
            raise ValueError(f"GraphedRenderer was built for {self.n} rays, got {rays.shape[0]}")
        self.rays.copy_(rays)
        self.ts.copy_(ts.view(-1))
        return self.replay()


@torch.no_grad()
def batched_inference(models, rays, ts, args):
    """``eval_satnerf.batched_inference``: ray-chunked no-grad rendering, per-key concatenation.  With ``args.use_graph`` the
    full-size chunks are replayed from a hipGraph (the ragged last chunk runs eagerly)."""
    chunk_size = args.chunk
    results = defaultdict(list)
    graphed = None
    validate_ts(ts, models)  # once for the whole image, not per chunk
    if getattr(args, "use_graph", False) and ts is not None and rays.shape[0] >= 2 * chunk_size and type(_rng) is _TorchRng:
        graphed = GraphedRenderer(models, args, chunk_size, rays.device)
    for i in range(0, rays.shape[0], chunk_size):
        r, t = rays[i:i + chunk_size], ts[i:i + chunk_size] if ts is not None else None
        if graphed is not None and r.shape[0] == chunk_size:
            out = {k: v.clone() for k, v in graphed(r, t).items()}
        else:
            out = render_rays(models, args, r, t, _ts_validated=True)
        for k, v in out.items():
            results[k] += [v]
    for k, v in results.items():
        results[k] = None if v[0] is None else torch.cat(v, 0)
    return results
