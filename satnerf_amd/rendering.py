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
    """'bf16x3' (parity mode, ~2e-6 vs fp32) or 'bf16' (single-pass throughput mode, ~1e-3)."""
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


_rng = _TorchRng()


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


def _inference(model, args, rays, z, ts, emb_weight, dir_cols, noise):
    """models/satnerf.inference (models/satnerf.py:4-79) for the points rays[:, 0:3] + rays[:, dir_cols] * z."""
    n, s = z.shape
    mode = _mode_of(args)
    hi, lo, l0 = model.packed(mode)
    albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, dir_cols[0]:dir_cols[1]], rays[:, 8:11], z, emb_weight, ts, n * s, s,
                                                 model.feat, model.t_embedding_dims, mode, hi, lo, l0)
    sk = model.sky_color
    sky = ops.sky(rays[:, 8:11], sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
    sigma, sun_v = sigma.view(n, s), sun_v.view(n, s)
    albedo = albedo.view(n, s, 3)
    use_noise = args.noise_std != 0
    weights, transparency, depth, rgb = ops.composite(z, sigma, noise if use_noise else None, args.noise_std, albedo, sun_v, sky)
    return {"rgb": rgb, "depth": depth, "weights": weights, "transparency": transparency, "albedo": albedo,
            "sun": sun_v.unsqueeze(-1), "sky": sky.unsqueeze(1).expand(n, s, 3), "beta": beta.view(n, s, 1)}


def render_rays(models, args, rays, ts):
    """Render a chunk of rays: stratified sampling -> fused Sat-NeRF MLP -> compositing [-> fine pass]."""
    n_samples, n_importance, variant = args.n_samples, args.n_importance, args.model
    if variant != "sat-nerf":
        raise NotImplementedError(f"model {variant}: only sat-nerf is built on the HIP path (SURVEY.md section 8)")
    if ts is None:
        raise TypeError("sat-nerf needs per-ray image indices ts (rendering.py:100 would fail in torch.cat)")
    if not rays.is_cuda:
        raise RuntimeError("rays must be on the GPU: satnerf_amd has no CPU path")
    if torch.is_grad_enabled() and any(p.requires_grad for p in models["coarse"].parameters()):
        from .autograd import render_rays_train

        return render_rays_train(models, args, rays, ts, _rng)
    rays = rays.contiguous().float()
    ts = ts.contiguous().long().view(-1)
    n, dev = rays.shape[0], rays.device
    emb = models["t"].weight.data if hasattr(models["t"], "weight") else models["t"]
    emb = emb.contiguous().float()

    z = ops.ray_sample(rays, _rng.rand(n, n_samples, dev), n_samples)  # rendering.py:62-78 (perturb = 1)
    result = {}

    def run(typ, z_cur):
        noise = _rng.randn(n, z_cur.shape[1], dev)  # models/satnerf.py:58 -- always drawn
        res = _inference(models[typ], args, rays, z_cur, ts, emb, (3, 6), noise)
        if args.sc_lambda > 0:  # solar correction: same depths along the sun direction (rendering.py:102-108)
            noise_sc = _rng.randn(n, z_cur.shape[1], dev)
            sc = _inference(models[typ], args, rays, z_cur, ts, emb, (8, 11), noise_sc)
            res["weights_sc"], res["transparency_sc"], res["sun_sc"] = sc["weights"], sc["transparency"], sc["sun"]
        for k, v in res.items():
            result[f"{k}_{typ}"] = v

    run("coarse", z)
    if n_importance > 0:  # rendering.py:118-156
        u = _rng.rand(n, n_importance, dev)
        z_fine = ops.sample_pdf_merge(z, result["weights_coarse"], u)
        run("fine", z_fine)
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


class GraphedRenderer:
    """``render_rays`` (no grad) for a FIXED chunk shape replayed from one hipGraph.

    A chunk is ~8 launches of 5-400 us; issued eagerly from Python they leave the GPU idle about half the time at 1024 rays
    (profiles/r01_bench_forward.json).  The captured graph contains the same launches (including the RNG draws, which advance
    torch's generator exactly as eager calls would), so results and the random stream are unchanged.  Outputs are views of
    static buffers: they are overwritten by the next call -- clone what must survive.  Weights may change between calls
    (the pack kernels are part of the graph) but not be re-allocated.
    """

    def __init__(self, models, args, n_rays, device):
        self.models, self.args, self.n = models, args, n_rays
        self.rays = torch.zeros(n_rays, 11, device=device)
        self.ts = torch.zeros(n_rays, dtype=torch.int64, device=device)
        self.graph, self.out = None, None

    def _run(self):
        mode = _mode_of(self.args)
        for typ in ("coarse", "fine"):
            if typ in self.models:
                self.models[typ].repack(mode)  # unconditional pack into fixed buffers: safe to capture
        return render_rays(self.models, self.args, self.rays, self.ts)

    @torch.no_grad()
    def __call__(self, rays, ts):
        if rays.shape[0] != self.n:
            raise ValueError(f"GraphedRenderer was built for {self.n} rays, got {rays.shape[0]}")
        self.rays.copy_(rays)
        self.ts.copy_(ts.view(-1))
        if self.graph is None:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._run()  # warm-up outside capture: lazy initialisation (LDS attributes, index maps)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self._run()
        self.graph.replay()
        return self.out


@torch.no_grad()
def batched_inference(models, rays, ts, args):
    """``eval_satnerf.batched_inference``: ray-chunked no-grad rendering, per-key concatenation.  With ``args.use_graph`` the
    full-size chunks are replayed from a hipGraph (the ragged last chunk runs eagerly)."""
    chunk_size = args.chunk
    results = defaultdict(list)
    graphed = None
    if getattr(args, "use_graph", False) and ts is not None and rays.shape[0] >= 2 * chunk_size and type(_rng) is _TorchRng:
        graphed = GraphedRenderer(models, args, chunk_size, rays.device)
    for i in range(0, rays.shape[0], chunk_size):
        r, t = rays[i:i + chunk_size], ts[i:i + chunk_size] if ts is not None else None
        if graphed is not None and r.shape[0] == chunk_size:
            out = {k: v.clone() for k, v in graphed(r, t).items()}
        else:
            out = render_rays(models, args, r, t)
        for k, v in out.items():
            results[k] += [v]
    for k, v in results.items():
        results[k] = None if v[0] is None else torch.cat(v, 0)
    return results
