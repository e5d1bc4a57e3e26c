"""Typed Python wrappers over the C ABI (one function per entry point of include/satrender.h).

Every wrapper validates device / dtype / contiguity (the library itself sees only raw pointers), enqueues on
torch's current HIP stream and returns torch tensors it allocated.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib

MODES = {"bf16": _lib.MODE_BF16, "f16": _lib.MODE_F16, "bf16x3": _lib.MODE_BF16X3}


class KernelTimer:
    """Brackets selected kernel launches with HIP events on torch's current stream (bench.py's roofline leg)."""

    def __init__(self):
        self.spans = []

    def span(self, name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.spans.append((name, e0, e1))
        return e0, e1

    def mean_ms(self, name):
        t = [a.elapsed_time(b) for n, a, b in self.spans if n == name]
        return sum(t) / len(t) if t else None


kernel_timer = None  # set to a KernelTimer to time the fused-MLP launches


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, name, dtype=torch.float32, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); satnerf_amd has no CPU path")
    _same_device(t, name)
    if t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _same_device(t, name):
    """Launches go to torch's CURRENT stream, i.e. to the current device: a tensor living on another GPU would be touched by the
    wrong device's kernel."""
    if t.device.index != torch.cuda.current_device():
        raise ValueError(f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                         "(wrap the call in torch.cuda.device(...) or torch.cuda.set_device)")


def _rows(t, name, min_cols):
    """A 2-D fp32 view whose rows may be strided (e.g. rays[:, 3:6]): returns (tensor, row stride in elements)."""
    if t.dim() != 2 or t.shape[1] < min_cols or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
        raise ValueError(f"{name} must be a GPU fp32 (N,>={min_cols}) tensor with unit inner stride")
    _same_device(t, name)
    return t, t.stride(0) if t.shape[0] > 1 else t.shape[1]


def pack_stream(flat, idx, scale, want_lo, f16=False):
    """Gather + scale + convert the weight stream: bf16 hi (and lo) planes, or fp16 (``f16``: the SR_MODE_F16 forward stream)."""
    n = idx.numel()
    hi = torch.empty(n, dtype=torch.int16, device=flat.device)
    lo = torch.empty(n, dtype=torch.int16, device=flat.device) if want_lo else None
    _lib.call("sr_pack_stream", _p(_chk(flat, "flat")), _p(_chk(idx, "idx", torch.int32)), _p(_chk(scale, "scale")), n, _p(hi), _p(lo),
              n if f16 else 0, _stream())
    return hi, lo


def pack_stream_into(flat, idx, scale, hi, lo, f16=False):
    _lib.call("sr_pack_stream", _p(_chk(flat, "flat")), _p(_chk(idx, "idx", torch.int32)), _p(_chk(scale, "scale")), idx.numel(), _p(hi), _p(lo),
              idx.numel() if f16 else 0, _stream())


def gather_scale_into(flat, idx, scale, out):
    _lib.call("sr_gather_scale_f32", _p(_chk(flat, "flat")), _p(_chk(idx, "idx", torch.int32)), _p(_chk(scale, "scale")), idx.numel(), _p(out),
              _stream())


def gather_scale(flat, idx, scale):
    out = torch.empty(idx.numel(), dtype=torch.float32, device=flat.device)
    _lib.call("sr_gather_scale_f32", _p(_chk(flat, "flat")), _p(_chk(idx, "idx", torch.int32)), _p(_chk(scale, "scale")), idx.numel(), _p(out),
              _stream())
    return out


def ray_sample(rays, u, n_samples):
    rays, stride = _rows(rays, "rays", 8)
    n = rays.shape[0]
    _chk(u, "u")
    if tuple(u.shape) != (n, n_samples):
        raise ValueError(f"u must be ({n},{n_samples}), got {tuple(u.shape)}")
    z = torch.empty(n, n_samples, dtype=torch.float32, device=rays.device)
    _lib.call("sr_ray_sample_fwd", _p(rays), stride, _p(u), n, n_samples, _p(z), _stream())
    return z


def sky(sun, w1, b1, w2, b2):
    sun, stride = _rows(sun, "sun", 3)
    n, hidden = sun.shape[0], w1.shape[0]
    out = torch.empty(n, 3, dtype=torch.float32, device=sun.device)
    _lib.call("sr_sky_fwd", _p(sun), stride, n, hidden, _p(_chk(w1, "w1")), _p(_chk(b1, "b1")), _p(_chk(w2, "w2")), _p(_chk(b2, "b2")), _p(out),
              _stream())
    return out


def satnerf_mlp(org, direction, sun, z, temb, ts, n_points, n_samples, feat, tau, mode, stream_hi, stream_lo, l0, acts=None, fmt=16):
    """Fused MLP over n_points sample points; returns (albedo (P,3), sigma (P), sun_v (P), beta (P)).  ``acts`` = training
    workspace (``acts_workspace(..., fmt)``) the activations are saved to in format ``fmt`` (16 | 8)."""
    org, so = _rows(org, "org", 3)
    sun, ss = _rows(sun, "sun", 3)
    sd = 0
    if direction is not None:
        direction, sd = _rows(direction, "dir", 3)
    if z is not None:
        _chk(z, "z")
    _chk(temb, "temb")
    if ts is not None:
        _chk(ts, "ts", torch.int64)
    dev = org.device
    albedo = torch.empty(n_points, 3, dtype=torch.float32, device=dev)
    sigma = torch.empty(n_points, dtype=torch.float32, device=dev)
    sun_v = torch.empty(n_points, dtype=torch.float32, device=dev)
    beta = torch.empty(n_points, dtype=torch.float32, device=dev)
    inp = _lib.MlpInputs(_p(org), so, _p(direction), sd, _p(sun), ss, _p(z), _p(temb), _p(ts), n_points, n_samples)
    ev = kernel_timer.span("mlp_fwd") if kernel_timer is not None else None
    if ev:
        ev[0].record()
    _lib.call("sr_satnerf_mlp_fwd", C.byref(inp), feat, tau, MODES[mode], _p(stream_hi), _p(stream_lo), _p(_chk(l0, "l0")), _p(albedo), _p(sigma),
              _p(sun_v), _p(beta), _p(acts), int(fmt), _stream())
    if ev:
        ev[1].record()
    return albedo, sigma, sun_v, beta


def render_fused_ok(feat, mode, n_samples):
    """True when sr_satnerf_render_fwd covers (feat, mode) and ``n_samples`` divides the points a workgroup owns."""
    per_block = _lib.lib().sr_render_points_per_block(int(feat), MODES[mode])
    return per_block > 0 and n_samples >= 2 and per_block % n_samples == 0


def render_fwd(rays, ts, temb, n_samples, feat, tau, mode, stream_hi, stream_lo, l0, sky_w1, sky_b1, sky_w2, sky_b2, z=None, u=None, noise=None,
               noise_std=0.0, seed=0, step_counter=None, tick=False, want_z=True, bank_chunks=0):
    """One launch: stratified sampling (or given depths ``z``) -> fused MLP -> sky head + compositing (sr_satnerf_render_fwd).
    Depths: ``z`` (N,S) given, else stratified with ``u`` (N,S), else jitter drawn in the kernel (``seed``, ``step_counter``, ``tick``).
    ``bank_chunks`` > 0: ``rays`` / ``ts`` are a bank of bank_chunks x N rows and the launch renders chunk step_counter[0] % bank_chunks.
    Returns dict(z, albedo (N,S,3), sun_v (N,S), beta (N,S), sky (N,3), weights, transparency (N,S), depth (N), rgb (N,3))."""
    rays, stride = _rows(rays, "rays", 11)
    n, s, dev = rays.shape[0], int(n_samples), rays.device
    if bank_chunks:
        if rays.shape[0] % bank_chunks or ts.numel() != rays.shape[0] or step_counter is None:
            raise ValueError("bank mode: rays / ts must hold bank_chunks x N rows and step_counter is required")
        n = rays.shape[0] // bank_chunks
    _chk(ts, "ts", torch.int64), _chk(temb, "temb")
    for t, nm in ((z, "z"), (u, "u"), (noise, "noise")):
        if t is not None and tuple(_chk(t, nm).shape) != (n, s):
            raise ValueError(f"{nm} must be ({n},{s}), got {tuple(t.shape)}")
    if tick and (step_counter is None or step_counter.numel() < 4):
        raise ValueError("tick=True needs step_counter = a zero-initialised float32 block of 4")
    e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
    out = {"z": z if z is not None else (e(n, s) if want_z else None), "albedo": e(n, s, 3), "sun_v": e(n, s), "beta": e(n, s), "sky": e(n, 3),
           "weights": e(n, s), "transparency": e(n, s), "depth": e(n), "rgb": e(n, 3)}
    args = _lib.RenderArgs(_p(rays), stride, _p(ts), _p(temb), n, s, _p(z), _p(u), int(seed) & 0xFFFFFFFFFFFFFFFF,
                           _p(_chk(step_counter, "step_counter", allow_none=True)), int(bool(tick)), _p(noise), float(noise_std), sky_w1.shape[0],
                           _p(_chk(sky_w1, "w1")), _p(_chk(sky_b1, "b1")), _p(_chk(sky_w2, "w2")), _p(_chk(sky_b2, "b2")), int(bank_chunks))
    outs = _lib.RenderOutputs(_p(out["z"]) if z is None else None, _p(out["albedo"]), None, _p(out["sun_v"]), _p(out["beta"]), _p(out["sky"]),
                              _p(out["weights"]), _p(out["transparency"]), _p(out["depth"]), _p(out["rgb"]))
    ev = kernel_timer.span("mlp_fwd") if kernel_timer is not None else None
    if ev:
        ev[0].record()
    _lib.call("sr_satnerf_render_fwd", C.byref(args), feat, tau, MODES[mode], _p(stream_hi), _p(stream_lo), _p(_chk(l0, "l0")), C.byref(outs), _stream())
    if ev:
        ev[1].record()
    return out


def render_train(rays, ts, temb, n_samples, feat, tau, mode, stream_hi, stream_lo, l0, sky_w1, sky_b1, sky_w2, sky_b2, target, acts, u=None,
                 noise=None, noise_std=0.0, seed=0, step_counter=None, sched=None, beta_min=0.05, want_z=False, tick=0, gather=None):
    """The training forward in ONE launch (sr_satnerf_render_train): stratified depths (``u`` (N,S) given, else drawn in the kernel from
    ``seed`` / ``step_counter``) -> fused MLP saving the 8-bit activations into ``acts`` -> sky head, compositing, colour loss and the
    compositing backward per ray.  Returns dict(albedo (N,S,3), sigma, sun_v, beta (N,S), sky (N,3), z (N,S) or None, loss (partial sums),
    rgb (N,3), d_sigma, d_sun, g_beta (N,S), d_albedo (N,S,3), d_sky (N,3)) -- what ``ray_setup`` + ``satnerf_mlp`` + ``render_loss``
    return, bit for bit.  ``tick`` = 2: the launch opens the step (advances ``step_counter`` and draws for the advanced value).
    ``gather`` = dict(idx, cursor, batches, out=(rays (N,11), rgbs (N,3), ts (N))): ``rays`` / ``ts`` / ``target`` are the resident bank and
    the launch samples batch cursor[0] of the epoch's shuffled ``idx`` itself, writing the batch rows to ``out``."""
    rays, stride = _rows(rays, "rays", 11)
    n, s, dev = rays.shape[0], int(n_samples), rays.device
    if gather is not None:
        o_rays, o_rgbs, o_ts = gather["out"]
        n = o_rays.shape[0]
        if (stride != 11 or tuple(_chk(o_rays, "out rays").shape) != (n, 11) or tuple(_chk(o_rgbs, "out rgbs").shape) != (n, 3)
                or tuple(_chk(o_ts, "out ts", torch.int64).shape) != (n,) or _chk(gather["idx"], "idx", torch.int64).numel() < int(gather["batches"]) * n
                or target.shape[0] != rays.shape[0] or ts.shape[0] != rays.shape[0] or u is not None or noise is not None):
            raise ValueError("gather: the bank (rays (R,11), ts (R), target (R,3)), idx (>= batches x N) and out = ((N,11), (N,3), (N,)) do not fit")
    _chk(ts, "ts", torch.int64), _chk(temb, "temb")
    for t, nm in ((u, "u"), (noise, "noise")):
        if t is not None and tuple(_chk(t, nm).shape) != (n, s):
            raise ValueError(f"{nm} must be ({n},{s}), got {tuple(t.shape)}")
    per_block = _lib.lib().sr_render_points_per_block(int(feat), MODES[mode])
    if per_block <= 0 or s > 64 or per_block % s:
        raise ValueError(f"no fused training forward for feat={feat}, mode={mode}, n_samples={s}")
    e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
    out = {"z": e(n, s) if want_z else None, "albedo": e(n, s, 3), "sigma": e(n, s), "sun_v": e(n, s), "beta": e(n, s), "sky": e(n, 3),
           "loss": e((n * s + per_block - 1) // per_block), "rgb": e(n, 3), "d_sigma": e(n, s), "d_albedo": e(n, s, 3), "d_sun": e(n, s),
           "g_beta": e(n, s), "d_sky": e(n, 3)}
    args = _lib.RenderArgs(_p(rays), stride, _p(ts), _p(temb), n, s, None, _p(u), int(seed) & 0xFFFFFFFFFFFFFFFF,
                           _p(_chk(step_counter, "step_counter", allow_none=True)), int(tick), _p(noise), float(noise_std), sky_w1.shape[0],
                           _p(_chk(sky_w1, "w1")), _p(_chk(sky_b1, "b1")), _p(_chk(sky_w2, "w2")), _p(_chk(sky_b2, "b2")), 0)
    outs = _lib.RenderOutputs(_p(out["z"]), _p(out["albedo"]), _p(out["sigma"]), _p(out["sun_v"]), _p(out["beta"]), _p(out["sky"]), None, None, None, None)
    g = gather or {}
    tr = _lib.TrainArgs(_p(_chk(target, "target")), _p(_chk(sched, "sched", allow_none=True)), float(beta_min), _p(out["loss"]), _p(out["rgb"]),
                        _p(out["d_sigma"]), _p(out["d_albedo"]), _p(out["d_sun"]), _p(out["g_beta"]), _p(out["d_sky"]),
                        _p(g.get("idx")), _p(_chk(g.get("cursor"), "cursor", allow_none=True)), int(g.get("batches", 0)),
                        *([_p(t) for t in g["out"]] if gather is not None else [None, None, None]))
    ev = kernel_timer.span("mlp_fwd") if kernel_timer is not None else None
    if ev:
        ev[0].record()
    _lib.call("sr_satnerf_render_train", C.byref(args), feat, tau, MODES[mode], _p(stream_hi), _p(stream_lo), _p(_chk(l0, "l0")), C.byref(outs),
              C.byref(tr), _p(acts), 8, _stream())
    if ev:
        ev[1].record()
    return out


def composite(z, sigma, noise, noise_std, albedo, sun_v, sky_rgb, clamp_rgb=True):
    n, s = z.shape
    dev = z.device
    _chk(z, "z"), _chk(sigma, "sigma")
    weights = torch.empty(n, s, dtype=torch.float32, device=dev)
    transp = torch.empty(n, s, dtype=torch.float32, device=dev)
    depth = torch.empty(n, dtype=torch.float32, device=dev)
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    _lib.call("sr_composite_fwd", _p(z), _p(sigma), _p(_chk(noise, "noise", allow_none=True)), float(noise_std), _p(_chk(albedo, "albedo")),
              _p(_chk(sun_v, "sun_v", allow_none=True)), _p(_chk(sky_rgb, "sky", allow_none=True)), n, s, int(clamp_rgb), _p(weights), _p(transp),
              _p(depth), _p(rgb), _stream())
    return weights, transp, depth, rgb


def composite_bwd(z, sigma, noise, noise_std, albedo, sun_v, sky_rgb, weights, transp, g_rgb, g_depth, g_weights, g_transp, clamp_rgb=True):
    n, s = z.shape
    dev = z.device
    d_sigma = torch.empty(n, s, dtype=torch.float32, device=dev)
    d_albedo = torch.empty(n, s, 3, dtype=torch.float32, device=dev)
    d_sun = torch.empty(n, s, dtype=torch.float32, device=dev)
    d_sky = torch.empty(n, 3, dtype=torch.float32, device=dev)
    opt = lambda t, nm: _p(_chk(t, nm, allow_none=True))  # noqa: E731
    _lib.call("sr_composite_bwd", _p(z), _p(sigma), opt(noise, "noise"), float(noise_std), _p(albedo), opt(sun_v, "sun_v"), opt(sky_rgb, "sky"),
              _p(weights), _p(transp), None, n, s, int(clamp_rgb), opt(g_rgb, "g_rgb"), opt(g_depth, "g_depth"), opt(g_weights, "g_weights"),
              opt(g_transp, "g_transparency"), _p(d_sigma), _p(d_albedo), _p(d_sun), _p(d_sky), _stream())
    return d_sigma, d_albedo, d_sun, d_sky


# ----------------------------------------------------------------------------------- layer-by-layer path (any width)
ACTS = {None: _lib.ACT_NONE, "none": _lib.ACT_NONE, "sin": _lib.ACT_SIN, "relu": _lib.ACT_RELU}
OUT_ACTS = {None: _lib.OUT_NONE, "none": _lib.OUT_NONE, "softplus": _lib.OUT_SOFTPLUS, "sigmoid": _lib.OUT_SIGMOID, "sigmoid_rgb": _lib.OUT_SIGMOID_RGB}


def _linear_srcs(srcs, n_points):
    """srcs: 1 or 2 tuples (x (rows, k) fp32 with unit inner stride, act, w0, row_div) -> ctypes array (keeps the tensors alive)."""
    if not 1 <= len(srcs) <= 2:
        raise ValueError("a linear layer takes one or two concatenated sources")
    arr = (_lib.LinearSrc * len(srcs))()
    for a, (x, act, w0, row_div) in zip(arr, srcs):
        if act not in ACTS:
            raise ValueError(f"unknown activation {act!r} (one of {sorted(k for k in ACTS if k)})")
        x, ld = _rows(x, "linear source", 1)
        if x.shape[0] * row_div < n_points:
            raise ValueError(f"linear source has {x.shape[0]} rows x row_div {row_div} < {n_points} points")
        a.x, a.ld, a.k, a.act, a.w0, a.row_div = x.data_ptr(), ld, x.shape[1], ACTS[act], float(w0), int(row_div)
    return arr


def linear_fwd(srcs, weight, bias, n_points, out_act=None):
    """One nn.Linear on the concatenation of ``srcs`` (activation applied on load) -> (P, n_out) fp32, see sr_linear_fwd."""
    if out_act not in OUT_ACTS:
        raise ValueError(f"unknown output activation {out_act!r} (one of {sorted(k for k in OUT_ACTS if k)})")
    arr = _linear_srcs(srcs, n_points)
    n_out = weight.shape[0]
    if weight.shape[1] != sum(x.shape[1] for x, *_ in srcs):
        raise ValueError(f"weight is {tuple(weight.shape)} but the sources have {sum(x.shape[1] for x, *_ in srcs)} columns")
    y = torch.empty(n_points, n_out, dtype=torch.float32, device=weight.device)
    _lib.call("sr_linear_fwd", arr, len(srcs), _p(_chk(weight, "weight")), _p(_chk(bias, "bias", allow_none=True)), n_points, n_out, OUT_ACTS[out_act],
              _p(y), n_out, _stream())
    return y


def linear_bwd_input(gy, y, out_act, weight, col0, target, n_points):
    """Gradient w.r.t. one source (pre-activation): (P, k); ``target`` = (x, act, w0, row_div) of that source."""
    arr = _linear_srcs([target], n_points if target[3] == 1 else target[0].shape[0] * target[3])
    k = target[0].shape[1]
    d = torch.empty(n_points, k, dtype=torch.float32, device=gy.device)
    _lib.call("sr_linear_bwd_input", _p(_chk(gy, "gy")), gy.shape[1], _p(_chk(y, "y", allow_none=True)), 0 if y is None else y.shape[1], OUT_ACTS[out_act],
              _p(_chk(weight, "weight")), weight.shape[1], col0, arr, n_points, weight.shape[0], _p(d), k, _stream())
    return d


def linear_bwd_weight(gy, y, out_act, srcs, n_points, n_out, want_bias=True):
    arr = _linear_srcs(srcs, n_points)
    k = sum(x.shape[1] for x, *_ in srcs)
    dw = torch.zeros(n_out, k, dtype=torch.float32, device=gy.device)
    db = torch.zeros(n_out, dtype=torch.float32, device=gy.device) if want_bias else None
    _lib.call("sr_linear_bwd_weight", _p(_chk(gy, "gy")), gy.shape[1], _p(_chk(y, "y", allow_none=True)), 0 if y is None else y.shape[1], OUT_ACTS[out_act],
              arr, len(srcs), n_points, n_out, _p(dw), _p(db), _stream())
    return dw, db


def positional_map(x, n_freqs):
    """``Mapping.forward`` (models/nerf.py:53-69): (rows, dim) -> (rows, 2*n_freqs*dim)."""
    x, ld = _rows(x, "x", 1)
    out = torch.empty(x.shape[0], 2 * n_freqs * x.shape[1], dtype=torch.float32, device=x.device)
    _lib.call("sr_positional_map", _p(x), ld, x.shape[1], x.shape[0], n_freqs, _p(out), _stream())
    return out


def points_along(rays, dir_col, z):
    """xyz (N*S, 3) = rays[:, 0:3] + rays[:, dir_col:dir_col+3] * z (rendering.py:81)."""
    rays, stride = _rows(rays, "rays", dir_col + 3)
    n, s = z.shape
    xyz = torch.empty(n * s, 3, dtype=torch.float32, device=rays.device)
    _lib.call("sr_points_along", _p(rays), stride, dir_col, _p(_chk(z, "z")), n, s, _p(xyz), _stream())
    return xyz


IMAGE_COLUMNS = {"rgb": (0, 3), "depth": (3, 4), "acc": (4, 5), "sun": (5, 6), "albedo": (6, 9), "beta": (9, 10), "sky": (10, 13)}


def composite_image(z, sigma, noise, noise_std, albedo, sun_v, beta, sky_rgb):
    """Compositing reduced to the per-pixel images of eval_satnerf.save_nerf_output_to_images: (N,13), see IMAGE_COLUMNS."""
    n, s = z.shape
    image = torch.empty(n, 13, dtype=torch.float32, device=z.device)
    _lib.call("sr_composite_image", _p(_chk(z, "z")), _p(_chk(sigma, "sigma")), _p(_chk(noise, "noise", allow_none=True)), float(noise_std),
              _p(_chk(albedo, "albedo")), _p(_chk(sun_v, "sun_v")), _p(_chk(beta, "beta")), _p(_chk(sky_rgb, "sky")), n, s, _p(image), _stream())
    return image


def latlonalt_from_depth(rays, depth, center, scene_range):
    """(lat, lon, alt) fp64 device tensors of the points rays_o + rays_d * depth, de-normalised by ``scene_range`` / ``center``."""
    import ctypes

    rays, stride = _rows(rays, "rays", 6)
    n = rays.shape[0]
    if depth.reshape(-1).shape[0] != n:
        raise ValueError("depth must have one value per ray")
    depth = _chk(depth.reshape(-1).contiguous().float(), "depth")
    c = (ctypes.c_double * 3)(*[float(v) for v in center])
    out = torch.empty(3, n, dtype=torch.float64, device=rays.device)
    _lib.call("sr_latlonalt_from_depth", _p(rays), stride, _p(depth), n, ctypes.addressof(c), float(scene_range), out[0].data_ptr(),
              out[1].data_ptr(), out[2].data_ptr(), _stream())
    return out[0], out[1], out[2]


RPC_KEYS = ("row_num", "row_den", "col_num", "col_den")
RPC_SCALARS = ("row_offset", "col_offset", "lat_offset", "lon_offset", "alt_offset", "row_scale", "col_scale", "lat_scale", "lon_scale", "alt_scale")


def rpc_rays(rpc, width, height, min_alt, max_alt, center, scene_range, sun_elevation_deg, sun_azimuth_deg, device, want_cache=False):
    """RPC ray generation of one image on the GPU (sr_rpc_rays): ``rpc`` = dict in rpcm's "rpcm" format (20-term lists row_num,
    row_den, col_num, col_den + the ten offsets / scales).  Returns (rays (H*W, 11) fp32, cache (H*W, 8) fp32 or None)."""
    import ctypes

    vals = []
    for k in RPC_KEYS:
        c = [float(v) for v in rpc[k]]
        if len(c) != 20:
            raise ValueError(f"rpc[{k!r}] must hold the 20 RPC00B coefficients, got {len(c)}")
        vals += c
    vals += [float(rpc[k]) for k in RPC_SCALARS]
    buf = (ctypes.c_double * 90)(*vals)
    ctr = (ctypes.c_double * 3)(*[float(v) for v in center])
    n = int(width) * int(height)
    dev = torch.device(device)
    with torch.cuda.device(dev):
        rays = torch.empty(n, 11, dtype=torch.float32, device=dev)
        cache = torch.empty(n, 8, dtype=torch.float32, device=dev) if want_cache else None
        _lib.call("sr_rpc_rays", ctypes.addressof(buf), int(width), int(height), float(min_alt), float(max_alt), ctypes.addressof(ctr),
                  float(scene_range), float(sun_elevation_deg), float(sun_azimuth_deg), _p(rays), _p(cache), _stream())
    return rays, cache


def sample_pdf(bins, weights, u, eps=1e-5):
    n, nb = bins.shape
    _chk(bins, "bins"), _chk(weights, "weights"), _chk(u, "u")
    if tuple(weights.shape) != (n, nb - 1) or u.shape[0] != n:
        raise ValueError(f"sample_pdf: weights must be ({n},{nb - 1}) and u ({n},I)")
    out = torch.empty(n, u.shape[1], dtype=torch.float32, device=bins.device)
    _lib.call("sr_sample_pdf", _p(bins), _p(weights), _p(u), n, nb, u.shape[1], float(eps), _p(out), _stream())
    return out


def sample_pdf_merge(z_coarse, weights_coarse, u, eps=1e-5):
    n, s = z_coarse.shape
    i = u.shape[1]
    _chk(z_coarse, "z_coarse"), _chk(weights_coarse, "weights_coarse"), _chk(u, "u")
    if tuple(weights_coarse.shape) != (n, s) or u.shape[0] != n:
        raise ValueError("sample_pdf_merge: shape mismatch")
    z_fine = torch.empty(n, s + i, dtype=torch.float32, device=z_coarse.device)
    _lib.call("sr_sample_pdf_merge", _p(z_coarse), _p(weights_coarse), _p(u), n, s, i, float(eps), _p(z_fine), _stream())
    return z_fine


# ------------------------------------------------------------------------------------------------ backward
def graph_capture(graph):
    """``torch.cuda.graph(graph)``; under ``torch.distributed`` the capture is thread-local, so the process group's watchdog
    thread (which polls events of earlier collectives) cannot invalidate a capture in progress on this thread."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return torch.cuda.graph(graph, capture_error_mode="thread_local")
    return torch.cuda.graph(graph)


def default_fmt(mode):
    """Workspace format of a numeric mode: the throughput mode trains on 8-bit saved state, the parity mode on 16-bit."""
    return 8 if mode in ("bf16", "f16") else 16


def _ws_empty(n, dtype, device, slot):
    """Workspace allocation; SATNERF_WS_PAD=<bytes> shifts workspace `slot` by slot * pad bytes (placement experiments)."""
    pad = int(os.environ.get("SATNERF_WS_PAD", "0")) * slot
    if not pad:
        return torch.empty(n, dtype=dtype, device=device)
    item = torch.empty((), dtype=dtype).element_size()
    return torch.empty(n + pad // item, dtype=dtype, device=device)[pad // item:]


def acts_workspace(n_points, feat, device, fmt=16):
    per_tile = _lib.lib().sr_act_elems_per_tile(feat, int(fmt))
    if per_tile <= 0:
        raise ValueError(f"unsupported workspace (feat={feat}, fmt={fmt})")
    return _ws_empty(_lib.lib().sr_workspace_tiles(n_points) * per_tile, torch.int16, device, 1)


def satnerf_mlp_bwd(feat, tau, n_points, bwd_stream, acts, albedo, sigma, sun_v, beta, g_albedo, g_sigma, g_sun_v, g_beta, want_dt=True, fmt=16):
    """dX chain: returns (dpre workspace, d_t (P,tau) or None); ``fmt`` = format of ``acts`` and of the returned workspace."""
    dev = albedo.device
    n_elems = _lib.lib().sr_dpre_workspace_elems(n_points, feat, int(fmt))  # (8-bit: + the table of exponent maxima behind the last tile)
    if n_elems <= 0:
        raise ValueError(f"unsupported workspace (feat={feat}, fmt={fmt})")
    dpre = _ws_empty(n_elems, torch.int16, dev, 2)
    d_t = torch.empty(n_points, tau, dtype=torch.float32, device=dev) if want_dt else None
    opt = lambda t, nm: _p(_chk(t, nm, allow_none=True))  # noqa: E731
    ev = kernel_timer.span("mlp_bwd") if kernel_timer is not None else None
    if ev:
        ev[0].record()
    _lib.call("sr_satnerf_mlp_bwd", feat, tau, n_points, _p(bwd_stream), _p(acts), _p(_chk(albedo, "albedo")), _p(_chk(sigma, "sigma")),
              _p(_chk(sun_v, "sun_v")), _p(_chk(beta, "beta")), opt(g_albedo, "g_albedo"), opt(g_sigma, "g_sigma"), opt(g_sun_v, "g_sun_v"),
              opt(g_beta, "g_beta"), _p(dpre), _p(d_t), int(fmt), _stream())
    if ev:
        ev[1].record()
    return dpre, d_t


_plan_cache = {}


def wgrad_plan(blocks, n_points, n_wg=0, fmt=16):
    """Split-K plan of the weight-gradient job table for ``n_points`` points (sr_wgrad_plan; ``fmt`` = workspace format of the kernel
    that will run it): returns (planned device table, total slices, span of a stream-K plan or 0).  Cached per (table, n_points, fmt): the copy to the device
    must not happen inside a graph capture."""
    key = (blocks.data_ptr(), int(n_points), int(n_wg), str(blocks.device), int(fmt))
    ent = _plan_cache.get(key)
    if ent is None:
        import ctypes

        host = _chk(blocks, "blocks", torch.int32).cpu().contiguous().clone()
        n_slices = ctypes.c_int(0)
        with torch.cuda.device(blocks.device):
            _lib.call("sr_wgrad_plan", host.data_ptr(), host.shape[0], n_points, n_wg, int(fmt), ctypes.addressof(n_slices))
        if len(_plan_cache) > 64:
            _plan_cache.clear()
        # (int 11 of the first row: the span of a stream-K plan, 0 otherwise -- the launch has to know which build of the kernel to run)
        ent = _plan_cache[key] = (host.to(blocks.device), int(n_slices.value), blocks, int(host[0, 11]))  # keeps `blocks` alive: data_ptr stays unique
    return ent[0], ent[1], ent[3]


def wgrad_partials(feat, tau, n_points, dpre, acts, blocks, fmt=16, loads=None):
    """Weight-gradient GEMMs only: returns (fp32 split-K slices, planned job table); reduce with grad_tail / unpack_grads.
    ``fmt`` = format of both workspaces; the 8-bit kernel also needs the per-block load table ``loads`` (packing.wgrad8_loads)."""
    plan, n_slices, span = wgrad_plan(blocks, n_points, fmt=fmt)
    block_floats = 256 * 256 + 256 * 32  # csrc/mlp_layout.h kWgBlockFloats
    partial = _ws_empty(n_slices * block_floats, torch.float32, dpre.device, 3)
    ev = kernel_timer.span("wgrad") if kernel_timer is not None else None
    if ev:
        ev[0].record()
    if int(fmt) == 8:
        _lib.call("sr_satnerf_wgrad8", feat, tau, n_points, _p(dpre), dpre.numel(), _p(acts), _p(plan), _p(_chk(loads, "loads", torch.int32)), plan.shape[0], n_slices,
                  span, _p(partial), _stream())
    else:
        _lib.call("sr_satnerf_wgrad", feat, tau, n_points, _p(dpre), _p(acts), _p(plan), plan.shape[0], n_slices, _p(partial), _stream())
    if ev:
        ev[1].record()
    return partial, plan


def satnerf_wgrad(feat, tau, n_points, dpre, acts, blocks, gidx, gscale, grad_flat, accumulate=True, fmt=16, loads=None):
    """Weight-gradient GEMMs + split-K reduction + scatter into the flat gradient buffer."""
    partial, plan = wgrad_partials(feat, tau, n_points, dpre, acts, blocks, fmt, loads)
    _lib.call("sr_unpack_grads", _p(partial), _p(_chk(gidx, "gidx", torch.int32)), _p(_chk(gscale, "gscale")), gidx.numel(), _p(plan),
              _p(_chk(grad_flat, "grad_flat")), int(accumulate), _stream())


def sky_bwd(sun, w1, b1, w2, sky_rgb, d_sky, g_w1, g_b1, g_w2, g_b2):
    sun, stride = _rows(sun, "sun", 3)
    _lib.call("sr_sky_bwd", _p(sun), stride, sun.shape[0], w1.shape[0], _p(w1), _p(b1), _p(w2), _p(_chk(sky_rgb, "sky")), _p(_chk(d_sky, "d_sky")),
              _p(g_w1), _p(g_b1), _p(g_w2), _p(g_b2), _stream())


def embedding_bwd(d_t, ts, n_rays, n_samples, tau, g_emb):
    _lib.call("sr_embedding_bwd", _p(_chk(d_t, "d_t")), _p(_chk(ts, "ts", torch.int64)), n_rays, n_samples, tau, _p(_chk(g_emb, "g_emb")), _stream())


def satnerf_loss(rgb, weights, beta, target, beta_min=0.05, grad_scale=1.0, sched=None):
    """Fused SatNerfLoss forward + gradient: returns (loss partial sums (ceil(N/4),) -- the loss is their sum --, g_rgb (N,3),
    g_weights (N,S), g_beta (N,S))."""
    n, s = weights.shape
    dev = rgb.device
    loss = torch.empty((n + 3) // 4, dtype=torch.float32, device=dev)
    g_rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    g_w = torch.empty(n, s, dtype=torch.float32, device=dev)
    g_b = torch.empty(n, s, dtype=torch.float32, device=dev)
    _lib.call("sr_satnerf_loss", _p(_chk(rgb, "rgb")), _p(_chk(weights, "weights")), _p(_chk(beta, "beta")), _p(_chk(target, "target")), n, s,
              float(beta_min), float(grad_scale), _p(_chk(sched, "sched", allow_none=True)), _p(loss), _p(g_rgb), _p(g_w), _p(g_b), _stream())
    return loss, g_rgb, g_w, g_b


def adam_step(params, grads, exp_avg, exp_avg_sq, step, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, zero_grad=True):
    """``step`` is the 1-based step count (bias corrections are computed on the host in fp64)."""
    _lib.call("sr_adam_step", _p(_chk(params, "params")), _p(_chk(grads, "grads")), _p(_chk(exp_avg, "exp_avg")), _p(_chk(exp_avg_sq, "exp_avg_sq")),
              params.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(grad_scale), int(step), int(zero_grad), _stream())


def pack_all(flat, idx, scale, hi, lo, f32_idx, f32_scale, f32_out, tick=None, n_f16=0):
    """``n_f16``: the first n_f16 stream elements (the forward stream) are written as fp16 (SR_MODE_F16), the rest as bf16."""
    _lib.call("sr_pack_all", _p(_chk(flat, "flat")), _p(_chk(idx, "idx", torch.int32)), _p(_chk(scale, "scale")), idx.numel(), _p(hi), _p(lo),
              _p(_chk(f32_idx, "f32_idx", torch.int32)), _p(_chk(f32_scale, "f32_scale")), f32_idx.numel(), _p(f32_out), _p(tick), int(n_f16),
              _stream())


def sc_loss(z, sigma, noise, noise_std, sun_v, lambda_sc):
    """Solar-correction terms of the pass along the sun direction: (loss_parts, d_sun_v (N,S))."""
    n, s = z.shape
    parts = torch.empty((n + 3) // 4, dtype=torch.float32, device=z.device)
    d_sun = torch.empty(n, s, dtype=torch.float32, device=z.device)
    _lib.call("sr_sc_loss", _p(_chk(z, "z")), _p(_chk(sigma, "sigma")), _p(_chk(noise, "noise", allow_none=True)), float(noise_std),
              _p(_chk(sun_v, "sun_v")), n, s, float(lambda_sc), _p(parts), _p(d_sun), _stream())
    return parts, d_sun


def depth_loss(depth, depths, lambda_ds, use_weights=True):
    """metrics.DepthLoss (coarse) value parts + gradient w.r.t. the rendered depth: (loss_parts, g_depth (N,))."""
    depths, stride = _rows(depths, "depths", 2 if use_weights else 1)
    n = depth.shape[0]
    parts = torch.empty((n + 255) // 256, dtype=torch.float32, device=depth.device)
    g = torch.empty(n, dtype=torch.float32, device=depth.device)
    _lib.call("sr_depth_loss", _p(_chk(depth, "depth")), _p(depths), stride, int(use_weights), n, float(lambda_ds), _p(parts), _p(g), _stream())
    return parts, g


def ray_setup(rays, u, n_samples, w1, b1, w2, b2, seed=0, step_counter=None, tick=False):
    """Fused sr_ray_sample_fwd + sr_sky_fwd: returns (z (N,S), sky (N,3)).  ``u`` = (N,S) uniform jitter, or None to draw it inside
    the kernel (Philox keyed by ``seed``, stepping with the device counter ``step_counter[0]``; ``tick``: the launch advances the
    counter itself -- ``step_counter`` is then a zero-initialised 4-float block)."""
    rays, stride = _rows(rays, "rays", 11)
    n = rays.shape[0]
    z = torch.empty(n, n_samples, dtype=torch.float32, device=rays.device)
    sky_rgb = torch.empty(n, 3, dtype=torch.float32, device=rays.device)
    w = (_p(_chk(w1, "w1")), _p(_chk(b1, "b1")), _p(_chk(w2, "w2")), _p(_chk(b2, "b2")))
    if u is None:
        if tick and (step_counter is None or step_counter.numel() < 4):
            raise ValueError("tick=True needs step_counter = a zero-initialised float32 block of 4")
        _lib.call("sr_ray_setup_rng", _p(rays), stride, int(seed) & 0xFFFFFFFFFFFFFFFF, _p(_chk(step_counter, "step_counter", allow_none=True)),
                  int(bool(tick)), n, n_samples, w1.shape[0], *w, _p(z), _p(sky_rgb), _stream())
    else:
        _lib.call("sr_ray_setup", _p(rays), stride, _p(_chk(u, "u")), n, n_samples, w1.shape[0], *w, _p(z), _p(sky_rgb), _stream())
    return z, sky_rgb


def render_loss(z, sigma, noise, noise_std, albedo, sun_v, beta, sky_rgb, target, beta_min=0.05, sched=None):
    """Fused compositing forward + SatNerf loss + compositing backward (S <= 64).
    Returns (loss partial sums, rgb (N,3), d_sigma (N,S), d_albedo (N,S,3), d_sun (N,S), g_beta (N,S), d_sky (N,3))."""
    n, s = z.shape
    dev = z.device
    e = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
    loss, rgb, d_sigma, d_albedo, d_sun, g_beta, d_sky = e((n + 3) // 4), e(n, 3), e(n, s), e(n, s, 3), e(n, s), e(n, s), e(n, 3)
    _lib.call("sr_render_loss", _p(_chk(z, "z")), _p(_chk(sigma, "sigma")), _p(_chk(noise, "noise", allow_none=True)), float(noise_std),
              _p(_chk(albedo, "albedo")), _p(_chk(sun_v, "sun_v")), _p(_chk(beta, "beta")), _p(_chk(sky_rgb, "sky")), _p(_chk(target, "target")), n, s,
              float(beta_min), _p(_chk(sched, "sched", allow_none=True)), _p(loss), _p(rgb), _p(d_sigma), _p(d_albedo), _p(d_sun), _p(g_beta), _p(d_sky),
              _stream())
    return loss, rgb, d_sigma, d_albedo, d_sun, g_beta, d_sky


def gather_batch(rays, rgbs, ts, idx, out=None, cursor=None, batches=0):
    """Rows ``idx`` of the ray bank -> (rays (B,11), ts (B,), rgbs (B,3)), one launch.  With ``cursor`` (zeros(4) float32) ``idx``
    holds a whole epoch of ``batches`` x B indices: the launch takes batch cursor[0] and advances the cursor (captured steps)."""
    n = idx.numel()
    if cursor is not None:
        if batches < 1 or n % batches or cursor.numel() < 4:
            raise ValueError("cursor mode: idx must hold batches x B indices and cursor 4 floats")
        n //= batches
    if rays.shape[1] != 11 or rgbs.shape[1] != 3:
        raise ValueError("gather_batch expects (N,11) rays and (N,3) rgbs")
    if out is None:
        out = (torch.empty(n, 11, dtype=torch.float32, device=rays.device), torch.empty(n, dtype=torch.int64, device=rays.device),
               torch.empty(n, 3, dtype=torch.float32, device=rays.device))
    _lib.call("sr_gather_batch", _p(_chk(rays, "rays")), _p(_chk(rgbs, "rgbs")), _p(_chk(ts, "ts", torch.int64)), _p(_chk(idx, "idx", torch.int64)), n,
              _p(_chk(out[0], "out_rays")), _p(_chk(out[2], "out_rgbs")), _p(_chk(out[1], "out_ts", torch.int64)),
              _p(_chk(cursor, "cursor", allow_none=True)), int(batches), _stream())
    return out


def gather_setup(rays, rgbs, ts, idx, out, n_samples, w1, b1, w2, b2, z, sky_rgb, seed, step_counter, step_offset=1, cursor=None, batches=0):
    """``gather_batch`` + ``ray_setup`` (in-kernel jitter) in one launch: rows -> ``out`` = (rays, ts, rgbs), depths -> ``z`` (B,S),
    sky colour -> ``sky_rgb`` (B,3); the jitter step is step_counter[0] + step_offset."""
    n = idx.numel()
    if cursor is not None:
        if batches < 1 or n % batches or cursor.numel() < 4:
            raise ValueError("cursor mode: idx must hold batches x B indices and cursor 4 floats")
        n //= batches
    if out[0].shape[0] != n or tuple(z.shape) != (n, n_samples) or tuple(sky_rgb.shape) != (n, 3):
        raise ValueError("gather_setup: output shapes do not match the batch")
    _lib.call("sr_gather_setup", _p(_chk(rays, "rays")), _p(_chk(rgbs, "rgbs")), _p(_chk(ts, "ts", torch.int64)), _p(_chk(idx, "idx", torch.int64)), n,
              _p(_chk(out[0], "out_rays")), _p(_chk(out[2], "out_rgbs")), _p(_chk(out[1], "out_ts", torch.int64)),
              _p(_chk(cursor, "cursor", allow_none=True)), int(batches), int(n_samples), w1.shape[0], _p(_chk(w1, "w1")), _p(_chk(b1, "b1")),
              _p(_chk(w2, "w2")), _p(_chk(b2, "b2")), _p(_chk(z, "z")), _p(_chk(sky_rgb, "sky")), int(seed) & 0xFFFFFFFFFFFFFFFF,
              _p(_chk(step_counter, "step_counter", allow_none=True)), int(step_offset), _stream())


def grad_tail(partial, plan, gidx, gscale, grad_flat, sun, w1, b1, w2, sky_rgb, d_sky, g_w1, g_b1, g_w2, g_b2, d_t, ts, n_rays,
              n_samples, tau, g_emb):
    sun, stride = _rows(sun, "sun", 3)
    _lib.call("sr_grad_tail", _p(partial), _p(_chk(gidx, "gidx", torch.int32)), _p(_chk(gscale, "gscale")), gidx.numel(), _p(plan), plan.shape[0],
              _p(_chk(grad_flat, "grad_flat")), 1, _p(sun), stride, n_rays, w1.shape[0], _p(w1), _p(b1), _p(w2), _p(_chk(sky_rgb, "sky")),
              _p(_chk(d_sky, "d_sky")), _p(g_w1), _p(g_b1), _p(g_w2), _p(g_b2), _p(_chk(d_t, "d_t")), _p(_chk(ts, "ts", torch.int64)), n_samples, tau,
              _p(_chk(g_emb, "g_emb")), _stream())


def grad_tail_adam(partial, plan, gidx, gscale, grad_flat, sun, w1, b1, w2, sky_rgb, d_sky, g_w1, g_b1, g_w2, g_b2, d_t, ts, n_rays, n_samples, tau,
                   g_emb, params, exp_avg, exp_avg_sq, late_idx, state, lr=-1.0, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, pack=None):
    """``grad_tail`` + ``adam_step_graph`` in one launch (sr_grad_tail_adam): ``params`` / ``exp_avg`` / ``exp_avg_sq`` are the flat buffers
    aligned with ``grad_flat`` (they may extend past it: ``late_idx`` addresses the embedding rows behind the model's parameters).
    ``pack`` (``SatNeRF.pack_scatter``): the launch also writes every updated parameter into the weight streams (no sr_pack_all next step)."""
    sun, stride = _rows(sun, "sun", 3)
    ps = None
    if pack is not None:
        if pack["map"].shape != (gidx.numel(), 2):
            raise ValueError("pack map does not match the parameter count")
        ps = _lib.PackScatter(_p(_chk(pack["map"], "pack map", torch.int32)), _p(pack["hi"]), _p(pack["lo"]), _p(_chk(pack["l0"], "pack l0")),
                              int(pack["n_f16"]), (C.c_float * 4)(*[float(x) for x in pack["scales"]]))
    _lib.call("sr_grad_tail_adam", _p(partial), _p(_chk(gidx, "gidx", torch.int32)), _p(_chk(gscale, "gscale")), gidx.numel(), _p(plan), plan.shape[0],
              _p(_chk(grad_flat, "grad_flat")), 1, _p(sun), stride, n_rays, w1.shape[0], _p(w1), _p(b1), _p(w2), _p(_chk(sky_rgb, "sky")),
              _p(_chk(d_sky, "d_sky")), _p(g_w1), _p(g_b1), _p(g_w2), _p(g_b2), _p(_chk(d_t, "d_t")), _p(_chk(ts, "ts", torch.int64)), n_samples, tau,
              _p(_chk(g_emb, "g_emb")), _p(_chk(params, "params")), _p(_chk(exp_avg, "exp_avg")), _p(_chk(exp_avg_sq, "exp_avg_sq")),
              _p(_chk(late_idx, "late_idx", torch.int32)), late_idx.numel(), _p(_chk(state, "state")), float(lr), float(betas[0]), float(betas[1]),
              float(eps), float(grad_scale), C.byref(ps) if ps is not None else None, _stream())


def _pack_scatter_struct(pack, n_model):
    if pack["map"].shape != (n_model, 2):
        raise ValueError("pack map does not match the parameter count")
    return _lib.PackScatter(_p(_chk(pack["map"], "pack map", torch.int32)), _p(pack["hi"]), _p(pack["lo"]), _p(_chk(pack["l0"], "pack l0")),
                            int(pack["n_f16"]), (C.c_float * 4)(*[float(x) for x in pack["scales"]]))


def adam_step_pack(params, grads, exp_avg, exp_avg_sq, state, pack=None, lr=-1.0, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, zero_grad=True):
    """``adam_step_graph`` that also writes the first ``pack['map'].shape[0]`` parameters (the coarse model's) into the weight streams
    (sr_adam_step_pack): the update launch of a data-parallel step, issued after the gradient all-reduce."""
    ps = _pack_scatter_struct(pack, pack["map"].shape[0]) if pack is not None else None
    _lib.call("sr_adam_step_pack", _p(_chk(params, "params")), _p(_chk(grads, "grads")), _p(_chk(exp_avg, "exp_avg")), _p(_chk(exp_avg_sq, "exp_avg_sq")),
              params.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(grad_scale), _p(_chk(state, "state")), int(zero_grad),
              0 if pack is None else pack["map"].shape[0], C.byref(ps) if ps is not None else None, _stream())


def adam_step_graph(params, grads, exp_avg, exp_avg_sq, state, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, zero_grad=True):
    _lib.call("sr_adam_step_graph", _p(_chk(params, "params")), _p(_chk(grads, "grads")), _p(_chk(exp_avg, "exp_avg")), _p(_chk(exp_avg_sq, "exp_avg_sq")),
              params.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(grad_scale), _p(_chk(state, "state")), int(zero_grad),
              _stream())
