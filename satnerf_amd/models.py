"""Model containers with the reference's constructor signatures and ``state_dict`` layout.

``SatNeRF`` mirrors ``models/satnerf.py:81-153`` (module tree, parameter names, shapes, construction order and
therefore init RNG stream) so Lightning checkpoints written by the reference load unchanged
(``eval_satnerf.py:82-90``).  The modules are parameter holders only: arithmetic runs in the HIP library.
All parameters are views into ONE flat fp32 buffer (``flat_params()``) -- the unit the weight-stream packer,
the fused optimizer and the data-parallel gradient all-reduce operate on.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import ops, packing


class Siren(nn.Module):
    """Placeholder for ``models/nerf.py:23-33`` (keeps the Sequential indices 0,2,4,... of the reference)."""

    def __init__(self, w0=1.0):
        super().__init__()
        self.w0 = w0

    def forward(self, x):  # pragma: no cover - never executed, the fused kernel applies sin(w0 x)
        raise RuntimeError("satnerf_amd modules hold parameters only; call SatNeRF.forward / render_rays")


def _sine_init(seq, first_only=False):
    """``sine_init`` / ``first_layer_sine_init`` (models/nerf.py:9-21) applied the way ``Module.apply`` walks."""
    with torch.no_grad():
        for m in ([seq[0]] if first_only else list(seq)):
            if hasattr(m, "weight"):
                n_in = m.weight.size(-1)
                if first_only:
                    m.weight.uniform_(-1 / n_in, 1 / n_in)
                else:
                    m.weight.uniform_(-np.sqrt(6 / n_in), np.sqrt(6 / n_in))


class _FlatParamModule(nn.Module):
    """Keeps every parameter a view of one flat buffer; survives ``.to()/.cuda()/.float()``."""

    def _flatten(self, buffer=None):
        """Re-point every parameter into one flat fp32 buffer (``buffer`` = a caller-owned 1-D tensor to adopt, e.g. the
        trainer's combined [model | embedding] buffer)."""
        params = list(self.parameters())
        n = sum(p.numel() for p in params)
        flat = torch.empty(n, dtype=torch.float32, device=params[0].device) if buffer is None else buffer
        assert flat.numel() == n and flat.dtype == torch.float32
        off = 0
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
        self._flat = flat
        self._flat_grad = None
        self._pack_cache = {}
        self._bumps = 0

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def weights_version(self):
        """Changes whenever the weights are modified in place, through a parameter view OR through the flat buffer
        (``p.data = view`` does not share version counters, so both are counted)."""
        return self._bumps + self._flat._version + sum(p._version for p in self.parameters())

    def mark_weights_changed(self):
        """For updaters that write the flat buffer through an alias autograd cannot see (fused optimizers)."""
        self._bumps += 1

    def flat_grads(self, buffer=None):
        """The flat gradient buffer; every ``p.grad`` is (re)bound as a view of it.  A parameter whose ``.grad`` was set
        to None (``zero_grad(set_to_none=True)``) gets its slice zeroed, which is what a fresh gradient means."""
        self.flat_params()
        if buffer is not None or self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat) if buffer is None else buffer
            for p in self.parameters():
                p.grad = None
        off = 0
        base = self._flat_grad.data_ptr()
        for p in self.parameters():
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                view = self._flat_grad[off:off + n].view(p.shape)
                view.zero_()
                p.grad = view
            off += n
        return self._flat_grad

    def flat_params(self):
        params = list(self.parameters())
        first, last = params[0], params[-1]
        ok = (first.data_ptr() == self._flat.data_ptr()
              and last.data_ptr() == self._flat.data_ptr() + (self._flat.numel() - last.numel()) * 4)
        if not ok:  # someone re-pointed a .data: fold the parameters back into one buffer
            self._flatten()
        return self._flat


def _stream_kind(mode):
    """Which packed forward stream a numeric mode reads: bf16 hi plane, bf16 hi + lo planes, or fp16."""
    return {"bf16": "bf16", "bf16x3": "x3", "f16": "f16"}[mode]


class SatNeRF(_FlatParamModule):
    number_of_outputs = 9  # rgb 3, sigma 1, sun visibility 1, sky rgb 3, beta 1 (models/satnerf.py:90)

    def __init__(self, layers=8, feat=256, mapping=False, mapping_sizes=[10, 4], skips=[4], siren=True, t_embedding_dims=16):
        super().__init__()
        if mapping or not siren:
            raise NotImplementedError("Sat-NeRF runs with mapping=False, siren=True (models/__init__.py:12); other variants are not built")
        if layers < 1 or feat < 2 or feat % 2:
            raise ValueError("SatNeRF needs layers >= 1 and an even feat")
        self.layers, self.skips, self.feat = layers, list(skips), feat
        # the fused register-resident kernel is built for the BASELINE shape; everything else (opt.py's default fc_units=512
        # included) runs layer by layer through satnerf_amd.generic
        self.fused = feat == 256 and layers == 8 and list(skips) == [4]
        # ... and a 512-wide build of the fused kernels (opt.py:50's default width): throughput arithmetic + 8-bit workspaces only
        self._fused_wide = feat == 512 and layers == 8 and list(skips) == [4]
        self.t_embedding_dims = t_embedding_dims
        self.mapping = [nn.Identity(), nn.Identity()]  # plain list, not registered (models/satnerf.py:101)
        self.input_sizes = [3, 0]
        self.rgb_padding = 0.001
        half = feat // 2
        nl = Siren()
        fc = [nn.Linear(3, feat), Siren(w0=30.0)]
        for i in range(1, layers):
            fc += [nn.Linear(feat + 3 if i in skips else feat, feat), nl]
        self.fc_net = nn.Sequential(*fc)
        self.sigma_from_xyz = nn.Sequential(nn.Linear(feat, 1), nn.Softplus())
        self.feats_from_xyz = nn.Linear(feat, feat)
        self.rgb_from_xyzdir = nn.Sequential(nn.Linear(feat, half), nl, nn.Linear(half, 3), nn.Sigmoid())
        self.sun_v_net = nn.Sequential(nn.Linear(feat + 3, half), Siren(), nn.Linear(half, half), nl, nn.Linear(half, half), nl,
                                       nn.Linear(half, 1), nn.Sigmoid())
        self.sky_color = nn.Sequential(nn.Linear(3, half), nn.ReLU(), nn.Linear(half, 3), nn.Sigmoid())
        _sine_init(self.fc_net)
        _sine_init(self.fc_net, first_only=True)
        _sine_init(self.sun_v_net)
        _sine_init(self.sun_v_net, first_only=True)
        self._make_beta_head(t_embedding_dims, feat, half, nl)
        self._flatten()

    def _make_beta_head(self, tau, feat, half, nl):
        self.beta_from_xyz = nn.Sequential(nn.Linear(tau + feat, half), nl, nn.Linear(half, 1), nn.Softplus())

    def fused_forward(self, mode):
        """True when a no-grad forward in numeric mode ``mode`` runs in the fused kernel (else: layer by layer, satnerf_amd.generic)."""
        return self.fused or (self._fused_wide and mode in ("bf16", "f16"))  # (256: bf16, f16 and bf16x3; 512: bf16, f16)

    def fused_training(self, mode, fmt):
        """True when forward + backward run in the fused kernels (256: every mode / format; 512: bf16 with the 8-bit workspaces)."""
        if int(fmt) == 32:  # parity-grade backward: the layer-by-layer path (train._fmt_of)
            return False
        return self.fused or (self._fused_wide and mode in ("bf16", "f16") and int(fmt) == 8)

    # ---- weight stream ------------------------------------------------------------------------------------
    def packed(self, mode):
        """(stream_hi, stream_lo | None, l0) for the current weights; rebuilt only when the flat buffer changed."""
        flat = self.flat_params()
        if not flat.is_cuda:
            raise RuntimeError("SatNeRF parameters are on the CPU: move the model to the GPU (.cuda()); there is no CPU path")
        key = _stream_kind(mode)
        ent = self._pack_cache.get(key)
        version = self.weights_version()
        if ent is not None and ent[0] == version and ent[1] == flat.data_ptr():
            return ent[2]
        maps = self._device_maps()
        hi, lo = ops.pack_stream(flat, maps["idx"], maps["scale"], want_lo=key == "x3", f16=key == "f16")
        l0 = ops.gather_scale(flat, maps["l0_idx"], maps["l0_scale"])
        self._pack_cache[key] = (version, flat.data_ptr(), (hi, lo, l0))
        return hi, lo, l0

    def repack(self, mode, backward=False, tick=None):
        """Re-run the pack kernel unconditionally INTO THE SAME device buffers (hipGraph-capturable: fixed addresses, no
        version checks on the captured path) and refresh the caches ``packed`` / ``packed_backward`` consult.  With
        ``backward`` the forward stream, the transposed stream and the fc_net.0 table are produced by ONE launch."""
        flat = self.flat_params()
        key = _stream_kind(mode)
        maps = self._device_maps()
        if backward and "bmaps" not in self._pack_cache:
            self.packed_backward()
        n_f = maps["idx"].numel()
        n_b = self._pack_cache["bmaps"]["idx"].numel() if backward else 0
        ck = ("buf", key, backward)
        bufs = self._pack_cache.get(ck)
        if bufs is None or bufs["hi"].device != flat.device:
            dev = flat.device
            bufs = {"hi": torch.empty(n_f + n_b, dtype=torch.int16, device=dev),
                    "lo": torch.empty(n_f + n_b, dtype=torch.int16, device=dev) if key == "x3" else None,
                    "l0": torch.empty(maps["l0_idx"].numel(), dtype=torch.float32, device=dev)}
            if backward:
                bm = self._pack_cache["bmaps"]
                bufs["idx"], bufs["scale"] = torch.cat([maps["idx"], bm["idx"]]), torch.cat([maps["scale"], bm["scale"]])
            else:
                bufs["idx"], bufs["scale"] = maps["idx"], maps["scale"]
            self._pack_cache[ck] = bufs
        ops.pack_all(flat, bufs["idx"], bufs["scale"], bufs["hi"], bufs["lo"], maps["l0_idx"], maps["l0_scale"], bufs["l0"], tick,
                     n_f16=n_f if key == "f16" else 0)  # (the transposed stream behind it stays bf16: the backward kernels' format)
        version = self.weights_version()
        self._pack_cache[key] = (version, flat.data_ptr(), (bufs["hi"][:n_f], bufs["lo"][:n_f] if key == "x3" else None, bufs["l0"]))
        if backward:
            self._pack_cache["bstream"] = (version, flat.data_ptr(), bufs["hi"][n_f:])

    def pack_scatter(self, mode):
        """What ``ops.grad_tail_adam(pack=...)`` needs so that the launch that updates the parameters ALSO refreshes the buffers
        ``repack(mode, backward=True)`` fills (forward stream | transposed stream, fc_net.0 table): the inverse scatter map of
        ``packing.pack_scatter_map`` on the device plus those buffers.  A captured single-GPU step then has no sr_pack_all launch;
        ``note_packed`` tells the caches afterwards."""
        key = _stream_kind(mode)
        bufs = self._pack_cache.get(("buf", key, True))
        if bufs is None:
            raise RuntimeError("pack_scatter needs the buffers of an earlier repack(mode, backward=True)")
        dev = bufs["hi"].device
        ent = self._pack_cache.get("scatter")
        if ent is None or ent[0].device != dev:
            m, scales = packing.pack_scatter_map(self.feat, self.t_embedding_dims)
            ent = (torch.from_numpy(m).to(dev), [float(x) for x in scales])
            self._pack_cache["scatter"] = ent
        n_f = self._device_maps()["idx"].numel()
        return {"map": ent[0], "scales": ent[1], "hi": bufs["hi"], "lo": bufs["lo"], "l0": bufs["l0"], "n_f16": n_f if key == "f16" else 0}

    def packed_static(self, mode):
        """(hi, lo | None, l0, transposed stream, backward maps) = views of the STATIC buffers ``repack(mode, backward=True)`` fills, without
        a version check: for steps whose own optimizer launch keeps those buffers current (``pack_scatter``)."""
        key = _stream_kind(mode)
        bufs = self._pack_cache.get(("buf", key, True))
        if bufs is None:
            raise RuntimeError("packed_static needs the buffers of an earlier repack(mode, backward=True)")
        n_f = self._device_maps()["idx"].numel()
        return bufs["hi"][:n_f], (bufs["lo"][:n_f] if key == "x3" else None), bufs["l0"], bufs["hi"][n_f:], self._pack_cache["bmaps"]

    def note_packed(self, mode):
        """The buffers of ``repack(mode, backward=True)`` hold the CURRENT weights (the optimizer launch scattered them): stamp the caches
        ``packed`` / ``packed_backward`` consult with the current version, as ``repack`` does after its launch."""
        key = _stream_kind(mode)
        bufs = self._pack_cache[("buf", key, True)]
        flat = self.flat_params()
        n_f = self._device_maps()["idx"].numel()
        version = self.weights_version()
        self._pack_cache[key] = (version, flat.data_ptr(), (bufs["hi"][:n_f], bufs["lo"][:n_f] if key == "x3" else None, bufs["l0"]))
        self._pack_cache["bstream"] = (version, flat.data_ptr(), bufs["hi"][n_f:])

    def _device_maps(self):
        dev = self._flat.device
        ent = self._pack_cache.get("maps")
        if ent is None or ent["idx"].device != dev:
            m = packing.forward_maps(self.feat, self.t_embedding_dims)
            ent = {k: torch.from_numpy(m[k]).to(dev) for k in ("idx", "scale", "l0_idx", "l0_scale")}
            self._pack_cache["maps"] = ent
        return ent

    def packed_backward(self):
        """Transposed (dX) weight stream for the current weights + the weight-gradient job table / scatter maps."""
        flat = self.flat_params()
        ent = self._pack_cache.get("bmaps")
        if ent is None or ent["idx"].device != flat.device:
            m = packing.backward_maps(self.feat, self.t_embedding_dims)
            ent = {k: torch.from_numpy(m[k]).to(flat.device) for k in ("idx", "scale", "blocks", "gidx", "gscale")}
            ent["loads8"] = torch.from_numpy(packing.wgrad8_loads(self.feat, self.t_embedding_dims)).to(flat.device)
            self._pack_cache["bmaps"] = ent
        cached = self._pack_cache.get("bstream")
        version = self.weights_version()
        if cached is None or cached[0] != version or cached[1] != flat.data_ptr():
            hi, _ = ops.pack_stream(flat, ent["idx"], ent["scale"], want_lo=False)
            cached = (version, flat.data_ptr(), hi)
            self._pack_cache["bstream"] = cached
        return cached[2], ent

    # ---- SatNeRF.forward (models/satnerf.py:156-208): points in, (B,9) out ---------------------------------
    def forward(self, input_xyz, input_dir=None, input_sun_dir=None, input_t=None, sigma_only=False, mlp_mode=None):
        from .rendering import default_mode

        if input_sun_dir is None or input_t is None:
            raise TypeError("SatNeRF.forward needs input_sun_dir and input_t (models/satnerf.py:199,204)")
        mode = mlp_mode or default_mode()
        xyz = input_xyz.contiguous().float()
        sun = input_sun_dir.contiguous().float()
        t = input_t.contiguous().float()
        b = xyz.shape[0]
        if self.fused_forward(mode):
            hi, lo, l0 = self.packed(mode)
            albedo, sigma, sun_v, beta = ops.satnerf_mlp(xyz, None, sun, None, t, None, b, 1, self.feat, self.t_embedding_dims, mode, hi, lo, l0)
        else:
            from .generic import satnerf_points

            with torch.no_grad():
                albedo, sigma, sun_v, beta = satnerf_points(self, xyz, sun, t, 1)
        if sigma_only:
            return sigma.unsqueeze(1)
        sk = self.sky_color
        sky = ops.sky(sun, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
        return torch.cat([albedo, sigma.unsqueeze(1), sun_v.unsqueeze(1), sky, beta.unsqueeze(1)], 1)


class ShadowNeRF(SatNeRF):
    """``models.snerf.ShadowNeRF`` (models/snerf.py:78-196): Sat-NeRF without the transient embedding and the uncertainty head --
    same trunk, density / albedo / sun-visibility / sky heads, same ``state_dict`` keys, same init RNG stream.  It runs on the
    Sat-NeRF kernels as is: the uncertainty head exists as frozen ZERO weights (its output is ignored, it receives no gradient
    -- g_beta = 0 -- and Adam leaves zeros at zero), the embedding is a 1-row zero table indexed by ts = 0; the head costs 5 % of
    the FLOPs, which is what a template flag on the fused kernels would save.  ``state_dict`` / ``load_state_dict`` hide the
    dummy head, so reference checkpoints (``nerf_coarse.*`` without ``beta_from_xyz``) load unchanged."""

    number_of_outputs = 8  # rgb 3, sigma 1, sun visibility 1, sky rgb 3 (models/snerf.py:85)

    def __init__(self, layers=8, feat=256, mapping=False, mapping_sizes=[10, 4], skips=[4], siren=True):
        super().__init__(layers=layers, feat=feat, mapping=mapping, mapping_sizes=mapping_sizes, skips=skips, siren=siren, t_embedding_dims=4)
        self._dummy = []  # plain list: the 1-row zero embedding must NOT become a submodule (parameters() / state_dict stay the reference's)
        self._register_state_dict_hook(ShadowNeRF._drop_beta)
        self._register_load_state_dict_pre_hook(self._inject_beta)

    def _make_beta_head(self, tau, feat, half, nl):
        with torch.random.fork_rng(devices=[]):  # the reference's ctor draws nothing here: keep torch's generator where it left it
            super()._make_beta_head(tau, feat, half, nl)
        for p in self.beta_from_xyz.parameters():
            p.data.zero_()
            p.requires_grad_(False)

    @staticmethod
    def _drop_beta(module, state_dict, prefix, local_metadata):
        for k in [k for k in state_dict if k.startswith(prefix + "beta_from_xyz.")]:
            del state_dict[k]
        return state_dict

    def _inject_beta(self, state_dict, prefix, *unused):
        for k, v in self.beta_from_xyz.state_dict().items():
            state_dict.setdefault(prefix + "beta_from_xyz." + k, torch.zeros_like(v))

    def dummy_embedding(self):
        """The 1-row zero embedding the Sat-NeRF kernels index with ts = 0 (it only feeds the dead uncertainty head)."""
        dev = self._flat.device
        if not self._dummy or self._dummy[0].weight.device != dev:
            with torch.random.fork_rng(devices=[]):
                emb = nn.Embedding(1, self.t_embedding_dims)
            emb.weight.data.zero_()
            self._dummy[:] = [emb.to(dev)]
        return self._dummy[0]

    def forward(self, input_xyz, input_dir=None, input_sun_dir=None, sigma_only=False, mlp_mode=None):
        if input_sun_dir is None:
            raise TypeError("ShadowNeRF.forward needs input_sun_dir (models/snerf.py:190)")
        t = torch.zeros(input_xyz.shape[0], self.t_embedding_dims, device=input_xyz.device)
        out = super().forward(input_xyz, input_dir, input_sun_dir, t, sigma_only=sigma_only, mlp_mode=mlp_mode)
        return out if sigma_only else out[:, :8].contiguous()


class NeRF(_FlatParamModule):
    """Classic NeRF (models/nerf.py:135-227): positional encoding, ReLU trunk, direction-conditioned colour head; same
    ``state_dict`` keys.  BASELINE configs[0] is a CPU plumbing run in the reference; here it runs layer by layer through the
    same MFMA GEMM as the non-256 Sat-NeRF widths (satnerf_amd.generic)."""

    number_of_outputs = 4
    fused = False

    def __init__(self, layers=8, feat=256, mapping=True, mapping_sizes=[10, 4], skips=[4], siren=False):
        super().__init__()
        if not mapping or siren:
            raise NotImplementedError("classic NeRF runs with mapping=True, siren=False (models/__init__.py:8)")
        self.layers, self.skips, self.feat = layers, list(skips), feat
        self.mapping_sizes = list(mapping_sizes)
        self.input_sizes = [3, 3]
        self.rgb_padding = 0.001
        in_xyz, in_dir = 2 * mapping_sizes[0] * 3, 2 * mapping_sizes[1] * 3
        nl = nn.ReLU()
        fc = [nn.Linear(in_xyz, feat), nl]
        for i in range(1, layers):
            fc += [nn.Linear(feat + in_xyz if i in skips else feat, feat), nl]
        self.fc_net = nn.Sequential(*fc)
        self.sigma_from_xyz = nn.Sequential(nn.Linear(feat, 1), nn.Softplus())
        self.feats_from_xyz = nn.Linear(feat, feat)
        self.rgb_from_xyzdir = nn.Sequential(nn.Linear(feat + in_dir, feat // 2), nl, nn.Linear(feat // 2, 3), nn.Sigmoid())
        self._flatten()

    def forward(self, input_xyz, input_dir=None, sigma_only=False):
        from .generic import nerf_points

        if input_dir is None and not sigma_only:
            raise TypeError("NeRF.forward needs input_dir (models/nerf.py:213)")
        xyz = input_xyz.contiguous().float()
        with torch.no_grad():
            rgb, sigma = nerf_points(self, xyz, None if input_dir is None else input_dir.contiguous().float(), 1, sigma_only=sigma_only)
        if sigma_only:
            return sigma.unsqueeze(1)
        return torch.cat([rgb, sigma.unsqueeze(1)], 1)


def load_model(args):
    """``models.load_model`` (models/__init__.py:6-15)."""
    if args.model == "sat-nerf":
        return SatNeRF(layers=args.fc_layers, feat=args.fc_units, t_embedding_dims=args.t_embbeding_tau)
    if args.model == "nerf":
        return NeRF(layers=args.fc_layers, feat=args.fc_units)
    if args.model == "s-nerf":
        return ShadowNeRF(layers=args.fc_layers, feat=args.fc_units)
    raise ValueError(f"model {args.model} is not valid")
