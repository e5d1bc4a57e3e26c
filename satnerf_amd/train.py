"""Minimal training harness for the GPU box: the step the reference's Lightning module performs (main.py:81-154).

    results = render_rays(models, args, rays, ts)        # with grad  (main.py:60-75,127)
    loss    = SatNerfLoss(results, rgbs)                 # metrics.py:21-25,56-73 (stays PyTorch)
    loss.backward(); Adam(lr=5e-4).step()                # main.py:83-84
    args.noise_std *= 0.9                                # main.py:132

Data parallelism (new capability): one process per GPU, models replicated, each rank renders its own ray
batch; gradients live in ONE flat fp32 buffer [coarse | fine | embedding] so a step needs exactly one RCCL all-reduce
(2.65 MB) and one fused Adam launch.  On an 8 x MI355X node the xGMI fabric is fully connected, so the small
all-reduce is latency bound; it is issued once per step on the compute stream right after backward.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


_VOTES = 0  # capture votes taken by this process (Trainer._any_rank): the rendezvous store's keys are job-global


def _fmt_of(args):
    """Format of the saved training state (include/satrender.h SR_FMT*): ``args.bwd_fmt`` (32 | 16 | 8) when given, else the
    numeric mode's default -- 8-bit for the throughput mode ``bf16``, 16-bit for the parity mode ``bf16x3``.
    32 = parity-grade backward: fp32 saved activations and 3-pass (hi/lo split) bf16 GEMMs for dX and dW, i.e. the
    layer-by-layer path (satnerf_amd.generic) in ``bf16x3`` -- gradients within 2e-4 of the reference's (the fused backward's
    single-pass bf16 GEMMs: 7e-3), several times slower."""
    from . import ops
    from .rendering import _mode_of

    fmt = getattr(args, "bwd_fmt", None)
    fmt = ops.default_fmt(_mode_of(args)) if fmt is None else int(fmt)
    if fmt not in (8, 16, 32):
        raise ValueError(f"bwd_fmt must be 8, 16 or 32, got {fmt}")
    if fmt == 32 and _mode_of(args) != "bf16x3":
        raise ValueError("bwd_fmt=32 (parity-grade backward) belongs to mlp_mode='bf16x3'")
    if fmt == 8 and _mode_of(args) not in ("bf16", "f16"):
        raise ValueError("the 8-bit workspace format belongs to mlp_mode='bf16' / 'f16'")
    return fmt


def quiet_gc():
    """Call once after set-up (models built, first step captured): ``gc.collect()`` then ``gc.freeze()``, so that the cyclic
    collector's full passes no longer walk torch's ~10^6 long-lived objects.  Each such pass pauses the host for tens of
    milliseconds -- long enough for the queued graph replays to drain and the GPU to idle (0.415 -> 0.53 ms per training step
    on MI355X, profiles/r02_ab_variants.txt)."""
    import gc

    gc.collect()
    gc.freeze()


def satnerf_loss(res, target, lambda_sc=0.0, beta_min=0.05):
    """``metrics.SatNerfLoss`` for the coarse model (metrics.py:21-34,56-73)."""
    beta = torch.sum(res["weights_coarse"].unsqueeze(-1) * res["beta_coarse"], -2) + beta_min
    loss = ((res["rgb_coarse"] - target) ** 2 / (2 * beta ** 2)).mean() + (3 + torch.log(beta).mean()) / 2
    if lambda_sc > 0:
        sun_sc = res["sun_sc_coarse"].squeeze(-1)
        term2 = torch.sum(torch.square(res["transparency_sc_coarse"].detach() - sun_sc), -1)
        term3 = 1 - torch.sum(res["weights_sc_coarse"].detach() * sun_sc, -1)
        loss = loss + lambda_sc / 3.0 * torch.mean(term2) + lambda_sc / 3.0 * torch.mean(term3)
    return loss


def snerf_loss(res, target, lambda_sc=0.0):
    """``metrics.SNerfLoss`` for the coarse model (metrics.py:36-54): plain MSE + the solar-correction terms; what a sat-nerf
    run trains with during its first two epochs (main.py:128-131)."""
    loss = torch.mean((res["rgb_coarse"] - target) ** 2)
    if lambda_sc > 0:
        sun_sc = res["sun_sc_coarse"].squeeze(-1)
        term2 = torch.sum(torch.square(res["transparency_sc_coarse"].detach() - sun_sc), -1)
        term3 = 1 - torch.sum(res["weights_sc_coarse"].detach() * sun_sc, -1)
        loss = loss + lambda_sc / 3.0 * torch.mean(term2) + lambda_sc / 3.0 * torch.mean(term3)
    return loss


def nerf_loss(res, target):
    """``metrics.NerfLoss`` (metrics.py:8-19): MSE of the coarse [+ fine] colour."""
    loss = torch.mean((res["rgb_coarse"] - target) ** 2)
    if "rgb_fine" in res:
        loss = loss + torch.mean((res["rgb_fine"] - target) ** 2)
    return loss


def depth_loss(res, targets, weights=1.0, lambda_ds=1.0):
    """``metrics.DepthLoss`` (metrics.py:75-92): lambda_ds/3 * mean(weights * (depth - target)^2), coarse [+ fine]."""
    lam = lambda_ds / 3.0
    loss = lam * torch.mean(weights * (res["depth_coarse"] - targets) ** 2)
    if "depth_fine" in res:
        loss = loss + lam * torch.mean(weights * (res["depth_fine"] - targets) ** 2)
    return loss


class FlatState:
    """One flat parameter buffer and one flat gradient buffer shared by a list of modules (models + embedding)."""

    def __init__(self, modules):
        self.modules = list(modules)
        sizes = [sum(p.numel() for p in m.parameters()) for m in self.modules]
        dev = next(self.modules[0].parameters()).device
        self.params = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        off = 0
        for m, n in zip(self.modules, sizes):
            pslice, gslice = self.params[off:off + n], self.grads[off:off + n]
            if hasattr(m, "_flatten"):  # satnerf_amd model: adopt the slices as its flat buffers
                m._flatten(buffer=_filled(pslice, m))
                m.flat_grads(buffer=gslice)
            else:  # plain module (nn.Embedding)
                o = 0
                for p in m.parameters():
                    k = p.numel()
                    pslice[o:o + k].copy_(p.data.reshape(-1))
                    p.data = pslice[o:o + k].view(p.shape)
                    p.grad = gslice[o:o + k].view(p.shape)
                    o += k
            off += n

    def zero_grad(self):
        self.grads.zero_()

    def allreduce_mean_(self, world_size):
        """Sum the flat gradient over ranks and divide by the world size (losses are batch means, metrics.py:11,23-24)."""
        if world_size > 1:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM)
            self.grads.mul_(1.0 / world_size)


def _filled(dst, module):
    flat = torch.cat([p.data.reshape(-1) for p in module.parameters()])
    dst.copy_(flat)
    return dst


def shard_rays(n_total, rank, world_size):
    """Contiguous, balanced ray ranges for evaluation sharding: returns (start, stop)."""
    base, rem = divmod(n_total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class _LazyLoss:
    """The fused loss kernel leaves per-block partial sums; the scalar is only reduced when somebody looks at it."""

    def __init__(self, parts):
        self.parts = parts

    def item(self):
        return float(self.parts.sum().item())

    def detach(self):
        return self.parts.sum()

    def __float__(self):
        return self.item()


class Trainer:
    """One training step = main.py:119-154: the sat-nerf colour batch and, when ``args.ds_lambda > 0`` and a depth batch is
    passed, the depth-supervision batch (main.py:134-141, metrics.DepthLoss) -- both backward passes accumulate into the same
    flat gradient buffer before the single all-reduce + Adam update.

    Fast path (default loss, no fine model; solar correction and depth supervision included): the step calls the HIP kernels directly -- no autograd
    graph, fused loss+gradient kernel, fused Adam over the flat buffer -- and, when ``noise_std == 0``, replays the whole
    forward+backward from ONE hipGraph (the step is ~25 launches of 5-350 us; eager launch gaps would dominate).
    Anything else goes through ``render_rays`` + autograd + the same flat buffers.

    Saved state: ``mlp_mode`` 'bf16' and 'f16' keep the activations / pre-activation gradients between the forward, dX and
    weight-gradient kernels in the 8-BIT workspaces (PHASE8 / MX8, csrc/mlp_layout.h) by default -- gradients within 2.1e-2 /
    1.8e-2 of the reference's instead of 1.4e-2 / 7.3e-3 with the 16-bit state; pass ``args.bwd_fmt = 16`` for the latter
    (~0.53 instead of ~0.42 ms per step), ``mlp_mode='bf16x3'`` (16-bit state) for outputs within 1e-4, ``bwd_fmt = 32`` for
    gradients within 2e-4 (``_fmt_of``).
    """

    def __init__(self, models, args, world_size=1, lr=5e-4, loss_fn=None, use_graph=True, steps_per_epoch=None, lr_gamma=0.9,
                 warmup_epochs=2, lr_steps_per_epoch=None):
        """``steps_per_epoch`` (= len(dataset) // batch_size, train_utils.py:14-15) switches the reference's schedule on:
        StepLR(step_size=1, gamma=``lr_gamma``) per epoch (main.py:86-94, train_utils.py:41-57) and ``metrics.SNerfLoss`` instead
        of ``SatNerfLoss`` while epoch < ``warmup_epochs`` (main.py:128-131 hard-codes 2).  None = constant rate, SatNerfLoss
        from the first step.  ``lr_steps_per_epoch`` (default: ``steps_per_epoch``) is the number of batches after which Lightning
        steps the StepLR: a DataLoader epoch has ceil(len(dataset) / batch_size) batches (drop_last=False, main.py:96-110) while the
        warm-up test divides by the floor -- pass both when the batch size does not divide the dataset."""
        self._snerf = args.model == "s-nerf"
        self._caller_args = args  # main.py:132 decays args.noise_std in place: the caller's object follows, also for s-nerf's private copy
        if args.model == "sat-nerf" and steps_per_epoch is None:
            import warnings

            warnings.warn("Trainer(steps_per_epoch=None): constant learning rate and SatNerfLoss from the first step; main.py decays the rate "
                          "by 0.9 per epoch and trains sat-nerf with SNerfLoss for 2 epochs -- pass steps_per_epoch = len(dataset) // batch_size",
                          stacklevel=2)
        if self._snerf:  # s-nerf trains on the Sat-NeRF kernels: dead uncertainty head, 1-row zero embedding, SNerfLoss throughout
            from .rendering import _as_satnerf

            models, args = _as_satnerf(models, args)
        self.models, self.args, self.world, self.lr = models, args, world_size, lr
        # the gradient all-reduce runs when there is more than one rank -- or, with SATNERF_FORCE_ALLREDUCE=1, also on a 1-rank
        # process group (how the single-GPU test box exercises the captured RCCL path)
        self._collective = world_size > 1 or (os.environ.get("SATNERF_FORCE_ALLREDUCE", "0") == "1" and dist.is_available() and dist.is_initialized())
        self.lr0, self.lr_gamma, self.steps_per_epoch, self.warmup_epochs = lr, lr_gamma, steps_per_epoch, warmup_epochs
        self.lr_steps_per_epoch = lr_steps_per_epoch if lr_steps_per_epoch else steps_per_epoch
        if args.model == "sat-nerf" and args.n_importance > 0 and loss_fn is None:
            # metrics.py:22 multiplies weights_fine (N,S+I,1) with beta_coarse (N,S,1): the reference cannot train this combination
            # either; training only the coarse model while eval renders the fine one would be silently wrong
            raise NotImplementedError("sat-nerf with n_importance > 0 has no trainable reference loss (metrics.py:22 shape error): pass loss_fn")
        mods = [models["coarse"]] + ([models["fine"]] if "fine" in models else []) + ([models["t"]] if "t" in models else [])
        self.state = FlatState(mods)
        p = self.state.params
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(p), torch.zeros_like(p)
        self.n_steps = 0
        # device-side schedule block (include/satrender.h `sched`): [0] step counter, ticked by the step's first launch (the in-graph
        # Adam's step count and the step of the in-kernel jitter RNG of captured steps), [1] learning rate, [2] SNerfLoss warm-up flag
        self.adam_state = torch.zeros(4, dtype=torch.float32, device=p.device)
        self.sched = self.adam_state
        self._sched_host = None
        self._apply_schedule()
        self._kernel_rng = False
        self._seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        self._adam_in_graph = False
        self.loss_fn = loss_fn
        from .rendering import _mode_of

        self.direct = (loss_fn is None and p.is_cuda and args.n_importance == 0 and args.model == "sat-nerf"
                       and hasattr(models["coarse"], "fused_training") and models["coarse"].fused_training(_mode_of(args), _fmt_of(args)))
        self.use_graph = use_graph and self.direct
        # parameters whose gradients arrive by float atomics (no weight-gradient GEMM produces them: the sky head; the embedding rows
        # behind the model in the flat buffer): the fused tail + Adam launch updates them from its last atomics block
        self._late_idx = None
        if self.direct and mods[0] is models["coarse"] and len(mods) == 2 and mods[1] is models.get("t"):
            import numpy as np

            from . import packing

            gidx = packing.backward_maps(models["coarse"].feat, models["coarse"].t_embedding_dims)["gidx"]
            late = np.concatenate([np.nonzero(gidx < 0)[0], np.arange(gidx.size, p.numel())])
            self._late_idx = torch.from_numpy(late.astype(np.int32)).to(p.device)
        self._graph, self._static, self._graph_banks = None, None, None
        self._pre_setup, self._pre_bufs = None, None
        self._pack_in_tail, self._gather_in_fwd, self._packed_version = False, None, None
        self.max_inflight = int(os.environ.get("SATNERF_MAX_INFLIGHT", "0"))
        self.pace_every = max(1, int(os.environ.get("SATNERF_PACE_EVERY", "1")))
        self.last_rgb = None
        self.last_loss = None
        # stratified jitter of the eager step (rendering.py:77): torch's generator by default; tools/convergence.py substitutes the
        # reference run's draws so that two trainings differ by their arithmetic only
        self.jitter = lambda n, s, device: torch.rand(n, s, device=device)

    # ---- forward + loss + backward on the current stream, gradients accumulate into the flat buffer -------------------
    def _forward_backward(self, rays, ts, rgbs, depth=None):
        """Colour pass [+ depth-supervision pass] [+ in-graph Adam]; returns the per-block loss partial sums."""
        from . import ops
        from .rendering import _mode_of

        if self._snerf:  # (also inside a captured step, whose static ts may hold a bank's image ids)
            ts = self._zero_ts(ts)
            if depth is not None:
                depth = (depth[0], self._zero_ts(depth[1]), depth[2])
        model, emb = self.models["coarse"], self.models["t"]
        args = self.args
        n, s = rays.shape[0], args.n_samples
        mode = _mode_of(args)
        feat, tau = model.feat, model.t_embedding_dims
        ticking = self._kernel_rng or self._adam_in_graph
        # r05, single-GPU captured step: no sr_pack_all launch -- the launch that updates the parameters (sr_grad_tail_adam) also writes
        # them into the weight streams, and the forward, now the step's first launch, ticks the step counter itself ("tick first")
        pit = self._pack_in_tail and ticking
        if pit:
            hi, lo, l0, bstream, maps = model.packed_static(mode)
        else:
            model.repack(mode, backward=True, tick=self.adam_state if ticking else None)
            hi, lo, l0 = model.packed(mode)
            bstream, maps = model.packed_backward()
        sk = model.sky_color
        # stratified jitter (rendering.py:77): torch's generator when run eagerly; inside a captured step the kernel draws it
        # itself (Philox keyed by the seed, stepping with the device counter) -- one launch and the graph's RNG bookkeeping less
        u = None if self._kernel_rng else self.jitter(n, s, rays.device)
        noise_std = float(args.noise_std)
        # models/satnerf.py:58 draws randn even when noise_std == 0; the draw is skipped then (results are identical)
        nz = torch.randn(n, s, device=rays.device) if noise_std != 0 else None
        fmt = _fmt_of(args)
        acts = ops.acts_workspace(n * s, feat, rays.device, fmt)
        sc_on = float(getattr(args, "sc_lambda", 0.0)) > 0
        # ONE launch for the forward (sr_satnerf_render_train): stratified depths + sky head in the prologue, MLP saving the 8-bit
        # state, compositing + colour loss + compositing backward in the epilogue -- r04 ran sr_ray_setup, the MLP and sr_render_loss
        # as three launches; the per-ray functions are the same, the results bit-identical.  SATNERF_TRAIN_FUSED=0: the three launches (A/B)
        fused = self._pre_setup is None and self._fused_forward()
        gather, self._gather_in_fwd = self._gather_in_fwd, None
        if pit and not fused:
            raise RuntimeError("pack-in-tail steps need the one-launch training forward (it ticks the step counter)")
        if fused:
            # gather: the captured step samples its batch INSIDE this launch (the bank's cursor over the epoch's shuffled rows): rays / ts /
            # rgbs -- the graph's static batch tensors -- are then OUTPUTS of the launch, read by the step's later launches
            src = (gather["bank"].rays, gather["bank"].ts, gather["bank"].rgbs) if gather is not None else (rays, ts, rgbs)
            r = ops.render_train(src[0], src[1], emb.weight.data, s, feat, tau, mode, hi, lo, l0, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data,
                                 sk[2].bias.data, src[2], acts, u=u, noise=nz, noise_std=noise_std, seed=self._seed, step_counter=self.adam_state,
                                 sched=self.sched, want_z=sc_on, tick=2 if pit else 0,
                                 gather=None if gather is None else dict(idx=gather["idx"], cursor=gather["cursor"], batches=gather["batches"],
                                                                         out=(rays, rgbs, ts)))
            z, sky, loss, self.last_rgb = r["z"], r["sky"], r["loss"], r["rgb"]
            albedo, sigma, sun_v, beta = r["albedo"].view(-1, 3), r["sigma"].view(-1), r["sun_v"].view(-1), r["beta"].view(-1)
            d_sigma, d_albedo, d_sun, g_beta, d_sky = r["d_sigma"], r["d_albedo"], r["d_sun"], r["g_beta"], r["d_sky"]
        else:
            if self._pre_setup is not None:  # a captured step whose gather launch already produced them (_gather_from_banks)
                (z, sky), self._pre_setup = self._pre_setup, None
            else:
                z, sky = ops.ray_setup(rays, u, s, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data, seed=self._seed,
                                       step_counter=self.adam_state)
            albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, n * s, s, feat, tau, mode,
                                                         hi, lo, l0, acts=acts, fmt=fmt)
            if s <= 64:  # one launch: compositing forward -> loss -> compositing backward
                loss, self.last_rgb, d_sigma, d_albedo, d_sun, g_beta, d_sky = ops.render_loss(z, sigma.view(n, s), nz, noise_std, albedo.view(n, s, 3),
                                                                                               sun_v.view(n, s), beta.view(n, s), sky, rgbs,
                                                                                               sched=self.sched)
            else:
                weights, transp, _, rgb = ops.composite(z, sigma.view(n, s), nz, noise_std, albedo.view(n, s, 3), sun_v.view(n, s), sky)
                loss, g_rgb, g_w, g_beta = ops.satnerf_loss(rgb, weights, beta.view(n, s), rgbs, sched=self.sched)
                d_sigma, d_albedo, d_sun, d_sky = ops.composite_bwd(z, sigma.view(n, s), nz, noise_std, albedo.view(n, s, 3), sun_v.view(n, s), sky,
                                                                   weights, transp, g_rgb, None, g_w, None)
                self.last_rgb = rgb
        dpre, d_t = ops.satnerf_mlp_bwd(feat, tau, n * s, bstream, acts, albedo, sigma, sun_v, beta, d_albedo, d_sigma, d_sun, g_beta.view(-1), fmt=fmt)
        partial, plan = ops.wgrad_partials(feat, tau, n * s, dpre, acts, maps["blocks"], fmt, maps["loads8"])
        # the colour pass's gradient tail comes LAST: the solar-correction and depth-supervision passes accumulate their weight gradients
        # into the (zeroed) flat buffer first, the tail adds the colour pass's on top -- and, on a single GPU under graph capture, applies
        # Adam in the same launch (sr_grad_tail_adam: the thread that reduces a parameter's split-K slices updates it and zeroes its
        # gradient; r04 ran sr_grad_tail and sr_adam_step_graph back to back over the same flat buffers)
        if sc_on:
            loss = torch.cat([loss.view(-1), self._sc_pass(rays, ts, z, noise_std).view(-1)])
        if depth is not None:
            loss = torch.cat([loss.view(-1), self._depth_pass(*depth, noise_std * 0.9).view(-1)])  # main.py:132 decays the noise first
        tail = (partial, plan, maps["gidx"], maps["gscale"], model.flat_grads(), rays[:, 8:11], sk[0].weight.data, sk[0].bias.data,
                sk[2].weight.data, sky, d_sky, sk[0].weight.grad, sk[0].bias.grad, sk[2].weight.grad, sk[2].bias.grad, d_t, ts, n, s, tau,
                emb.weight.grad)
        if self._adam_in_graph and not self._collective and self._late_idx is not None and os.environ.get("SATNERF_TAIL_ADAM", "1") != "0":
            # lr < 0: the kernel reads the current rate from sched[1], so a scheduler can change it under graph replay
            ops.grad_tail_adam(*tail, self.state.params, self.exp_avg, self.exp_avg_sq, self._late_idx, self.adam_state, lr=-1.0,
                               grad_scale=1.0 / self.world, pack=model.pack_scatter(mode) if pit else None)
            return loss
        if pit and not self._collective:
            raise RuntimeError("single-GPU pack-in-tail steps end in sr_grad_tail_adam")
        # data parallel (r06): the N > 1 step is the N = 1 step split at the collective -- sr_grad_tail (the reduction) | all-reduce of the flat
        # gradient | sr_adam_step_pack (Adam + the re-pack of the weight streams, `_update_and_pack`): no sr_pack_all, no separate Adam launch
        ops.grad_tail(*tail)
        if self._adam_in_graph:  # the update rides in the same graph (the RCCL all-reduce captured with it, or the A/B switch above)
            if self._collective:
                dist.all_reduce(self.state.grads, op=dist.ReduceOp.SUM)
            if pit:
                self._update_and_pack()
            else:
                ops.adam_step_graph(self.state.params, self.state.grads, self.exp_avg, self.exp_avg_sq, self.adam_state, lr=-1.0,
                                    grad_scale=1.0 / self.world, zero_grad=True)
        return loss

    def _update_and_pack(self):
        """(data-parallel pack-in-tail steps, after the all-reduce) Adam over the flat buffers with the device-side step count / rate, each
        coarse-model parameter written into the weight streams by the same launch."""
        from . import ops
        from .rendering import _mode_of

        ops.adam_step_pack(self.state.params, self.state.grads, self.exp_avg, self.exp_avg_sq, self.adam_state,
                           pack=self.models["coarse"].pack_scatter(_mode_of(self.args)), lr=-1.0, grad_scale=1.0 / self.world, zero_grad=True)

    def _fused_forward(self):
        """True when the colour pass's forward is ONE launch (sr_satnerf_render_train): 8-bit saved state, <= 64 samples dividing a
        workgroup's points, a generated-core build for (width, mode)."""
        from . import ops
        from .rendering import _mode_of

        s, mode = self.args.n_samples, _mode_of(self.args)
        return (_fmt_of(self.args) == 8 and s <= 64 and ops.render_fused_ok(self.models["coarse"].feat, mode, s)
                and os.environ.get("SATNERF_TRAIN_FUSED", "1") != "0" and os.environ.get("SATNERF_FWD_V1", "0") != "1")

    def _sc_pass(self, rays, ts, z, noise_std):
        """Solar correction (rendering.py:102-108, metrics.py:27-34): the SAME depths along the sun direction; transparency and
        weights of that pass are detached, so only sun visibility receives a gradient."""
        from . import ops
        from .rendering import _mode_of

        model, emb, args = self.models["coarse"], self.models["t"], self.args
        n, s = z.shape
        mode = _mode_of(args)
        feat, tau = model.feat, model.t_embedding_dims
        hi, lo, l0 = model.packed(mode)
        bstream, maps = model.packed_backward()
        nz = torch.randn(n, s, device=rays.device) if noise_std != 0 else None
        fmt = _fmt_of(args)
        acts = ops.acts_workspace(n * s, feat, rays.device, fmt)
        albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, 8:11], rays[:, 8:11], z, emb.weight.data, ts, n * s, s, feat, tau, mode,
                                                     hi, lo, l0, acts=acts, fmt=fmt)
        loss, d_sun = ops.sc_loss(z, sigma.view(n, s), nz, noise_std, sun_v.view(n, s), float(args.sc_lambda))
        dpre, _ = ops.satnerf_mlp_bwd(feat, tau, n * s, bstream, acts, albedo, sigma, sun_v, beta, None, None, d_sun, None, want_dt=False, fmt=fmt)
        ops.satnerf_wgrad(feat, tau, n * s, dpre, acts, maps["blocks"], maps["gidx"], maps["gscale"], model.flat_grads(), accumulate=True, fmt=fmt,
                          loads=maps["loads8"])
        return loss

    def _depth_pass(self, rays, ts, depths, noise_std):
        """Depth supervision (main.py:134-141): render the depth batch, loss = ds_lambda/3 * mean(w * (depth - target)^2)
        (metrics.py:75-92); only sigma receives a gradient, so the MLP backward runs with the other head gradients absent."""
        from . import ops
        from .rendering import _mode_of

        model, emb, args = self.models["coarse"], self.models["t"], self.args
        n, s = rays.shape[0], args.n_samples
        mode = _mode_of(args)
        feat, tau = model.feat, model.t_embedding_dims
        hi, lo, l0 = model.packed(mode)
        bstream, maps = model.packed_backward()
        sk = model.sky_color
        u = None if self._kernel_rng else self.jitter(n, s, rays.device)
        nz = torch.randn(n, s, device=rays.device) if noise_std != 0 else None
        z, sky = ops.ray_setup(rays, u, s, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data, seed=self._seed + 1,
                               step_counter=self.adam_state)
        fmt = _fmt_of(args)
        acts = ops.acts_workspace(n * s, feat, rays.device, fmt)
        albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, n * s, s, feat, tau, mode,
                                                     hi, lo, l0, acts=acts, fmt=fmt)
        weights, transp, depth, _ = ops.composite(z, sigma.view(n, s), nz, noise_std, albedo.view(n, s, 3), sun_v.view(n, s), sky)
        loss, g_depth = ops.depth_loss(depth, depths, float(args.ds_lambda), use_weights=not getattr(args, "ds_noweights", False))
        d_sigma, _, _, _ = ops.composite_bwd(z, sigma.view(n, s), nz, noise_std, albedo.view(n, s, 3), sun_v.view(n, s), sky, weights, transp,
                                             None, g_depth, None, None)
        dpre, _ = ops.satnerf_mlp_bwd(feat, tau, n * s, bstream, acts, albedo, sigma, sun_v, beta, None, d_sigma, None, None, want_dt=False, fmt=fmt)
        ops.satnerf_wgrad(feat, tau, n * s, dpre, acts, maps["blocks"], maps["gidx"], maps["gscale"], model.flat_grads(), accumulate=True, fmt=fmt,
                          loads=maps["loads8"])
        return loss

    # ---- schedule (main.py:86-94,128-131) ------------------------------------------------------------------------------------
    def current_epoch(self):
        """``NeRF_pl.get_current_epoch`` (train_utils.py:14-15) of the step about to run: main.py:121 counts the step first
        (``self.train_steps += 1``), so step k (0-based) sees epoch (k + 1) // steps_per_epoch."""
        return 0 if not self.steps_per_epoch else (self.n_steps + 1) // int(self.steps_per_epoch)

    def _pace(self):
        """Optional bound on the graph replays the host keeps in flight (``max_inflight`` / SATNERF_MAX_INFLIGHT, default 0 =
        unbounded): an event is recorded every ``pace_every``-th step and the host waits for the one recorded ``max_inflight`` steps earlier
        (events are created up front: creating one costs milliseconds).  Bounded latency for interactive use; it does not change
        the throughput (profiles/r02_ab_variants.txt: the 0.54 ms episodes seen 50-200 ms into a run are the box's power
        management, with or without a bound)."""
        every = self.pace_every
        k = self.max_inflight // every
        if k <= 0 or self.n_steps % every:
            return
        ring = self.__dict__.get("_pace_ring")
        if ring is None:
            ring = self._pace_ring = [torch.cuda.Event() for _ in range(k)]
            for ev in ring:
                ev.record()
        ev = ring[(self.n_steps // every) % k]
        ev.synchronize()  # recorded max_inflight steps ago, before that step's replay: every earlier replay has completed
        ev.record()

    def _zero_ts(self, ts):
        n = ts.shape[0]
        cache = self.__dict__.setdefault("_zero_ts_cache", {})
        if n not in cache:
            cache[n] = torch.zeros(n, dtype=torch.int64, device=self.state.params.device)
        return cache[n]

    def warming_up(self):
        """True while the reference trains with SNerfLoss (main.py:128: sat-nerf, epoch < 2)."""
        if self._snerf:  # metrics.load_loss: s-nerf trains with SNerfLoss from start to end (metrics.py:94-98)
            return True
        return bool(self.steps_per_epoch) and self.args.model == "sat-nerf" and self.current_epoch() < self.warmup_epochs

    def _apply_schedule(self):
        """Write [lr, warm-up flag] of the step about to run into the device-side schedule block when they changed (once per epoch)."""
        if self.steps_per_epoch:
            # StepLR(step_size=1, gamma) is stepped by Lightning AFTER an epoch's last batch: step k (0-based) trains at
            # gamma ** (k // steps_per_epoch); main.py:121's += 1 only shifts the SNerfLoss test (current_epoch / warming_up)
            self.lr = self.lr0 * self.lr_gamma ** (self.n_steps // int(self.lr_steps_per_epoch))
        host = (float(self.lr), 1.0 if self.warming_up() else 0.0)
        if host != self._sched_host:
            self.adam_state[1:3].copy_(torch.tensor(host, dtype=torch.float32))
            self._sched_host = host

    def set_lr(self, lr):
        self.lr0 = self.lr = float(lr)
        self._apply_schedule()

    def save_ckpt(self, path):
        """A checkpoint ``eval_satnerf.load_nerf`` / ``checkpoint.load_ckpt`` can read: Lightning's ``state_dict`` key prefixes
        (main.py:51-58: nerf_coarse. / nerf_fine. / embedding_t.) plus the optimizer moments and the step count."""
        sd = {}
        for prefix, key in (("nerf_coarse.", "coarse"), ("nerf_fine.", "fine"), ("embedding_t.", "t")):
            if key in self.models and not (self._snerf and key == "t"):
                sd.update({prefix + k: v.detach().cpu().clone() for k, v in self.models[key].state_dict().items()})
        torch.save({"state_dict": sd, "global_step": self.n_steps, "epoch": self.current_epoch(),
                    "optimizer": {"exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "lr": self.lr, "step": self.n_steps,
                                  "lr0": self.lr0, "device_step": float(self.adam_state[0].item()), "noise_std": float(self.args.noise_std),
                                  "jitter_seed": int(self._seed)}}, path)

    def load_ckpt(self, path):
        """Resume (main.py:251 ``resume_from_checkpoint``): restore what ``save_ckpt`` wrote -- weights by Lightning's key prefixes, the
        Adam moments, the step count (host side: schedule; device side: bias corrections and the jitter stream of captured steps) -- so that
        the next step is the one the saved run would have taken.  A checkpoint without an ``optimizer`` entry (a reference ``epoch=N.ckpt``:
        Lightning keeps its optimizer state under other keys) restores the weights and the step count only; the moments restart at zero.
        NOT part of a checkpoint: the data position (a RayBank's epoch permutation and cursor) -- as in the reference, whose resumed
        DataLoader reshuffles; "the step the saved run would have taken" therefore holds for batches the caller supplies."""
        # weights_only=False as in checkpoint.load_ckpt: a Lightning checkpoint pickles its hparams Namespace and callback states
        ck = torch.load(path, map_location="cpu", weights_only=False)
        sd = ck["state_dict"]
        for prefix, key in (("nerf_coarse.", "coarse"), ("nerf_fine.", "fine"), ("embedding_t.", "t")):
            if key in self.models and not (self._snerf and key == "t"):
                part = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
                if not part:
                    raise KeyError(f"checkpoint holds no '{prefix}*' entries")
                self.models[key].load_state_dict(part)
        self.state.zero_grad()
        self.n_steps = int(ck.get("global_step", 0))
        opt = ck.get("optimizer")
        if opt is not None and "exp_avg" in opt:
            if opt["exp_avg"].numel() != self.exp_avg.numel():
                raise ValueError("the checkpoint's optimizer state does not fit this model set")
            self.exp_avg.copy_(opt["exp_avg"]), self.exp_avg_sq.copy_(opt["exp_avg_sq"])
            self.lr0 = float(opt.get("lr0", self.lr0))
            self.adam_state[0] = float(opt.get("device_step", opt.get("step", self.n_steps)))
            if "jitter_seed" in opt and int(opt["jitter_seed"]) != self._seed:
                self._seed = int(opt["jitter_seed"])  # the in-kernel jitter is keyed by (seed, step): the graph holds the seed as a constant
                self._graph = None
            if "noise_std" in opt:
                self.args.noise_std = float(opt["noise_std"])
                if self._caller_args is not self.args:
                    self._caller_args.noise_std = self.args.noise_std
        else:
            self.exp_avg.zero_(), self.exp_avg_sq.zero_()
            self.adam_state[0] = float(self.n_steps)
        self.adam_state.view(torch.int32)[3] = 0  # (the arrival counter of the tick / tail launches)
        self._sched_host = None
        self._apply_schedule()
        for m in self.state.modules:  # the packed weight streams follow at the next step (version check / repack)
            if hasattr(m, "mark_weights_changed"):
                m.mark_weights_changed()

    def _gather_from_banks(self):
        """(inside the captured step) every bank gathers its next batch into the static inputs and moves its device cursor on"""
        from . import ops

        for k, b in enumerate(self._graph_banks):
            idx, cursor, batches = b.graph_source()
            out = self._static[3 * k:3 * k + 3]
            if (k == 0 and self._kernel_rng and self._fused_forward() and not self._snerf and type(b).__name__ == "RayBank"
                    and os.environ.get("SATNERF_GATHER_IN_FWD", "1") != "0"):
                # the colour batch is sampled by the forward launch itself (sr_satnerf_render_train's gather): no launch here
                self._gather_in_fwd = dict(bank=b, idx=idx, cursor=cursor, batches=batches)
            elif k == 0 and self._kernel_rng and not self._fused_forward():
                # the colour batch: gather + stratified depths + sky colour in ONE launch (sr_gather_setup); _forward_backward then
                # skips its ray set-up launch (when the forward is not the one-launch training render, which sets the rays up itself).
                # step_offset 1: this runs before sr_pack_all ticks the step counter
                model, n, s = self.models["coarse"], out[0].shape[0], self.args.n_samples
                if self._pre_bufs is None or self._pre_bufs[0].shape != (n, s):
                    dev = out[0].device
                    self._pre_bufs = (torch.empty(n, s, device=dev), torch.empty(n, 3, device=dev))
                sk = model.sky_color
                ops.gather_setup(b.rays, b.rgbs, b.ts, idx, out, s, sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data,
                                 self._pre_bufs[0], self._pre_bufs[1], self._seed, self.adam_state, step_offset=1, cursor=cursor, batches=batches)
                self._pre_setup = self._pre_bufs
            else:
                ops.gather_batch(b.rays, b.rgbs, b.ts, idx, out=out, cursor=cursor, batches=batches)

    def _capture(self, inputs, banks=None):
        self._static = tuple(t.clone() for t in inputs)
        self._graph_banks = tuple(banks) if banks else None
        # data parallel: with the RCCL backend ("nccl") the gradient all-reduce and the Adam update CAN be captured into the step's graph
        # (NCCL / RCCL collectives are capturable): a step is then one replay, no eager launches between steps.  It is OPT-IN
        # (SATNERF_GRAPH_ALLREDUCE=1) until a job with two or more GPUs has run it: so far it has only executed on a 1-rank group
        # (tests/test_hip_training.py), and a collective that misbehaves inside a replayed graph hangs the job instead of raising
        # (ADVICE r03).  Default, other backends (gloo cannot be captured), or a capture that fails on any rank: the eager all-reduce +
        # Adam issued after the replay.
        capture_collective = (self._collective and os.environ.get("SATNERF_GRAPH_ALLREDUCE", "0") == "1" and dist.is_initialized()
                              and dist.get_backend() == "nccl" and not getattr(self, "_collective_capture_failed", False))
        self._adam_in_graph = (not self._collective) or capture_collective
        self._kernel_rng = float(self.args.noise_std) == 0.0  # (a noisy step still draws randn from torch's generator)
        # the launch that updates the parameters re-packs the weight streams and the forward opens the step (no sr_pack_all launch): on a
        # single GPU that launch is the gradient tail (sr_grad_tail_adam), with a collective it is sr_adam_step_pack behind the all-reduce
        # (in the graph when the collective is captured, eagerly after the replay otherwise).  SATNERF_DP_PACK=0: the r05 N > 1 step (A/B)
        self._pack_in_tail = (self._late_idx is not None and self._kernel_rng and self._fused_forward()
                              and os.environ.get("SATNERF_TAIL_ADAM", "1") != "0" and os.environ.get("SATNERF_TAIL_PACK", "1") != "0"
                              and (not self._collective or os.environ.get("SATNERF_DP_PACK", "1") != "0"))
        if self._pack_in_tail:
            self._repack_static()
        snapshot = (self.state.params.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.adam_state.clone())
        def run():
            if self._graph_banks:
                self._gather_from_banks()
            return self._forward_backward(*self._static[:3], depth=self._static[3:] or None)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # warm-up on a side stream: lazy inits (LDS attributes, maps) happen outside capture
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.state.zero_grad()
        if self._adam_in_graph:  # the warm-up passes stepped the optimizer: roll them back
            self.state.params.copy_(snapshot[0]), self.exp_avg.copy_(snapshot[1]), self.exp_avg_sq.copy_(snapshot[2])
        self.adam_state.copy_(snapshot[3])  # ... and ticked the step counter (the jitter RNG's step) in every configuration
        for b in self._graph_banks or ():    # ... and moved the banks' cursors
            b.graph_reset()
        if self._pack_in_tail:               # ... and left the warm-up's weights in the streams
            self._repack_static()
        from . import ops

        self._graph = torch.cuda.CUDAGraph()
        failed = False
        try:
            with ops.graph_capture(self._graph):
                self._static_loss = run()
        except RuntimeError:
            if not capture_collective:
                raise
            failed = True
        if capture_collective and self._any_rank(failed):
            # the collective could not be captured on SOME rank: every rank drops its graph and re-captures without it (eager all-reduce +
            # Adam after the replay), together -- the warm-up passes of the retry issue collectives, so a rank retrying alone would
            # leave the job with mismatched collectives.  The vote goes through the rendezvous store, not a collective: the
            # RCCL communicator of a failed capture is not to be trusted with it.
            self._collective_capture_failed = True
            self._graph = None
            torch.cuda.synchronize()
            self.state.zero_grad()
            return self._capture(inputs, banks=banks)
        self.state.zero_grad()  # capture does not execute

    def _repack_static(self):
        """(pack-in-tail steps) one eager sr_pack_all into the static stream buffers: at capture time, and whenever somebody other than the
        step's own optimizer launch changed the weights (load_state_dict, a manual edit) -- the step itself keeps them current."""
        from .rendering import _mode_of

        model = self.models["coarse"]
        model.repack(_mode_of(self.args), backward=True, tick=None)
        self._packed_version = model.weights_version()

    def _any_rank(self, flag):
        """Logical OR of ``flag`` over the ranks, host-side and without a collective: every rank adds its flag and a tick to two counters
        of the job's rendezvous store (the TCPStore torch.distributed was initialised through) and waits for all ticks.  A single process
        answers for itself; if the store cannot be reached the vote falls back to a gloo side group."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return bool(flag)
        world = dist.get_world_size()
        # one key per vote of the JOB, not of this Trainer: the store and its counters outlive a trainer, and a second trainer re-using
        # vote #1 would find the first one's ticks already at `world` and read `failed` before the other ranks have added theirs
        # (ADVICE r04).  Captures -- of every trainer a process creates -- happen in the same order on every rank: a process-wide
        # counter names the same vote everywhere.
        global _VOTES
        _VOTES += 1
        try:
            import time

            store = dist.distributed_c10d._get_default_store()
            key = f"satnerf_amd/capture_vote/{_VOTES}"
            store.add(key + "/failed", 1 if flag else 0)
            store.add(key + "/ticks", 1)
            deadline = time.time() + 300.0
            while int(store.add(key + "/ticks", 0)) < world:
                if time.time() > deadline:
                    raise TimeoutError("capture vote: a rank never arrived")
                time.sleep(0.002)
            return int(store.add(key + "/failed", 0)) > 0
        except (AttributeError, RuntimeError):
            if getattr(self, "_vote_group", None) is None:
                self._vote_group = dist.new_group(backend="gloo")
            t = torch.tensor([1 if flag else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._vote_group)
            return bool(t.item())

    def _sampler_in_forward(self, banks):
        """True when a captured step can sample its colour batch inside the forward launch (sr_satnerf_render_train's gather)."""
        return (self._fused_forward() and not self._snerf and type(banks[0]).__name__ == "RayBank"
                and os.environ.get("SATNERF_GATHER_IN_FWD", "1") != "0")

    def step_from_bank(self, bank, depth_bank=None):
        """One step on the banks' next batches; with a captured graph the batches are gathered straight into its static inputs."""
        shapes = (bank.batch_size,) + ((depth_bank.batch_size,) if depth_bank is not None else ())
        from .rendering import validate_ts

        for b in (bank, depth_bank):  # image indices are checked once per bank (nn.Embedding would raise on a bad one)
            if b is not None and not self._snerf and not getattr(b, "_ts_validated", False):
                validate_ts(b.ts, self.models)
                b._ts_validated = True
        banks = (bank,) + ((depth_bank,) if depth_bank is not None else ())
        if (self.direct and self.use_graph and float(self.args.noise_std) == 0.0 and all(b.drop_last for b in banks)
                and os.environ.get("SATNERF_GRAPH_SAMPLER", "1" if self._sampler_in_forward(banks) else "0") == "1"):
            # the captured step samples for itself (device cursors over the epoch's shuffled indices): a step is ONE graph replay with no
            # eager launch and no host-side index arithmetic.  ON by default when the forward launch can do the sampling itself
            # (_sampler_in_forward: r05, sr_satnerf_render_train's gather); otherwise opt-in (SATNERF_GRAPH_SAMPLER=1) -- as separate
            # gather launches in the graph it is exactly as fast as the eager gather in front of the replay (r02: 0.420-0.425 ms either way)
            if self._graph is None or getattr(self, "_graph_banks", None) is None or tuple(map(id, self._graph_banks)) != tuple(map(id, banks)):
                first = [b.gather(b.graph_source()[0][:b.batch_size]) for b in banks]
                self._apply_schedule()
                self._capture(tuple(t for batch in first for t in batch), banks=banks)
            out = self.step(*self._static[:3], depth=self._static[3:] or None, _inputs_in_place=True)
            for b in banks:
                b.graph_advance()
            return out
        if self.direct and self._graph is not None and tuple(t.shape[0] for t in self._static[::3]) == shapes:
            idx = [bank.next_indices()] + ([depth_bank.next_indices()] if depth_bank is not None else [])
            if all(i.numel() == n for i, n in zip(idx, shapes)):
                bank.gather(idx[0], out=self._static[:3])
                if depth_bank is not None:
                    depth_bank.gather(idx[1], out=self._static[3:])
                return self.step(*self._static[:3], depth=self._static[3:] or None, _inputs_in_place=True)
            # a short last batch (drop_last=False): gather it normally; step() re-captures / runs it with its own shape
            batches = [bank.gather(idx[0])] + ([depth_bank.gather(idx[1])] if depth_bank is not None else [])
            return self.step(*batches[0], depth=batches[1] if depth_bank is not None else None)
        return self.step(*bank.next_batch(), depth=None if depth_bank is None else depth_bank.next_batch())

    def step(self, rays, ts, rgbs, depth=None, _inputs_in_place=False, validate=True):
        """``depth`` = (rays (M,11), ts (M,), depths (M,2) = [target depth, weight]) of the depth-supervision batch or None
        (pass None once past ``ds_drop``: main.py:138 stops adding the term).  ``validate=False`` skips the range check of the image
        indices (one device reduction + a host sync per step, rendering.validate_ts) for callers that have checked their ``ts`` once:
        the host then runs ahead of the replayed step again."""
        from . import ops

        if depth is not None and not float(getattr(self.args, "ds_lambda", 0.0)) > 0:
            raise ValueError("a depth batch was passed but args.ds_lambda is not > 0 (main.py:51)")
        if self._snerf:  # image indices are not an input of s-nerf: every ray reads row 0 of the zero embedding
            ts = self._zero_ts(ts)
            if depth is not None:
                depth = (depth[0], self._zero_ts(depth[1]), depth[2])
        if not _inputs_in_place and validate:
            from .rendering import validate_ts

            validate_ts(ts, self.models)
            if depth is not None:
                validate_ts(depth[1], self.models)
        self._apply_schedule()
        if self.direct:
            inputs = (rays, ts, rgbs) + (tuple(depth) if depth is not None else ())
            if self.use_graph and float(self.args.noise_std) == 0.0:
                stale = getattr(self, "_graph_banks", None) is not None and not _inputs_in_place  # that graph gathers from its banks
                if self._graph is None or stale or [t.shape for t in self._static] != [t.shape for t in inputs]:
                    self._capture(inputs)
                if not _inputs_in_place:
                    for dst, src in zip(self._static, inputs):
                        dst.copy_(src)
                if self._pack_in_tail and self.models["coarse"].weights_version() != self._packed_version:
                    self._repack_static()  # the weights changed behind the step's back
                self._pace()
                self._graph.replay()
                loss = self._static_loss
            else:
                inputs = tuple(t.contiguous() for t in inputs)
                if self._pack_in_tail and self.models["coarse"].weights_version() != self._packed_version:
                    self._repack_static()
                loss = self._forward_backward(*inputs[:3], depth=inputs[3:] or None)
            in_graph = self._adam_in_graph and self._graph is not None and self.use_graph and float(self.args.noise_std) == 0.0
            if self._collective and not self._adam_in_graph:
                dist.all_reduce(self.state.grads, op=dist.ReduceOp.SUM)
            self.n_steps += 1
            if not in_graph and not self._adam_in_graph:  # (an eager direct step after a capture already stepped Adam)
                if self._pack_in_tail:  # the forward ticked the device-side step count; the update launch also re-packs the streams
                    self._update_and_pack()
                else:
                    ops.adam_step(self.state.params, self.state.grads, self.exp_avg, self.exp_avg_sq, self.n_steps, lr=self.lr,
                                  grad_scale=1.0 / self.world, zero_grad=True)
            loss = _LazyLoss(loss)
        else:
            from .rendering import render_rays

            res = render_rays(self.models, self.args, rays, ts)
            if self.loss_fn is not None:
                loss_fn = self.loss_fn
            elif self.args.model == "nerf":
                loss_fn = nerf_loss
            elif self.warming_up():  # metrics.SNerfLoss for the first epochs (main.py:128-131)
                loss_fn = lambda r, t: snerf_loss(r, t, getattr(self.args, "sc_lambda", 0.0))  # noqa: E731
            else:
                loss_fn = lambda r, t: satnerf_loss(r, t, getattr(self.args, "sc_lambda", 0.0))  # noqa: E731  metrics.load_loss
            loss = loss_fn(res, rgbs)
            if depth is not None:
                d_rays, d_ts, d_depths = depth
                noise_std = self.args.noise_std
                self.args.noise_std = noise_std * 0.9  # main.py:132 decays the noise before the depth batch is rendered
                try:
                    res_d = render_rays(self.models, self.args, d_rays, d_ts)
                finally:
                    self.args.noise_std = noise_std
                w = 1.0 if getattr(self.args, "ds_noweights", False) else d_depths[:, 1]
                loss = loss + depth_loss(res_d, d_depths[:, 0], w, float(self.args.ds_lambda))
            loss.backward()
            if self._collective:
                dist.all_reduce(self.state.grads, op=dist.ReduceOp.SUM)
            if self.state.params.is_cuda:
                self.n_steps += 1
                ops.adam_step(self.state.params, self.state.grads, self.exp_avg, self.exp_avg_sq, self.n_steps, lr=self.lr,
                              grad_scale=1.0 / self.world, zero_grad=True)
            else:
                raise RuntimeError("training needs a GPU: satnerf_amd has no CPU path")
        for m in self.state.modules:
            if hasattr(m, "mark_weights_changed"):
                m.mark_weights_changed()
        if self.direct and self._pack_in_tail:  # ... and the step's update launch re-packed the streams
            from .rendering import _mode_of

            self.models["coarse"].note_packed(_mode_of(self.args))
            self._packed_version = self.models["coarse"].weights_version()
        self.args.noise_std *= 0.9  # main.py:132
        if self._caller_args is not self.args:
            self._caller_args.noise_std = self.args.noise_std
        self.last_loss = loss if isinstance(loss, _LazyLoss) else loss.detach()
        return self.last_loss
