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

import torch
import torch.distributed as dist


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


class Trainer:
    def __init__(self, models, args, world_size=1, lr=5e-4, loss_fn=None):
        self.models, self.args, self.world = models, args, world_size
        mods = [models["coarse"]] + ([models["fine"]] if "fine" in models else []) + [models["t"]]
        self.state = FlatState(mods)
        self._leaf = torch.nn.Parameter(self.state.params)  # shares storage with every module's parameters
        self._leaf.grad = self.state.grads
        fused = self.state.params.is_cuda
        self.opt = torch.optim.Adam([self._leaf], lr=lr, fused=fused)  # main.py:84
        self.loss_fn = loss_fn or (lambda res, tgt: satnerf_loss(res, tgt, getattr(args, "sc_lambda", 0.0)))
        self.last_loss = None

    def step(self, rays, ts, rgbs):
        from .rendering import render_rays

        self.state.zero_grad()
        res = render_rays(self.models, self.args, rays, ts)
        loss = self.loss_fn(res, rgbs)
        loss.backward()
        self.state.allreduce_mean_(self.world)
        self.opt.step()
        for m in self.state.modules:
            if hasattr(m, "mark_weights_changed"):
                m.mark_weights_changed()
        self.args.noise_std *= 0.9  # main.py:132
        self.last_loss = loss.detach()
        return self.last_loss
