"""r05: the captured single-GPU training step in FOUR launches (forward, dX, weight gradients, tail) -- VERDICT r04 Next #3.

* the batch sampler rides in the forward launch (sr_satnerf_render_train's gather: replaces sr_gather_batch = the DataLoader of
  main.py:96-110),
* the forward opens the step: it ticks the device-side step counter itself and draws for the ticked value (tick == 2),
* the tail launch that applies Adam also writes every updated parameter into the packed weight streams (sr_grad_tail_adam's pack: no
  sr_pack_all at the start of the next step).
Each piece against the launches it replaces, bit for bit; then whole trainers against each other."""
import os

import pytest
import torch

from oracle import satnerf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def maxnorm_rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _setup(mode="bf16", feat=256):
    from satnerf_amd.models import load_model

    torch.manual_seed(0)
    args = O.default_args(mlp_mode=mode, fc_units=feat)
    model = load_model(args).to(DEV)
    emb = torch.nn.Embedding(30, 4).to(DEV)
    model.repack(mode, backward=True)
    sk = model.sky_color
    return args, model, emb, (sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)


@pytest.mark.parametrize("feat,mode,n", [(256, "bf16", 128), (256, "f16", 36), (512, "bf16", 22)])
def test_forward_launch_samples_its_batch_like_gather_batch(feat, mode, n):
    """sr_satnerf_render_train with gather_idx: batch cursor[0] of the epoch's shuffled rows, straight from the bank -- every output, the
    saved activations and the batch rows it leaves for the later launches equal sr_gather_batch + the plain launch; the cursor moves on
    modulo the epoch's batches."""
    from satnerf_amd import ops

    s, tau, batches = 64, 4, 3
    args, model, emb, w = _setup(mode, feat)
    hi, lo, l0 = model.packed(mode)
    n_bank = batches * n + 17
    rays, ts = O.synthetic_rays(n_bank, seed=31)
    rays, ts = rays.to(DEV), ts.to(DEV)
    rgbs = torch.rand(n_bank, 3, device=DEV)
    idx = torch.randperm(n_bank, device=DEV)[:batches * n].contiguous()
    cursor = torch.zeros(4, device=DEV)
    sched = torch.tensor([5.0, 5e-4, 0.0, 0.0], device=DEV)
    for k in range(batches + 1):  # one lap and the wrap-around
        b = k % batches
        assert int(cursor[0].item()) == b and int(cursor.view(torch.int32)[3].item()) == 0
        want_rows = ops.gather_batch(rays, rgbs, ts, idx[b * n:(b + 1) * n].contiguous())
        acts_a = ops.acts_workspace(n * s, feat, DEV, 8).zero_()
        acts_b = torch.zeros_like(acts_a)
        ra = ops.render_train(want_rows[0], want_rows[1], emb.weight.data, s, feat, tau, mode, hi, lo, l0, *w, want_rows[2], acts_a, seed=77,
                              step_counter=sched, sched=sched, want_z=True)
        out = (torch.empty(n, 11, device=DEV), torch.empty(n, 3, device=DEV), torch.empty(n, dtype=torch.int64, device=DEV))
        rb = ops.render_train(rays, ts, emb.weight.data, s, feat, tau, mode, hi, lo, l0, *w, rgbs, acts_b, seed=77, step_counter=sched, sched=sched,
                              want_z=True, gather=dict(idx=idx, cursor=cursor, batches=batches, out=out))
        torch.cuda.synchronize()
        assert torch.equal(out[0], want_rows[0]) and torch.equal(out[1], want_rows[2]) and torch.equal(out[2], want_rows[1])
        for key in ra:
            if key == "loss":
                assert torch.equal(ra[key], rb[key])
            else:
                assert torch.equal(ra[key], rb[key]), (key, k)
        tiles = (n * s + 31) // 32
        per_tile = acts_a.numel() // ops._lib.lib().sr_workspace_tiles(n * s)
        assert torch.equal(acts_a[:tiles * per_tile], acts_b[:tiles * per_tile])


def test_forward_launch_opens_the_step():
    """tick == 2: the launch advances the step counter once and draws its jitter for the ADVANCED value -- the depths of a launch that
    found the counter at c equal those of a plain launch at c + 1 (what followed sr_pack_all's tick in r04)."""
    from satnerf_amd import ops

    n, s, tau, mode, feat = 100, 64, 4, "bf16", 256
    args, model, emb, w = _setup(mode, feat)
    hi, lo, l0 = model.packed(mode)
    rays, ts = O.synthetic_rays(n, seed=32)
    rays, ts = rays.to(DEV), ts.to(DEV)
    target = torch.rand(n, 3, device=DEV)
    for c in (0.0, 41.0):
        st_a = torch.tensor([c + 1.0, 5e-4, 0.0, 0.0], device=DEV)
        st_b = torch.tensor([c, 5e-4, 0.0, 0.0], device=DEV)
        acts = ops.acts_workspace(n * s, feat, DEV, 8)
        ra = ops.render_train(rays, ts, emb.weight.data, s, feat, tau, mode, hi, lo, l0, *w, target, acts, seed=5, step_counter=st_a, sched=st_a, want_z=True)
        rb = ops.render_train(rays, ts, emb.weight.data, s, feat, tau, mode, hi, lo, l0, *w, target, acts, seed=5, step_counter=st_b, sched=st_b, want_z=True,
                              tick=2)
        torch.cuda.synchronize()
        assert st_a[0].item() == c + 1.0 and st_b[0].item() == c + 1.0 and int(st_b.view(torch.int32)[3].item()) == 0
        for key in ("z", "rgb", "d_sigma", "albedo"):
            assert torch.equal(ra[key], rb[key]), key
    # a plain render launch has no "tick first"
    import ctypes as C

    e = lambda *sh: torch.empty(*sh, device=DEV)  # noqa: E731
    a = ops._lib.RenderArgs(ops._p(rays), 11, ops._p(ts), ops._p(emb.weight.data), n, s, None, None, 1, ops._p(st_b), 2, None, 0.0, 128,
                            *[ops._p(t) for t in w], 0)
    o = ops._lib.RenderOutputs(None, ops._p(e(n, s, 3)), None, ops._p(e(n, s)), ops._p(e(n, s)), ops._p(e(n, 3)), ops._p(e(n, s)), ops._p(e(n, s)),
                               ops._p(e(n)), ops._p(e(n, 3)))
    with pytest.raises(ops._lib.SatRenderError, match="tick"):
        ops._lib.call("sr_satnerf_render_fwd", C.byref(a), feat, tau, ops.MODES[mode], ops._p(hi), None, ops._p(l0), C.byref(o), ops._stream())


@pytest.mark.parametrize("mode", ["bf16", "f16", "bf16x3"])
def test_tail_launch_keeps_the_weight_streams_current(monkeypatch, mode):
    """sr_grad_tail_adam with `pack`: after the launch the forward stream (fp16 in f16 mode; hi and lo planes in bf16x3), the transposed
    stream and the fc_net.0 table hold exactly what sr_pack_all produces from the updated parameters."""
    from satnerf_amd import ops
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16" if mode == "bf16x3" else mode)  # (the step itself runs the 8-bit-state kernels)
    model = load_model(args).to(DEV)
    models = {"coarse": model, "t": torch.nn.Embedding(30, 4).to(DEV)}
    tr = Trainer(models, args, use_graph=False)
    model.repack(mode, backward=True)  # the buffers the tail is to keep current
    pack = model.pack_scatter(mode)
    assert (pack["lo"] is not None) == (mode == "bf16x3") and (pack["n_f16"] > 0) == (mode == "f16")
    before = (pack["hi"].clone(), pack["l0"].clone())
    orig = ops.grad_tail_adam
    monkeypatch.setattr(ops, "grad_tail_adam", lambda *a, **k: orig(*a, **{**k, "pack": pack}))
    tr._adam_in_graph = True  # the eager step ends in sr_grad_tail_adam
    tr.adam_state[0] = 0.0
    rays, ts = O.synthetic_rays(96, seed=33)
    for _ in range(2):
        tr._forward_backward(rays.to(DEV), ts.to(DEV), torch.rand(96, 3, device=DEV))
    torch.cuda.synchronize()
    assert not torch.equal(before[0], pack["hi"]) and not torch.equal(before[1], pack["l0"])  # (the parameters moved)
    from satnerf_amd.models import _stream_kind

    bufs = model._pack_cache[("buf", _stream_kind(mode), True)]
    maps = model._device_maps()
    hi2 = torch.full_like(bufs["hi"], -1)
    lo2 = torch.full_like(bufs["lo"], -1) if bufs["lo"] is not None else None
    l02 = torch.full_like(bufs["l0"], -1.0)
    ops.pack_all(model.flat_params(), bufs["idx"], bufs["scale"], hi2, lo2, maps["l0_idx"], maps["l0_scale"], l02, None, n_f16=pack["n_f16"])
    torch.cuda.synchronize()
    assert torch.equal(hi2, bufs["hi"]) and torch.equal(l02, bufs["l0"])
    if lo2 is not None:
        assert torch.equal(lo2, bufs["lo"])


def test_four_launch_step_trains_like_the_six_launch_step(monkeypatch):
    """Two trainers on the same bank, seeds and device-side sampler: the r05 default (sampler in the forward launch, forward ticks, tail
    packs) against the r04 launch sequence (sr_gather_batch, sr_pack_all, forward, dX, weight gradients, tail + Adam).  After the first
    step the MLP parameters are bit-identical (same batch, same jitter, same sums, same Adam), the streams of the fused step equal a
    fresh pack of its parameters after every step, and the runs stay together."""
    from satnerf_amd import ops
    from satnerf_amd.data import RayBank
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    n_bank, bs = 4 * 128 + 9, 128
    rays, ts = O.synthetic_rays(n_bank, seed=34)
    rgbs = torch.rand(n_bank, 3, generator=torch.Generator().manual_seed(35))
    runs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("SATNERF_TAIL_PACK", fused)
        monkeypatch.setenv("SATNERF_GATHER_IN_FWD", fused)
        monkeypatch.setenv("SATNERF_GRAPH_SAMPLER", "1")
        torch.manual_seed(0)
        args = O.default_args(mlp_mode="bf16")
        model = load_model(args).to(DEV)
        tr = Trainer({"coarse": model, "t": torch.nn.Embedding(30, 4).to(DEV)}, args, steps_per_epoch=1000)
        bank = RayBank(rays.to(DEV), rgbs.to(DEV), ts.to(DEV), bs, seed=9)
        snaps, losses, batches = [], [], []
        for k in range(6):  # crosses an epoch boundary (4 batches per epoch)
            losses.append(tr.step_from_bank(bank).item())
            torch.cuda.synchronize()
            snaps.append(tr.state.params.clone())
            batches.append(tuple(t.clone() for t in tr._static[:3]))
            if fused == "1":
                from satnerf_amd.models import _stream_kind

                bufs = model._pack_cache[("buf", _stream_kind("bf16"), True)]
                maps = model._device_maps()
                hi2, l02 = torch.empty_like(bufs["hi"]), torch.empty_like(bufs["l0"])
                ops.pack_all(model.flat_params(), bufs["idx"], bufs["scale"], hi2, None, maps["l0_idx"], maps["l0_scale"], l02, None)
                assert torch.equal(hi2, bufs["hi"]) and torch.equal(l02, bufs["l0"]), k
                hi, _, l0 = model.packed("bf16")  # the caches follow: no stale stream for an eval between steps
                assert hi.data_ptr() == bufs["hi"].data_ptr() and l0.data_ptr() == bufs["l0"].data_ptr()
        assert tr._graph is not None and tr._pack_in_tail == (fused == "1") and int(tr.adam_state[0].item()) == 6
        runs.append((snaps, losses, batches, tr._late_idx.long()))
    (sa, la, ba, late), (sb, lb, bb, _) = runs
    for x, y in zip(ba, bb):  # the same batches, in the same order, as the graph's static tensors
        assert all(torch.equal(p, q) for p, q in zip(x, y))
    mlp = torch.ones(sa[0].numel(), dtype=torch.bool, device=DEV)
    mlp[late] = False
    assert torch.equal(sa[0][mlp], sb[0][mlp]) and maxnorm_rel(sa[0][late].cpu(), sb[0][late].cpu()) < 1e-5
    assert max(abs(a - b) for a, b in zip(la, lb)) < 1e-4 * max(lb), (la, lb)
    assert maxnorm_rel(sa[-1].cpu(), sb[-1].cpu()) < 5e-3


def test_weights_changed_behind_the_steps_back_are_repacked():
    """A pack-in-tail trainer whose weights are overwritten between two replays (load_state_dict) trains on the NEW weights: the step
    re-packs eagerly when the model's version moved without it."""
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    model = load_model(args).to(DEV)
    tr = Trainer({"coarse": model, "t": torch.nn.Embedding(30, 4).to(DEV)}, args, steps_per_epoch=1000)
    rays, ts = O.synthetic_rays(128, seed=36)
    rays, ts, target = rays.to(DEV), ts.to(DEV), torch.rand(128, 3, device=DEV)
    tr.step(rays, ts, target, validate=False)
    assert tr._pack_in_tail
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    l1 = tr.step(rays, ts, target, validate=False).item()
    with torch.no_grad():
        for p_ in model.parameters():
            p_.mul_(0.5)
    l_half = tr.step(rays, ts, target, validate=False).item()
    model.load_state_dict(sd)
    tr.exp_avg.zero_(), tr.exp_avg_sq.zero_()
    l2 = tr.step(rays, ts, target, validate=False).item()
    assert abs(l_half - l1) > 1e-3 * abs(l1)          # the halved weights were seen ...
    assert abs(l2 - l1) < 2e-2 * abs(l1), (l1, l2)    # ... and so were the restored ones (same weights as before step 2, other jitter)


def test_late_parameters_of_the_fused_tail_under_stress():
    """inside sr_grad_tail_adam the sky head and the embedding rows are updated by the LAST of the blocks whose float
    atomics produce their gradients; it must see every other block's atomics.  The launch of a real step is recorded and repeated 400 times
    from the same optimizer state beside a stream of unrelated memory traffic; every repetition's late parameters must equal the reference
    -- sr_grad_tail followed by sr_adam_step_graph on the same inputs, where the kernel boundary orders the atomics -- to float-atomics
    rounding.  A late parameter updated from a PARTIAL gradient (one block's atomics missing) differs at the 1e-2 level."""
    from satnerf_amd import ops
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
    tr = Trainer(models, args, use_graph=True, steps_per_epoch=1000)
    n = 1024
    rays, ts = O.synthetic_rays(n, seed=77)
    rays, ts, tgt = rays.to(DEV), ts.to(DEV), torch.rand(n, 3, device=DEV)
    seen = {}
    real = ops.grad_tail_adam

    def spy(*a, **k):
        seen["a"], seen["k"] = a, k
        return real(*a, **k)

    ops.grad_tail_adam = spy
    try:
        for _ in range(3):
            tr.step(rays, ts, tgt, validate=False)
        torch.cuda.synchronize()
    finally:
        ops.grad_tail_adam = real
    a, k = seen["a"], dict(seen["k"])
    k["pack"] = None
    tail_args, (params, m, v, late, state) = a[:21], a[21:26]
    late = late.long()
    snap = (params.clone(), m.clone(), v.clone())

    def restore():
        params.copy_(snap[0]), m.copy_(snap[1]), v.copy_(snap[2])
        tr.state.grads.zero_()

    # reference: the unfused pair (the atomics are complete at the kernel boundary)
    restore()
    ops.grad_tail(*tail_args)
    assert tr.state.grads.numel() == params.numel()
    ops.adam_step_graph(params, tr.state.grads, m, v, state, lr=-1.0, grad_scale=1.0, zero_grad=True)
    torch.cuda.synchronize()
    want = params[late].clone()
    moved = (want - snap[0][late]).abs().max().item()
    assert moved > 0
    noise_src = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    noise_dst = torch.empty_like(noise_src)
    worst = 0.0
    for rep in range(400):
        restore()
        if rep % 2:
            noise_dst.copy_(noise_src)  # dirty lines in every L2 while the tail runs behind it
        real(*a[:21], params, m, v, a[24], state, **k)
        got = params[late]
        worst = max(worst, (got - want).abs().max().item())
        if rep % 50 == 49:
            torch.cuda.synchronize()
            assert float(tr.state.grads.abs().max()) == 0.0 and int(state.view(torch.int32)[3].item()) == 0
    torch.cuda.synchronize()
    print(f"late parameters over 400 launches: worst |difference| {worst:.2e} of an update of {moved:.2e}")
    assert worst <= 1e-3 * moved + 1e-9
    restore()
