"""Training-step semantics of the HIP path (main.py:81-154): the fused Adam against torch.optim.Adam, the reference's schedule
(StepLR per epoch, SNerfLoss warm-up) under graph replay, depth supervision at BASELINE configs[3] size, and the data-parallel
Trainer.step with two ranks on one GPU."""
import os
import socket

import pytest
import torch

from oracle import satnerf_oracle as O
from tests.helpers import load_golden, golden_draws, make_models, maxnorm_rel
from tests.test_hip_parity import DEV, build_models

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("graph", [False, True])
def test_fused_adam_matches_torch_adam(graph):
    """sr_adam_step / sr_adam_step_graph == torch.optim.Adam(lr, betas (0.9, 0.999), eps 1e-8) on identical gradients over 12 steps,
    including a learning-rate change midway (the graph variant reads step and rate from the device-side schedule block)."""
    from satnerf_amd import ops

    n = 662537 + 120
    g = torch.Generator().manual_seed(3)
    p0 = (torch.rand(n, generator=g) - 0.5) * 0.3
    ref_p = torch.nn.Parameter(p0.clone().double())  # fp64 reference of the same recurrence
    ref32 = torch.nn.Parameter(p0.clone().to(DEV))   # torch's own fp32 GPU Adam
    opt64 = torch.optim.Adam([ref_p], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    opt32 = torch.optim.Adam([ref32], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
    p = p0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sched = torch.zeros(4, device=DEV)
    lr = 5e-4
    for step in range(1, 13):
        if step == 7:  # StepLR(gamma=0.9) at an epoch boundary
            lr *= 0.9
            for o in (opt64, opt32):
                o.param_groups[0]["lr"] = lr
        grad = torch.randn(n, generator=g) * (10.0 ** torch.randint(-6, 1, (n,), generator=g).float())  # gradients over 6 decades
        ref_p.grad, ref32.grad = grad.double(), grad.to(DEV)
        opt64.step(), opt32.step()
        gdev = (grad * 2.0).to(DEV)  # the kernel undoes this with grad_scale = 0.5 (the 1 / world_size of the data-parallel step)
        if graph:
            sched[0] = float(step)
            sched[1] = lr
            ops.adam_step_graph(p, gdev, m, v, sched, lr=-1.0, grad_scale=0.5, zero_grad=True)
        else:
            ops.adam_step(p, gdev, m, v, step, lr=lr, grad_scale=0.5, zero_grad=True)
        assert float(gdev.abs().max()) == 0.0  # zero_grad
    err64 = maxnorm_rel(p.cpu(), ref_p.detach())
    err32 = maxnorm_rel(p.cpu(), ref32.detach().cpu())
    upd = maxnorm_rel((p.cpu() - p0), (ref_p.detach() - p0.double()))  # on the update itself, not hidden behind the parameter's magnitude
    print(f"adam graph={graph}: vs fp64 {err64:.1e}, vs torch fp32 {err32:.1e}, update {upd:.1e}")
    assert err64 < 1e-6 and err32 < 1e-6 and upd < 1e-4  # (fp32 parameter storage: 12 roundings of |p| ~ 0.15 against updates of ~6e-3)


def test_schedule_lr_decay_and_snerf_warmup_under_graph_replay():
    """steps_per_epoch switches on StepLR(0.9)/epoch and the SNerfLoss warm-up (main.py:86-94,128-131).  The captured step must
    follow both without re-capture: loss values equal the torch formulations on the trainer's own rendering, the parameter
    update of a step scales with the scheduled rate."""
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    model = load_model(args).to(DEV)
    emb = torch.nn.Embedding(30, 4).to(DEV)
    tr = Trainer({"coarse": model, "t": emb}, args, steps_per_epoch=3, warmup_epochs=2)
    rays, ts = O.synthetic_rays(256, seed=3)
    rays, ts = rays.to(DEV), ts.to(DEV)
    target = (torch.rand(256, 3, generator=torch.Generator().manual_seed(4)) * 0.2 + 0.4).to(DEV)
    seen = []
    for k in range(9):
        before = tr.state.params.clone()
        loss = tr.step(rays, ts, target).item()
        epoch = (k + 1) // 3  # what main.py:121-128 tests for the SNerfLoss warm-up (train_steps is incremented first)
        lr_epoch = k // 3     # StepLR is stepped after an epoch's LAST batch: step k trains at gamma ** (k // steps_per_epoch)
        assert tr._graph is not None  # one capture serves every epoch
        assert abs(tr.lr - 5e-4 * 0.9 ** lr_epoch) < 1e-12
        assert float(tr.sched[1]) == pytest.approx(5e-4 * 0.9 ** lr_epoch, rel=1e-6) and float(tr.sched[2]) == (1.0 if epoch < 2 else 0.0)
        rgb = tr.last_rgb
        mse = torch.mean((rgb - target) ** 2).item()
        if epoch < 2:  # metrics.SNerfLoss (lambda_sc = 0): plain MSE
            assert loss == pytest.approx(mse, rel=1e-4), (k, loss, mse)
        else:          # metrics.SatNerfLoss: > the MSE (log-beta term + 3/2)
            assert loss > 1.0 > mse
        step_size = (tr.state.params - before).abs().max().item()
        seen.append((lr_epoch, step_size))
    # Adam's first updates are ~lr per coordinate: the largest update of an epoch's first step tracks the decayed rate
    assert seen[0][1] == pytest.approx(5e-4, rel=0.05)
    assert all(s <= 5e-4 * 0.9 ** e * 1.2 for e, s in seen)


def test_snerf_warmup_loss_matches_reference_golden_and_autograd_path():
    """(i) the torch SNerfLoss formulation on the reference's own batched results == the stored metrics.SNerfLoss value;
    (ii) the kernel-direct step in warm-up mode == render_rays under autograd + that formulation (value and gradients)."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer, snerf_loss

    g = load_golden("batched_losses")
    args = O.default_args(chunk=100, sc_lambda=0.05, mlp_mode="bf16x3")
    models = build_models(args)
    draws = [x.to(DEV) for x in golden_draws(g)]
    rays, ts = g["rays"].to(DEV), g["ts"].to(DEV)
    with torch.no_grad(), rendering.replay_rng(draws):
        outs = [rendering.render_rays(models, args, rays[i:i + 100], ts[i:i + 100]) for i in range(0, 250, 100)]
    res = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    l_sn = snerf_loss(res, g["target"].to(DEV), lambda_sc=0.05)
    assert abs(l_sn.item() - float(g["loss_snerf"])) < 1e-4 * abs(float(g["loss_snerf"]))

    args = O.default_args(mlp_mode="bf16x3", sc_lambda=0.1)
    params = O.procedural_satnerf_params(256, 4, seed=111)
    m = load_model(args)
    m.load_state_dict(params)
    emb = torch.nn.Embedding(30, 4)
    emb.load_state_dict({"weight": O.procedural_uniform((30, 4), 1.0, 112)})
    models = {"coarse": m.to(DEV), "t": emb.to(DEV)}
    rays, ts = O.synthetic_rays(160, seed=113)
    rays, ts = rays.to(DEV), ts.to(DEV)
    target = torch.rand(160, 3, generator=torch.Generator().manual_seed(114)).to(DEV)
    tr = Trainer(models, args, use_graph=False, steps_per_epoch=1000)  # epoch 0: warm-up
    assert tr.direct and tr.warming_up()
    torch.manual_seed(11)
    parts = tr._forward_backward(rays, ts, target)
    g_direct = tr.state.grads.clone()
    tr.state.zero_grad()
    torch.manual_seed(11)
    u = torch.rand(160, 64, device=DEV)
    with rendering.replay_rng([u, torch.zeros_like(u), torch.zeros_like(u)]):
        res = rendering.render_rays(models, args, rays, ts)
    loss = snerf_loss(res, target, 0.1)
    loss.backward()
    assert abs(parts.sum().item() - loss.item()) < 1e-4 * abs(loss.item())
    assert maxnorm_rel(g_direct.cpu(), tr.state.grads.cpu()) < 1e-4
    # no gradient reaches the uncertainty head while the warm-up lasts
    sd = dict(models["coarse"].named_parameters())
    assert float(sd["beta_from_xyz.2.weight"].grad.abs().max()) == 0.0


def test_depth_supervision_at_config4_size():
    """BASELINE configs[3]: 4096 colour + 4096 depth rays per step (main.py:134-141, metrics.DepthLoss).  Properties at full
    size + 32 sampled depth rays against the oracle; the step trains."""
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16x3", ds_lambda=1000.0)
    model = load_model(args)
    params = O.procedural_satnerf_params(256, 4, seed=121)
    model.load_state_dict(params)
    embw = O.procedural_uniform((30, 4), 1.0, 122)
    emb = torch.nn.Embedding(30, 4)
    emb.load_state_dict({"weight": embw})
    models = {"coarse": model.to(DEV), "t": emb.to(DEV)}
    n = 4096
    rays, ts = O.synthetic_rays(n, seed=123)
    d_rays, d_ts = O.synthetic_rays(n, seed=124)
    gen = torch.Generator().manual_seed(125)
    target = torch.rand(n, 3, generator=gen)
    depths = torch.stack([torch.rand(n, generator=gen) * 0.5 + 0.2, torch.rand(n, generator=gen) + 0.5], 1)  # [target depth, weight]
    # depth rendering of the depth batch vs the oracle on 32 sampled rays (same draws)
    u, nz = torch.rand(n, 64, device=DEV), torch.zeros(n, 64, device=DEV)
    with torch.no_grad(), rendering.replay_rng([u, nz]):
        res = rendering.render_rays(models, args, d_rays.to(DEV), d_ts.to(DEV))
    assert res["depth_coarse"].shape == (n,) and torch.isfinite(res["depth_coarse"]).all()
    assert (res["depth_coarse"] >= 0).all() and (res["depth_coarse"] <= d_rays[:, 7].max().item() + 1e-4).all()  # within [near, far]
    pick = torch.arange(0, n, 128)[:32]
    want = O.render_rays({"coarse": params, "t": embw}, args, d_rays[pick], d_ts[pick], O.ReplayRng([u[pick].cpu(), nz[pick].cpu()]))
    assert maxnorm_rel(res["depth_coarse"][pick.to(DEV)].cpu(), want["depth_coarse"]) < 1e-4
    lam = 1000.0 / 3.0
    l_ref = lam * torch.mean(depths[pick, 1] * (want["depth_coarse"] - depths[pick, 0]) ** 2)
    l_hip = lam * torch.mean(depths[pick, 1].to(DEV) * (res["depth_coarse"][pick.to(DEV)] - depths[pick, 0].to(DEV)) ** 2)
    assert abs(l_hip.item() - l_ref.item()) < 1e-3 * abs(l_ref.item())
    # the full-size step: colour + depth batches, one flat gradient, loss parts = [colour, depth]
    targs = O.default_args(mlp_mode="bf16", ds_lambda=1000.0)
    tr = Trainer(models, targs)
    dev = lambda *t: tuple(x.to(DEV) for x in t)  # noqa: E731
    losses = [tr.step(*dev(rays, ts, target), depth=dev(d_rays, d_ts, depths)).item() for _ in range(8)]
    assert tr.direct and tr._graph is not None
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


def _rank_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the one GPU: gloo stages through the host
    torch.manual_seed(0)  # identical init on every rank
    args = O.default_args(mlp_mode="bf16", bwd_fmt=16)
    models = {"coarse": load_model(args).to("cuda:0"), "t": torch.nn.Embedding(30, 4).to("cuda:0")}
    tr = Trainer(models, args, world_size=world)
    rays, ts = O.synthetic_rays(128 * world, seed=31)
    target = torch.rand(128 * world, 3, generator=torch.Generator().manual_seed(32))
    sl = slice(128 * rank, 128 * (rank + 1))
    torch.manual_seed(100 + rank)  # per-rank sampling jitter
    losses = [tr.step(rays[sl].cuda(), ts[sl].cuda(), target[sl].cuda()).item() for _ in range(3)]
    torch.save({"params": tr.state.params.cpu(), "losses": losses, "graphed": tr._graph is not None}, out_path.format(rank))
    dist.destroy_process_group()


def test_two_rank_trainer_step_on_one_gpu(tmp_path):
    """Trainer.step with world_size 2 (the path bench.py runs at N > 1: captured forward/backward per rank, all-reduce of the flat
    gradient, Adam with grad_scale 1/2): replicas bit-identical after 3 steps, and equal to ONE process that accumulates the two
    ranks' batches into the flat gradient and steps with grad_scale 1/2."""
    import torch.multiprocessing as mp

    from satnerf_amd import ops
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    port, out = _free_port(), str(tmp_path / "rank{}.pt")
    mp.spawn(_rank_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out.format(0)), torch.load(out.format(1))
    assert r0["graphed"] and r1["graphed"]
    assert torch.equal(r0["params"], r1["params"])  # replicas stay bit-identical
    # single-process twin: same init, same per-rank draws (the captured step draws its jitter in-kernel from the seed and the
    # step counter), rank batches accumulated into one gradient, one Adam step with grad_scale = 1/2
    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16", bwd_fmt=16)
    models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
    rays, ts = O.synthetic_rays(256, seed=31)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(32))
    tr = Trainer(models, args, world_size=2, use_graph=False)  # (constructed under the same torch seed as the ranks: same jitter key)
    tr._kernel_rng = True  # what the captured rank steps use
    for step in range(3):
        for rank in range(2):
            sl = slice(128 * rank, 128 * (rank + 1))
            tr.adam_state[0] = float(step)  # sr_pack_all ticks it to step + 1 at the start of each pass, as in a rank's step
            tr._forward_backward(rays[sl].to(DEV), ts[sl].to(DEV), target[sl].to(DEV))
        tr.n_steps += 1
        ops.adam_step(tr.state.params, tr.state.grads, tr.exp_avg, tr.exp_avg_sq, tr.n_steps, lr=tr.lr, grad_scale=0.5, zero_grad=True)
        for m in tr.state.modules:
            if hasattr(m, "mark_weights_changed"):
                m.mark_weights_changed()
    err = maxnorm_rel(tr.state.params.cpu(), r0["params"])
    upd = maxnorm_rel(tr.state.params.cpu() - _initial_params(args), r0["params"] - _initial_params(args))
    print(f"2-rank vs accumulated single process: params {err:.1e}, update {upd:.1e}")
    assert err < 1e-6 and upd < 1e-3


def _rank_worker8(rank, world, port, out_path, dp_pack):
    """two ranks on the one GPU (gloo), the benchmarked arithmetic (8-bit state, one-launch training forward): with dp_pack the r06 step --
    forward (ticks) | dX | wgrad | sr_grad_tail in the replayed graph, all-reduce, sr_adam_step_pack -- without it the r05 step
    (sr_pack_all first, sr_adam_step last)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SATNERF_DP_PACK="1" if dp_pack else "0")
    import torch.distributed as dist

    from satnerf_amd import ops
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    models = {"coarse": load_model(args).to("cuda:0"), "t": torch.nn.Embedding(30, 4).to("cuda:0")}
    tr = Trainer(models, args, world_size=world, steps_per_epoch=1000)
    rays, ts = O.synthetic_rays(128 * world, seed=31)
    target = torch.rand(128 * world, 3, generator=torch.Generator().manual_seed(32))
    sl = slice(128 * rank, 128 * (rank + 1))
    calls = {"pack_all": 0, "adam_pack": 0}
    real_pack, real_adam = ops.pack_all, ops.adam_step_pack
    ops.pack_all = lambda *a, **k: (calls.__setitem__("pack_all", calls["pack_all"] + 1), real_pack(*a, **k))[1]
    ops.adam_step_pack = lambda *a, **k: (calls.__setitem__("adam_pack", calls["adam_pack"] + 1), real_adam(*a, **k))[1]
    losses, streams_ok = [], True
    for k in range(3):
        losses.append(tr.step(rays[sl].cuda(), ts[sl].cuda(), target[sl].cuda(), validate=False).item())
        if k == 0:
            calls["pack_all_at_capture"] = calls["pack_all"]  # (the capture packs the static streams eagerly; the steps themselves must not)
    torch.cuda.synchronize()
    if dp_pack:  # the streams the update launch left equal a fresh sr_pack_all of the updated parameters, bit for bit
        model = models["coarse"]
        hi, lo, l0, bstream, _ = model.packed_static("bf16")
        kept = [t.clone() for t in (hi, l0, bstream)]
        ops.pack_all, ops.adam_step_pack = real_pack, real_adam
        model.mark_weights_changed()
        model.repack("bf16", backward=True, tick=None)
        hi2, _, l02 = model.packed("bf16")
        b2, _ = model.packed_backward()
        streams_ok = all(torch.equal(a, b) for a, b in zip(kept, (hi2, l02, b2)))
    torch.save({"params": tr.state.params.cpu(), "losses": losses, "graphed": tr._graph is not None, "pit": tr._pack_in_tail, "calls": calls,
                "streams_ok": streams_ok, "step": float(tr.adam_state[0].item())}, out_path.format(rank))
    dist.destroy_process_group()


def test_two_rank_step_is_the_single_gpu_step_plus_one_collective(tmp_path):
    """r06: at N > 1 the captured step is forward | dX | wgrad | sr_grad_tail, then the all-reduce and ONE update launch
    (sr_adam_step_pack: Adam + re-pack) -- no sr_pack_all in the step, no separate Adam launch.  Two gloo ranks on the one GPU: replicas
    bit-identical, the weight streams equal a fresh pack of the updated parameters, and the parameters equal the r05 N > 1 step's
    (SATNERF_DP_PACK=0: sr_pack_all ... sr_adam_step) -- the same sums and the same adam_one, so bit for bit where no float atomics
    feed the gradient and to rounding for the sky head / embedding rows."""
    import torch.multiprocessing as mp

    res = {}
    for dp_pack in (True, False):
        port, out = _free_port(), str(tmp_path / ("new{}.pt" if dp_pack else "old{}.pt"))
        mp.spawn(_rank_worker8, args=(2, port, out, dp_pack), nprocs=2, join=True)
        res[dp_pack] = [torch.load(out.format(r)) for r in range(2)]
    new, old = res[True], res[False]
    assert all(r["graphed"] for r in new + old)
    assert all(r["pit"] for r in new) and not any(r["pit"] for r in old)
    assert all(r["step"] == 3.0 for r in new + old)
    for r in new:  # the capture packs the static streams eagerly (sr_pack_all); steps 2 and 3 add no sr_pack_all, every step ONE update launch
        assert r["calls"]["adam_pack"] == 3 and r["calls"]["pack_all"] == r["calls"]["pack_all_at_capture"] and r["streams_ok"], r["calls"]
    assert torch.equal(new[0]["params"], new[1]["params"]) and torch.equal(old[0]["params"], old[1]["params"])
    err = maxnorm_rel(new[0]["params"], old[0]["params"])
    print(f"r06 vs r05 two-rank step after 3 steps: {err:.1e}")
    assert err < 5e-4  # (atomics order; Adam's first updates amplify last-bit differences, see the 1-rank nccl test below)
    assert new[0]["losses"] == pytest.approx(old[0]["losses"], rel=1e-4)
    # ... and to ONE process that accumulates the two ranks' batches into the flat gradient and steps with grad_scale 1/2 (the union batch):
    # same init, same in-kernel jitter (keyed by the seed and the step count the forward / sr_pack_all advances)
    from satnerf_amd import ops
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
    rays, ts = O.synthetic_rays(256, seed=31)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(32))
    tr = Trainer(models, args, world_size=2, use_graph=False, steps_per_epoch=1000)
    tr._kernel_rng = True
    for step in range(3):
        for rank in range(2):
            sl = slice(128 * rank, 128 * (rank + 1))
            tr.adam_state[0] = float(step)  # sr_pack_all ticks it to step + 1 at the start of each pass, as a rank's forward does
            tr._forward_backward(rays[sl].to(DEV), ts[sl].to(DEV), target[sl].to(DEV))
        tr.n_steps += 1
        tr._apply_schedule()
        ops.adam_step(tr.state.params, tr.state.grads, tr.exp_avg, tr.exp_avg_sq, tr.n_steps, lr=tr.lr, grad_scale=0.5, zero_grad=True)
        for m in tr.state.modules:
            if hasattr(m, "mark_weights_changed"):
                m.mark_weights_changed()
    err1 = maxnorm_rel(tr.state.params.cpu(), new[0]["params"])
    upd1 = maxnorm_rel(tr.state.params.cpu() - _initial_params(args), new[0]["params"] - _initial_params(args))
    print(f"r06 two-rank step vs accumulated single process: params {err1:.1e}, update {upd1:.1e}")
    # measured: parameters 5.4e-5 (the r05-sequence comparison above: 8.1e-5), update 0.10 of the largest update -- Adam's first steps move every
    # element by ~lr whatever its gradient's size, so the float atomics' last bits (sky head, embedding rows) decide the sign where a gradient is ~0
    assert err1 < 5e-4 and upd1 < 0.3


def _nccl_one_rank_worker(rank, port, out_path, force):
    import torch.distributed as dist

    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SATNERF_FORCE_ALLREDUCE="1" if force else "0",
                      SATNERF_GRAPH_ALLREDUCE="1")  # (the captured collective is opt-in until a multi-GPU job has run it)
    torch.cuda.set_device(0)
    if force:
        dist.init_process_group("nccl", rank=0, world_size=1)
    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
    tr = Trainer(models, args, world_size=1)
    rays, ts = O.synthetic_rays(256, seed=33)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(34))
    losses = [tr.step(rays.cuda(), ts.cuda(), target.cuda()).item() for _ in range(4)]
    torch.cuda.synchronize()
    torch.save({"params": tr.state.params.cpu(), "losses": losses, "graphed": tr._graph is not None, "collective": tr._collective,
                "adam_in_graph": tr._adam_in_graph, "capture_failed": getattr(tr, "_collective_capture_failed", False)}, out_path)
    if force:
        dist.destroy_process_group()


def test_rccl_allreduce_is_captured_into_the_step_graph(tmp_path):
    """The data-parallel step with SATNERF_GRAPH_ALLREDUCE=1 and the RCCL backend: the all-reduce of the flat gradient and the Adam update
    are captured into the step's hipGraph (opt-in until a job with two or more GPUs has run it; the default issues both eagerly after the
    replay).  Exercised on the single GPU with a 1-rank "nccl" process group (SATNERF_FORCE_ALLREDUCE=1 makes the
    1-rank trainer issue the collective): the capture succeeds, replays run, and the result equals the plain single-GPU trainer's
    (an all-reduce over one rank is the identity, grad_scale 1)."""
    import torch.multiprocessing as mp

    a, b = str(tmp_path / "rccl.pt"), str(tmp_path / "plain.pt")
    mp.spawn(_nccl_one_rank_worker, args=(_free_port(), a, True), nprocs=1, join=True)
    mp.spawn(_nccl_one_rank_worker, args=(_free_port(), b, False), nprocs=1, join=True)
    ra, rb = torch.load(a), torch.load(b)
    assert ra["graphed"] and ra["collective"] and ra["adam_in_graph"] and not ra["capture_failed"]
    assert rb["graphed"] and not rb["collective"]
    # (two runs of the step are equal to rounding, not bit for bit: the embedding / loss reductions use atomics; r05: the plain trainer
    # applies Adam inside the gradient-tail launch, the collective one in its own launch after the all-reduce -- another atomics order,
    # and Adam's first updates (+-lr whatever a gradient's size) amplify last-bit differences wherever a gradient is near zero:
    # measured 5.6e-5 after four steps)
    assert maxnorm_rel(ra["params"], rb["params"]) < 5e-4 and ra["losses"] == pytest.approx(rb["losses"], rel=1e-4)


def _initial_params(args):
    from satnerf_amd.models import load_model

    torch.manual_seed(0)
    m, e = load_model(args), torch.nn.Embedding(30, 4)
    return torch.cat([torch.cat([p.detach().reshape(-1) for p in m.parameters()]), e.weight.detach().reshape(-1)])


def test_bad_image_index_raises_like_nn_embedding():
    """``nn.Embedding`` raises IndexError on ts >= t_embbeding_vocab (rendering.py:100); the fused kernels index the table directly,
    so render_rays / batched_inference / the Trainer check the range up front instead of reading out of bounds."""
    from satnerf_amd import rendering
    from satnerf_amd.data import RayBank
    from satnerf_amd.train import Trainer

    args = O.default_args(mlp_mode="bf16")
    models = build_models(args)
    rays, ts = O.synthetic_rays(64, seed=5)
    bad = ts.clone()
    bad[7] = 30
    with pytest.raises(IndexError):
        with torch.no_grad():
            rendering.render_rays(models, args, rays.to(DEV), bad.to(DEV))
    with pytest.raises(IndexError):
        rendering.batched_inference(models, rays.to(DEV), (-bad).to(DEV), args)
    tr = Trainer(models, args)
    with pytest.raises(IndexError):
        tr.step_from_bank(RayBank(rays.to(DEV), torch.rand(64, 3, device=DEV), bad.to(DEV), 32))
    with torch.no_grad():  # a valid batch still renders
        assert torch.isfinite(rendering.render_rays(models, args, rays.to(DEV), ts.to(DEV))["rgb_coarse"]).all()


def test_step_from_bank_handles_a_short_last_batch():
    """drop_last=False: the 44-ray tail of a 300-ray bank must not replay the captured 128-ray step on stale inputs."""
    from satnerf_amd.data import RayBank
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="bf16")
    models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
    rays, ts = O.synthetic_rays(300, seed=6)
    bank = RayBank(rays.to(DEV), torch.rand(300, 3, device=DEV), ts.to(DEV), 128, drop_last=False)
    tr = Trainer(models, args)
    sizes = []
    for _ in range(7):  # 128, 128, 44 | 128, 128, 44 | 128
        tr.step_from_bank(bank)
        sizes.append(tr._static[0].shape[0])
    assert sizes == [128, 128, 44, 128, 128, 44, 128]
    assert torch.isfinite(tr.state.params).all() and tr.n_steps == 7


def test_captured_step_samples_its_own_batches_from_the_bank():
    """Trainer.step_from_bank with a graph: the captured step's first launch gathers batch cursor[0] of the epoch's shuffled index
    buffer and advances the device cursor (sr_gather_batch cursor mode) -- a step is one graph replay.  The batches the steps
    trained on are exactly the bank's shuffled epochs, in order, every ray once per epoch, reshuffled between epochs; and the
    losses equal those of a twin trainer fed the same batches through the eager gather."""
    from satnerf_amd.data import RayBank
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    n_bank, bs = 5 * 64 + 20, 64  # 5 full batches per epoch, 20 rays dropped per epoch (drop_last)
    rays, ts = O.synthetic_rays(n_bank, seed=201)
    rays[:, 6] = torch.arange(n_bank) * 1e-6  # tag every ray (near): tells which rows a step saw
    rgbs = torch.rand(n_bank, 3, generator=torch.Generator().manual_seed(202))

    def make():
        torch.manual_seed(0)
        args = O.default_args(mlp_mode="bf16")
        models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
        return Trainer(models, args), RayBank(rays.to(DEV), rgbs.to(DEV), ts.to(DEV), bs, seed=9)

    tr, bank = make()
    seen, losses = [], []
    os.environ["SATNERF_GRAPH_SAMPLER"] = "1"  # opt-in
    try:
        for _ in range(12):
            losses.append(tr.step_from_bank(bank).item())
            seen.append((tr._static[0][:, 6] * 1e6).round().long().cpu())
    finally:
        del os.environ["SATNERF_GRAPH_SAMPLER"]
    assert tr._graph is not None and tr._graph_banks == (bank,)
    epochs = [torch.cat(seen[e * 5:(e + 1) * 5]) for e in range(2)]
    for e in epochs:
        assert e.numel() == 5 * bs and e.unique().numel() == 5 * bs  # every ray at most once per epoch
    assert not torch.equal(epochs[0], epochs[1])                     # reshuffled
    assert int(bank._gcursor[0].item()) == 12 % 5 and int(bank._gcursor[3].item()) == 0
    # twin: same seeds, batches through the eager gather of the same index sequence (in-kernel jitter is keyed by the step)
    tr2, bank2 = make()
    bank2._new_epoch()  # (the graph sampler drew one extra shuffle for the capture's first gather + reset)
    losses2 = []
    for k in range(12):
        if k % 5 == 0:
            bank2._new_epoch()
        idx = bank2._perm[(k % 5) * bs:(k % 5 + 1) * bs]
        assert torch.equal((bank2.rays[idx][:, 6] * 1e6).round().long().cpu(), seen[k])
        losses2.append(tr2.step(*bank2.gather(idx)).item())
    assert max(abs(a - b) for a, b in zip(losses, losses2)) < 1e-5 * max(losses), (losses, losses2)


def test_f16_forward_trains_kernel_direct():
    """mlp_mode='f16': fp16 forward, bf16 backward, 8-bit saved state -- the kernel-direct captured step trains, and the 16-bit
    workspaces are refused (they hold bf16 operands for the backward kernels)."""
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    args = O.default_args(mlp_mode="f16")
    tr = Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, args)
    assert tr.direct
    rays, ts = O.synthetic_rays(256, seed=3)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(4)) * 0.2 + 0.4
    losses = [tr.step(rays.to(DEV), ts.to(DEV), target.to(DEV)).item() for _ in range(12)]
    assert tr._graph is not None and all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    # ... and with the 16-bit saved state (bf16 copies of the identity stages for the bf16 backward kernels)
    tr16 = Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, O.default_args(mlp_mode="f16", bwd_fmt=16))
    assert tr16.direct
    losses = [tr16.step(rays.to(DEV), ts.to(DEV), target.to(DEV)).item() for _ in range(12)]
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("mode,fmt", [("bf16", 8), ("bf16", 16), ("bf16x3", 16)])
def test_workspaces_hold_the_tail_waves_tiles(monkeypatch, mode, fmt):
    """The forward and dX kernels run whole workgroups (8 or 4 waves = tiles): with 1001 rays x 64 samples = 2,002 tiles the last
    workgroup's waves past the last point still store their tile.  The workspaces are sized by sr_workspace_tiles (a multiple of 8), so
    nothing is written behind them: canaries right after both workspaces stay intact over a training step."""
    from satnerf_amd import _lib, ops
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    assert _lib.lib().sr_workspace_tiles(1001 * 64) == 2008 and _lib.lib().sr_workspace_tiles(32) == 8
    canaries = []
    real_empty = ops._ws_empty

    def guarded(n, dtype, device, slot):
        if slot not in (1, 2):
            return real_empty(n, dtype, device, slot)
        buf = torch.empty(n + 4096, dtype=dtype, device=device)
        buf[n:] = 0x5A5A if dtype == torch.int16 else 0
        canaries.append(buf[n:])
        return buf[:n]

    monkeypatch.setattr(ops, "_ws_empty", guarded)
    torch.manual_seed(0)
    args = O.default_args(mlp_mode=mode, bwd_fmt=fmt)
    models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
    rays, ts = O.synthetic_rays(1001, seed=3)
    tr = Trainer(models, args, use_graph=False)
    tr.step(rays.to(DEV), ts.to(DEV), torch.rand(1001, 3, device=DEV))
    torch.cuda.synchronize()
    assert len(canaries) >= 2
    for c in canaries:
        assert bool((c == 0x5A5A).all()), "a kernel wrote behind its workspace"
    assert torch.isfinite(tr.state.params).all()


@pytest.mark.parametrize("feat,mode,n_rays,warm", [(256, "bf16", 130, 0.0), (256, "f16", 64, 1.0), (512, "bf16", 37, 0.0)])
def test_one_launch_training_forward_is_bit_identical_to_the_three_launches(feat, mode, n_rays, warm):
    """sr_satnerf_render_train (stratified depths + sky head in the prologue, MLP saving the 8-bit state, compositing + colour loss +
    compositing backward in the epilogue) against sr_ray_setup -> sr_satnerf_mlp_fwd -> sr_render_loss: the same per-ray device functions
    on the same fp32 values, so every output -- and the saved activations -- must agree bit for bit (Next #3)."""
    from satnerf_amd import ops
    from satnerf_amd.models import load_model

    torch.manual_seed(0)
    s, tau = 64, 4
    args = O.default_args(mlp_mode=mode, fc_units=feat)
    model = load_model(args).to(DEV)
    emb = torch.nn.Embedding(30, tau).to(DEV)
    rays, ts = O.synthetic_rays(n_rays, seed=9)
    rays, ts = rays.to(DEV), ts.to(DEV)
    target = torch.rand(n_rays, 3, device=DEV)
    sched = torch.tensor([3.0, 5e-4, warm, 0.0], device=DEV)
    model.repack(mode, backward=True)
    hi, lo, l0 = model.packed(mode)
    sk = model.sky_color
    w = (sk[0].weight.data, sk[0].bias.data, sk[2].weight.data, sk[2].bias.data)
    for u in (torch.rand(n_rays, s, device=DEV), None):  # given jitter / drawn in the kernel (Philox keyed by seed and sched[0])
        acts_a = ops.acts_workspace(n_rays * s, feat, DEV, 8).zero_()
        acts_b = torch.zeros_like(acts_a)
        z, sky = ops.ray_setup(rays, u, s, *w, seed=77, step_counter=sched)
        albedo, sigma, sun_v, beta = ops.satnerf_mlp(rays[:, 0:3], rays[:, 3:6], rays[:, 8:11], z, emb.weight.data, ts, n_rays * s, s, feat, tau, mode,
                                                     hi, lo, l0, acts=acts_a, fmt=8)
        loss, rgb, d_sigma, d_albedo, d_sun, g_beta, d_sky = ops.render_loss(z, sigma.view(n_rays, s), None, 0.0, albedo.view(n_rays, s, 3),
                                                                             sun_v.view(n_rays, s), beta.view(n_rays, s), sky, target, sched=sched)
        r = ops.render_train(rays, ts, emb.weight.data, s, feat, tau, mode, hi, lo, l0, *w, target, acts_b, u=u, seed=77, step_counter=sched,
                             sched=sched, want_z=True)
        torch.cuda.synchronize()
        want = dict(z=z, sky=sky, albedo=albedo.view(n_rays, s, 3), sigma=sigma.view(n_rays, s), sun_v=sun_v.view(n_rays, s), beta=beta.view(n_rays, s),
                    rgb=rgb, d_sigma=d_sigma, d_albedo=d_albedo, d_sun=d_sun, g_beta=g_beta, d_sky=d_sky)
        for k, v in want.items():
            assert torch.equal(r[k], v), (k, u is None, (r[k] - v).abs().max().item())
        # the loss: the same per-ray terms, summed per workgroup instead of per 4 rays
        assert abs(r["loss"].sum().item() - loss.sum().item()) <= 1e-6 * abs(loss.sum().item())
        tiles = (n_rays * s + 31) // 32  # (the padding tiles of the last workgroup hold whatever its idle waves computed)
        per_tile = acts_a.numel() // ((tiles + 7) // 8 * 8)
        assert torch.equal(acts_a[:tiles * per_tile], acts_b[:tiles * per_tile])


def test_trainer_step_with_one_launch_forward_matches_three_launch_step(monkeypatch):
    """The kernel-direct step with the fused forward (default) against SATNERF_TRAIN_FUSED=0 on identical draws: identical MLP gradients."""
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    rays, ts = O.synthetic_rays(200, seed=5)
    target = torch.rand(200, 3, generator=torch.Generator().manual_seed(6))
    u = torch.rand(200, 64, generator=torch.Generator().manual_seed(7))
    grads = []
    for fused in ("1", "0"):
        monkeypatch.setenv("SATNERF_TRAIN_FUSED", fused)
        torch.manual_seed(0)
        args = O.default_args(mlp_mode="bf16")
        tr = Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, args, use_graph=False)
        assert tr._fused_forward() == (fused == "1")
        tr.jitter = lambda n, s, device: u.to(device)
        loss = tr._forward_backward(rays.to(DEV), ts.to(DEV), target.to(DEV))
        grads.append((loss.sum().item(), tr.state.grads.clone()))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-6 * abs(grads[1][0])
    # (sky head and embedding gradients are accumulated with float atomics: equal to rounding, not bit for bit)
    assert maxnorm_rel(grads[0][1].cpu(), grads[1][1].cpu()) < 1e-6


def test_fused_tail_adam_matches_tail_then_adam(monkeypatch):
    """sr_grad_tail_adam (split-K reduction + scatter + Adam per parameter in one launch, the sky head and embedding rows by the last
    atomics block) against sr_grad_tail followed by sr_adam_step_graph: graph-replayed steps of two trainers on identical batches and
    in-kernel jitter (same seed, same device step counter) end with the same parameters and moments, the gradient buffer zeroed."""
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    rays, ts = O.synthetic_rays(256, seed=5)
    rays, ts = rays.to(DEV), ts.to(DEV)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(6)).to(DEV)
    out = []
    for fused in ("1", "0"):
        monkeypatch.setenv("SATNERF_TAIL_ADAM", fused)
        torch.manual_seed(0)
        args = O.default_args(mlp_mode="bf16")
        tr = Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, args, use_graph=True, steps_per_epoch=1000)
        assert tr._late_idx is not None and tr._late_idx.numel() == 899 + 30 * 4  # sky head (3*128 + 128 + 3*128 + 3) + the embedding
        snaps = []
        for _ in range(3):
            tr.step(rays, ts, target, validate=False)
            torch.cuda.synchronize()
            assert float(tr.state.grads.abs().max()) == 0.0  # zeroed for the next step, the atomics' slots included
            assert int(tr.adam_state.view(torch.int32)[3].item()) == 0  # the arrival counter is back at zero
            snaps.append((tr.state.params.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone()))
        assert tr._graph is not None and tr._adam_in_graph
        out.append(snaps)
    late = tr._late_idx.long()
    mlp = torch.ones(out[0][0][0].numel(), dtype=torch.bool, device=DEV)
    mlp[late] = False
    # after ONE step (same parameters in, same gradients): the weight-gradient parameters bit for bit -- same sums, same adam_one; the
    # atomics' parameters to rounding.  Later steps only to a tolerance: the embedding rows differ in their last bits after step 1, and
    # Adam's early updates (+-lr whatever the gradient's size) amplify that wherever a gradient is near zero
    for a, b in zip(out[0][0], out[1][0]):
        assert torch.equal(a[mlp], b[mlp])
        assert maxnorm_rel(a[late].cpu(), b[late].cpu()) < 1e-5
    assert maxnorm_rel(out[0][2][0].cpu(), out[1][2][0].cpu()) < 2e-3


def test_resume_from_checkpoint_continues_the_saved_run(tmp_path):
    """Trainer.save_ckpt / load_ckpt (main.py:241-251 ModelCheckpoint + resume_from_checkpoint): a fresh trainer that loads the checkpoint takes
    the steps the saved run goes on to take -- same weights, moments, schedule position, device-side step (Adam's bias corrections, the key
    of the in-kernel jitter) -- so after three more steps on the same batches the weight-gradient parameters agree bit for bit; and
    checkpoint.load_ckpt reads the same file into an evaluation model by Lightning's key prefixes."""
    from satnerf_amd.checkpoint import load_ckpt
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    batches = []
    for k in range(6):
        rays, ts = O.synthetic_rays(128, seed=300 + k)
        batches.append((rays.to(DEV), ts.to(DEV), torch.rand(128, 3, generator=torch.Generator().manual_seed(400 + k)).to(DEV)))
    args = O.default_args(mlp_mode="bf16")

    def make(seed):
        torch.manual_seed(seed)
        return Trainer({"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}, args, steps_per_epoch=2)

    tr = make(0)
    for b in batches[:3]:
        tr.step(*b, validate=False)
    path = str(tmp_path / "epoch=1.ckpt")
    tr.save_ckpt(path)
    epoch3 = tr.current_epoch()
    for b in batches[3:]:
        la = tr.step(*b, validate=False).item()
    torch.cuda.synchronize()

    tr2 = make(123)  # other initial weights, other jitter seed: everything must come from the file
    tr2.load_ckpt(path)
    assert tr2.n_steps == 3 and tr2.current_epoch() == epoch3 and abs(tr2.lr - tr.lr0 * 0.9) < 1e-12 and tr2.adam_state[0].item() == 3.0
    for b in batches[3:]:
        lb = tr2.step(*b, validate=False).item()
    torch.cuda.synchronize()
    late = tr._late_idx.long()
    mlp = torch.ones(tr.state.params.numel(), dtype=torch.bool, device=DEV)
    mlp[late] = False
    # (the sky head and the embedding rows take their gradients by float atomics: equal to rounding, and they feed the later steps)
    assert maxnorm_rel(tr2.state.params[mlp].cpu(), tr.state.params[mlp].cpu()) < 2e-3
    assert abs(la - lb) < 1e-4 * abs(la) and tr2.n_steps == tr.n_steps == 6
    ev = load_model(args).to(DEV)
    load_ckpt(ev, path, model_name="nerf_coarse")
    assert all(torch.isfinite(p_).all() for p_ in ev.parameters())
