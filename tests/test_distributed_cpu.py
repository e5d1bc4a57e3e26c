"""Host logic of the data-parallel path on CPU: 2 processes, gloo (the GPU run uses the same code over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import satnerf_oracle as O
from satnerf_amd.models import load_model
from satnerf_amd.train import FlatState, shard_rays


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)  # identical init on every rank
    model = load_model(O.default_args())
    emb = torch.nn.Embedding(30, 4)
    st = FlatState([model, emb])
    # every parameter and every .grad is a view of the two flat buffers
    assert st.params.numel() == 662537 + 120
    assert next(model.parameters()).data_ptr() == st.params.data_ptr()
    assert emb.weight.data_ptr() == st.params.data_ptr() + 662537 * 4
    assert emb.weight.grad.data_ptr() == st.grads.data_ptr() + 662537 * 4
    for i, p in enumerate(model.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    emb.weight.grad.fill_(10.0 * (rank + 1))
    st.allreduce_mean_(world)
    want = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p.grad, want * (i + 1))) for i, p in enumerate(model.parameters()))
    ok = ok and torch.allclose(emb.weight.grad, torch.full_like(emb.weight.grad, 10.0 * want))
    # replicas stay identical after an optimizer step on the reduced gradient
    leaf = torch.nn.Parameter(st.params)
    leaf.grad = st.grads
    torch.optim.Adam([leaf], lr=5e-4).step()
    gathered = [torch.empty_like(st.params) for _ in range(world)]
    dist.all_gather(gathered, st.params)
    ok = ok and all(torch.equal(g, gathered[0]) for g in gathered)
    # the optimizer wrote through to the module parameters
    ok = ok and model.state_dict()["fc_net.0.weight"].data_ptr() == st.params.data_ptr()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_shard_rays_covers_everything_once():
    for n in (0, 1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_rays(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _shard_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satnerf_amd.rendering import batched_inference_sharded

    n = 1003  # ragged: 502 + 501
    rays = torch.arange(n * 11, dtype=torch.float32).view(n, 11)
    ts = torch.arange(n)

    def fake_render(models, r, t, args):  # stands in for the GPU renderer: per-ray outputs that identify the ray
        return {"rgb_coarse": r[:, :3] * 2 + t[:, None].float(), "depth_coarse": r[:, 7] + 1, "weights_coarse": r[:, :4].repeat(1, 16), "sky": None}

    res = batched_inference_sharded({}, rays, ts, None, render_fn=fake_render)
    full = fake_render({}, rays, ts, None)
    ok = all((res[k] is None and full[k] is None) or torch.equal(res[k], full[k]) for k in full)
    out[rank] = bool(ok and res["weights_coarse"].shape == (n, 64))
    dist.destroy_process_group()


def test_sharded_batched_inference_gathers_contiguous_row_blocks():
    """Evaluation sharding: every rank ends up with the whole image, in ray order, ragged shares included."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_shard_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def _vote_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satnerf_amd.train import Trainer

    tr = Trainer.__new__(Trainer)  # the vote needs no trainer state: only the process group's rendezvous store
    votes = [tr._any_rank(False), tr._any_rank(rank == 1), tr._any_rank(True), tr._any_rank(False)]
    # a SECOND trainer of the same job (re-created, or another test sharing the process group): its votes must not meet the first
    # trainer's counters in the store (keys are numbered per process, not per trainer).  Rank 0 arrives late on purpose: a
    # re-used key would already hold `world` ticks and rank 1 would read `failed` before rank 0 has added its flag
    tr2 = Trainer.__new__(Trainer)
    if rank == 0:
        import time

        time.sleep(0.3)
    votes += [tr2._any_rank(rank == 0), tr2._any_rank(False)]
    out[rank] = votes
    dist.destroy_process_group()


def test_capture_failure_vote_is_the_same_on_every_rank():
    """Trainer._any_rank: if the RCCL all-reduce cannot be captured on ANY rank, EVERY rank must re-capture without it (a rank retrying
    alone would issue warm-up collectives nobody matches).  The vote goes through the rendezvous store, not a collective."""
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_vote_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: [False, True, True, False, True, False], 1: [False, True, True, False, True, False]}
