"""Sampling law of the on-device ray bank (needs device tensors -> GPU-marked; the permutation logic itself is trivial)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ray_bank_visits_every_ray_once_per_epoch_and_shards_across_ranks():
    from satnerf_amd.data import RayBank

    dev = "cuda:0"
    n, b = 1000, 64
    rays = torch.arange(n, device=dev, dtype=torch.float32).unsqueeze(1).repeat(1, 11)
    rgbs = torch.zeros(n, 3, device=dev)
    ts = torch.arange(n, device=dev) % 7
    banks = [RayBank(rays, rgbs, ts, b, seed=5, rank=r, world_size=2) for r in range(2)]
    seen = []
    for bank in banks:
        assert len(bank) == (n // 2) // b
        got = torch.cat([bank.next_batch()[0][:, 0] for _ in range(len(bank))])
        assert got.numel() == len(bank) * b and got.unique().numel() == got.numel()
        seen.append(got)
    assert torch.cat(seen).unique().numel() == 2 * len(banks[0]) * b  # the two ranks never draw the same ray in an epoch
    r, t, c = banks[0].next_batch()  # rolls into the next epoch with a fresh permutation
    assert banks[0].epoch == 1 and r.shape == (b, 11) and t.dtype == torch.int64 and c.shape == (b, 3)
    assert torch.equal(t, r[:, 0].long() % 7)
