"""The table of exponent maxima the dX kernel leaves behind the dpre workspace (csrc/mlp_layout.h, SR_FMT8): every byte is recomputed on
the host from the workspace bytes the same launch wrote.  The 4-wave weight-gradient kernel fits fp16's range from this table (r04
scanned the exponent bytes itself), so a wrong entry would silently flush or overflow a block's rows."""
import numpy as np
import pytest
import torch

from oracle import satnerf_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.mark.parametrize("feat,n_rays", [(256, 301), (512, 70)])
def test_dx_kernel_leaves_the_exponent_maxima(monkeypatch, feat, n_rays):
    from satnerf_amd import _lib, ops, packing
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    torch.manual_seed(0)
    s = 64
    args = O.default_args(mlp_mode="bf16", fc_units=feat)
    models = {"coarse": load_model(args).to(DEV), "t": torch.nn.Embedding(30, 4).to(DEV)}
    rays, ts = O.synthetic_rays(n_rays, seed=3)
    seen = {}
    real = ops.wgrad_partials

    def spy(feat_, tau, n_points, dpre, acts, *a, **k):
        seen["dpre"], seen["acts"], seen["n"] = dpre, acts, n_points
        return real(feat_, tau, n_points, dpre, acts, *a, **k)

    monkeypatch.setattr(ops, "wgrad_partials", spy)
    tr = Trainer(models, args, use_graph=False)
    assert tr.direct
    tr._forward_backward(rays.to(DEV), ts.to(DEV), torch.rand(n_rays, 3, device=DEV))
    torch.cuda.synchronize()
    n_points = seen["n"]
    assert n_points == n_rays * s
    lib = _lib.lib()
    tiles = lib.sr_workspace_tiles(n_points)
    g8 = packing.fmt8_geometry(feat)
    dk, ak = packing.dpre8_units(feat), lib.sr_act_elems_per_tile(feat, 8) // 512 - 1   # (the size query assumes two aux fragments; tau 4 has one)
    dpre = seen["dpre"].cpu().numpy().view(np.uint8)
    assert dpre.size == lib.sr_dpre_workspace_elems(n_points, feat, 8) * 2 == tiles * dk * 1024 + (tiles // 4 * 16 + 1023) // 1024 * 1024
    body = dpre[:tiles * dk * 1024].reshape(tiles, dk, 64, 16)
    table = dpre[tiles * dk * 1024:][:tiles // 4 * 16].reshape(tiles // 4, 16)
    acts = seen["acts"].cpu().numpy().view(np.uint8)[:tiles * ak * 1024].reshape(tiles, ak, 64, 16)
    mt, gpu = g8["MT"], g8["GROUPS_PER_UNIT"]
    want = np.zeros((tiles // 4, 16), np.uint8)
    per4 = lambda x: x.reshape(tiles // 4, -1).max(1)  # noqa: E731
    for g in range(14):
        n_bytes = mt if g < 9 else g8["MTH"]
        unit = body[:, g8["D8_SCALE"] + g // gpu, :, (g % gpu) * mt:(g % gpu) * mt + n_bytes]
        want[:, g] = per4(unit)
    want[:, packing.EMAX_FEATS] = per4(acts[:, 1 + g8["A8_SCALE"], :, :mt])
    raw = np.stack([body[:, g8["D8_SIGMA"]], body[:, g8["D8_HEAD"]]], 1).reshape(tiles, -1).view(np.uint16)
    want[:, packing.EMAX_RAW] = per4(((raw & 0x7fff) >> 7).astype(np.uint8))
    used = ((n_points + 31) // 32 + 3) // 4   # entries the weight-gradient kernel reads: those of real tiles (a 4-wave workgroup at width 512
    table, want = table[:used], want[:used]    # covers 4 tiles, so the workspace's padding tiles may never be visited)
    bad = np.argwhere(table != want)
    assert bad.size == 0, (bad[:10], table[bad[:5, 0]], want[bad[:5, 0]])
    # sanity of the fixture itself: the exponents are not all one value, and the gradients are in the range the kernel's clamp assumes
    assert want[:, :14].max() > want[:, :14].min() and 32 < int(want[:, :14].max()) < 254
