"""Host logic of the weight-stream packer, validated on CPU through a lane-accurate emulation of the kernel."""
import numpy as np
import pytest
import torch

from oracle import satnerf_oracle as O
from satnerf_amd import packing
from tests.mfma_emulator import Emulator


def test_slot_permutation_is_a_bijection_per_32_block():
    f = packing.slot_to_feat(np.arange(256))
    assert sorted(f.tolist()) == list(range(256))
    for b in range(8):
        assert sorted(f[32 * b:32 * b + 32].tolist()) == list(range(32 * b, 32 * b + 32))


def test_param_shapes_match_reference_state_dict_layout():
    want = O.satnerf_param_shapes(256, 4)
    got = packing.satnerf_param_shapes(256, 4)
    assert list(want.items()) == [(k, tuple(v)) for k, v in got.items()]
    assert packing.forward_maps(256, 4)["n_params"] == 662537


def test_every_mlp_parameter_appears_exactly_once():
    m = packing.forward_maps(256, 4)
    used = np.concatenate([m["idx"][m["idx"] >= 0], m["l0_idx"]])
    counts = np.bincount(used, minlength=m["n_params"])
    off = m["offsets"]
    sky = np.zeros(m["n_params"], bool)
    for k in ("sky_color.0.weight", "sky_color.0.bias", "sky_color.2.weight", "sky_color.2.bias"):
        o, shp = off[k]
        sky[o:o + int(np.prod(shp))] = True
    assert (counts[~sky] == 1).all() and (counts[sky] == 0).all()  # the sky head runs per ray in its own kernel


@pytest.mark.parametrize("tau,bf16,tol", [(4, False, 2e-6), (16, False, 2e-6), (4, True, 2e-2)])
def test_emulated_kernel_dataflow_matches_oracle(tau, bf16, tol):
    p = O.procedural_satnerf_params(256, tau, seed=3)
    flat = torch.cat([p[k].reshape(-1) for k in packing.satnerf_param_shapes(256, tau)]).numpy()
    g = torch.Generator().manual_seed(4)
    xyz = torch.rand(32, 3, generator=g) * 2 - 1
    sun = torch.randn(32, 3, generator=g)
    sun = sun / sun.norm(dim=1, keepdim=True)
    t = torch.rand(32, tau, generator=g) * 2 - 1
    ref = O.satnerf_mlp({k: v.double() for k, v in p.items()}, xyz.double(), sun.double(), t.double()).numpy()
    alb, sig, sv, beta = Emulator(flat, 256, tau, bf16).forward_tile(xyz.double().numpy(), sun.double().numpy(), t.double().numpy())
    assert np.abs(alb - ref[:, :3]).max() < tol
    assert np.abs(sig - ref[:, 3]).max() < tol
    assert np.abs(sv - ref[:, 4]).max() < tol
    assert np.abs(beta - ref[:, 8]).max() < tol


@pytest.mark.parametrize("tau", [4, 16])
def test_emulated_backward_dataflow_matches_autograd(tau):
    """Transposed stream + weight-gradient job table + gradient scatter map, validated through the lane-accurate emulation
    of csrc/mlp_bwd.hip and csrc/wgrad.hip against autograd through the oracle (fp64)."""
    p = O.procedural_satnerf_params(256, tau, seed=3)
    shapes = packing.satnerf_param_shapes(256, tau)
    flat = torch.cat([p[k].reshape(-1) for k in shapes]).numpy()
    g = torch.Generator().manual_seed(5)
    xyz = (torch.rand(32, 3, generator=g) * 2 - 1).double()
    sun = torch.randn(32, 3, generator=g).double()
    sun = sun / sun.norm(dim=1, keepdim=True)
    t = (torch.rand(32, tau, generator=g) * 2 - 1).double().requires_grad_(True)
    ga, gs, gv, gb = (torch.randn(32, 3, generator=g).double(), torch.randn(32, generator=g).double(), torch.randn(32, generator=g).double(),
                      torch.randn(32, generator=g).double())
    pd = {k: v.double().requires_grad_(True) for k, v in p.items()}
    out = O.satnerf_mlp(pd, xyz, sun, t)
    ((out[:, :3] * ga).sum() + (out[:, 3] * gs).sum() + (out[:, 4] * gv).sum() + (out[:, 8] * gb).sum()).backward()
    em = Emulator(flat, 256, tau)
    em.forward_tile(xyz.numpy(), sun.numpy(), t.detach().numpy())
    grad, d_t = em.backward_tile(ga.numpy(), gs.numpy(), gv.numpy(), gb.numpy())
    bm = packing.backward_maps(256, tau)
    assert bm["blocks"].shape == (14, 12) and (bm["gidx"] < 0).sum() == 899  # only the sky head is produced elsewhere
    for k, (o, shp) in bm["offsets"].items():
        if k.startswith("sky"):
            continue
        n = int(np.prod(shp))
        want = pd[k].grad.reshape(-1).numpy()
        assert np.abs(grad[o:o + n] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-30), k
    assert np.abs(d_t - t.grad.numpy()).max() <= 1e-5 * np.abs(t.grad.numpy()).max()


@pytest.mark.parametrize("feat,tau", [(256, 4), (256, 16), (512, 4)])
def test_pack_scatter_map_is_the_inverse_of_the_gather_maps(feat, tau):
    """packing.pack_scatter_map (what sr_grad_tail_adam's `pack` walks: parameter -> its places in the packed streams) against the gather
    maps sr_pack_all runs: scattering every parameter reproduces the gathered forward stream | transposed stream and the fc_net.0 table
    exactly (the same fp32 products), every live stream element is written exactly once, constant-zero elements never."""
    import numpy as np

    fm, bm = packing.forward_maps(feat, tau), packing.backward_maps(feat, tau)
    m, scales = packing.pack_scatter_map(feat, tau)
    n = int(bm["n_params"])
    assert m.shape == (n, 2) and m.dtype == np.int32 and scales.shape == (4,)
    src = np.random.default_rng(5).standard_normal(n).astype(np.float32)
    idx = np.concatenate([fm["idx"], bm["idx"]])
    scale = np.concatenate([fm["scale"], bm["scale"]]).astype(np.float32)
    want = np.where(idx >= 0, src[np.maximum(idx, 0)] * scale, np.float32(0)).astype(np.float32)
    want_l0 = np.where(fm["l0_idx"] >= 0, src[np.maximum(fm["l0_idx"], 0)] * fm["l0_scale"].astype(np.float32), np.float32(0)).astype(np.float32)
    got, got_l0 = np.zeros_like(want), np.zeros_like(want_l0)
    hits, hits_l0 = np.zeros(want.size, np.int32), np.zeros(want_l0.size, np.int32)
    for o in range(2):
        c = m[:, o]
        ok = c >= 0
        pos, si, is_l0 = c & ((1 << packing.PACK_POS_BITS) - 1), (c >> packing.PACK_POS_BITS) & 3, (c & packing.PACK_L0_FLAG) != 0
        v = (src * scales[si]).astype(np.float32)
        s_ = ok & ~is_l0
        got[pos[s_]] = v[s_]
        np.add.at(hits, pos[s_], 1)
        t_ = ok & is_l0
        got_l0[pos[t_]] = v[t_]
        np.add.at(hits_l0, pos[t_], 1)
    assert np.array_equal(got, want) and np.array_equal(got_l0, want_l0)
    live = (idx >= 0) & (scale != 0)
    assert np.array_equal(hits, live.astype(np.int32)) and np.array_equal(hits_l0, ((fm["l0_idx"] >= 0) & (fm["l0_scale"] != 0)).astype(np.int32))


@pytest.mark.parametrize("feat", [256, 512])
def test_duty_tables_of_the_weight_gradient_kernel_cover_every_operand_once(feat):
    """packing.wgrad9_duties / wgrad9_variants (r06: thin streams for the one-row head blocks): in EVERY block each row and column fragment is
    decoded by exactly one duty that its wave's stream executes, into the LDS fragment the MFMAs read it from; the raw fragment of a block with
    a bf16 row sits on wave 0, the aux fragment on wave 1; a thin block is exactly a block whose row operand is one raw fragment over PHASE8
    columns, and its variants hold the stream shapes gen/wgrad9_loop.py generates."""
    import numpy as np

    bm = packing.backward_maps(feat, 4)
    auxs = bm["auxs"]
    duties = packing.wgrad9_duties(feat, 4).reshape(-1, 4, 5, 4)
    variants = packing.wgrad9_variants(feat, 4)
    thin = packing.wgrad9_thin_blocks(feat, 4)
    assert len(thin) == (2 if feat == 256 else 5)
    for b, (rows, cols) in enumerate(zip(bm["block_rows"], bm["block_cols"])):
        is_thin = b in thin
        assert is_thin == bool(variants[b].any())
        assert is_thin == (len(rows) == 1 and packing.dpre8_source(rows[0], feat)["codec"] == packing.RAW16 and bm["blocks"][b, 8] == packing.KIND_PHASE)
        got = {}   # LDS fragment -> (source, unit) of the duty that fills it
        for w in range(4):
            v = int(variants[b, w])
            n_df, raw = (4, True) if v == 0 else packing.WG9_THIN[v - 1][:2]
            for k in list(range(n_df)) + ([4] if raw else []):
                src, unit, dst, _ = (int(x) for x in duties[b, w, k])
                if dst == packing.WG9_DUMP_FRAG:
                    continue
                assert dst not in got, (b, w, k, "two duties fill one fragment")
                got[dst] = (src, unit)
                if k < 4:
                    assert dst % 2 == 0 and dst + 1 not in got   # a double fragment fills an even / odd pair
                    if v:   # thin streams read every double fragment from the activation workspace
                        assert src == packing.SRC_ACTS and dst >= 16
        want = {}
        pos = 0
        while pos < len(rows):
            d = packing.dpre8_source(rows[pos], feat)
            want[pos] = (packing.SRC_DPRE, d["unit"])
            pos += 1 if d["codec"] == packing.RAW16 else 2
        pos = 0
        while pos < len(cols):
            want[16 + pos] = (packing.SRC_ACTS, packing.act8_source(cols[pos], auxs, feat)["unit"])
            pos += 2
        for a in range(auxs):
            want[32 + a] = (packing.SRC_ACTS, a)
        assert got == want, (b, sorted(set(got) ^ set(want)))
        if any(packing.dpre8_source(r, feat)["codec"] == packing.RAW16 for r in rows):
            assert int(duties[b, 0, 4, 0]) == packing.SRC_DPRE   # the bf16 row fragment: wave 0's raw duty
        assert tuple(int(x) for x in duties[b, 1, 4][:3]) == (packing.SRC_ACTS, 0, 32)   # the aux fragment: wave 1's
    # the thin variants are the shapes the generator emits, in the kernel's order
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("w9gen", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "satnerf_amd", "csrc", "gen", "wgrad9_loop.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert list(gen.THIN) == list(packing.WG9_THIN)
    loads = packing.wgrad8_loads(feat, 4)
    assert loads.shape[1] == packing.WG8_LOAD_INTS == 113 and np.array_equal(loads[:, 109:113], variants)
