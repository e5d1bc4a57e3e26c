"""Host-side model container: reference-compatible state_dict layout, flat parameter buffer, error behaviour (CPU)."""
import pytest
import torch

from oracle import satnerf_oracle as O
from satnerf_amd import rendering
from satnerf_amd.models import SatNeRF, load_model


def test_state_dict_keys_shapes_and_order_match_reference():
    m = SatNeRF(feat=256, t_embedding_dims=4)
    want = O.satnerf_param_shapes(256, 4)
    sd = m.state_dict()
    assert list(sd.keys()) == list(want.keys())
    assert all(tuple(sd[k].shape) == want[k] for k in want)
    assert m.number_of_outputs == 9


def test_parameters_alias_one_flat_buffer_and_survive_load_and_to():
    m = load_model(O.default_args())
    flat = m.flat_params()
    assert flat.numel() == 662537
    p = O.procedural_satnerf_params(256, 4, seed=1)
    v0 = m.weights_version()
    m.load_state_dict(p)
    assert m.weights_version() > v0  # the pack cache keys on this
    v1 = m.weights_version()
    m.flat_params().mul_(1.0)  # a fused optimizer updates the flat buffer directly
    assert m.weights_version() > v1
    assert torch.equal(m.flat_params()[:768], p["fc_net.0.weight"].reshape(-1))
    m2 = m.double().float()
    assert m2.flat_params().data_ptr() == next(m2.parameters()).data_ptr()
    assert torch.equal(m2.state_dict()["beta_from_xyz.2.bias"], p["beta_from_xyz.2.bias"])


def test_init_ranges_follow_siren_init():
    torch.manual_seed(0)
    m = SatNeRF(feat=256, t_embedding_dims=4)
    sd = m.state_dict()
    assert sd["fc_net.0.weight"].abs().max() <= 1 / 3
    assert sd["fc_net.2.weight"].abs().max() <= (6 / 256) ** 0.5
    assert sd["fc_net.8.weight"].abs().max() <= (6 / 259) ** 0.5
    assert sd["sun_v_net.0.weight"].abs().max() <= 1 / 259
    assert sd["feats_from_xyz.weight"].abs().max() <= (1 / 256) ** 0.5 + 1e-6


def test_cpu_inputs_fail_loudly_no_fallback():
    args = O.default_args()
    m = load_model(args)
    rays, ts = O.synthetic_rays(8)
    with pytest.raises(RuntimeError):
        rendering.render_rays({"coarse": m, "t": torch.nn.Embedding(30, 4)}, args, rays, ts)
    with pytest.raises(TypeError):
        rendering.render_rays({"coarse": m, "t": torch.nn.Embedding(30, 4)}, args, rays, None)
    sn = load_model(O.default_args(model="s-nerf"))  # ShadowNeRF: the reference's keys / parameter count, no uncertainty head in sight
    assert list(sn.state_dict()) == list(O.snerf_param_shapes(256)) and sn.number_of_outputs == 8
    assert sum(p.numel() for p in sn.parameters() if p.requires_grad) == sum(int(torch.tensor(v).prod()) for v in O.snerf_param_shapes(256).values())
    sn.load_state_dict(O.procedural_snerf_params(256, seed=2))  # strict load of the reference layout
    sn.dummy_embedding()
    assert list(sn.state_dict()) == list(O.snerf_param_shapes(256)) and sn._flat.numel() == 662537  # (the dummy stays outside)
    with pytest.raises(RuntimeError):
        rendering.render_rays({"coarse": sn}, O.default_args(model="s-nerf"), rays, ts)
    with pytest.raises(ValueError):
        load_model(O.default_args(model="bogus"))


def test_lightning_style_checkpoint_loads_by_reference_key_layout(tmp_path):
    """A checkpoint written the way NeRF_pl does (main.py:51-58 attribute prefixes) loads through the eval_satnerf.py:23-93 twins."""
    import json

    from satnerf_amd import checkpoint

    args = O.default_args(n_importance=64)
    coarse, fine = O.procedural_satnerf_params(256, 4, seed=1), O.procedural_satnerf_params(256, 4, seed=2)
    emb = O.procedural_uniform((30, 4), 1.0, 7)
    sd = {f"nerf_coarse.{k}": v for k, v in coarse.items()}
    sd.update({f"nerf_fine.{k}": v for k, v in fine.items()})
    sd["embedding_t.weight"] = emb
    run = "run0"
    (tmp_path / "logs" / run).mkdir(parents=True)
    (tmp_path / "ckpts" / run).mkdir(parents=True)
    torch.save({"state_dict": sd, "epoch": 3}, tmp_path / "ckpts" / run / "epoch=3.ckpt")
    json.dump(vars(args), open(tmp_path / "logs" / run / "opts.json", "w"))
    models, loaded_args = checkpoint.load_nerf(run, str(tmp_path / "logs"), str(tmp_path / "ckpts"), 3, device="cpu")
    assert loaded_args.n_importance == 64 and set(models) == {"coarse", "fine", "t"}
    assert all(torch.equal(models["coarse"].state_dict()[k], v) for k, v in coarse.items())
    assert all(torch.equal(models["fine"].state_dict()[k], v) for k, v in fine.items())  # NOT the coarse weights / random init
    assert torch.equal(models["t"].weight.data, emb)
    assert models["coarse"].flat_params().numel() == 662537  # still one flat buffer after load
    with pytest.raises(FileNotFoundError):
        checkpoint.load_nerf(run, str(tmp_path / "logs"), str(tmp_path / "ckpts"), 9, device="cpu")


def test_rays_from_reference_cache_format(tmp_path):
    """data.rays_from_cache: the reference's <img_id>.data cache (torch.save of (HW,8) ECEF rays) -> normalised (HW,11) rays with
    the sun direction, as SatelliteDataset.load_data assembles them (datasets/satellite.py:185-227,232-244)."""
    import math

    from satnerf_amd.data import rays_from_cache, sun_direction

    g = torch.Generator().manual_seed(5)
    center, rng = [794000.0, -5455000.0, 3200000.0], 431.7
    o = torch.tensor(center, dtype=torch.float64) + (torch.rand(50, 3, generator=g, dtype=torch.float64) - 0.5) * 2 * rng
    d = torch.nn.functional.normalize(torch.randn(50, 3, generator=g, dtype=torch.float64), dim=1)
    ecef = torch.cat([o, d, torch.zeros(50, 1, dtype=torch.float64), 300 + 50 * torch.rand(50, 1, generator=g, dtype=torch.float64)], 1)
    path = tmp_path / "JAX_068_001_RGB.data"
    torch.save(ecef, path)
    rays = rays_from_cache(str(path), center, rng, 47.5, 153.2)
    assert rays.shape == (50, 11) and rays.dtype == torch.float32
    assert torch.allclose(rays[:, 0:3].double(), (o - torch.tensor(center, dtype=torch.float64)) / rng, atol=1e-6)
    assert torch.allclose(rays[:, 3:6].double(), d, atol=1e-7) and torch.allclose(rays[:, 7].double(), ecef[:, 7] / rng, atol=1e-6)
    el, az = math.radians(47.5), math.radians(153.2)
    want = torch.tensor([math.sin(az) * math.cos(el), math.cos(az) * math.cos(el), math.sin(el)])
    assert torch.allclose(rays[:, 8:11], want.expand(50, 3), atol=1e-7) and abs(sun_direction(47.5, 153.2).norm().item() - 1) < 1e-6
    assert torch.equal(torch.load(path), ecef)  # the cache itself is not modified
    with pytest.raises(ValueError):
        rays_from_cache(torch.zeros(4, 11), center, rng, 1, 2)


def test_package_synthetic_rays_follow_the_survey_recipe():
    """satnerf_amd.data.synthetic_rays / default_args (what bench.py's GPU legs use) equal the oracle's recipe."""
    from satnerf_amd import data

    for n, seed in ((257, 20240628), (64, 3)):
        r1, t1 = data.synthetic_rays(n, seed=seed)
        r2, t2 = O.synthetic_rays(n, seed=seed)
        assert torch.equal(t1, t2) and torch.allclose(r1, r2, atol=1e-7) and r1.dtype == torch.float32
        assert torch.allclose(r1[:, 3:6].norm(dim=1), torch.ones(n), atol=1e-6) and (r1[:, 6] == 0).all()
    a, b = data.default_args(n_samples=50), O.default_args(n_samples=50)
    assert all(getattr(a, k) == v for k, v in vars(b).items())
