"""RPC ray generation (datasets/satellite.py:18-65).  The reference's localisation lives in the third-party
``rpcm`` package, absent offline: the oracle restates the published RPC00B model (oracle/rpc_oracle.py, "parity unpinned"); these
tests check its internal consistency on CPU and the HIP kernel against it on the GPU."""
import numpy as np
import pytest
import torch

from oracle import rpc_oracle as R

CENTER, RANGE = [796912.4, -5453871.2, 3200310.9], 310.0


def _ecef_center(rpc):
    return [float(v) for v in R.latlon_to_ecef(np.float64(rpc["lat_offset"]), np.float64(rpc["lon_offset"]), np.float64(10.0))]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_localization_inverts_the_projection(seed):
    rpc = R.synthetic_rpc(seed)
    cols, rows = np.meshgrid(np.arange(0, 512, 37.0), np.arange(0, 512, 41.0))
    for alt in (-40.0, 0.0, 95.0):
        lon, lat = R.localization(rpc, cols.ravel(), rows.ravel(), np.full(cols.size, alt))
        c, r = R.projection(rpc, lon, lat, np.full(cols.size, alt))
        assert np.abs(c - cols.ravel()).max() < 1e-6 and np.abs(r - rows.ravel()).max() < 1e-6
    # altitude parallax: the same pixel at two altitudes lands on two different ground points, and rays are unit vectors
    rays = R.get_rays(cols.ravel(), rows.ravel(), rpc, -25.0, 60.0)
    assert rays.shape == (cols.size, 8) and rays.dtype == np.float32
    assert np.allclose(np.linalg.norm(rays[:, 3:6].astype(np.float64), axis=1), 1.0, atol=1e-6)
    assert (rays[:, 6] == 0).all() and (rays[:, 7] > 85.0).all() and (rays[:, 7] < 200.0).all()  # 85 m of altitude, oblique view


def test_oracle_rescale_rpc_maps_the_downscaled_grid():
    """sat_utils.rescale_rpc (sat_utils.py:44-57): pixel (c, r) of the half-size image sees the ground point of pixel (2c, 2r)."""
    rpc = R.synthetic_rpc(3)
    half = R.rescale_rpc(rpc, 0.5)
    c, r = np.array([10.0, 100.0, 200.0]), np.array([7.0, 50.0, 250.0])
    a = R.localization(rpc, 2 * c, 2 * r, np.full(3, 12.0))
    b = R.localization(half, c, r, np.full(3, 12.0))
    assert np.abs(a[0] - b[0]).max() < 1e-10 and np.abs(a[1] - b[1]).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("seed,h,w,down", [(0, 64, 96, 1.0), (4, 200, 120, 2.0)])
def test_hip_rpc_rays_match_the_oracle(seed, h, w, down, tmp_path):
    from satnerf_amd import data

    rpc = R.synthetic_rpc(seed, height=h, width=w)
    center = _ecef_center(rpc)
    path = str(tmp_path / "cache" / "IMG_007.data")
    rays = data.rays_from_rpc(rpc, h, w, -25.0, 60.0, center, RANGE, 52.0, 141.0, device="cuda:0", img_downscale=down, cache_path=path)
    hh, ww = int(h // down), int(w // down)
    want = R.image_rays(R.rescale_rpc(rpc, 1.0 / down), hh, ww, -25.0, 60.0, center, RANGE, 52.0, 141.0)
    got = rays.cpu().numpy()
    assert got.shape == want.shape == (hh * ww, 11)
    # ECEF origins are ~6e6 m in fp32 (0.5 m ulp) before the centre is subtracted: one ulp of that, divided by the range, is the
    # resolution of BOTH paths; they must agree to it, directions / far / sun to fp32 rounding
    assert np.abs(got[:, 0:3] - want[:, 0:3]).max() <= 0.5 / RANGE + 1e-6
    assert np.abs(got[:, 3:6] - want[:, 3:6]).max() < 2e-6
    assert np.abs(got[:, 6:8] - want[:, 6:8]).max() < 2e-6 and np.abs(got[:, 8:11] - want[:, 8:11]).max() < 1e-6
    # the cache file is what the reference would have written (torch.save of the (H*W, 8) fp32 rays) and feeds rays_from_cache
    cache = torch.load(path)
    assert cache.shape == (hh * ww, 8) and cache.dtype == torch.float32
    ref8 = R.get_rays(*[g.flatten() for g in np.meshgrid(np.arange(ww), np.arange(hh))], R.rescale_rpc(rpc, 1.0 / down), -25.0, 60.0)
    assert np.abs(cache.numpy()[:, :3] - ref8[:, :3]).max() <= 0.5 and np.abs(cache.numpy()[:, 3:] - ref8[:, 3:]).max() < 1e-4
    again = data.rays_from_cache(path, center, RANGE, 52.0, 141.0)
    assert np.abs(again.numpy() - got).max() < 1e-6
