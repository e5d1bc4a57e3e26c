"""GPU-resident ray bank + on-device batch sampler.

Replaces ``DataLoader(train_dataset, shuffle=True, num_workers=4, batch_size=B, pin_memory=True)`` (main.py:96-110) over the
``(sum HW, 11)`` ray tensor ``SatelliteDataset`` builds (datasets/satellite.py:160-216, items ``{"rays": (11,), "rgbs": (3,),
"ts": (1,)}``, :347-350).  At > 1 M rays/s a per-item ``__getitem__`` + collate + H2D copy is two orders of magnitude too
slow; here the whole bank lives in HBM (45 B/ray: a 100 M-ray scene is 4.5 GB of 288 GB), an epoch is one ``randperm`` on the
device and a batch is one row gather.  Same sampling law as the DataLoader: every ray exactly once per epoch, in random order,
the last short batch dropped (``drop_last``) or kept.  Ranks draw disjoint strided shares of each epoch's permutation.
"""
from __future__ import annotations

import torch


class RayBank:
    def __init__(self, rays, rgbs, ts, batch_size, seed=0, rank=0, world_size=1, drop_last=True):
        if not (rays.is_cuda and rgbs.is_cuda and ts.is_cuda):
            raise ValueError("RayBank tensors must already be on the GPU")
        n = rays.shape[0]
        if rgbs.shape[0] != n or ts.reshape(-1).shape[0] != n:
            raise ValueError("rays / rgbs / ts disagree on the number of rays")
        self.rays, self.rgbs, self.ts = rays.contiguous().float(), rgbs.contiguous().float(), ts.reshape(-1).contiguous().long()
        self.batch_size, self.rank, self.world, self.drop_last = batch_size, rank, world_size, drop_last
        self.gen = torch.Generator(device=rays.device)
        self.gen.manual_seed(seed)  # same seed on every rank: the ranks slice ONE permutation
        self.epoch, self._perm, self._pos = -1, None, 0

    def __len__(self):
        share = (self.rays.shape[0] - self.rank + self.world - 1) // self.world
        return share // self.batch_size if self.drop_last else (share + self.batch_size - 1) // self.batch_size

    def _new_epoch(self):
        self.epoch += 1
        perm = torch.randperm(self.rays.shape[0], device=self.rays.device, generator=self.gen)
        self._perm, self._pos = perm[self.rank::self.world], 0

    def next_indices(self):
        if self._perm is None or self._pos + (self.batch_size if self.drop_last else 1) > self._perm.numel():
            self._new_epoch()
        idx = self._perm[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        return idx

    # ---- in-graph sampler: the captured training step gathers its own batch (no host work between replays) ----------------------
    def graph_source(self):
        """(epoch index buffer (batches * B,) int64, cursor zeros(4) float32, batches) for ``ops.gather_batch(..., cursor=...)`` inside a
        captured step: the launch takes batch cursor[0] of the buffer and advances the cursor; ``graph_advance()`` after every
        replay keeps the host in step and reshuffles the buffer IN PLACE when an epoch ends.  Needs drop_last (full batches)."""
        if not self.drop_last:
            raise ValueError("the in-graph sampler serves full batches only (drop_last=True)")
        if getattr(self, "_gperm", None) is None:
            self._gbatches = len(self)
            if self._gbatches < 1:
                raise ValueError("the bank holds less than one batch")
            self._gperm = torch.empty(self._gbatches * self.batch_size, dtype=torch.int64, device=self.rays.device)
            self._gcursor = torch.zeros(4, dtype=torch.float32, device=self.rays.device)
            self.graph_reset()
        return self._gperm, self._gcursor, self._gbatches

    def graph_reset(self):
        """Start a fresh epoch at batch 0 (also after the capture's warm-up passes moved the cursor)."""
        self._new_epoch()
        self._gperm.copy_(self._perm[:self._gperm.numel()])
        self._gcursor.zero_()
        self._gpos = 0

    def graph_advance(self):
        """Host mirror of the device cursor: call once per replay; the next epoch's shuffle is written when the last batch is out."""
        self._gpos += 1
        if self._gpos == self._gbatches:
            self._new_epoch()
            self._gperm.copy_(self._perm[:self._gperm.numel()])  # stream-ordered after the replay that used the old epoch
            self._gpos = 0

    def gather(self, idx, out=None):
        """Rows ``idx`` -> (rays (B,11), ts (B,), rgbs (B,3)) on the device; ``out`` = three preallocated tensors of exactly that
        size to gather into (e.g. the static inputs of a captured hipGraph -- no intermediate copy)."""
        from . import ops

        if out is not None and out[0].shape[0] != idx.numel():
            raise ValueError(f"gather target holds {out[0].shape[0]} rays but the batch has {idx.numel()}")
        return ops.gather_batch(self.rays, self.rgbs, self.ts, idx.contiguous(), out)

    def next_batch(self, out=None):
        """The next batch of the shuffled epoch (``gather`` of ``next_indices``); a short last batch (drop_last=False) that does
        not fit ``out`` is returned in fresh tensors."""
        idx = self.next_indices()
        if out is not None and out[0].shape[0] != idx.numel():
            out = None
        return self.gather(idx, out)


class DepthBank(RayBank):
    """Ray bank of the depth-supervision dataset (``SatelliteDataset_depth``, datasets/satellite_depth.py: items
    ``{"rays": (11,), "depths": (2,) = [target depth, weight], "ts": (1,)}``; main.py:103-109 gives it its own shuffled
    DataLoader).  Batches come out as (rays (B,11), ts (B,), depths (B,3)) with a zero third column -- the gather kernel
    moves 3-float targets -- which is what ``Trainer.step(..., depth=...)`` takes."""

    def __init__(self, rays, depths, ts, batch_size, **kw):
        if depths.dim() != 2 or depths.shape[1] != 2:
            raise ValueError("depths must be (N, 2) = [target depth, weight]")
        super().__init__(rays, torch.cat([depths.float(), torch.zeros_like(depths[:, :1], dtype=torch.float32)], 1), ts, batch_size, **kw)


def sun_direction(sun_elevation_deg, sun_azimuth_deg):
    """Unit sun vector of one image (``SatelliteDataset.get_sun_dirs``, datasets/satellite.py:232-244)."""
    import math

    el, az = math.radians(float(sun_elevation_deg)), math.radians(float(sun_azimuth_deg))
    return torch.tensor([math.sin(az) * math.cos(el), math.cos(az) * math.cos(el), math.sin(el)], dtype=torch.float32)


def rays_from_cache(cached_rays, center, scene_range, sun_elevation_deg, sun_azimuth_deg, device=None):
    """The (HW, 11) fp32 ray block of one image from the reference's ``<cache_dir>/<img_id>.data`` file (``torch.save`` of the
    (HW, 8) ECEF rays ``get_rays`` produced; datasets/satellite.py:185-196): scene normalisation of origin / near / far
    (``normalize_rays``, :218-227) and the per-image sun direction appended (:199-211).  ``cached_rays`` = a path or the
    loaded tensor; arithmetic in the cache's own dtype before the final cast, like the reference.  Host-side data plumbing
    for ``RayBank``: the RPC localisation that writes the cache is not rebuilt."""
    rays = torch.load(cached_rays) if isinstance(cached_rays, (str, bytes)) or hasattr(cached_rays, "__fspath__") else cached_rays
    if rays.dim() != 2 or rays.shape[1] != 8:
        raise ValueError(f"cached rays must be (HW, 8), got {tuple(rays.shape)}")
    rays = rays.clone()
    for c in range(3):
        rays[:, c] -= center[c]
        rays[:, c] /= scene_range
    rays[:, 6] /= scene_range
    rays[:, 7] /= scene_range
    sun = sun_direction(sun_elevation_deg, sun_azimuth_deg).to(rays.dtype).expand(rays.shape[0], 3)
    out = torch.hstack([rays, sun]).type(torch.float32)
    return out if device is None else out.to(device)


def rescale_rpc(rpc, alpha):
    """``sat_utils.rescale_rpc`` (sat_utils.py:44-57) on an "rpcm"-format dict: the camera of the image resized by ``alpha``."""
    out = dict(rpc)
    for k in ("row_scale", "col_scale", "row_offset", "col_offset"):
        out[k] = float(rpc[k]) * float(alpha)
    return out


def rays_from_rpc(rpc, height, width, min_alt, max_alt, center, scene_range, sun_elevation_deg, sun_azimuth_deg, device="cuda",
                  img_downscale=1.0, cache_path=None):
    """The (H*W, 11) ray block of one image straight from its RPC camera, on the GPU (``SatelliteDataset.load_data``,
    datasets/satellite.py:185-211, for an image without a ``.data`` cache): ``get_rays`` (:18-65) on the pixel grid of the
    down-scaled image, ``normalize_rays``, sun direction.  ``rpc`` = the JSON's "rpc" dict (rpcm format); ``height`` / ``width``
    = the FULL-resolution size stored in the JSON (the reference divides them by ``img_downscale`` itself, :191-192).
    ``cache_path``: also write the reference-compatible ``<cache_dir>/<img_id>.data`` file (torch.save of the (H*W, 8) rays)."""
    import os

    from . import ops

    h, w = int(height // img_downscale), int(width // img_downscale)
    rays, cache = ops.rpc_rays(rescale_rpc(rpc, 1.0 / img_downscale), w, h, min_alt, max_alt, center, scene_range, sun_elevation_deg,
                               sun_azimuth_deg, device, want_cache=cache_path is not None)
    if cache_path is not None:
        os.makedirs(os.path.dirname(os.path.abspath(cache_path)), exist_ok=True)
        torch.save(cache.cpu(), cache_path)
    return rays


def synthetic_rays(n_rays, seed=20240628, n_images=19, far_lo=0.5, far_hi=1.0):
    """Synthetic sat-nerf ray batch for benchmarks (no dataset ships offline): origins U[-1,1]^3, unit directions, near = 0
    (datasets/satellite.py:60), far U[far_lo, far_hi] (scene-normalised, :225-226), one sun direction per synthetic image id
    from random (azimuth, elevation in [30, 80] deg) as in ``get_sun_dirs`` (:239-241), ts ~ randint(n_images).
    Returns (rays (N,11) fp32 on the CPU, ts (N,) int64)."""
    import math

    g = torch.Generator().manual_seed(seed)
    o = torch.rand(n_rays, 3, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=1)
    far = far_lo + (far_hi - far_lo) * torch.rand(n_rays, 1, generator=g)
    az = torch.rand(n_images, generator=g) * 2 * math.pi
    el = math.radians(30) + torch.rand(n_images, generator=g) * math.radians(50)
    sun = torch.stack([torch.sin(az) * torch.cos(el), torch.cos(az) * torch.cos(el), torch.sin(el)], 1)
    ts = torch.randint(0, n_images, (n_rays,), generator=g)
    return torch.cat([o, d, torch.zeros(n_rays, 1), far, sun[ts]], 1).float(), ts


def default_args(**kw):
    """The ``args`` attributes the hot path reads, with the BASELINE configs[1] values (opt.py defaults except fc_units=256)."""
    from types import SimpleNamespace

    a = dict(model="sat-nerf", n_samples=64, n_importance=0, chunk=5120, noise_std=0.0, sc_lambda=0.0, ds_lambda=0.0, fc_layers=8,
             fc_units=256, t_embbeding_tau=4, t_embbeding_vocab=30)
    a.update(kw)
    return SimpleNamespace(**a)
