#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  The reference has no tests
or known-answer vectors for this path, so these vectors -- outputs of
``rendering.render_rays`` / ``models.satnerf.SatNeRF.forward`` / ``rendering.sample_pdf`` /
``eval_satnerf.batched_inference`` / ``metrics.*Loss`` on fixed inputs -- are what pins the oracle.

Only DATA is written (inputs, captured random draws, expected outputs).  Model weights are not
stored: they are procedural (``oracle.satnerf_oracle.procedural_satnerf_params``: an integer hash
mapped to the SIREN init ranges) and are loaded into the reference modules via ``load_state_dict``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py            # rewrite every fixture
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --check    # regenerate in memory, compare with the committed files bit for bit
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py --check --replay
                                                                            # ... feeding the STORED draws to the reference instead of drawing

Every case seeds torch's generator for itself (``CASES``: crc32 of its name), so a case regenerates the same arrays whatever ran
before it and ``--only`` reproduces exactly what a full run writes.  Archives are written with fixed zip timestamps: the same arrays
give the same file bytes.
"""
import argparse
import importlib.machinery
import io
import os
import sys
import types
import zipfile
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)  # reference first: it has top-level `models`, `datasets`, `rendering`


class _Stub(types.ModuleType):
    """Inert stand-in for an absent third-party import of eval_satnerf.py / metrics.py (never called)."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []
        self.__spec__ = importlib.machinery.ModuleSpec(name, None)

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Stub(f"{self.__name__}.{item}")

    def __call__(self, *a, **k):
        return _Stub(self.__name__ + "()")


for _m in ("kornia", "kornia.losses", "rasterio", "rpcm", "cv2", "torchvision", "torchvision.transforms", "numba",
           "osgeo", "fire", "pytorch_lightning", "plyflatten", "utm", "pyproj", "srtm4", "affine", "PIL", "PIL.Image",
           "rasterio.enums", "rasterio.warp", "plyflatten.utils", "osgeo.gdal"):
    sys.modules.setdefault(_m, _Stub(_m))

import rendering as ref_rendering  # noqa: E402
from models import load_model as ref_load_model  # noqa: E402
from models import satnerf as ref_satnerf  # noqa: E402

from oracle import satnerf_oracle as O  # noqa: E402


MODE = {"check": False, "replay": False, "failures": []}


def _stored(name):
    z = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


class Capture:
    """Record every draw of the three RNG entry points on the path (rendering.py:33,77; models/*.py randn).  With ``--replay`` the
    draws are not drawn but taken, in order, from the committed fixture ``fixture`` (keys ``keys``, default draw0, draw1, ...): the
    reference then runs on the STORED draws, and ``save`` compares what it returns with the stored outputs."""

    def __init__(self, fixture=None, keys=None):
        self.replay = None
        if MODE["replay"] and fixture is not None:
            st = _stored(fixture)
            if keys is None:
                keys = []
                while f"draw{len(keys)}" in st:
                    keys.append(f"draw{len(keys)}")
            self.replay = [torch.from_numpy(st[k]) for k in keys]

    def __enter__(self):
        self.draws = []
        self._orig = (torch.rand_like, torch.randn, torch.rand)

        def wrap(fn):
            def inner(*a, **k):
                if self.replay is not None:
                    assert self.replay, "the reference draws more often than the fixture stores draws"
                    want = fn(*a, **k)  # (shape / dtype check; the generator's state is irrelevant in this mode)
                    out = self.replay.pop(0)
                    assert out.shape == want.shape and out.dtype == want.dtype, (out.shape, want.shape)
                else:
                    out = fn(*a, **k)
                self.draws.append(out.clone())
                return out
            return inner

        torch.rand_like, torch.randn, torch.rand = (wrap(f) for f in self._orig)
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.randn, torch.rand = self._orig
        if exc[0] is None and self.replay is not None:
            assert not self.replay, f"{len(self.replay)} stored draws were never consumed"


def ref_models(args, seed_coarse=1, seed_fine=2, emb_seed=7):
    models = {}
    if args.model == "sat-nerf":
        mk = lambda s: O.procedural_satnerf_params(args.fc_units, args.t_embbeding_tau, seed=s)  # noqa: E731
    elif args.model == "s-nerf":
        mk = lambda s: O.procedural_snerf_params(args.fc_units, seed=s)  # noqa: E731
    else:
        mk = lambda s: O.procedural_nerf_params(args.fc_units, seed=s)  # noqa: E731
    m = ref_load_model(args)
    m.load_state_dict(mk(seed_coarse))
    models["coarse"] = m
    if args.n_importance > 0:
        f = ref_load_model(args)
        f.load_state_dict(mk(seed_fine))
        models["fine"] = f
    if args.model == "sat-nerf":
        emb = torch.nn.Embedding(args.t_embbeding_vocab, args.t_embbeding_tau)
        emb.load_state_dict({"weight": O.procedural_uniform((args.t_embbeding_vocab, args.t_embbeding_tau), 1.0, emb_seed)})
        models["t"] = emb
    return models


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    if MODE["check"]:
        st = _stored(name)
        bad = sorted(set(st) ^ set(out))
        for k in sorted(set(st) & set(out)):
            a, b = np.asarray(out[k]), st[k]
            if a.shape != b.shape or a.dtype != b.dtype or a.tobytes() != b.tobytes():
                bad.append(k)
        print(f"{name}.npz  {'OK: ' + str(len(out)) + ' arrays bit-equal' if not bad else 'MISMATCH: ' + ', '.join(bad)}"
              + ("  (stored draws replayed)" if MODE["replay"] else ""))
        if bad:
            MODE["failures"].append((name, bad))
        return
    # np.savez_compressed with fixed member timestamps: equal arrays -> equal file bytes
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED) as zf:
        for k, v in out.items():
            buf = io.BytesIO()
            np.lib.format.write_array(buf, np.asanyarray(v), allow_pickle=False)
            info = zipfile.ZipInfo(k + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            zf.writestr(info, buf.getvalue())
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def render_case(name, n_rays, ray_seed, **kw):
    args = O.default_args(**kw)
    classic = args.model == "nerf"
    rays, ts = O.synthetic_rays(n_rays, seed=ray_seed, classic=classic)
    models = ref_models(args)
    with torch.no_grad(), Capture(name) as cap:
        res = ref_rendering.render_rays(models, args, rays, ts)
    arrays = {"rays": rays, "cfg": np.array(repr(vars(args)))}
    if ts is not None:
        arrays["ts"] = ts
    for i, d in enumerate(cap.draws):
        arrays[f"draw{i}"] = d
    for k, v in res.items():
        arrays["out_" + k] = v.contiguous()
    save(name, **arrays)


def snerf_case():
    """s-nerf (models/snerf.py, rendering.py:85-96): render_rays with solar correction, SNerfLoss and a few of its gradients."""
    import metrics as ref_metrics  # (imports the stubs above)

    args = O.default_args(model="s-nerf", sc_lambda=0.05)
    rays, ts = O.synthetic_rays(48, seed=31)
    models = ref_models(args)
    with Capture("snerf_sc") as cap:
        res = ref_rendering.render_rays(models, args, rays, ts)
    target = torch.rand(48, 3, generator=torch.Generator().manual_seed(32))
    loss, _ = ref_metrics.SNerfLoss(lambda_sc=0.05)(res, target)
    loss.backward()
    grads = {"grad_" + k: v.grad for k, v in models["coarse"].named_parameters()
             if k in ("fc_net.0.weight", "fc_net.10.weight", "sigma_from_xyz.0.weight", "rgb_from_xyzdir.0.weight", "sun_v_net.4.weight",
                      "sun_v_net.6.weight", "sky_color.0.weight", "sky_color.2.bias")}
    arrays = {"rays": rays, "ts": ts, "target": target, "loss": loss.detach(), "cfg": np.array(repr(vars(args)))}
    arrays.update({f"draw{i}": d for i, d in enumerate(cap.draws)})
    arrays.update({"out_" + k: v.detach().contiguous() for k, v in res.items()})
    # ShadowNeRF.forward alone (B,8) + sigma_only
    g = torch.Generator().manual_seed(33)
    xyz = torch.rand(131, 3, generator=g) * 2 - 1
    sun = torch.nn.functional.normalize(torch.randn(131, 3, generator=g), dim=1)
    with torch.no_grad():
        arrays["fwd_xyz"], arrays["fwd_sun"] = xyz, sun
        arrays["fwd_out"] = models["coarse"](xyz, input_sun_dir=sun)
        arrays["fwd_sigma_only"] = models["coarse"](xyz, input_sun_dir=sun, sigma_only=True)
    arrays["state_keys"] = np.array(list(models["coarse"].state_dict().keys()))
    save("snerf_sc", **arrays, **grads)


def latlonalt_case():
    """datasets/satellite.py:246-275 (depth -> ECEF -> lat/lon/alt, fp64) on synthetic rays around a JAX-like scene centre."""
    from types import SimpleNamespace

    import sat_utils as ref_sat_utils
    from datasets.satellite import SatelliteDataset

    rays, _ = O.synthetic_rays(200, seed=41)
    g = torch.Generator().manual_seed(42)
    depth = 0.05 + 0.9 * torch.rand(200, generator=g)
    lat0, lon0, alt0 = 30.3165, -81.6633, 12.0
    cx, cy, cz = ref_sat_utils.latlon_to_ecef_custom(lat0, lon0, alt0)
    ds = SimpleNamespace(center=np.array([cx, cy, cz], dtype=np.float64), range=np.float64(431.7))
    lats, lons, alts = SatelliteDataset.get_latlonalt_from_nerf_prediction(ds, rays, depth)
    save("latlonalt", rays=rays, depth=depth, center=ds.center, range=ds.range, lats=lats, lons=lons, alts=alts)


def rpc_rays_case():
    """datasets/satellite.py:18-65 (``get_rays``), :218-227 (``normalize_rays``), :229-244 (``get_sun_dirs``) and sat_utils.py:44-57
    (``rescale_rpc``) RUN FROM THE REFERENCE on a synthetic RPC00B camera.  The one third-party call on the path,
    ``rpcm.RPCModel.localization`` (rpcm is absent offline), is supplied by a duck-typed rpc object whose ``localization`` is the
    oracle's Newton iteration: everything the reference itself computes -- ECEF conversion, near / far / direction, the float32 cast,
    the in-place scene normalisation, the sun direction, the RPC rescaling -- is pinned by this fixture; only rpcm's own iteration
    (checked by round trip through the published projection, tests/test_rpc.py) stays unpinned."""
    from types import SimpleNamespace

    import sat_utils as ref_sat_utils
    from datasets.satellite import SatelliteDataset
    from datasets.satellite import get_rays as ref_get_rays

    from oracle import rpc_oracle as R

    class DuckRPC:  # the attributes sat_utils.rescale_rpc touches + the method get_rays calls
        def __init__(self, d):
            self.d = dict(d)
            for k in ("row_scale", "col_scale", "row_offset", "col_offset"):
                setattr(self, k, float(d[k]))

        def _dict(self):
            return dict(self.d, row_scale=self.row_scale, col_scale=self.col_scale, row_offset=self.row_offset, col_offset=self.col_offset)

        def localization(self, cols, rows, alts):
            return R.localization(self._dict(), cols, rows, alts)

    arrays = {}
    for tag, (seed, h, w, down, min_alt, max_alt, el, az) in {"a": (0, 48, 64, 1.0, -25.0, 60.0, 52.0, 141.0),
                                                              "b": (4, 90, 70, 2.0, -31.5, 77.25, 38.5, 203.25)}.items():
        rpc_d = R.synthetic_rpc(seed, height=h, width=w)
        rpc = ref_sat_utils.rescale_rpc(DuckRPC(rpc_d), 1.0 / down)                      # sat_utils.py:44-57
        hh, ww = int(h // down), int(w // down)
        cols, rows = np.meshgrid(np.arange(ww), np.arange(hh))                            # datasets/satellite.py:193
        rays8 = ref_get_rays(cols.flatten(), rows.flatten(), rpc, min_alt, max_alt)       # :18-65
        cx, cy, cz = ref_sat_utils.latlon_to_ecef_custom(rpc_d["lat_offset"], rpc_d["lon_offset"], 10.0)
        ds = SimpleNamespace(center=torch.tensor([cx, cy, cz], dtype=torch.float32), range=torch.tensor(310.0 + 7 * seed))  # :160-163 loads them as float tensors
        rays_n = SatelliteDataset.normalize_rays(ds, rays8.clone())                       # :218-227
        sun = SatelliteDataset.get_sun_dirs(ds, el, az, rays_n.shape[0])                  # :229-244
        rays11 = torch.hstack([rays_n, sun]).type(torch.FloatTensor)                      # :213-214
        arrays.update({f"{tag}_seed": np.int64(seed), f"{tag}_hw": np.array([h, w]), f"{tag}_down": np.float64(down),
                       f"{tag}_alts": np.array([min_alt, max_alt]), f"{tag}_sun": np.array([el, az]),
                       f"{tag}_center": ds.center.numpy().astype(np.float64), f"{tag}_range": np.float64(ds.range.item()),
                       f"{tag}_rescaled": np.array([rpc.row_scale, rpc.col_scale, rpc.row_offset, rpc.col_offset]),
                       f"{tag}_rays8": rays8, f"{tag}_rays11": rays11})
    save("rpc_rays", **arrays)


def mlp_forward_case():
    """SatNeRF.forward alone on a ragged batch of points"""
    args = O.default_args()
    model = ref_models(args)["coarse"]
    g = torch.Generator().manual_seed(21)
    xyz = torch.rand(257, 3, generator=g) * 2 - 1
    sun = torch.randn(257, 3, generator=g)
    sun = sun / sun.norm(dim=1, keepdim=True)
    t = torch.rand(257, 4, generator=g) * 2 - 1
    with torch.no_grad():
        out = model(xyz, input_sun_dir=sun, input_t=t)
        sig = model(xyz, input_sun_dir=sun, input_t=t, sigma_only=True)
    save("mlp_forward", xyz=xyz, sun=sun, t=t, out=out, sigma_only=sig)


def sample_pdf_case():
    """sample_pdf alone, random and deterministic"""
    g = torch.Generator().manual_seed(22)
    bins = torch.sort(torch.rand(33, 63, generator=g), -1)[0]
    w = torch.rand(33, 62, generator=g) ** 4
    w[3] = 0.0  # an all-zero row: every bin hits the eps floor
    w[4, 10:40] = 0.0  # flat stretches inside the cdf (denom<eps branch)
    with Capture("sample_pdf", ["u"]) as cap:
        z_rand = ref_rendering.sample_pdf(bins, w, 48, det=False)
    z_det = ref_rendering.sample_pdf(bins, w, 48, det=True)
    save("sample_pdf", bins=bins, weights=w, u=cap.draws[0], z_rand=z_rand, z_det=z_det)


def composite_extreme_case():
    """compositing alone with extreme sigmas, through the reference's inference() around a canned model"""

    class Canned(torch.nn.Module):
        number_of_outputs = 9

        def __init__(self, table):
            super().__init__()
            self.table, self.pos = table, 0

        def forward(self, x, input_dir=None, input_sun_dir=None, input_t=None):
            out = self.table[self.pos : self.pos + x.shape[0]]
            self.pos += x.shape[0]
            return out

    g = torch.Generator().manual_seed(23)
    n, s = 29, 64
    raw = torch.rand(n * s, 9, generator=g)
    sig = torch.exp(torch.randn(n, s, generator=g) * 4)  # 1e-7 .. 1e7
    sig[0] = 0.0
    sig[1] = 1e-12
    sig[2] = 1e9
    sig[3, :32] = -5.0  # relu clips
    raw[:, 3] = sig.reshape(-1)
    z = torch.sort(torch.rand(n, s, generator=g), -1)[0]
    z[5, 10] = z[5, 11]  # a zero-length interval
    a2 = O.default_args(noise_std=0.7, chunk=1000)
    with torch.no_grad(), Capture("composite_extreme", ["noise"]) as cap:
        res = ref_satnerf.inference(Canned(raw), a2, torch.zeros(n, s, 3), z, sun_d=torch.zeros(n, 3), rays_t=torch.zeros(n, 4))
    save("composite_extreme", raw=raw.view(n, s, 9), z=z, noise=cap.draws[0], noise_std=np.float32(0.7),
         **{"out_" + k: v.contiguous() for k, v in res.items()})


def backward_case():
    """backward: grads of sum(rgb)+sum(depth)+sum(beta*w) wrt representative params + the embedding"""
    args = O.default_args()
    rays, ts = O.synthetic_rays(64, seed=24)
    models = ref_models(args)
    with Capture("backward") as cap:
        res = ref_rendering.render_rays(models, args, rays, ts)
    loss = res["rgb_coarse"].sum() + res["depth_coarse"].sum() + (res["weights_coarse"].unsqueeze(-1) * res["beta_coarse"]).sum()
    loss.backward()
    grads = {"grad_" + k: v.grad for k, v in models["coarse"].named_parameters()
             if k in ("fc_net.0.weight", "fc_net.0.bias", "fc_net.8.weight", "fc_net.14.bias", "sigma_from_xyz.0.weight",
                      "feats_from_xyz.weight", "rgb_from_xyzdir.2.weight", "sun_v_net.0.weight", "sun_v_net.6.bias",
                      "sky_color.0.weight", "sky_color.2.bias", "beta_from_xyz.0.weight", "beta_from_xyz.2.weight")}
    save("backward", rays=rays, ts=ts, loss=loss.detach(), grad_embedding=models["t"].weight.grad,
         **{f"draw{i}": d for i, d in enumerate(cap.draws)}, **grads)


def batched_losses_case():
    """batched_inference with a ragged last chunk + the three losses and a few loss gradients"""
    import eval_satnerf as ref_eval  # noqa: E402  (imports the stubs above)
    import metrics as ref_metrics  # noqa: E402

    args = O.default_args(chunk=100, sc_lambda=0.05)
    rays, ts = O.synthetic_rays(250, seed=25)
    models = ref_models(args)
    with Capture("batched_losses") as cap:
        res = ref_eval.batched_inference(models, rays, ts, args)
    arrays = {"rays": rays, "ts": ts}
    arrays.update({f"draw{i}": d for i, d in enumerate(cap.draws)})
    arrays.update({"out_" + k: v for k, v in res.items() if k in ("rgb_coarse", "depth_coarse", "weights_coarse", "sun_sc_coarse")})
    g = torch.Generator().manual_seed(26)
    target = torch.rand(250, 3, generator=g)
    dtarget, dweights = torch.rand(250, generator=g), torch.rand(250, generator=g)
    # the train-time twin of batched_inference (main.py:60-75) = the same loop with grad

    class Replay:
        def __init__(self, draws):
            self.d, self.i = list(draws), 0

        def __enter__(self):
            self._orig = (torch.rand_like, torch.randn, torch.rand)

            def nxt(*a, **k):
                out = self.d[self.i]
                self.i += 1
                return out.clone()

            torch.rand_like = torch.randn = torch.rand = nxt
            return self

        def __exit__(self, *exc):
            torch.rand_like, torch.randn, torch.rand = self._orig

    with Replay(cap.draws):
        outs = [ref_rendering.render_rays(models, args, rays[i:i + 100], ts[i:i + 100]) for i in range(0, 250, 100)]
    resg = {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
    l_sat, _ = ref_metrics.SatNerfLoss(lambda_sc=0.05)(resg, target)
    l_s, _ = ref_metrics.SNerfLoss(lambda_sc=0.05)(resg, target)
    l_d, _ = ref_metrics.DepthLoss(lambda_ds=1000.0)(resg, dtarget, dweights)
    (l_sat + l_d).backward()
    arrays.update(target=target, dtarget=dtarget, dweights=dweights, loss_satnerf=l_sat.detach(), loss_snerf=l_s.detach(),
                  loss_depth=l_d.detach(),
                  grad_fc_net_6_weight=models["coarse"].fc_net[6].weight.grad,
                  grad_beta_2_weight=models["coarse"].beta_from_xyz[2].weight.grad,
                  grad_sun_v_0_bias=models["coarse"].sun_v_net[0].bias.grad,
                  grad_embedding=models["t"].weight.grad)
    save("batched_losses", **arrays)


# every fixture and the function that makes it; a case seeds torch's global generator with crc32(fixture name) before it runs, so its
# draws do not depend on which cases ran before it
CASES = {
    "latlonalt": latlonalt_case,
    "rpc_rays": rpc_rays_case,
    "snerf_sc": snerf_case,
    # full render_rays variants
    "satnerf_coarse": lambda: render_case("satnerf_coarse", 96, 11),
    "satnerf_sc": lambda: render_case("satnerf_sc", 40, 12, sc_lambda=0.1),
    "satnerf_fine": lambda: render_case("satnerf_fine", 40, 13, n_importance=64),
    "satnerf_noise": lambda: render_case("satnerf_noise", 40, 14, noise_std=0.5),
    "satnerf_s128": lambda: render_case("satnerf_s128", 24, 15, n_samples=128),
    "satnerf_feat512": lambda: render_case("satnerf_feat512", 24, 16, fc_units=512, t_embbeding_tau=16),
    "satnerf_s50_ragged": lambda: render_case("satnerf_s50_ragged", 37, 17, n_samples=50, chunk=999),
    "nerf_coarse_fine": lambda: render_case("nerf_coarse_fine", 32, 18, model="nerf", n_importance=32),
    "mlp_forward": mlp_forward_case,
    "sample_pdf": sample_pdf_case,
    "composite_extreme": composite_extreme_case,
    "backward": backward_case,
    "batched_losses": batched_losses_case,
}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--only", default=None, help="comma-separated fixture names (default: all): " + " ".join(CASES))
    ap.add_argument("--check", action="store_true", help="write nothing: regenerate in memory and compare with the committed fixtures, bit for bit")
    ap.add_argument("--replay", action="store_true", help="with --check: feed the reference the fixtures' STORED draws instead of drawing")
    a = ap.parse_args()
    assert a.check or not a.replay, "--replay belongs to --check"
    MODE["check"], MODE["replay"] = a.check, a.replay
    torch.set_num_threads(8)
    aliases = {"snerf": "snerf_sc", "rpc": "rpc_rays"}
    names = list(CASES) if a.only is None else [aliases.get(n, n) for n in a.only.split(",")]
    for name in names:
        torch.manual_seed(zlib.crc32(name.encode()))
        CASES[name]()
    extra = sorted(set(f[:-4] for f in os.listdir(HERE) if f.endswith(".npz")) - set(CASES))
    assert not extra, f"fixtures without a recipe: {extra}"
    if MODE["failures"]:
        print("FAILED:", MODE["failures"])
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
