"""Convergence / depth-MAE check of the throughput arithmetic (SURVEY.md 8(c) last row).

north_star asks for "DSM MAE on JAX_068 within 2 cm of the reference".  The dataset is absent offline, so the stand-in is a
synthetic scene with KNOWN geometry -- a height field seen by 19 tilted near-nadir "images", colour = albedo(x, y) x sun shading,
sparse depth supervision as in BASELINE configs[3] (main.py:134-141) -- trained from the same seeded init, per seed:
  (a) the kernel-direct HIP Trainer in the throughput modes (bf16 MFMA + 8-bit saved state; + 16-bit state; f16 operands),
  (b) the oracle in fp32 with torch.optim.Adam (the reference's arithmetic; run on the GPU through torch -- same code as on the CPU),
  (c) the oracle again with OTHER stratified draws: the noise floor of "two trainings of this scene",
(a) and (b) on IDENTICAL stratified draws (generated once on the CPU, fed to both), so they differ by their arithmetic only; then
rendered on a fixed ray set with identical draws.  Reported, in metres at a 175 m scene range (depths are
normalised by the scene range, datasets/satellite.py:225-226):
  mae_truth_*      altitude-like MAE of the rendered depth against the true surface, per training arithmetic
  delta_mae_m      |mae_truth_hip - mae_truth_ref|                       <- the "within 2 cm of the reference" quantity
  mae_between_m    mean |depth_hip_trained - depth_ref_trained|          (pointwise; training dynamics amplify rounding)
  mae_infer_bf16_m mean |bf16 inference - fp32 inference| of the fp32-TRAINED weights (and the same for f16 and bf16x3)
The oracle is test infrastructure: this script (and tests/test_hip_convergence.py) are the only users here.

    python tools/convergence.py [--steps 1500] [--batch 256] [--seeds 0,1,2]        # prints one JSON line
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SCENE_RANGE_M = 175.0
N_IMAGES = 19


def height(x, y):
    """Normalised surface height: rolling terrain + one flat-roofed block."""
    return 0.12 * torch.sin(2.5 * x) * torch.cos(2.1 * y) + 0.10 * ((x > 0.15) & (x < 0.55) & (y > -0.4) & (y < 0.1)).float()


def albedo(x, y):
    return torch.stack([0.5 + 0.3 * torch.sin(3.0 * x + 1.0), 0.45 + 0.3 * torch.cos(2.0 * y - 0.5), 0.4 + 0.25 * torch.sin(2.0 * (x + y))], -1)


def make_scene(n_rays, seed):
    """(rays (N,11), ts (N,), rgbs (N,3), true depth (N,)) in normalised units: origins on the plane z = 0.6 above the scene,
    view directions tilted up to ~17 degrees per image, near = 0, far = 1.2, sun per image as in SURVEY.md 8(d)."""
    g = torch.Generator().manual_seed(seed)
    ts = torch.randint(0, N_IMAGES, (n_rays,), generator=g)
    gi = torch.Generator().manual_seed(20240628)
    tilt = (torch.rand(N_IMAGES, 2, generator=gi) - 0.5) * 0.6
    az = torch.rand(N_IMAGES, generator=gi) * 2 * math.pi
    el = math.radians(30) + torch.rand(N_IMAGES, generator=gi) * math.radians(50)
    sun_tab = torch.stack([torch.sin(az) * torch.cos(el), torch.cos(az) * torch.cos(el), torch.sin(el)], 1)
    xy = torch.rand(n_rays, 2, generator=g) * 1.6 - 0.8
    o = torch.cat([xy, torch.full((n_rays, 1), 0.6)], 1)
    d = torch.cat([tilt[ts], -torch.ones(n_rays, 1)], 1)
    d = d / d.norm(dim=1, keepdim=True)
    t = torch.full((n_rays,), 0.6)  # ray / surface intersection by fixed-point iteration (the surface is a graph, tilts are small)
    for _ in range(30):
        p = o + d * t[:, None]
        t = (0.6 - height(p[:, 0], p[:, 1])) / (-d[:, 2])
    p = o + d * t[:, None]
    eps = 1e-3  # surface normal by central differences -> Lambert shading with an ambient floor
    hx = (height(p[:, 0] + eps, p[:, 1]) - height(p[:, 0] - eps, p[:, 1])) / (2 * eps)
    hy = (height(p[:, 0], p[:, 1] + eps) - height(p[:, 0], p[:, 1] - eps)) / (2 * eps)
    nrm = torch.stack([-hx, -hy, torch.ones_like(hx)], 1)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    shade = 0.55 + 0.45 * torch.clamp((nrm * sun_tab[ts]).sum(1), min=0.0)
    rgbs = (albedo(p[:, 0], p[:, 1]) * shade[:, None]).clamp(0, 1)
    rays = torch.cat([o, d, torch.zeros(n_rays, 1), torch.full((n_rays, 1), 1.2), sun_tab[ts]], 1).float()
    return rays, ts, rgbs.float(), t.float()


def _draws(steps, batch, seed):
    """the stratified jitter of every step (colour batch, depth batch), drawn once on the CPU: every training of a seed uses them"""
    g = torch.Generator().manual_seed(seed + 10)
    return [(torch.rand(batch, 64, generator=g), torch.rand(batch, 64, generator=g)) for _ in range(steps)]


def train_ref(init, emb_init, data, draws, steps, batch, ds_lambda, dev, verbose=False):
    """the reference arithmetic: the oracle (fp32 torch restatement of rendering.py / metrics.py) + torch.optim.Adam, on `dev`"""
    from oracle import satnerf_oracle as O

    rays, ts, rgbs, d_rays, d_ts, depths = (t.to(dev) for t in data)
    args_ref = O.default_args(ds_lambda=ds_lambda)
    po = {k: v.clone().to(dev).requires_grad_(True) for k, v in init.items()}
    eo = emb_init.clone().to(dev).requires_grad_(True)
    opt = torch.optim.Adam(list(po.values()) + [eo], lr=5e-4)
    zeros = torch.zeros(batch, 64, device=dev)
    for k in range(steps):
        sl = slice(k * batch, (k + 1) * batch)
        mo = {"coarse": po, "t": eo}
        u_c, u_d = draws[k][0].to(dev), draws[k][1].to(dev)
        l_c = O.satnerf_loss(O.render_rays(mo, args_ref, rays[sl], ts[sl], O.ReplayRng([u_c, zeros])), rgbs[sl])
        l_d = O.depth_loss(O.render_rays(mo, args_ref, d_rays[sl], d_ts[sl], O.ReplayRng([u_d, zeros])), depths[sl, 0], depths[sl, 1], ds_lambda)
        opt.zero_grad()
        (l_c + l_d).backward()
        opt.step()
        if verbose and k % 250 == 0:
            print(f"ref step {k}: loss {(l_c + l_d).item():.4f}", file=sys.stderr)
    return {k: v.detach().cpu() for k, v in po.items()}, eo.detach().cpu(), (l_c + l_d).item()


def train_hip(init, emb_init, data, draws, steps, batch, ds_lambda, dev, mode, bwd_fmt=None, verbose=False):
    """the kernel-direct HIP Trainer (eager: every step its own batch), fed the SAME jitter draws"""
    from oracle import satnerf_oracle as O
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    rays, ts, rgbs, d_rays, d_ts, depths = (t.to(dev) for t in data)
    kw = {} if bwd_fmt is None else {"bwd_fmt": bwd_fmt}
    args_hip = O.default_args(mlp_mode=mode, ds_lambda=ds_lambda, **kw)
    model = load_model(args_hip)
    model.load_state_dict(init)
    emb = torch.nn.Embedding(30, 4)
    emb.load_state_dict({"weight": emb_init})
    models = {"coarse": model.to(dev), "t": emb.to(dev)}
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = Trainer(models, args_hip, use_graph=False)
    assert tr.direct
    queue = []
    tr.jitter = lambda n, s, device: queue.pop(0).to(device)
    for k in range(steps):
        sl = slice(k * batch, (k + 1) * batch)
        queue[:] = [draws[k][0], draws[k][1]]
        loss = tr.step(rays[sl], ts[sl], rgbs[sl], depth=(d_rays[sl], d_ts[sl], depths[sl]), _inputs_in_place=True)  # (indices are valid by construction)
        if verbose and k % 250 == 0:
            print(f"hip[{mode}] step {k}: loss {loss.item():.4f}", file=sys.stderr)
    torch.cuda.synchronize()
    return models, loss.item()


def run(steps=1500, batch=256, n_eval=2048, seed=0, verbose=False, modes=(("bf16", None),), floor=True):
    """One seed: fp32 reference, [fp32 reference with OTHER jitter draws = the noise floor], and one HIP training per entry of `modes`
    ((mlp_mode, bwd_fmt)); all from the same init, all HIP runs and the reference on the same draws.  Returns a dict of metres."""
    from oracle import satnerf_oracle as O
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model

    dev = torch.device("cuda:0")
    ds_lambda = 1000.0  # run_all.sh:80
    n_bank = steps * batch
    rays, ts, rgbs, depth = make_scene(n_bank, seed=seed + 1)
    d_rays, d_ts, _, d_depth = make_scene(n_bank, seed=seed + 2)  # the depth-supervision batch: its own rays (satellite_depth.py)
    depths = torch.stack([d_depth, torch.ones_like(d_depth)], 1)
    data = (rays, ts, rgbs, d_rays, d_ts, depths)
    ev_rays, ev_ts, _, ev_depth = make_scene(n_eval, seed=seed + 3)
    gen = torch.Generator().manual_seed(seed + 4)
    ev_u, ev_noise = torch.rand(n_eval, 64, generator=gen), torch.zeros(n_eval, 64)
    torch.manual_seed(seed)
    m0 = load_model(O.default_args())
    init = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    emb_init = torch.nn.Embedding(30, 4).weight.detach().clone()
    draws = _draws(steps, batch, seed)
    m = SCENE_RANGE_M
    mae = lambda a, b: float((a - b).abs().mean()) * m  # noqa: E731
    args_ref = O.default_args(ds_lambda=ds_lambda)

    def ref_depth(po, eo):
        with torch.no_grad():
            return O.render_rays({"coarse": po, "t": eo}, args_ref, ev_rays, ev_ts, O.ReplayRng([ev_u, ev_noise]))["depth_coarse"]

    def hip_depth(mods, mode):
        a = O.default_args(mlp_mode=mode)
        with torch.no_grad(), rendering.replay_rng([ev_u.to(dev), ev_noise.to(dev)]):
            return rendering.render_rays(mods, a, ev_rays.to(dev), ev_ts.to(dev))["depth_coarse"].cpu()

    t0 = time.time()
    po, eo, ref_loss = train_ref(init, emb_init, data, draws, steps, batch, ds_lambda, dev, verbose)
    t_ref = time.time() - t0
    d_ref = ref_depth(po, eo)
    out = {"seed": seed, "steps": steps, "batch": batch, "n_eval": n_eval, "scene_range_m": m, "mae_truth_ref_m": mae(d_ref, ev_depth),
           "final_loss_ref": ref_loss, "train_s_ref_gpu_fp32": t_ref, "hip": {}}
    if floor:  # the same arithmetic, other jitter draws: what "two trainings of this scene" differ by when nothing but the sampling noise differs
        po2, eo2, _ = train_ref(init, emb_init, data, _draws(steps, batch, seed + 1000), steps, batch, ds_lambda, dev, False)
        d_ref2 = ref_depth(po2, eo2)
        out["floor_mae_truth_m"] = mae(d_ref2, ev_depth)
        out["floor_delta_mae_m"] = abs(out["floor_mae_truth_m"] - out["mae_truth_ref_m"])
        out["floor_mae_between_m"] = mae(d_ref2, d_ref)
    for mode, fmt in modes:
        t0 = time.time()
        models, hip_loss = train_hip(init, emb_init, data, draws, steps, batch, ds_lambda, dev, mode, fmt, verbose)
        d_hip = hip_depth(models, mode)
        key = mode + (f"_state{fmt}" if fmt else "")
        out["hip"][key] = {"mae_truth_m": mae(d_hip, ev_depth), "delta_mae_m": abs(mae(d_hip, ev_depth) - out["mae_truth_ref_m"]),
                           "mae_between_m": mae(d_hip, d_ref), "final_loss": hip_loss, "train_s": time.time() - t0}
    # the fp32-trained weights through the HIP inference path in the three arithmetic modes
    m2 = load_model(O.default_args())
    m2.load_state_dict(po)
    e2 = torch.nn.Embedding(30, 4)
    e2.load_state_dict({"weight": eo})
    mods2 = {"coarse": m2.to(dev), "t": e2.to(dev)}
    for mode in ("bf16", "f16", "bf16x3"):
        out[f"mae_infer_{mode}_m"] = mae(hip_depth(mods2, mode), d_ref)
    return out


def run_seeds(seeds=(0, 1, 2), **kw):
    rows = [run(seed=s, **kw) for s in seeds]
    keys = rows[0]["hip"].keys()
    mean = lambda xs: sum(xs) / len(xs)  # noqa: E731
    summary = {"seeds": list(seeds), "steps": rows[0]["steps"], "batch": rows[0]["batch"],
               "mean_mae_truth_ref_m": mean([r["mae_truth_ref_m"] for r in rows]),
               "mean_delta_mae_m": {k: mean([r["hip"][k]["delta_mae_m"] for r in rows]) for k in keys},
               "delta_of_seed_mean_mae_m": {k: abs(mean([r["hip"][k]["mae_truth_m"] for r in rows]) - mean([r["mae_truth_ref_m"] for r in rows])) for k in keys}}
    if "floor_delta_mae_m" in rows[0]:
        summary["mean_floor_delta_mae_m"] = mean([r["floor_delta_mae_m"] for r in rows])
        summary["mean_floor_mae_between_m"] = mean([r["floor_mae_between_m"] for r in rows])
    return {"summary": summary, "rows": rows}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--eval", type=int, default=2048)
    ap.add_argument("--seeds", default="0,1,2")
    a = ap.parse_args()
    res = run_seeds(tuple(int(x) for x in a.seeds.split(",")), steps=a.steps, batch=a.batch, n_eval=a.eval, verbose=True,
                    modes=(("bf16", None), ("bf16", 16), ("f16", None), ("bf16x3", None)))
    print(json.dumps(res))
