"""Long-horizon convergence curve of the throughput arithmetic against the fp32 reference arithmetic (VERDICT r03 task 7).

tools/convergence.py compares the two after 1,500 steps, where both models are still at ~1.2 m altitude MAE.  This script trains ONE
seed for >= 20,000 steps on the same synthetic height-field scene (depth supervision as BASELINE configs[3]) and evaluates, at a list of
checkpoints, the altitude-like MAE against the true surface of
  ref   the fp32 oracle + torch.optim.Adam (the reference's arithmetic, on the GPU through torch),
  hip   the kernel-direct HIP Trainer in the benchmarked arithmetic (bf16 MFMA, 8-bit saved state, fp16-operand weight gradients),
  floor the fp32 oracle again with OTHER stratified jitter (what two trainings differ by when only the sampling noise differs),
ref and hip on IDENTICAL jitter (drawn per step from a seeded CPU generator), cycling over a bank of rays (epochs, as a real scene).
Metres at a 175 m scene range.  The oracle is test infrastructure; this script is a measurement tool, not a product path.

    python tools/convergence_long.py [--steps 20000] [--batch 256] [--bank-steps 2000] [--seed 0] > profiles/r04_convergence_long.json
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.convergence import SCENE_RANGE_M, make_scene  # noqa: E402


def jitter(seed, k, batch):
    g = torch.Generator().manual_seed(seed * 1000003 + k)
    return torch.rand(batch, 64, generator=g), torch.rand(batch, 64, generator=g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--bank-steps", type=int, default=2000, help="batches in the ray bank (one epoch)")
    ap.add_argument("--eval", type=int, default=2048)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--checkpoints", default="500,1500,3000,5000,10000,15000,20000")
    ap.add_argument("--no-floor", action="store_true")
    a = ap.parse_args()
    from oracle import satnerf_oracle as O
    from satnerf_amd import rendering
    from satnerf_amd.models import load_model
    from satnerf_amd.train import Trainer

    dev = torch.device("cuda:0")
    ds_lambda, batch, seed = 1000.0, a.batch, a.seed
    cps = sorted({int(c) for c in a.checkpoints.split(",") if int(c) <= a.steps} | {a.steps})
    n_bank = a.bank_steps * batch
    rays, ts, rgbs, _ = make_scene(n_bank, seed=seed + 1)
    d_rays, d_ts, _, d_depth = make_scene(n_bank, seed=seed + 2)
    depths = torch.stack([d_depth, torch.ones_like(d_depth)], 1)
    data = tuple(t.to(dev) for t in (rays, ts, rgbs, d_rays, d_ts, depths))
    ev_rays, ev_ts, _, ev_depth = make_scene(a.eval, seed=seed + 3)
    gen = torch.Generator().manual_seed(seed + 4)
    ev_u, ev_noise = torch.rand(a.eval, 64, generator=gen), torch.zeros(a.eval, 64)
    torch.manual_seed(seed)
    init = {k: v.detach().clone() for k, v in load_model(O.default_args()).state_dict().items()}
    emb_init = torch.nn.Embedding(30, 4).weight.detach().clone()
    args_ref = O.default_args(ds_lambda=ds_lambda)
    mae = lambda x, y: float((x - y).abs().mean()) * SCENE_RANGE_M  # noqa: E731
    ev = tuple(t.to(dev) for t in (ev_rays, ev_ts, ev_u, ev_noise))

    def ref_depth(po, eo):
        with torch.no_grad():
            return O.render_rays({"coarse": po, "t": eo}, args_ref, ev[0], ev[1], O.ReplayRng([ev[2], ev[3]]))["depth_coarse"].cpu()

    def train_ref(jseed, tag):
        rays, ts, rgbs, d_rays, d_ts, depths = data
        po = {k: v.clone().to(dev).requires_grad_(True) for k, v in init.items()}
        eo = emb_init.clone().to(dev).requires_grad_(True)
        opt = torch.optim.Adam(list(po.values()) + [eo], lr=5e-4)
        zeros = torch.zeros(batch, 64, device=dev)
        out, t0 = {}, time.time()
        for k in range(a.steps):
            b = k % a.bank_steps
            sl = slice(b * batch, (b + 1) * batch)
            u_c, u_d = (u.to(dev) for u in jitter(jseed, k, batch))
            mo = {"coarse": po, "t": eo}
            l_c = O.satnerf_loss(O.render_rays(mo, args_ref, rays[sl], ts[sl], O.ReplayRng([u_c, zeros])), rgbs[sl])
            l_d = O.depth_loss(O.render_rays(mo, args_ref, d_rays[sl], d_ts[sl], O.ReplayRng([u_d, zeros])), depths[sl, 0], depths[sl, 1], ds_lambda)
            opt.zero_grad()
            (l_c + l_d).backward()
            opt.step()
            if k + 1 in cps:
                out[k + 1] = ref_depth({n: v.detach() for n, v in po.items()}, eo.detach())
                print(f"{tag} step {k + 1}: loss {(l_c + l_d).item():.4f}  MAE vs truth {mae(out[k + 1], ev_depth):.3f} m  ({time.time() - t0:.0f} s)", file=sys.stderr)
        return out

    def train_hip(mode, fmt):
        rays, ts, rgbs, d_rays, d_ts, depths = data
        kw = {} if fmt is None else {"bwd_fmt": fmt}
        args_hip = O.default_args(mlp_mode=mode, ds_lambda=ds_lambda, **kw)
        model = load_model(args_hip)
        model.load_state_dict(init)
        emb = torch.nn.Embedding(30, 4)
        emb.load_state_dict({"weight": emb_init})
        models = {"coarse": model.to(dev), "t": emb.to(dev)}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tr = Trainer(models, args_hip, use_graph=False)
        queue = []
        tr.jitter = lambda n, s, device: queue.pop(0).to(device)
        out, t0 = {}, time.time()
        for k in range(a.steps):
            b = k % a.bank_steps
            sl = slice(b * batch, (b + 1) * batch)
            queue[:] = list(jitter(seed, k, batch))
            loss = tr.step(rays[sl], ts[sl], rgbs[sl], depth=(d_rays[sl], d_ts[sl], depths[sl]), _inputs_in_place=True)
            if k + 1 in cps:
                ah = O.default_args(mlp_mode=mode)
                with torch.no_grad(), rendering.replay_rng([ev[2], ev[3]]):
                    out[k + 1] = rendering.render_rays(models, ah, ev[0], ev[1])["depth_coarse"].cpu()
                print(f"hip[{mode}] step {k + 1}: loss {loss.item():.4f}  MAE vs truth {mae(out[k + 1], ev_depth):.3f} m  ({time.time() - t0:.0f} s)", file=sys.stderr)
        return out

    hip = train_hip("bf16", None)
    ref = train_ref(seed, "ref")
    floor = None if a.no_floor else train_ref(seed + 1000, "floor")
    rows = []
    for c in cps:
        row = {"step": c, "mae_truth_ref_m": mae(ref[c], ev_depth), "mae_truth_hip_m": mae(hip[c], ev_depth),
               "delta_mae_m": abs(mae(hip[c], ev_depth) - mae(ref[c], ev_depth)), "mae_between_m": mae(hip[c], ref[c])}
        if floor is not None:
            row.update(mae_truth_floor_m=mae(floor[c], ev_depth), floor_delta_mae_m=abs(mae(floor[c], ev_depth) - mae(ref[c], ev_depth)),
                       floor_mae_between_m=mae(floor[c], ref[c]))
        rows.append(row)
    print(json.dumps({"seed": seed, "steps": a.steps, "batch": batch, "bank_steps": a.bank_steps, "scene_range_m": SCENE_RANGE_M,
                      "hip": "mlp_mode=bf16, 8-bit saved state, kernel-direct Trainer (eager)", "curve": rows}))


if __name__ == "__main__":
    main()
