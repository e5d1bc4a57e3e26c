"""Shared helpers for the parity tests: golden-fixture loading and error metrics."""
import ast
import os

import numpy as np
import torch

from oracle import satnerf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    out = {k: (torch.from_numpy(v) if v.dtype.kind in "fi" and v.ndim > 0 else v) for k, v in d.items()}
    return out


def golden_draws(g):
    i, draws = 0, []
    while f"draw{i}" in g:
        draws.append(g[f"draw{i}"])
        i += 1
    return draws


def golden_cfg(g):
    return O.default_args(**ast.literal_eval(str(g["cfg"])))


def maxnorm_rel(a, b):
    """max|a-b| / max|b| -- the metric of SURVEY.md section 8(c) (element-wise rel is ill-conditioned on ~1e-5 weights)."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    denom = b.abs().max().item()
    return ((a - b).abs().max().item() / denom) if denom > 0 else (a - b).abs().max().item()


def make_models(args, device="cpu", seed_coarse=1, seed_fine=2, emb_seed=7):
    """Procedural parameter dicts identical to the ones make_golden.py loaded into the reference modules."""
    if args.model == "sat-nerf":
        mk = lambda s: O.procedural_satnerf_params(args.fc_units, args.t_embbeding_tau, seed=s)  # noqa: E731
    elif args.model == "s-nerf":
        mk = lambda s: O.procedural_snerf_params(args.fc_units, seed=s)  # noqa: E731
    else:
        mk = lambda s: O.procedural_nerf_params(args.fc_units, seed=s)  # noqa: E731
    models = {"coarse": {k: v.to(device) for k, v in mk(seed_coarse).items()}}
    if args.n_importance > 0:
        models["fine"] = {k: v.to(device) for k, v in mk(seed_fine).items()}
    if args.model == "sat-nerf":
        models["t"] = O.procedural_uniform((args.t_embbeding_vocab, args.t_embbeding_tau), 1.0, emb_seed).to(device)
    return models
