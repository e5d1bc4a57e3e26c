"""Checkpoint loading with the reference's conventions (eval_satnerf.py:23-93).

The reference trains under pytorch-lightning: a checkpoint is ``{'state_dict': {...}}`` whose keys carry the attribute names
of ``NeRF_pl`` (main.py:51-58): ``nerf_coarse.<key>``, ``nerf_fine.<key>``, ``embedding_t.weight``.  ``load_ckpt`` strips the
prefix and copies by key, which works unchanged for ``satnerf_amd.models.SatNeRF`` because its ``state_dict`` keys and shapes
are the reference's.  One deliberate deviation: ``load_nerf`` loads the ``nerf_fine.*`` weights into the FINE model
(eval_satnerf.py:86 loads them into the coarse one, leaving the fine model at its random init -- SURVEY.md section 4).
"""
from __future__ import annotations

import argparse
import json
import os

import torch

from .models import load_model


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=()):
    """eval_satnerf.extract_model_state_dict: entries of ``model_name.`` with the prefix stripped."""
    checkpoint = torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=False)
    if "state_dict" in checkpoint:  # a pytorch-lightning checkpoint
        checkpoint = checkpoint["state_dict"]
    out = {}
    for k, v in checkpoint.items():
        if not k.startswith(model_name):
            continue
        k = k[len(model_name) + 1:]
        if any(k.startswith(p) for p in prefixes_to_ignore):
            continue
        out[k] = v
    return out


def load_ckpt(model, ckpt_path, model_name="model", prefixes_to_ignore=()):
    """eval_satnerf.load_ckpt: update the model's state dict with the checkpoint's entries and load it."""
    model_dict = model.state_dict()
    model_dict.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore))
    model.load_state_dict(model_dict)


def load_nerf(run_id, logs_dir, ckpts_dir, epoch_number, device="cuda"):
    """eval_satnerf.load_nerf: (models dict, args) for ``{logs_dir}/{run_id}/opts.json`` + ``{ckpts_dir}/{run_id}/epoch=N.ckpt``."""
    with open(os.path.join(logs_dir, run_id, "opts.json")) as f:
        args = argparse.Namespace(**json.load(f))
    path = os.path.join(ckpts_dir, f"{run_id}/epoch={epoch_number}.ckpt")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Could not find checkpoint {path}")
    models = {}
    coarse = load_model(args)
    load_ckpt(coarse, path, model_name="nerf_coarse")
    models["coarse"] = coarse.to(device).eval()
    if args.n_importance > 0:
        fine = load_model(args)
        load_ckpt(fine, path, model_name="nerf_fine")
        models["fine"] = fine.to(device).eval()
    if args.model == "sat-nerf":
        emb = torch.nn.Embedding(args.t_embbeding_vocab, args.t_embbeding_tau)
        load_ckpt(emb, path, model_name="embedding_t")
        models["t"] = emb.to(device).eval()
    return models, args
