"""The BENCHED configuration held to the oracle directly (Next #4).

`bench.py`'s headline step is 1024 rays x 64 samples, mlp_mode='bf16', 8-bit saved state, fp16-operand weight gradients with a
per-workgroup range fit: at that size a weight-gradient slice is ~110 point tiles deep and the four-deep slice loop, the range fit
and the split-K reduction all run as they do in the measured step.  `Trainer._forward_backward` (the kernel-direct step, eager) is
compared against autograd through `oracle.satnerf_oracle` (the pinned fp32 restatement of rendering.py:52-158 +
metrics.SatNerfLoss) on IDENTICAL stratified draws (the trainer's jitter hook), with the reference's own initialisation
(models/satnerf.py:104-153 ctor, as bench.py builds it) -- per parameter tensor, max-norm relative, gates ~1.5x the measured error
(GATES below).  Cases: the headline arithmetic, the fp16-operand forward, the reference's own width 512; and BASELINE configs[3] size
(4096 colour + 4096 depth-supervision rays per step), every tensor.
"""
import pytest
import torch

from oracle import satnerf_oracle as O
from tests.helpers import maxnorm_rel

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

# Per-tensor gates = 1.5 x the max-norm relative error of each gradient tensor measured on MI355X by this file's own printout (the r06 run
# is committed as profiles/r06_benched_shape_errors.json; tools/make_benched_gates.py wrote the literal below from it; DESIGN.md section 2
# summarises it; a run on a GPU box records its own errors in gpurun_out/benched_shape_errors.json); a tensor not listed is held to
# DEFAULT_GATE.  Keys: (case, tensor name).
DEFAULT_GATE = 3.0e-2
GATES = {
    # bf16/256/1024x64: measured max 1.60e-02, flat 2-norm 4.42e-03
    ('bf16/256/1024x64', 'beta_from_xyz.0.bias'): 4.0e-03,  # measured 2.61e-03
    ('bf16/256/1024x64', 'beta_from_xyz.0.weight'): 6.6e-03,  # measured 4.38e-03
    ('bf16/256/1024x64', 'beta_from_xyz.2.bias'): 2.0e-04,  # measured 1.55e-05
    ('bf16/256/1024x64', 'beta_from_xyz.2.weight'): 7.5e-03,  # measured 4.99e-03
    ('bf16/256/1024x64', 'embedding_t.weight'): 2.4e-03,  # measured 1.60e-03
    ('bf16/256/1024x64', 'fc_net.0.bias'): 2.5e-02,  # measured 1.60e-02
    ('bf16/256/1024x64', 'fc_net.0.weight'): 8.7e-03,  # measured 5.74e-03
    ('bf16/256/1024x64', 'fc_net.10.bias'): 5.6e-03,  # measured 3.69e-03
    ('bf16/256/1024x64', 'fc_net.10.weight'): 7.6e-03,  # measured 5.06e-03
    ('bf16/256/1024x64', 'fc_net.12.bias'): 4.7e-03,  # measured 3.09e-03
    ('bf16/256/1024x64', 'fc_net.12.weight'): 8.2e-03,  # measured 5.46e-03
    ('bf16/256/1024x64', 'fc_net.14.bias'): 3.5e-03,  # measured 2.27e-03
    ('bf16/256/1024x64', 'fc_net.14.weight'): 8.7e-03,  # measured 5.75e-03
    ('bf16/256/1024x64', 'fc_net.2.bias'): 8.3e-03,  # measured 5.52e-03
    ('bf16/256/1024x64', 'fc_net.2.weight'): 1.2e-02,  # measured 7.41e-03
    ('bf16/256/1024x64', 'fc_net.4.bias'): 8.7e-03,  # measured 5.78e-03
    ('bf16/256/1024x64', 'fc_net.4.weight'): 1.7e-02,  # measured 1.12e-02
    ('bf16/256/1024x64', 'fc_net.6.bias'): 8.6e-03,  # measured 5.72e-03
    ('bf16/256/1024x64', 'fc_net.6.weight'): 1.2e-02,  # measured 7.62e-03
    ('bf16/256/1024x64', 'fc_net.8.bias'): 6.1e-03,  # measured 4.06e-03
    ('bf16/256/1024x64', 'fc_net.8.weight'): 8.4e-03,  # measured 5.58e-03
    ('bf16/256/1024x64', 'feats_from_xyz.bias'): 3.2e-03,  # measured 2.12e-03
    ('bf16/256/1024x64', 'feats_from_xyz.weight'): 8.8e-03,  # measured 5.81e-03
    ('bf16/256/1024x64', 'rgb_from_xyzdir.0.bias'): 3.2e-03,  # measured 2.10e-03
    ('bf16/256/1024x64', 'rgb_from_xyzdir.0.weight'): 8.4e-03,  # measured 5.57e-03
    ('bf16/256/1024x64', 'rgb_from_xyzdir.2.bias'): 2.0e-04,  # measured 1.30e-04
    ('bf16/256/1024x64', 'rgb_from_xyzdir.2.weight'): 1.2e-02,  # measured 7.87e-03
    ('bf16/256/1024x64', 'sigma_from_xyz.0.bias'): 4.3e-03,  # measured 2.84e-03
    ('bf16/256/1024x64', 'sigma_from_xyz.0.weight'): 9.4e-03,  # measured 6.23e-03
    ('bf16/256/1024x64', 'sky_color.0.bias'): 3.7e-04,  # measured 2.44e-04
    ('bf16/256/1024x64', 'sky_color.0.weight'): 3.8e-04,  # measured 2.52e-04
    ('bf16/256/1024x64', 'sky_color.2.bias'): 6.5e-04,  # measured 4.29e-04
    ('bf16/256/1024x64', 'sky_color.2.weight'): 6.6e-04,  # measured 4.34e-04
    ('bf16/256/1024x64', 'sun_v_net.0.bias'): 5.1e-03,  # measured 3.36e-03
    ('bf16/256/1024x64', 'sun_v_net.0.weight'): 5.1e-03,  # measured 3.38e-03
    ('bf16/256/1024x64', 'sun_v_net.2.bias'): 3.4e-03,  # measured 2.22e-03
    ('bf16/256/1024x64', 'sun_v_net.2.weight'): 2.4e-02,  # measured 1.54e-02
    ('bf16/256/1024x64', 'sun_v_net.4.bias'): 3.8e-03,  # measured 2.53e-03
    ('bf16/256/1024x64', 'sun_v_net.4.weight'): 9.2e-03,  # measured 6.07e-03
    ('bf16/256/1024x64', 'sun_v_net.6.bias'): 2.5e-04,  # measured 1.61e-04
    ('bf16/256/1024x64', 'sun_v_net.6.weight'): 5.7e-03,  # measured 3.75e-03
    # bf16/256/C4: measured max 8.21e-03, flat 2-norm 4.62e-03
    ('bf16/256/C4', 'beta_from_xyz.0.bias'): 4.3e-03,  # measured 2.86e-03
    ('bf16/256/C4', 'beta_from_xyz.0.weight'): 4.5e-03,  # measured 2.98e-03
    ('bf16/256/C4', 'beta_from_xyz.2.bias'): 2.0e-04,  # measured 8.44e-05
    ('bf16/256/C4', 'beta_from_xyz.2.weight'): 3.9e-03,  # measured 2.60e-03
    ('bf16/256/C4', 'embedding_t.weight'): 3.0e-03,  # measured 1.98e-03
    ('bf16/256/C4', 'fc_net.0.bias'): 1.3e-02,  # measured 8.21e-03
    ('bf16/256/C4', 'fc_net.0.weight'): 7.4e-03,  # measured 4.91e-03
    ('bf16/256/C4', 'fc_net.10.bias'): 4.2e-03,  # measured 2.73e-03
    ('bf16/256/C4', 'fc_net.10.weight'): 4.2e-03,  # measured 2.78e-03
    ('bf16/256/C4', 'fc_net.12.bias'): 3.9e-03,  # measured 2.56e-03
    ('bf16/256/C4', 'fc_net.12.weight'): 6.2e-03,  # measured 4.13e-03
    ('bf16/256/C4', 'fc_net.14.bias'): 3.2e-03,  # measured 2.09e-03
    ('bf16/256/C4', 'fc_net.14.weight'): 6.1e-03,  # measured 4.03e-03
    ('bf16/256/C4', 'fc_net.2.bias'): 6.1e-03,  # measured 4.02e-03
    ('bf16/256/C4', 'fc_net.2.weight'): 5.0e-03,  # measured 3.28e-03
    ('bf16/256/C4', 'fc_net.4.bias'): 7.6e-03,  # measured 5.00e-03
    ('bf16/256/C4', 'fc_net.4.weight'): 7.8e-03,  # measured 5.19e-03
    ('bf16/256/C4', 'fc_net.6.bias'): 6.1e-03,  # measured 4.00e-03
    ('bf16/256/C4', 'fc_net.6.weight'): 6.4e-03,  # measured 4.27e-03
    ('bf16/256/C4', 'fc_net.8.bias'): 4.7e-03,  # measured 3.07e-03
    ('bf16/256/C4', 'fc_net.8.weight'): 4.5e-03,  # measured 2.95e-03
    ('bf16/256/C4', 'feats_from_xyz.bias'): 4.3e-03,  # measured 2.84e-03
    ('bf16/256/C4', 'feats_from_xyz.weight'): 6.4e-03,  # measured 4.21e-03
    ('bf16/256/C4', 'rgb_from_xyzdir.0.bias'): 3.1e-03,  # measured 2.05e-03
    ('bf16/256/C4', 'rgb_from_xyzdir.0.weight'): 6.0e-03,  # measured 3.96e-03
    ('bf16/256/C4', 'rgb_from_xyzdir.2.bias'): 2.2e-04,  # measured 1.45e-04
    ('bf16/256/C4', 'rgb_from_xyzdir.2.weight'): 7.3e-03,  # measured 4.82e-03
    ('bf16/256/C4', 'sigma_from_xyz.0.bias'): 2.0e-04,  # measured 1.25e-04
    ('bf16/256/C4', 'sigma_from_xyz.0.weight'): 5.5e-03,  # measured 3.62e-03
    ('bf16/256/C4', 'sky_color.0.bias'): 3.3e-04,  # measured 2.17e-04
    ('bf16/256/C4', 'sky_color.0.weight'): 3.4e-04,  # measured 2.24e-04
    ('bf16/256/C4', 'sky_color.2.bias'): 4.3e-04,  # measured 2.84e-04
    ('bf16/256/C4', 'sky_color.2.weight'): 4.5e-04,  # measured 2.97e-04
    ('bf16/256/C4', 'sun_v_net.0.bias'): 5.0e-03,  # measured 3.32e-03
    ('bf16/256/C4', 'sun_v_net.0.weight'): 5.1e-03,  # measured 3.34e-03
    ('bf16/256/C4', 'sun_v_net.2.bias'): 3.9e-03,  # measured 2.55e-03
    ('bf16/256/C4', 'sun_v_net.2.weight'): 1.1e-02,  # measured 6.96e-03
    ('bf16/256/C4', 'sun_v_net.4.bias'): 3.5e-03,  # measured 2.27e-03
    ('bf16/256/C4', 'sun_v_net.4.weight'): 5.5e-03,  # measured 3.63e-03
    ('bf16/256/C4', 'sun_v_net.6.bias'): 2.0e-04,  # measured 4.56e-05
    ('bf16/256/C4', 'sun_v_net.6.weight'): 3.4e-03,  # measured 2.26e-03
    # bf16/512/1024x64: measured max 3.36e-02, flat 2-norm 5.00e-03
    ('bf16/512/1024x64', 'beta_from_xyz.0.bias'): 4.4e-03,  # measured 2.89e-03
    ('bf16/512/1024x64', 'beta_from_xyz.0.weight'): 4.6e-03,  # measured 3.01e-03
    ('bf16/512/1024x64', 'beta_from_xyz.2.bias'): 2.0e-04,  # measured 3.41e-05
    ('bf16/512/1024x64', 'beta_from_xyz.2.weight'): 6.9e-03,  # measured 4.58e-03
    ('bf16/512/1024x64', 'embedding_t.weight'): 2.3e-03,  # measured 1.50e-03
    ('bf16/512/1024x64', 'fc_net.0.bias'): 1.2e-02,  # measured 7.77e-03
    ('bf16/512/1024x64', 'fc_net.0.weight'): 1.2e-02,  # measured 7.80e-03
    ('bf16/512/1024x64', 'fc_net.10.bias'): 6.7e-03,  # measured 4.41e-03
    ('bf16/512/1024x64', 'fc_net.10.weight'): 1.4e-02,  # measured 8.93e-03
    ('bf16/512/1024x64', 'fc_net.12.bias'): 5.1e-03,  # measured 3.34e-03
    ('bf16/512/1024x64', 'fc_net.12.weight'): 8.7e-03,  # measured 5.76e-03
    ('bf16/512/1024x64', 'fc_net.14.bias'): 5.1e-03,  # measured 3.36e-03
    ('bf16/512/1024x64', 'fc_net.14.weight'): 9.4e-03,  # measured 6.25e-03
    ('bf16/512/1024x64', 'fc_net.2.bias'): 6.2e-03,  # measured 4.11e-03
    ('bf16/512/1024x64', 'fc_net.2.weight'): 7.3e-03,  # measured 4.86e-03
    ('bf16/512/1024x64', 'fc_net.4.bias'): 6.5e-03,  # measured 4.27e-03
    ('bf16/512/1024x64', 'fc_net.4.weight'): 1.2e-02,  # measured 7.58e-03
    ('bf16/512/1024x64', 'fc_net.6.bias'): 7.4e-03,  # measured 4.88e-03
    ('bf16/512/1024x64', 'fc_net.6.weight'): 1.3e-02,  # measured 8.03e-03
    ('bf16/512/1024x64', 'fc_net.8.bias'): 5.5e-03,  # measured 3.61e-03
    ('bf16/512/1024x64', 'fc_net.8.weight'): 1.2e-02,  # measured 7.47e-03
    ('bf16/512/1024x64', 'feats_from_xyz.bias'): 3.6e-03,  # measured 2.35e-03
    ('bf16/512/1024x64', 'feats_from_xyz.weight'): 7.6e-03,  # measured 5.03e-03
    ('bf16/512/1024x64', 'rgb_from_xyzdir.0.bias'): 2.9e-03,  # measured 1.88e-03
    ('bf16/512/1024x64', 'rgb_from_xyzdir.0.weight'): 8.8e-03,  # measured 5.82e-03
    ('bf16/512/1024x64', 'rgb_from_xyzdir.2.bias'): 5.9e-04,  # measured 3.88e-04
    ('bf16/512/1024x64', 'rgb_from_xyzdir.2.weight'): 1.5e-02,  # measured 9.82e-03
    ('bf16/512/1024x64', 'sigma_from_xyz.0.bias'): 6.4e-03,  # measured 4.26e-03
    ('bf16/512/1024x64', 'sigma_from_xyz.0.weight'): 6.7e-03,  # measured 4.44e-03
    ('bf16/512/1024x64', 'sky_color.0.bias'): 2.4e-04,  # measured 1.60e-04
    ('bf16/512/1024x64', 'sky_color.0.weight'): 2.6e-04,  # measured 1.71e-04
    ('bf16/512/1024x64', 'sky_color.2.bias'): 3.8e-04,  # measured 2.49e-04
    ('bf16/512/1024x64', 'sky_color.2.weight'): 3.8e-04,  # measured 2.52e-04
    ('bf16/512/1024x64', 'sun_v_net.0.bias'): 4.9e-03,  # measured 3.24e-03
    ('bf16/512/1024x64', 'sun_v_net.0.weight'): 5.1e-03,  # measured 3.40e-03
    ('bf16/512/1024x64', 'sun_v_net.2.bias'): 3.5e-03,  # measured 2.28e-03
    ('bf16/512/1024x64', 'sun_v_net.2.weight'): 5.1e-02,  # measured 3.36e-02
    ('bf16/512/1024x64', 'sun_v_net.4.bias'): 5.0e-03,  # measured 3.30e-03
    ('bf16/512/1024x64', 'sun_v_net.4.weight'): 1.4e-02,  # measured 9.07e-03
    ('bf16/512/1024x64', 'sun_v_net.6.bias'): 2.0e-04,  # measured 1.78e-05
    ('bf16/512/1024x64', 'sun_v_net.6.weight'): 7.0e-03,  # measured 4.65e-03
    # f16/256/1024x64: measured max 1.40e-02, flat 2-norm 3.68e-03
    ('f16/256/1024x64', 'beta_from_xyz.0.bias'): 4.0e-03,  # measured 2.61e-03
    ('f16/256/1024x64', 'beta_from_xyz.0.weight'): 7.0e-03,  # measured 4.62e-03
    ('f16/256/1024x64', 'beta_from_xyz.2.bias'): 2.0e-04,  # measured 9.07e-06
    ('f16/256/1024x64', 'beta_from_xyz.2.weight'): 6.0e-03,  # measured 3.95e-03
    ('f16/256/1024x64', 'embedding_t.weight'): 2.6e-03,  # measured 1.67e-03
    ('f16/256/1024x64', 'fc_net.0.bias'): 1.8e-02,  # measured 1.19e-02
    ('f16/256/1024x64', 'fc_net.0.weight'): 7.9e-03,  # measured 5.26e-03
    ('f16/256/1024x64', 'fc_net.10.bias'): 5.6e-03,  # measured 3.72e-03
    ('f16/256/1024x64', 'fc_net.10.weight'): 7.3e-03,  # measured 4.86e-03
    ('f16/256/1024x64', 'fc_net.12.bias'): 4.6e-03,  # measured 3.06e-03
    ('f16/256/1024x64', 'fc_net.12.weight'): 6.6e-03,  # measured 4.34e-03
    ('f16/256/1024x64', 'fc_net.14.bias'): 3.7e-03,  # measured 2.42e-03
    ('f16/256/1024x64', 'fc_net.14.weight'): 4.6e-03,  # measured 3.02e-03
    ('f16/256/1024x64', 'fc_net.2.bias'): 7.0e-03,  # measured 4.62e-03
    ('f16/256/1024x64', 'fc_net.2.weight'): 1.0e-02,  # measured 6.63e-03
    ('f16/256/1024x64', 'fc_net.4.bias'): 7.9e-03,  # measured 5.23e-03
    ('f16/256/1024x64', 'fc_net.4.weight'): 1.4e-02,  # measured 9.09e-03
    ('f16/256/1024x64', 'fc_net.6.bias'): 9.5e-03,  # measured 6.31e-03
    ('f16/256/1024x64', 'fc_net.6.weight'): 9.8e-03,  # measured 6.52e-03
    ('f16/256/1024x64', 'fc_net.8.bias'): 5.4e-03,  # measured 3.59e-03
    ('f16/256/1024x64', 'fc_net.8.weight'): 7.0e-03,  # measured 4.61e-03
    ('f16/256/1024x64', 'feats_from_xyz.bias'): 3.1e-03,  # measured 2.04e-03
    ('f16/256/1024x64', 'feats_from_xyz.weight'): 4.1e-03,  # measured 2.71e-03
    ('f16/256/1024x64', 'rgb_from_xyzdir.0.bias'): 3.4e-03,  # measured 2.25e-03
    ('f16/256/1024x64', 'rgb_from_xyzdir.0.weight'): 4.2e-03,  # measured 2.74e-03
    ('f16/256/1024x64', 'rgb_from_xyzdir.2.bias'): 2.5e-04,  # measured 1.61e-04
    ('f16/256/1024x64', 'rgb_from_xyzdir.2.weight'): 1.8e-02,  # measured 1.18e-02
    ('f16/256/1024x64', 'sigma_from_xyz.0.bias'): 3.1e-04,  # measured 2.06e-04
    ('f16/256/1024x64', 'sigma_from_xyz.0.weight'): 1.5e-03,  # measured 9.71e-04
    ('f16/256/1024x64', 'sky_color.0.bias'): 2.0e-04,  # measured 1.20e-04
    ('f16/256/1024x64', 'sky_color.0.weight'): 2.0e-04,  # measured 1.19e-04
    ('f16/256/1024x64', 'sky_color.2.bias'): 2.0e-04,  # measured 1.29e-04
    ('f16/256/1024x64', 'sky_color.2.weight'): 2.0e-04,  # measured 1.28e-04
    ('f16/256/1024x64', 'sun_v_net.0.bias'): 5.1e-03,  # measured 3.39e-03
    ('f16/256/1024x64', 'sun_v_net.0.weight'): 5.2e-03,  # measured 3.40e-03
    ('f16/256/1024x64', 'sun_v_net.2.bias'): 3.3e-03,  # measured 2.17e-03
    ('f16/256/1024x64', 'sun_v_net.2.weight'): 2.1e-02,  # measured 1.40e-02
    ('f16/256/1024x64', 'sun_v_net.4.bias'): 3.4e-03,  # measured 2.26e-03
    ('f16/256/1024x64', 'sun_v_net.4.weight'): 7.7e-03,  # measured 5.12e-03
    ('f16/256/1024x64', 'sun_v_net.6.bias'): 2.0e-04,  # measured 6.36e-05
    ('f16/256/1024x64', 'sun_v_net.6.weight'): 3.6e-03,  # measured 2.39e-03
}


def _record(case, errs, rel2):
    """keep the measured errors next to the run (gpurun_out/ travels back from the GPU box)"""
    import json
    import os

    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(d):
        return
    path = os.path.join(d, "benched_shape_errors.json")
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        doc = {}
    doc[case] = {"errors": {k: float(f"{e:.3e}") for k, e in errs.items()}, "flat_rel2": float(f"{rel2:.3e}")}
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)


def _check(case, errs, rel2, rel2_gate):
    print(f"\nper-tensor gradient error, case {case}:")
    for k, e in errs.items():
        print(f"    ({case!r}, {k!r}): {e:.2e},")
    print(f"    flat gradient, relative 2-norm error: {rel2:.2e}")
    _record(case, errs, rel2)
    bad = {k: (e, GATES.get((case, k), DEFAULT_GATE)) for k, e in errs.items() if not e < GATES.get((case, k), DEFAULT_GATE)}
    assert not bad, bad
    assert rel2 < rel2_gate, rel2


def _models(args, seed=0):
    from satnerf_amd.models import load_model

    torch.manual_seed(seed)
    model = load_model(args)
    emb = torch.nn.Embedding(args.t_embbeding_vocab, args.t_embbeding_tau)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    embw = emb.weight.detach().clone()
    return {"coarse": model.to(DEV).train(), "t": emb.to(DEV)}, params, embw


def _oracle_grads(params, embw, args, rays, ts, u, loss_of):
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    res = O.render_rays({"coarse": po, "t": eo}, args, rays, ts, O.ReplayRng([u, torch.zeros_like(u)]))
    loss = loss_of(res)
    loss.backward()
    return loss.item(), {k: v.grad for k, v in po.items()}, eo.grad


@pytest.mark.parametrize("mode,feat", [("bf16", 256), ("f16", 256), ("bf16", 512)])
def test_benched_step_gradients_match_oracle_per_tensor(mode, feat):
    """1024 x 64 -- the benched shape -- in the headline arithmetic (bf16, 8-bit state), with fp16 forward operands (`train_f16`) and at the
    reference's own width (fc_units 512, opt.py:50: `train_width512`)."""
    from satnerf_amd.train import Trainer, _fmt_of

    n, s = 1024, 64
    args = O.default_args(mlp_mode=mode, fc_units=feat)
    models, params, embw = _models(args)
    rays, ts = O.synthetic_rays(n)  # the SURVEY 8(d) recipe bench.py's bank is drawn from
    g = torch.Generator().manual_seed(99)
    target = torch.rand(n, 3, generator=g)
    u = torch.rand(n, s, generator=g)

    tr = Trainer(models, args, use_graph=False)
    assert tr.direct and _fmt_of(args) == 8  # the kernel-direct step, the benched saved-state format
    tr.jitter = lambda n_, s_, device: u.to(device)
    loss = tr._forward_backward(rays.to(DEV), ts.to(DEV), target.to(DEV))
    torch.cuda.synchronize()

    lo, go, ge = _oracle_grads(params, embw, O.default_args(fc_units=feat), rays, ts, u, lambda r: O.satnerf_loss(r, target))
    assert abs(loss.sum().item() - lo) < 5e-3 * abs(lo), (loss.sum().item(), lo)
    sd = dict(models["coarse"].named_parameters())
    assert set(sd) == set(go)
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), go[k]) for k in go}
    errs["embedding_t.weight"] = maxnorm_rel(models["t"].weight.grad.cpu(), ge)
    assert all(torch.isfinite(sd[k].grad).all() for k in sd)
    # relative error of the whole flat gradient in the 2-norm: what an optimizer step sees
    flat_o = torch.cat([go[k].reshape(-1) for k in sd] + [ge.reshape(-1)]).double()
    flat_h = torch.cat([sd[k].grad.reshape(-1).cpu() for k in sd] + [models["t"].weight.grad.reshape(-1).cpu()]).double()
    rel2 = ((flat_h - flat_o).norm() / flat_o.norm()).item()
    _check(f"{mode}/{feat}/1024x64", errs, rel2, 2e-2)


def test_c4_sized_step_gradients_match_oracle():
    """BASELINE configs[3]: 4096 colour rays + 4096 depth-supervision rays per step (ds_lambda 1000, run_all.sh:80): every tensor."""
    from satnerf_amd.train import Trainer

    n, s = 4096, 64
    args = O.default_args(mlp_mode="bf16", ds_lambda=1000.0)
    models, params, embw = _models(args, seed=1)
    rays, ts = O.synthetic_rays(n, seed=41)
    d_rays, d_ts = O.synthetic_rays(n, seed=42)
    g = torch.Generator().manual_seed(43)
    target = torch.rand(n, 3, generator=g)
    depths = torch.stack([0.2 + 0.5 * torch.rand(n, generator=g), 0.5 + torch.rand(n, generator=g)], 1)
    u_c, u_d = torch.rand(n, s, generator=g), torch.rand(n, s, generator=g)

    tr = Trainer(models, args, use_graph=False)
    assert tr.direct
    draws = iter([u_c, u_d])
    tr.jitter = lambda n_, s_, device: next(draws).to(device)
    loss = tr._forward_backward(rays.to(DEV), ts.to(DEV), target.to(DEV), depth=(d_rays.to(DEV), d_ts.to(DEV), depths.to(DEV)))
    torch.cuda.synchronize()

    oargs = O.default_args()
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eo = embw.clone().requires_grad_(True)
    res_c = O.render_rays({"coarse": po, "t": eo}, oargs, rays, ts, O.ReplayRng([u_c, torch.zeros_like(u_c)]))
    res_d = O.render_rays({"coarse": po, "t": eo}, oargs, d_rays, d_ts, O.ReplayRng([u_d, torch.zeros_like(u_d)]))
    lo = O.satnerf_loss(res_c, target) + O.depth_loss(res_d, depths[:, 0], depths[:, 1], lambda_ds=1000.0)
    lo.backward()
    assert abs(loss.sum().item() - lo.item()) < 5e-3 * abs(lo.item()), (loss.sum().item(), lo.item())
    sd = dict(models["coarse"].named_parameters())
    errs = {k: maxnorm_rel(sd[k].grad.cpu(), po[k].grad) for k in sd}
    errs["embedding_t.weight"] = maxnorm_rel(models["t"].weight.grad.cpu(), eo.grad)
    flat_o = torch.cat([po[k].grad.reshape(-1) for k in sd] + [eo.grad.reshape(-1)]).double()
    flat_h = torch.cat([sd[k].grad.reshape(-1).cpu() for k in sd] + [models["t"].weight.grad.reshape(-1).cpu()]).double()
    rel2 = ((flat_h - flat_o).norm() / flat_o.norm()).item()
    _check("bf16/256/C4", errs, rel2, 2e-2)
