"""Depth-MAE / convergence check of the throughput arithmetic (north_star: "DSM MAE within 2 cm of the reference"; VERDICT g1).

tools/convergence.py trains the same seeded synthetic scene (known height field, 19 tilted views, depth supervision as in BASELINE
configs[3]) with (a) the kernel-direct HIP Trainer in bf16 with the 8-bit saved state and (b) the fp32 oracle + torch.optim.Adam on
IDENTICAL stratified draws, for three seeds, and renders a fixed ray set with identical draws.  Gate: the altitude-like MAE against
the true surface differs by < 2 cm between the two arithmetics, averaged over the seeds (metres at a 175 m scene range).
profiles/r03_convergence.json holds the recorded 3 x 1500-step run with the fp32-vs-fp32 noise floor."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_training_reaches_the_reference_depth_mae():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convergence

    res = convergence.run_seeds((0, 1, 2), steps=600, batch=256, n_eval=2048, modes=(("bf16", None),), floor=False)
    print(json.dumps(res["summary"]))
    for r in res["rows"]:  # both arithmetics learn the surface (untrained: ~40 m)
        assert r["hip"]["bf16"]["mae_truth_m"] < 5.0 and r["mae_truth_ref_m"] < 5.0, r
    # the north_star quantity, over the seed mean
    assert res["summary"]["mean_delta_mae_m"]["bf16"] < 0.02, res["summary"]
    # inference of the fp32-TRAINED weights: the parity mode reproduces fp32 depths to 1e-4 m, f16 to < 1 cm; single-pass bf16 moves
    # them by ~3 cm (SURVEY.md section 6 predicted 1.7-2.2 cm at init): DSM extraction should run in bf16x3 or f16
    r0 = res["rows"][0]
    assert r0["mae_infer_bf16x3_m"] < 0.002 and r0["mae_infer_f16_m"] < 0.02 and r0["mae_infer_bf16_m"] < 0.10, r0
