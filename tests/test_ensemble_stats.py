"""The statistics of the g1 ensemble study (tools/convergence_ensemble.py: Welch's difference of means with a 95 % interval) against known
answers; no GPU.  The study itself runs on the GPU box (profiles/r06_convergence_ensemble.json), its short version under -m gpu
(tests/test_hip_convergence.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_welch_interval_and_the_three_readings():
    import convergence_ensemble as ce

    w = ce.welch([1.0, 2.0, 3.0], [2.0, 3.0, 4.0])   # equal variances 1, n = 3 each: se^2 = 2 / 3, df = 4, t = 2.776
    assert abs(w["delta"] + 1.0) < 1e-12 and abs(w["df"] - 4.0) < 1e-9 and abs(w["se"] - (2.0 / 3.0) ** 0.5) < 1e-12
    assert abs(w["ci95"][1] - (-1.0 + 2.776 * (2.0 / 3.0) ** 0.5)) < 1e-9
    assert abs(ce.t975(4) - 2.776) < 1e-9 and abs(ce.t975(14) - 2.145) < 1e-9 and 1.96 < ce.t975(1e6) < 1.961
    assert ce.t975(7) > ce.t975(7.5) > ce.t975(8)   # interpolated between the table's rows
    # unequal variances: Welch-Satterthwaite df lies between min(n) - 1 and n_a + n_b - 2
    w2 = ce.welch([0.0, 10.0, 20.0, 30.0], [4.9, 5.0, 5.1, 5.0, 5.0, 5.0])
    assert 3.0 <= w2["df"] <= 8.0 and w2["ci95"][0] < w2["delta"] < w2["ci95"][1]
    tight = ce.summarise([0.500, 0.501, 0.499, 0.500], [0.505, 0.506, 0.504, 0.505])
    assert tight["ci_inside_bar"] and tight["ci_excludes_zero"] and not tight["bar_inside_ci"]          # a 5-mm difference, established, inside 2 cm
    loose = ce.summarise([0.45, 0.47, 0.50, 0.44], [0.46, 0.45, 0.52, 0.41])
    assert not loose["ci_inside_bar"] and not loose["ci_excludes_zero"] and loose["bar_inside_ci"]        # too small a study to say


def test_committed_study_is_consistent_with_its_runs():
    path = os.path.join(ROOT, "profiles", "r06_convergence_ensemble.json")
    if not os.path.exists(path):
        import pytest

        pytest.skip("study not recorded yet")
    import convergence_ensemble as ce

    doc = json.load(open(path))
    hip, ref = [r["final_m"] for r in doc["hip_runs"]], [r["final_m"] for r in doc["ref_runs"]]
    s = ce.summarise(hip, ref)
    assert len(hip) >= 8 and len(ref) >= 8 and abs(s["delta_mean_m"] - doc["summary"]["delta_mean_m"]) < 1e-12
    assert s["ci95_m"] == doc["summary"]["ci95_m"]
    for r in doc["hip_runs"] + doc["ref_runs"]:   # final metric = mean of the last three checkpoints
        cps = sorted(int(k) for k in r["checkpoints"])
        assert len(cps) == 3 and cps[-1] == r["steps"] and abs(sum(r["checkpoints"].values()) / 3 - r["final_m"]) < 1e-12
    assert len({r["seed"] for r in doc["hip_runs"] + doc["ref_runs"]}) == len(hip) + len(ref)   # no two runs share a seed
