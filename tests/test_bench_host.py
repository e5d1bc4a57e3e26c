"""Host-side pieces of bench.py that need no GPU: the CPU baseline leg (the oracle timed as bench.py times it: the whole batch,
the intra-op pool swept over the physical-core counts, the best size timed for a bounded budget; r05 dropped the all-threads child)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_cpu_baseline_reports_the_sweep_and_its_best_pool(monkeypatch):
    monkeypatch.setattr(os, "cpu_count", lambda: 3)  # every probed size clamps to the host's 3 threads: one sweep entry
    r = bench.cpu_baseline("forward", 16, 8, budget_s=0.5)
    assert r["kind"] == "port" and r["unit"] == "rays/s" and r["host_cpus"] == 3
    assert r["value"] > 0 and r["cores"] == 3 and list(r["sweep_rays_per_s"]) == ["3"]
    assert "16 rays x 8 samples" in r["sample"] and "forward" in r["sample"]


def test_cpu_baseline_times_the_training_pass_within_its_budget(monkeypatch):
    import time

    monkeypatch.setattr(os, "cpu_count", lambda: 2)
    t0 = time.time()
    r = bench.cpu_baseline("train", 8, 4, budget_s=0.3)
    assert r["value"] > 0 and "train" in r["sample"] and time.time() - t0 < 60
