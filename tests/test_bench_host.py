"""Host-side pieces of bench.py that need no GPU: the CPU baseline leg (the oracle timed as bench.py times it: the whole batch,
the intra-op pool swept over the physical-core counts, the best size timed for a bounded budget; r06: the batch sharded over processes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_cpu_baseline_reports_the_sweep_and_the_sharded_leg(monkeypatch):
    monkeypatch.setattr(os, "cpu_count", lambda: 3)  # every probed size clamps to the host's 3 threads: one sweep entry
    r = bench.cpu_baseline("forward", 16, 8, budget_s=0.5)
    assert r["kind"] == "port" and r["unit"] == "rays/s" and r["host_cpus"] == 3
    sp = r["single_process"]
    assert sp["value"] > 0 and sp["threads"] == 3 and list(sp["sweep_rays_per_s"]) == ["3"]
    # r06: the batch sharded over processes (here 2 x 1 thread, 8 rays each), started together; `value` = the better of the two legs
    sh = r["sharded"]
    assert sh["processes"] == 2 and sh["threads_per_process"] == 1 and sh["rays_per_process"] == 8 and sh["value"] > 0
    assert "replicas" not in r   # (one whole batch per process: measured once, 575 rays/s on 256 threads, dropped)
    assert r["value"] == max(sp["value"], sh["value"]) and r["cores"] in (3, 2)
    assert "x 8 samples" in r["sample"] and "forward" in r["sample"]


def test_cpu_baseline_times_the_training_pass_within_its_budget(monkeypatch):
    import time

    monkeypatch.setattr(os, "cpu_count", lambda: 1)  # (one hardware thread: no sharded leg)
    t0 = time.time()
    r = bench.cpu_baseline("train", 8, 4, budget_s=0.3)
    assert r["value"] > 0 and "train" in r["sample"] and "sharded" not in r and time.time() - t0 < 60
