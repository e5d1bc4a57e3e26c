"""Host-side pieces of bench.py that need no GPU: the CPU baseline leg (oracle timed as bench.py times it) and its time-boxed
all-threads figure (a pass at the GPU boxes' 256 threads takes ~100 s: it runs in a child process that is killed at the limit)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_cpu_baseline_reports_both_thread_counts(monkeypatch):
    monkeypatch.setattr(os, "cpu_count", lambda: 3)  # the probed pool size IS all threads: one figure, no child process
    r = bench.cpu_baseline("forward", 16, 8, budget_s=0.5)
    assert r["kind"] == "port" and r["unit"] == "rays/s" and r["host_cpus"] == 3 and r["all_threads"] == 3
    assert r["value"] > 0 and 1 <= r["cores"] <= 3
    assert r["value_all_threads"] is None or r["value_all_threads"] > 0
    assert (r["value_all_threads"] is None) == (r["all_threads_note"] is not None)


def test_all_threads_leg_gives_up_at_its_limit(monkeypatch):
    monkeypatch.setattr(os, "cpu_count", lambda: 100)  # none of the probed pool sizes: the child process runs
    monkeypatch.setattr(bench, "ALL_THREADS_LIMIT_S", 0.2)  # shorter than the child's interpreter start-up
    r = bench.cpu_baseline("forward", 16, 8, budget_s=0.3)
    assert r["value_all_threads"] is None and "did not finish" in r["all_threads_note"] and r["value"] > 0
