"""bench.py started plainly with --gpus N > 1 must become the launcher of its N ranks (before anything touches a
GPU), relay rank 0's single JSON line, and never print a 1-GPU line for an N-GPU request.  --launch-check keeps the
ranks off the GPU and the engine, so this runs on the CPU (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def test_plain_invocation_for_two_gpus_starts_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == [0, 1]
    # every rank got its share of the host, not the whole of it
    assert int(d["dut_threads"]) <= max(2, int(d["host_budget"]) // 2)


def test_three_ranks_and_an_explicit_thread_count_is_kept():
    r = _run(["--gpus", "3", "--launch-check"], env={"DUT_THREADS": "5"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["ranks"] == [0, 1, 2] and d["dut_threads"] == "5"


def test_a_launcher_that_cannot_start_gives_no_line_and_a_failure_status():
    r = _run(["--gpus", "2", "--launch-check"], env={"BENCH_LAUNCHER": "no.such.launcher"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_a_rank_count_that_differs_from_the_request_is_refused():
    # an outer launcher that gives one rank to a --gpus 8 request: no n_gpus = 1 line, a failure status
    r = _run(["--gpus", "8", "--launch-check"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
