"""bench.py --gpus N must run N ranks by itself and report the world size it actually ran on (CPU test of the launcher and
collection logic: `--stub gloo` replaces the GPU work by a rendezvous + barrier + max-over-ranks of a fake time)."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_runs_n_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--stub", "gloo"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["probes_total"] == 64 and line["config"]["world_size_seen_by_rank0"] == 2


def test_weak_scaling_multiplies_the_probes():
    r = _run(["--gpus", "2", "--stub", "gloo", "--scaling", "weak", "--probes", "8"])
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["probes_total"] == 16


def test_world_size_mismatch_is_an_error_not_a_silent_single_gpu_number():
    r = _run(["--gpus", "8", "--stub", "gloo"], env_extra={"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_single_rank_needs_no_rendezvous():
    r = _run(["--stub", "gloo"])
    assert r.returncode == 0, r.stderr
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1
