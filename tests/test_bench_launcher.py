"""bench.py's own launcher (`python bench.py --gpus N` with no launcher environment): N fresh child processes, one rank each,
started before anything touches the GPU; rank 0's JSON line relayed.  Exercised here on CPU with the stand-in renderer
(--stub: the real exchange plan of raytracedggx_amd.strips over gloo, every frame's apron rows and the assembled frame checked
inside the ranks), world size 2 and 3."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_launches_its_own_ranks(world):
    r = _run(["--gpus", str(world), "--stub", "--steps", "4", "--warmup", "2", "--width", "16", "--height", "120"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0: %r" % r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 4 and out["warmup"] == 2
    assert out["scaling"] == "strong" and out["unit"] == "Mrays/s" and out["value"] > 0 and out["ms_per_step"] > 0
    assert out["config"]["parallelism"] == "row strips x%d" % world


def test_bench_single_rank_stub_and_world_size_mismatch():
    r = _run(["--gpus", "1", "--stub", "--steps", "2", "--warmup", "1", "--width", "16", "--height", "60"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    # under a real launcher the environment decides: a mismatch with --gpus is an error, not a silent single-GPU run
    r = _run(["--gpus", "4", "--stub", "--steps", "1", "--warmup", "0"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stdout + r.stderr)


def test_a_failing_rank_fails_the_launcher():
    # strips thinner than the history apron: every rank raises; the launcher must not report success
    r = _run(["--gpus", "2", "--stub", "--steps", "1", "--warmup", "0", "--width", "16", "--height", "20"])
    assert r.returncode != 0
