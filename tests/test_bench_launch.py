"""CPU: `python bench.py --gpus 2` without a launcher must become two ranks (--gpus was parsed and
ignored).  Runs the same self-launch path on the gloo backend with a tiny model (`--plumbing`)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing", "--steps", "2"] + list(extra),
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_self_launches_two_ranks():
    out = run_bench("--gpus", "2")
    assert out["n_gpus"] == 2 and out["plumbing"] is True and out["value"] > 0


def test_single_rank_default():
    out = run_bench()
    assert out["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    env = {k: v for k, v in os.environ.items()}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing", "--gpus", "2"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
