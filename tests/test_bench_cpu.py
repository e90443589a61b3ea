"""CPU: bench.py's own launcher. `python bench.py --gpus N` without torch.distributed.run around it must start N ranks
itself (`--gpus` used to be parsed and ignored). The hidden --selftest-launch mode runs only the rendezvous, one
all-reduce of ones and the JSON line, over gloo, so this runs in the GPU-less build container."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_starts_two_ranks():
    r = _run(["--gpus", "2", "--selftest-launch", "gloo"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_more_gpus_than_visible_fails_loudly():
    r = _run(["--gpus", "3"], env={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0 and "--gpus 3" in (r.stderr + r.stdout)


def test_a_dead_rank_fails_the_job():
    """A rank that dies takes the job down with a non-zero code instead of leaving the others in a collective."""
    r = _run(["--gpus", "2", "--selftest-launch", "no_such_backend"])
    assert r.returncode != 0
