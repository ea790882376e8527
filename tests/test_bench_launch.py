"""CPU: `python bench.py --gpus N` starts N ranks by itself when no launcher did (the flag used to be
parsed and ignored), and keeps working under torch.distributed.run.  --dry-run rehearses launch, gloo rendezvous and the
one-JSON-line contract without a GPU; the measured path itself needs an MI355X."""
import json
import os
import subprocess
import sys

from _helpers import ROOT


def _run(cmd, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    return r.returncode, [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stderr


def test_gpus_flag_spawns_one_rank_per_gpu():
    rc, lines, err = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert rc == 0, err
    assert len(lines) == 1                     # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1


def test_single_rank_default_and_launcher_mismatch():
    rc, lines, err = _run([sys.executable, "bench.py", "--dry-run"])
    assert rc == 0 and json.loads(lines[0])["n_gpus"] == 1
    rc, lines, err = _run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert rc != 0 and "started 1 ranks" in err


def test_under_torch_distributed_run():
    rc, lines, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                           "127.0.0.1", "--master-port", "29631", "bench.py", "--gpus", "2", "--dry-run"])
    assert rc == 0, err
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_a_failing_rank_fails_the_launch():
    # without --dry-run there is no GPU here: every rank exits non-zero, the parent must too (and must not hang)
    rc, lines, err = _run([sys.executable, "bench.py", "--gpus", "2", "--extras", ""])
    assert rc != 0 and not lines
