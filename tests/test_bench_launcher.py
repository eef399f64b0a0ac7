"""bench.py never reports GPUs that are not there: `--gpus N` without a launcher starts N ranks itself (and fails when it
cannot get N devices), a launcher that started another number of ranks is refused, and the JSON line carries what the
collective really saw (`ranks_seen`, `devices`).  CPU: this container has no GPU, so every N > 1 run must end non-zero
without printing a line."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_gpus_2_without_a_launcher_cannot_print_n_gpus_2_from_one_process():
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu"], {})
    assert p.returncode != 0
    assert b'"n_gpus"' not in p.stdout


def test_a_launcher_with_another_rank_count_is_refused():
    p = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and b"refusing" in p.stderr and b'"n_gpus"' not in p.stdout
