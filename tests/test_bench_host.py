"""bench.py's launcher logic on a machine without GPUs: `--gpus N` without a launcher becomes the launcher (one rank per GPU under
torch.distributed.run on 127.0.0.1, the driver's own command line), a WORLD_SIZE that disagrees with --gpus is refused, and ranks
refuse to start on a node with fewer GPUs than ranks -- each before any CUDA call."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2", "--no-cpu-baseline"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=4 but --gpus 2" in (r.stderr + r.stdout)


def test_gpus_n_spawns_n_ranks_which_check_the_gpu_count():
    r = _run(["--gpus", "2", "--no-cpu-baseline"])
    out = r.stderr + r.stdout
    assert "[bench] spawning 2 ranks" in out and "torch.distributed.run" in out and "--master-addr 127.0.0.1" in out
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "--gpus 2 but only" in out          # printed by the spawned ranks, not by the launcher
