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


def test_gpus_8_launcher_command_and_thread_cap():
    """BASELINE configs[3] / the driver's SCALE run: `--gpus 8` becomes `torch.distributed.run --nnodes=1 --nproc-per-node=8` on
    127.0.0.1 with the caller's flags passed through, and every rank caps its host threads at its share of the node."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert bench.host_threads_per_rank(8, cpus=256) == 32 and bench.host_threads_per_rank(8, cpus=64) == 8
    assert bench.host_threads_per_rank(8, cpus=4) == 1 and bench.host_threads_per_rank(1, cpus=256) == 32
    r = _run(["--gpus", "8", "--no-cpu-baseline"])
    out = r.stderr + r.stdout
    assert "[bench] spawning 8 ranks" in out and "--nproc-per-node=8" in out


def test_executed_flops_of_the_trimmed_last_block():
    """`whole_path.executed_pflop_per_image` (round 6): the last block launches only the S0 noise rows' post-attention work, so per forward
    (S - S0) rows skip out-proj + MLP (169,869,312 FLOPs per row) and their attention queries (12,288 S per row).  At the headline geometry
    that is 0.72 % of the algorithmic count, and it must equal what the FLOP model says those launches were worth."""
    import bench
    alg = bench.flops_image(1024, 1024, 40, 512, 272, 4.0)
    trim = bench.flops_trimmed_last_block(1024, 1024, 40, 512, 272, 4.0)
    assert 0.0070 < trim / alg < 0.0075
    # per-row terms against flops_forward's own coefficients: 226,492,416 per row = QKV 56,623,104 + the rest; 12,288 S^2 = 4 S^2 * 3072
    assert 226_492_416 - 2 * 3072 * 9216 == 169_869_312
    S0, S = 4096, 8192 + 512
    one = bench.flops_trimmed_last_block(1024, 1024, 1, 512, 272, 1.0)
    assert one == (S - S0) * (169_869_312 + 12_288 * S)
    assert bench.flops_trimmed_last_block(1024, 1024, 40, 512, 272, 1.0) == 40 * one      # CFG off: one forward per step
