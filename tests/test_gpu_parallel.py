"""`-m gpu` multi-rank test of the data-parallel host path ON THE HIP KERNELS over RCCL (backend "nccl"):
2 ranks x tiny images through parallel.run_data_parallel and the CFG-pair exchange, bit-for-bit against a
single-process run.  Skipped on boxes with fewer than 2 GPUs (the gloo tests cover the plumbing on CPU).
Reference: scripts/inference/inference_pica.py:217-220,251-261 (manual --start_idx/--end_idx sharding)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
H = W = 64
T_P, T_N, NSP, STEPS = 24, 16, 8, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _engine(dev):
    from physicedit_amd import synth
    from physicedit_amd.dit import QwenImageDiTEngine
    sd = synth.make_state_dict(synth.dit_layout(1), 1234)
    ad = synth.make_state_dict(synth.adapter_layout(), 4321)
    return QwenImageDiTEngine(sd, ad, device=dev)


def _prompts():
    from physicedit_amd import synth
    return (synth.make_prompt_emb(7, T_P), synth.make_prompt_emb(8, T_N),
            synth.make_special_token_mask(T_P, NSP), synth.make_special_token_mask(T_N, NSP))


def _edit_unit(loop, dev, u):
    """One work unit = one edited image: noise seed is the UNIT id, so any world size edits the same images."""
    from physicedit_amd import synth
    pe_p, pe_n, m_p, m_n = _prompts()
    return loop(synth.make_noise(100 + u, H, W), pe_p.to(dev), pe_n.to(dev), m_p, m_n, H, W, num_inference_steps=STEPS,
                cfg_scale=4.0).clone()


def _worker(rank, world, port, n_units, mode, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from physicedit_amd import parallel
    from physicedit_amd.pipeline import DenoiseLoop
    eng = _engine(dev)
    if mode == "dp":
        loop = DenoiseLoop(eng)
        res = parallel.run_data_parallel(n_units, lambda u: _edit_unit(loop, dev, u))
        torch.cuda.synchronize()
        q.put((rank, [r.float().cpu().numpy() for r in res]))
    else:   # CFG pair split through the PRODUCT loop: rank 0 runs the positive forward, rank 1 the negative one, per-step
        #         all-gather of noise_pred inside DenoiseLoop (pipeline.py)
        ex = parallel.CfgPairExchange.make_pairs()
        lat = _edit_unit(DenoiseLoop(eng, cfg_pair=ex), dev, 0)
        torch.cuda.synchronize()
        q.put((rank, lat.float().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(mode, n_units):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_units, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return got


def _need_two_gpus():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")


@pytest.mark.parametrize("n_units", [2, 3])
def test_rccl_data_parallel_matches_single_rank(n_units):
    _need_two_gpus()
    from physicedit_amd.pipeline import DenoiseLoop
    got = _spawn("dp", n_units)
    dev = torch.device("cuda", 0)
    loop = DenoiseLoop(_engine(dev))
    single = [_edit_unit(loop, dev, u).float().cpu() for u in range(n_units)]
    for rank in (0, 1):
        assert len(got[rank]) == n_units
        for u in range(n_units):
            assert torch.equal(torch.from_numpy(got[rank][u]), single[u]), (rank, u)


def test_rccl_cfg_pair_split_matches_single_rank():
    _need_two_gpus()
    from physicedit_amd.pipeline import DenoiseLoop
    got = _spawn("cfg", 1)
    dev = torch.device("cuda", 0)
    ref = _edit_unit(DenoiseLoop(_engine(dev)), dev, 0).float().cpu()
    assert torch.equal(torch.from_numpy(got[0]), ref) and torch.equal(torch.from_numpy(got[1]), ref)


def test_rccl_path_executes_at_world_1():
    """The driver's N > 1 scaling runs must not be the first execution of the RCCL path.  On a 1-GPU box everything but the rank
    count can run: `bench.py --gpus 1 --force-dist` under torch.distributed.run initialises the nccl (= RCCL) process group,
    closes the batch with the all-gather of the final latents, and takes the max-over-ranks time through barrier + all-reduce.
    Reduced model (1 layer, 2 steps): the collectives and the launcher are what is under test."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--layers", "1",
           "--steps", "1", "--warmup", "0", "--inference-steps", "2", "--no-cpu-baseline", "--no-secondary"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["config"]["rccl_ranks"] == 1
    assert "all_gather" in out["config"]["batch_closing_collective"]
    assert len(out["config"]["per_rank_elapsed_s"]) == 1 and out["config"]["finite_outputs"] is True
    assert out["value"] > 0
