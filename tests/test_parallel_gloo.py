"""Multi-process CPU tests (gloo, world_size 2) of the N>1 host path: unit sharding + final all-gather,
and the CFG-pair exchange, checked against a single-process run of the same units (oracle as the
compute stand-in: these tests exercise the distributed plumbing, not the HIP kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle.physicedit_oracle as O
from physicedit_amd import parallel, synth

BF = torch.bfloat16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unit_result(u: int) -> torch.Tensor:
    """One 'edited image' of the plumbing test: 2 Euler steps of a 0-layer DiT on unit-seeded noise."""
    sd = synth.make_state_dict(synth.dit_layout(0), 1234)
    noise = synth.make_noise(100 + u, 64, 64)
    pe = synth.make_prompt_emb(7, 16)
    return O.denoise_loop(sd, None, noise, pe, None, None, None, 64, 64, 2, cfg_scale=1.0)


def _worker_dp(rank, world, port, n_units, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    res = parallel.run_data_parallel(n_units, _unit_result)
    # numpy, not torch tensors: a tensor travels through mp.Queue as a shared-memory handle that dies with this
    # process, which races with the parent's q.get(); an ndarray is pickled by value
    q.put((rank, [r.float().numpy() for r in res]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_units", [3, 4, 1])
def test_data_parallel_gather_matches_single_process(n_units):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dp, args=(r, world, port, n_units, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = [_unit_result(u).float() for u in range(n_units)]
    for rank in range(world):
        assert len(got[rank]) == n_units
        for u in range(n_units):
            assert torch.equal(torch.from_numpy(got[rank][u]), single[u]), (rank, u)   # same seeds, same math: bit-identical


def test_shard_units():
    assert parallel.shard_units(16, 3, 8) == [3, 11]
    assert parallel.shard_units(3, 2, 8) == [2]
    assert parallel.shard_units(3, 5, 8) == []
    allu = sorted(u for r in range(8) for u in parallel.shard_units(16, r, 8))
    assert allu == list(range(16))
    with pytest.raises(ValueError):
        parallel.shard_units(4, 2, 2)


def _worker_cfg(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    ex = parallel.CfgPairExchange.make_pairs()
    sd = synth.make_state_dict(synth.dit_layout(0), 1234)
    noise = synth.make_noise(0, 64, 64)
    pe = synth.make_prompt_emb(7 + ex.role, 16 if ex.role == 0 else 12)      # role 0 = posi prompt, 1 = nega
    tab = O.FlowMatchTables(2, dynamic_shift_len=16)
    lat = noise.clone()
    for i, t in enumerate(tab.timesteps):
        pred = O.model_fn(sd, None, lat, t.unsqueeze(0).to(BF), pe, None, 64, 64)
        posi, nega = ex.exchange(pred)
        lat = tab.step(nega + 4.0 * (posi - nega), i, lat)
    q.put((rank, lat.float().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_pair_split_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cfg, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sd = synth.make_state_dict(synth.dit_layout(0), 1234)
    ref = O.denoise_loop(sd, None, synth.make_noise(0, 64, 64), synth.make_prompt_emb(7, 16), synth.make_prompt_emb(8, 12),
                         None, None, 64, 64, 2, cfg_scale=4.0).float()
    assert torch.equal(torch.from_numpy(got[0]), ref) and torch.equal(torch.from_numpy(got[1]), ref)
