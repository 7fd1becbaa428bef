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


# ------------------------------------------------------------------------------------------------
# product entry: parallel.edit_batch through a pipeline object, DenoiseLoop(cfg_pair=...)
# ------------------------------------------------------------------------------------------------
class OracleEngine:
    """CPU stand-in for physicedit_amd.dit.QwenImageDiTEngine (same methods DenoiseLoop calls), the oracle as the compute."""
    device = torch.device("cpu")
    version, fp8 = 0, False

    def __init__(self):
        self.sd = synth.make_state_dict(synth.dit_layout(0), 1234)
        self._eligen_words = None
        self.forwards = 0

    def bind(self, *a):
        pass

    def prepare(self, ts):
        pass

    def forward(self, latents, t, prompt_emb, idx, edits, step=None, out=None, controls=None, **kw):
        self.forwards += 1
        h8, w8 = latents.shape[-2:]
        pred = O.model_fn(self.sd, None, latents, t, prompt_emb, None, h8 * 8, w8 * 8)
        if out is not None:
            out.copy_(pred)
            return out
        return pred


def _cpu_cfg_euler(posi, nega, latents, cfg_scale, dsigma, out=None, **kw):
    """stand-in for ops.cfg_euler_step (a HIP kernel): any deterministic function of the same inputs serves the plumbing tests"""
    pred = posi if nega is None else (nega.float() + cfg_scale * (posi.float() - nega.float())).to(BF)
    res = (latents.float() + pred.float() * dsigma).to(BF)
    if out is not None:
        out.copy_(res)
        return out
    return res


class StubPipe:
    """The attributes / methods of QwenImagePhysicPipeline that parallel.edit_batch touches, over the REAL DenoiseLoop."""

    def __init__(self):
        import physicedit_amd.pipeline as PL
        PL.ops.cfg_euler_step = _cpu_cfg_euler
        self.PL = PL
        self.device, self.torch_dtype, self.cfg_pair = torch.device("cpu"), BF, None
        self.dit = OracleEngine()
        self.last_latents = None

    def check_resize_height_width(self, h, w):
        return (h + 15) // 16 * 16, (w + 15) // 16 * 16

    def __call__(self, prompt, seed=0, height=64, width=64, num_inference_steps=2, cfg_scale=4.0, **kw):
        from PIL import Image
        height, width = self.check_resize_height_width(height, width)
        loop = self.PL.DenoiseLoop(self.dit, cfg_pair=self.cfg_pair)
        lat = loop(synth.make_noise(seed, height, width), synth.make_prompt_emb(len(prompt), 16), synth.make_prompt_emb(8, 12),
                   None, None, height, width, num_inference_steps=num_inference_steps, cfg_scale=cfg_scale)
        self.last_latents = lat
        u8 = ((lat.float()[0, :3].clamp(-2, 2) + 2) * 63.75).to(torch.uint8).permute(1, 2, 0)
        u8 = u8.repeat_interleave(8, 0).repeat_interleave(8, 1).contiguous()
        return Image.fromarray(u8.numpy())


JOBS = [dict(prompt="a" * (3 + i), seed=10 + i, height=64, width=64 if i % 3 else 96) for i in range(5)]
# BASELINE configs[3]'s literal shape: 16 images (CFG pairs) for 8 ranks; and fewer images than ranks, split over CFG pairs
JOBS16 = [dict(prompt="b" * (2 + i % 5), seed=100 + i, height=64, width=64) for i in range(16)]
JOBS4 = [dict(prompt="c" * (2 + i), seed=200 + i, height=64, width=64) for i in range(4)]
JOB_SETS = {"5": JOBS, "16": JOBS16, "4": JOBS4}


def _worker_edit(rank, world, port, split, gather, q, jobs_key="5"):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1 if world > 4 else 2)
    pipe = StubPipe()
    jobs = JOB_SETS[jobs_key]
    res = parallel.edit_batch(pipe, jobs, split_cfg=split, gather=gather)
    if split:       # a second batch on the same pipe reuses the pair communicators (one make_pairs per pipe and group)
        first = getattr(pipe, "_cfg_pair_cache", None)
        parallel.edit_batch(pipe, jobs[:world // 2], split_cfg=True, gather="none")
        assert first is not None and pipe._cfg_pair_cache is first
    out = [np.asarray(r) if gather == "image" else r.float().numpy() for r in res]
    q.put((rank, out, pipe.dit.forwards))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,split,gather,jobs_key", [(2, False, "image", "5"), (2, True, "latents", "5"), (4, True, "image", "5"),
                                                         (8, False, "latents", "16"), (8, True, "latents", "4")])
def test_edit_batch_matches_single_process(world, split, gather, jobs_key):
    """also at world 8: configs[3]'s literal shape (16 jobs, 2 per rank, one closing all-gather of the latents) and 4 jobs on 4 CFG
    pairs (split_cfg: every rank runs ONE forward per step of its pair's image)"""
    import numpy as np
    JOBS = JOB_SETS[jobs_key]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_edit, args=(r, world, port, split, gather, q, jobs_key)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (o, f) for r, o, f in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pipe = StubPipe()
    single = parallel.edit_batch(pipe, JOBS, gather=gather)        # no process group: this process edits every job
    single = [np.asarray(r) if gather == "image" else r.float().numpy() for r in single]
    for rank in range(world):
        assert len(got[rank][0]) == len(JOBS)
        for u in range(len(JOBS)):
            assert np.array_equal(got[rank][0][u], single[u]), (rank, u)        # job order, bit-identical, any world size
    # work really was divided: 2 steps x 2 forwards per job; a split pair runs ONE forward per step and rank (+ the second, ungathered
    # batch of world / 2 jobs the split workers run to check the communicator cache: one more job per pair)
    lanes = world // 2 if split else world
    for rank in range(world):
        lane = rank // 2 if split else rank
        n_mine = len([u for u in range(len(JOBS)) if u % lanes == lane]) + (1 if split else 0)
        assert got[rank][1] == n_mine * 2 * (1 if split else 2), (rank, got[rank][1])


def _worker_unseeded_pair(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(1000 + rank)               # the two processes' global RNGs disagree, as they do in real life
    pipe = StubPipe()
    seen = []
    orig = StubPipe.__call__
    pipe_call = lambda **kw: (seen.append(kw.get("seed")), orig(pipe, **kw))[1]
    pipe.__class__ = type("RecPipe", (StubPipe,), {"__call__": lambda self, **kw: pipe_call(**kw)})
    res = parallel.edit_batch(pipe, [dict(prompt="abc", seed=None, height=64, width=64)], split_cfg=True, gather="latents")
    q.put((rank, seen, res[0].float().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_split_cfg_unseeded_job_gets_one_seed_per_pair():
    """a job with seed=None (the pipeline default) under split_cfg would draw its noise from each process's own global
    RNG and the pair would silently combine forwards of different latents.  The pair's even rank now decides the seed."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_unseeded_pair, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (s_, lat) for r, s_, lat in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0] and got[0][0][0] is not None          # both ranks ran the job with the SAME, concrete seed
    assert np.array_equal(got[0][1], got[1][1])


def _worker_mismatched_pair(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = parallel.CfgPairExchange.make_pairs()
    out = []
    # (what each rank holds for the same job, what the pair must settle on)
    for mine in ((7, 7), (None, 5), (9, None), (3, 4)):
        try:
            out.append(cfg.agree_on_seed(mine[rank]))
        except ValueError as e:
            out.append("error: " + str(e)[:40])
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_pair_seed_agreement_is_verified():
    """with an explicit seed agree_on_seed returned early and trusted both ranks to hold the same jobs list; two ranks with
    different seeds denoised different latents and combined mismatched predictions, and a (None, seed) pair dead-locked.  Now one
    object all-gather inside the pair per image: equal seeds pass, a missing one takes the even rank's, different ones raise on BOTH."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mismatched_pair, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    assert got[0][0] == 7 and got[0][2] == 9 and isinstance(got[0][1], int) and got[0][3].startswith("error")


def _worker_loop_pair(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import physicedit_amd.pipeline as PL
    PL.ops.cfg_euler_step = _cpu_cfg_euler
    ex = parallel.CfgPairExchange.make_pairs()
    eng = OracleEngine()
    loop = PL.DenoiseLoop(eng, cfg_pair=ex)
    lat = loop(synth.make_noise(3, 64, 64), synth.make_prompt_emb(7, 16), synth.make_prompt_emb(8, 12), None, None, 64, 64,
               num_inference_steps=3, cfg_scale=4.0)
    lat1 = loop(synth.make_noise(3, 64, 64), synth.make_prompt_emb(7, 16), synth.make_prompt_emb(8, 12), None, None, 64, 64,
                num_inference_steps=2, cfg_scale=1.0)            # CFG off: the pair is ignored, every rank runs the one forward
    q.put((rank, lat.float().numpy(), lat1.float().numpy(), eng.forwards))
    dist.barrier()
    dist.destroy_process_group()


def test_denoise_loop_cfg_pair_is_product_code():
    """DenoiseLoop(cfg_pair=CfgPairExchange): each rank of a pair runs ONE forward per step and both end with the latents of the
    unsplit loop."""
    import physicedit_amd.pipeline as PL
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_loop_pair, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (a, b, f) for r, a, b, f in (q.get(timeout=300) for _ in range(2))}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    PL.ops.cfg_euler_step = _cpu_cfg_euler
    eng = OracleEngine()
    loop = PL.DenoiseLoop(eng)
    ref = loop(synth.make_noise(3, 64, 64), synth.make_prompt_emb(7, 16), synth.make_prompt_emb(8, 12), None, None, 64, 64,
               num_inference_steps=3, cfg_scale=4.0).float()
    ref1 = loop(synth.make_noise(3, 64, 64), synth.make_prompt_emb(7, 16), synth.make_prompt_emb(8, 12), None, None, 64, 64,
                num_inference_steps=2, cfg_scale=1.0).float()
    for r in (0, 1):
        assert torch.equal(torch.from_numpy(got[r][0]), ref) and torch.equal(torch.from_numpy(got[r][1]), ref1)
        assert got[r][2] == 3 + 2          # one forward per step with the pair, one per step without CFG
