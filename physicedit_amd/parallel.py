"""Data-parallel sharding of the edit workload over the GPUs of one node (one process per GPU).

The hot path has NO data-path collective: images are independent, and inside
an image the two CFG branches only meet in the 512 KiB `noise_pred` combine of each step.  The
reference shards by hand with `--start_idx/--end_idx` processes
(scripts/inference/inference_pica.py:217-220,251-261).  Here:

  * `edit_batch`       the PRODUCT entry: a list of edit jobs (keyword dicts of the pipeline's `__call__`:
                       prompt, edit_image, seed, height, width, ...) through `QwenImagePhysicPipeline`,
                       jobs round-robin over the ranks (or over rank PAIRS with the CFG pair of an image
                       split inside the pair), ONE closing all-gather of the edited images (uint8) or of
                       the final latents, results in job order on every rank;
  * `shard_units`      static round-robin of work units (images) over ranks; weights are replicated
                       (41.5 GB << 288 GB HBM), seeds/noise belong to the unit, not to the rank, so any
                       world size produces the same images;
  * `gather_units`     ONE all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) of per-unit
                       result tensors, returned in unit order on every rank;
  * `CfgPairExchange`  latency mode for fewer images than GPUs: ranks (2k, 2k+1) run the posi / nega
                       forward of the SAME image and all-gather `noise_pred` inside the pair each step
                       (`DenoiseLoop(cfg_pair=...)` is the product code that uses it).

Backend-agnostic: torch.distributed with "nccl" (= RCCL on ROCm) for device tensors, "gloo" for CPU.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_units(n_units: int, rank: int, world: int) -> List[int]:
    """Unit ids owned by `rank`: u with u % world == rank (static round-robin)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_units, world))


def _gather_by_owner(local: Dict[int, torch.Tensor], ids: Sequence[int], owner: Callable[[int], int], shape, dtype, device,
                     group=None, always_collective: bool = False) -> Dict[int, torch.Tensor]:
    """ONE all-gather of the per-unit tensors `ids` (all of `shape` / `dtype`); unit u is contributed by rank owner(u)
    (every rank evaluates the same `owner`).  Returns {unit: tensor} for all `ids` on every rank.  A single-rank job skips the
    collective unless `always_collective` (and a process group exists): that is how a 1-GPU box executes the RCCL call."""
    rank, world = world_info(group)
    if world == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        return {u: local[u] for u in ids}
    mine = [u for u in ids if owner(u) == rank]
    per_rank = max(sum(1 for u in ids if owner(u) == r) for r in range(world))
    buf = torch.zeros((max(per_rank, 1),) + tuple(shape), dtype=dtype, device=device)
    for i, u in enumerate(mine):
        buf[i].copy_(local[u].reshape(shape))
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    res, seen = {}, [0] * world
    for u in ids:
        r = owner(u)
        res[u] = out[r][seen[r]]
        seen[r] += 1
    return res


def gather_units(local: Sequence[torch.Tensor], n_units: int, group=None, always_collective: bool = False) -> List[torch.Tensor]:
    """All-gather per-unit result tensors (same shape/dtype for every unit).  `local` holds this rank's
    units in `shard_units` order.  Returns all `n_units` tensors in unit order on every rank."""
    rank, world = world_info(group)
    if world == 1 and not (always_collective and dist.is_available() and dist.is_initialized()):
        assert len(local) == n_units
        return list(local)
    mine = shard_units(n_units, rank, world)
    assert len(local) == len(mine), (len(local), len(mine))
    # shape/dtype agreement: a rank without units still needs a template -> broadcast from rank 0's first unit
    meta = [None]
    if rank == 0:
        assert len(local) > 0
        meta = [(tuple(local[0].shape), local[0].dtype)]
    dist.broadcast_object_list(meta, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    shape, dtype = meta[0]
    dev = local[0].device if len(local) else _default_device()
    got = _gather_by_owner(dict(zip(mine, local)), list(range(n_units)), lambda u: u % world, shape, dtype, dev, group,
                           always_collective)
    return [got[u] for u in range(n_units)]


def _default_device():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def run_data_parallel(n_units: int, fn: Callable[[int], torch.Tensor], group=None) -> List[torch.Tensor]:
    """Run `fn(unit)` for this rank's units and return every unit's result, in unit order, on all ranks."""
    rank, world = world_info(group)
    local = [fn(u) for u in shard_units(n_units, rank, world)]
    return gather_units(local, n_units, group)


class CfgPairExchange:
    """Split one image's CFG pair over two ranks.  role 0 runs the positive forward, role 1 the negative;
    `exchange(pred)` all-gathers the two `noise_pred` tensors inside the pair so both ranks apply the same
    CFG combine + Euler update (qwen_image_physical.py:653-660) and keep identical latents."""

    def __init__(self, pair_group, role: int):
        if role not in (0, 1):
            raise ValueError("role must be 0 (posi) or 1 (nega)")
        self.group = pair_group
        self.role = role

    @staticmethod
    def make_pairs(group=None) -> "CfgPairExchange":
        """Collective over `group` (default: the world): every rank of it must call it.  Ranks (2k, 2k+1) OF THE GROUP form pair k;
        the pair groups are built from their global ranks, as dist.new_group wants them.  Creates world / 2 communicators: call it
        once and keep the result (edit_batch caches it on the pipe)."""
        rank, world = world_info(group)
        if world % 2:
            raise ValueError("CFG-pair split needs an even world size")
        to_global = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
        mine = None
        for k in range(world // 2):
            g = dist.new_group([to_global(2 * k), to_global(2 * k + 1)])       # every rank of the default group takes part in every call
            if rank // 2 == k:
                mine = g
        return CfgPairExchange(mine, rank % 2)

    def agree_on_seed(self, seed):
        """The two ranks of a pair must draw the SAME noise (and the same RandomCrop in training mode): a job without a seed would
        take it from each process's own global RNG, and two ranks that were handed different job lists would silently denoise
        different latents and combine mismatched predictions.  One small object all-gather inside the pair per image (beside 40
        per-step all-gathers of `noise_pred`) settles both: the even rank's seed decides (a fresh random one when it has none), and
        two DIFFERENT explicit seeds are an error on both ranks instead of a wrong image."""
        if self.group is None and dist.get_world_size() != 2:
            # the default group is a pair only in a 2-rank world: anywhere else the gather would span every rank of the job
            raise ValueError("CfgPairExchange without a pair group is only valid in a 2-rank world (use make_pairs)")
        mine = seed
        if mine is None and self.role == 0:
            mine = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        both = [None, None]
        dist.all_gather_object(both, (seed is not None, mine), group=self.group)
        (even_explicit, even_seed), (odd_explicit, odd_seed) = both
        if even_explicit and odd_explicit and even_seed != odd_seed:
            raise ValueError(f"the two ranks of a CFG pair hold different seeds for the same job ({even_seed} vs {odd_seed}): "
                             "every rank must pass the same jobs list")
        return even_seed

    def exchange(self, pred: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        both = [torch.empty_like(pred), torch.empty_like(pred)]
        dist.all_gather(both, pred.contiguous(), group=self.group)
        return both[0], both[1]


def edit_batch(pipe, jobs: Sequence[dict], group=None, split_cfg: bool = False, gather: str = "image",
               cfg_pair: Optional[CfgPairExchange] = None) -> List:
    """Edit a batch of images on all ranks of the job (the multi-GPU form of scripts/inference/inference_pica.py's
    `for idx in range(start_idx, end_idx): image = pipe(prompt, edit_image=..., seed=..., ...)`, :251-296, which the
    reference spreads over GPUs by starting one process per index range by hand, :217-220).

    `pipe`   a `diffsynth.pipelines.qwen_image_physical.QwenImagePhysicPipeline` (same weights on every rank);
    `jobs`   one keyword dict per image for `pipe.__call__` (every rank passes the SAME list); job j is unit j: its seed,
             prompt and image travel with it, so the edited images do not depend on the world size;
    `split_cfg`  False: job j runs on rank j % world.  True (even world size): ranks (2k, 2k+1) form pair k, job j runs on pair
             j % (world / 2), the positive forward on the even rank and the negative one on the odd rank with a per-step
             all-gather of `noise_pred` inside the pair (`CfgPairExchange`; pass a `cfg_pair` made earlier to reuse its groups);
             halves the latency of an image when there are fewer images than GPUs;
    `gather` "image": ONE all-gather of the decoded uint8 images closes the batch -> list of PIL images in job order on every
             rank;  "latents": the all-gather moves the final latents instead (512 KiB instead of 3 MiB per 1024^2 image, the
             collective BASELINE.json's north star names) -> list of [1,16,H/8,W/8] tensors;  "none": no collective, this
             rank's own results only (the reference's behaviour: every process saves its own files) -> {job index: PIL image}.
    Jobs whose results differ in shape are gathered per shape (one collective per distinct size)."""
    from PIL import Image
    rank, world = world_info(group)
    if gather not in ("image", "latents", "none"):
        raise ValueError("gather must be 'image', 'latents' or 'none'")
    n = len(jobs)
    if split_cfg and world > 1:
        if world % 2:
            raise ValueError("split_cfg needs an even world size")
        if cfg_pair is None:
            # one set of pair communicators per (pipe, group), not one per call
            cache = getattr(pipe, "_cfg_pair_cache", None)
            if cache is None or cache[0] is not group:
                cache = (group, CfgPairExchange.make_pairs(group))
                try:
                    pipe._cfg_pair_cache = cache
                except AttributeError:
                    pass
            cfg_pair = cache[1]
        lanes, lane = world // 2, rank // 2
        owner = lambda u: 2 * (u % lanes)            # the pair's even rank contributes the (identical) result
    else:
        cfg_pair = None
        lanes, lane = world, rank
        owner = lambda u: u % lanes
    mine = [u for u in range(n) if u % lanes == lane]
    prev_pair = getattr(pipe, "cfg_pair", None)
    pipe.cfg_pair = cfg_pair
    images: Dict[int, torch.Tensor] = {}
    latents: Dict[int, torch.Tensor] = {}
    pil: Dict[int, "Image.Image"] = {}
    try:
        for u in mine:
            job = jobs[u]
            if cfg_pair is not None:
                # both ranks of the pair must start from identical latents: an unseeded job gets the even rank's seed
                job = dict(job, seed=cfg_pair.agree_on_seed(job.get("seed")))
            img = pipe(**job)
            pil[u] = img
            if gather == "latents":
                latents[u] = pipe.last_latents.detach().clone()
            elif gather == "image":
                import numpy as np
                images[u] = torch.from_numpy(np.asarray(img).copy()).to(pipe.device)
    finally:
        pipe.cfg_pair = prev_pair
    if gather == "none":
        return pil
    if world == 1:
        return [pil[u] for u in range(n)] if gather == "image" else [latents[u] for u in range(n)]
    # result shapes are a function of the job alone (ShapeChecker, :673-680): every rank groups the units the same way
    def out_hw(job):
        h, w = pipe.check_resize_height_width(job.get("height", 1328), job.get("width", 1328))
        return int(h), int(w)
    classes: Dict[Tuple[int, int], List[int]] = {}
    for u in range(n):
        classes.setdefault(out_hw(jobs[u]), []).append(u)
    got: Dict[int, torch.Tensor] = {}
    for (h, w), ids in sorted(classes.items()):
        if gather == "image":
            got.update(_gather_by_owner(images, ids, owner, (h, w, 3), torch.uint8, pipe.device, group))
        else:
            got.update(_gather_by_owner(latents, ids, owner, (1, 16, h // 8, w // 8), pipe.torch_dtype, pipe.device, group))
    if gather == "latents":
        return [got[u] for u in range(n)]
    return [pil[u] if u in pil else Image.fromarray(got[u].cpu().numpy()) for u in range(n)]
