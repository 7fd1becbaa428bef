"""Data-parallel sharding of the edit workload over the GPUs of one node (one process per GPU).

The hot path has NO data-path collective: images are independent, and inside
an image the two CFG branches only meet in the 512 KiB `noise_pred` combine of each step.  The
reference shards by hand with `--start_idx/--end_idx` processes
(scripts/inference/inference_pica.py:217-220,251-261).  Here:

  * `shard_units`      static round-robin of work units (images) over ranks; weights are replicated
                       (41.5 GB << 288 GB HBM), seeds/noise belong to the unit, not to the rank, so any
                       world size produces the same images;
  * `gather_units`     ONE all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) of the decoded
                       latents / images at the end of the batch, returned in unit order on every rank;
  * `CfgPairExchange`  optional latency mode for fewer images than GPUs: ranks (2k, 2k+1) run the posi /
                       nega forward of the SAME image and all-gather `noise_pred` inside the pair each step.

Backend-agnostic: torch.distributed with "nccl" (= RCCL on ROCm) for device tensors, "gloo" for CPU.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

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


def gather_units(local: Sequence[torch.Tensor], n_units: int, group=None) -> List[torch.Tensor]:
    """All-gather per-unit result tensors (same shape/dtype for every unit).  `local` holds this rank's
    units in `shard_units` order.  Returns all `n_units` tensors in unit order on every rank."""
    rank, world = world_info(group)
    if world == 1:
        assert len(local) == n_units
        return list(local)
    mine = shard_units(n_units, rank, world)
    assert len(local) == len(mine), (len(local), len(mine))
    per_rank = (n_units + world - 1) // world
    # shape/dtype agreement: a rank without units still needs a template -> broadcast from rank 0's first unit
    meta = [None]
    if rank == 0:
        assert len(local) > 0
        meta = [(tuple(local[0].shape), local[0].dtype)]
    dist.broadcast_object_list(meta, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    shape, dtype = meta[0]
    dev = local[0].device if len(local) else _default_device()
    buf = torch.zeros((per_rank,) + tuple(shape), dtype=dtype, device=dev)
    for i, t in enumerate(local):
        buf[i].copy_(t.reshape(shape))
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return [out[u % world][u // world] for u in range(n_units)]


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
    def make_pairs() -> "CfgPairExchange":
        """Collective: every rank must call it.  Ranks (2k, 2k+1) form pair k."""
        rank, world = world_info()
        if world % 2:
            raise ValueError("CFG-pair split needs an even world size")
        mine = None
        for k in range(world // 2):
            g = dist.new_group([2 * k, 2 * k + 1])
            if rank // 2 == k:
                mine = g
        return CfgPairExchange(mine, rank % 2)

    def exchange(self, pred: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        both = [torch.empty_like(pred), torch.empty_like(pred)]
        dist.all_gather(both, pred.contiguous(), group=self.group)
        return both[0], both[1]
