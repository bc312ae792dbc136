"""Multi-GPU sharding of the restart axis: one process per GPU, torch.distributed (RCCL on ROCm).

The sieve's R candidates (misc/vpsieve_vbmc.m:74-78) are independent, so rank g evaluates the
candidates i = g (mod G) on its own GPU against a full replica of the GP (25.6 MB at the headline
shape) and the only exchange is an all-gather of the R ELCBO values -- 8*R bytes, latency-bound
on xGMI.  All-gather (not all-reduce) so that every rank holds the identical vector and performs
the identical stable sort: the sieve order is index-identical on all ranks and to the 1-GPU run.
"""
from __future__ import annotations

import numpy as np


def make_allgather(group=None, device=None):
    """Returns allgather(local_values, local_indices, R) -> full length-R vector on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)

    def allgather(local, idx, R):
        per = (R + world - 1) // world
        buf = torch.full((per,), float("nan"), dtype=torch.float64, device=device)
        if len(local):
            buf[: len(local)] = torch.as_tensor(np.asarray(local, dtype=np.float64), device=device)
        out = torch.empty(per * world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(out, buf, group=group)
        out = out.cpu().numpy().reshape(world, per)
        full = np.full(R, np.nan)
        for g in range(world):
            ids = np.arange(R)[g::world]
            full[ids] = out[g, : len(ids)]
        return full

    return allgather


def shard_spec(group=None, device=None):
    """(rank, world, allgather) for vbmc_amd.optimize.sieve_evaluate / vpsieve_vbmc."""
    import torch.distributed as dist

    return dist.get_rank(group), dist.get_world_size(group), make_allgather(group, device)
