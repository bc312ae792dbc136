"""Multi-GPU sharding: one process per GPU, torch.distributed (RCCL on ROCm).  Restart axis first (below); when there are
fewer restarts than GPUs, ONE evaluation is sharded along the GP hyper-sample axis and the entropy's sample chunks
(ShardExchange + vbmc_amd.elbo.negelcbo_shard).

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


class ShardExchange:
    """The one exchange of a hyper-sample / sample-chunk sharded evaluation (vbmc_amd.elbo.negelcbo_shard): every rank
    contributes one block of n doubles of device memory, every rank receives the world blocks in rank order.  With the
    nccl backend (RCCL over xGMI) the blocks never leave the devices (all_gather_into_tensor on the device tensors); with
    gloo (CPU tests / ranks sharing one GPU) they are staged through the host.  Buffers are kept between calls."""

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.is_cuda = self.device.type == "cuda"    # a CPU device only for the gloo tests of the exchange itself
        self.world = dist.get_world_size(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self.send = self.recv = None

    def _sync(self):
        if self.is_cuda:
            self.torch.cuda.synchronize(self.device)

    def send_buffer(self, n):
        if self.send is None or self.send.numel() != n:
            self.send = self.torch.empty(n, dtype=self.torch.float64, device=self.device)
            self.recv = self.torch.empty(n * self.world, dtype=self.torch.float64, device=self.device)
            self._sync()
        return self.send.data_ptr()

    def all_gather(self):
        if self.on_device:
            self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        else:
            h = self.send.cpu()
            hr = self.torch.empty(h.numel() * self.world, dtype=self.torch.float64)
            self.dist.all_gather_into_tensor(hr, h, group=self.group)
            self.recv.copy_(hr)
        self._sync()   # the library reads recv on its own stream
        return self.recv.data_ptr()


def shard_spec(group=None, device=None):
    """(rank, world, allgather) for vbmc_amd.optimize.sieve_evaluate / vpsieve_vbmc."""
    import torch.distributed as dist

    return dist.get_rank(group), dist.get_world_size(group), make_allgather(group, device)
