"""Multi-GPU plumbing of the self-play path: one process per GPU, games sharded by index, no data-path
collective.  RCCL (torch.distributed backend "nccl") is used for exactly two things, as BASELINE.json's
north_star states: broadcasting the weight blob after a training epoch and gathering per-rank episode
statistics.  The reference itself is single-device; the same code runs on gloo for the
CPU tests.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend, device=None):
    """Join the process group when launched by torch.distributed.run; no-op for a single process."""
    w, _, _ = world()
    if w > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return w > 1


def shard_range(total, rank, world_size):
    """Games [lo, hi) of rank: contiguous by game index so that outputs concatenate in order."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(blob, src=0, device=None):
    """blob: fp32 tensor (same numel on every rank; contents only matter on `src`).  In place."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def blob_tensor(spec, weights, device):
    """Rank-0 weights dict (or None on other ranks) -> fp32 tensor on `device` ready for broadcast."""
    from . import netspec
    t = torch.empty(netspec.num_params(spec), dtype=torch.float32, device=device)
    if weights is not None:
        t.copy_(torch.from_numpy(netspec.pack_blob(spec, weights)))
    return t


def gather_stats(values, device=None):
    """values: list/array of per-rank numbers -> [world, len] float64 array on every rank."""
    v = torch.tensor(np.asarray(values, np.float64), dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(out, v)
        return torch.stack(out).cpu().numpy()
    return v.cpu().numpy()[None]


def max_over_ranks(x, device=None):
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
