"""Multi-GPU host side of the self-play path: one process per GPU, games sharded by index, no collective on
the data path.  What north_star names - the weight broadcast after a training epoch and the
gather of per-rank episode statistics - runs over RCCL THROUGH THE C ABI (mz_comm_init / mz_broadcast_weights /
mz_gather_stats, include/muzero_hip.h): `NativeComm`.  torch.distributed is used for one thing only, the
rendezvous that hands rank 0's 128-byte RCCL id to the other ranks (the c10d store torchrun already runs).
`HostComm` offers the same four operations over a torch.distributed process group (gloo): it is what the CPU
tests run (no RCCL without a GPU; tests/test_dist_gloo.py also drives NativeComm itself over a stand-in context whose
collectives are HostComm's).  bench.py never substitutes it: if RCCL cannot start, the run fails.
The reference itself is single-device (ModelController.java:121,136).
"""
import os

import numpy as np


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_range(total, rank, world_size):
    """Games [lo, hi) of rank: contiguous by game index so that outputs concatenate in order."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rendezvous_store(rank, world_size):
    """The key-value store of the launcher (env:// rendezvous: MASTER_ADDR / MASTER_PORT, or the agent's own store
    under torch.distributed.run).  No process group is created."""
    from torch.distributed import rendezvous
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    store, _, _ = next(rendezvous("env://", rank=rank, world_size=world_size))
    return store


def exchange_unique_id(store, rank, make_id, key="mz_rccl_unique_id"):
    """Rank 0 publishes make_id() (mz_comm_unique_id), everyone returns the same bytes."""
    if rank == 0:
        store.set(key, make_id())
    return bytes(store.get(key))


class NativeComm:
    """RCCL over xGMI behind the C ABI: the communicator lives in the mz_ctx."""
    kind = "rccl (C ABI: mz_comm_init / mz_broadcast_weights / mz_gather_stats)"

    def __init__(self, ctx, rank, world_size, store=None, make_id=None):
        """make_id: what rank 0 calls for the 128-byte id (default mz_comm_unique_id; the CPU tests pass a stand-in)."""
        if make_id is None:
            from . import capi
            make_id = capi.comm_unique_id
        self.ctx, self.rank, self.world = ctx, rank, world_size
        if world_size > 1:
            uid = exchange_unique_id(store if store is not None else rendezvous_store(rank, world_size), rank, make_id)
        else:
            uid = make_id()
        ctx.comm_init(uid, rank, world_size)

    def broadcast_weights(self, root=0):
        """In place on the context's fp32 master blob; non-root ranks repack (mz_commit_weights) inside."""
        self.ctx.broadcast_weights(root)

    def gather_stats(self, values):
        return self.ctx.gather_stats(values, self.world)

    def max_over_ranks(self, x):
        return float(self.gather_stats([float(x)]).max())

    def barrier(self):
        self.ctx.comm_barrier()

    def close(self):
        self.ctx.comm_destroy()


class HostComm:
    """The same operations over a torch.distributed process group (gloo on CPU; "nccl" = RCCL on a GPU box)."""

    def __init__(self, backend="gloo", device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.device = torch, dist, device
        w, r, _ = world()
        self.rank, self.world = r, w
        if w > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group(backend)
        self.kind = f"torch.distributed ({backend})"

    @property
    def active(self):
        return self.world > 1 and self.dist.is_initialized()

    def blob_tensor(self, spec, weights):
        """Rank-0 weights dict (None elsewhere) -> fp32 tensor ready for the broadcast."""
        from . import netspec
        t = self.torch.empty(netspec.num_params(spec), dtype=self.torch.float32, device=self.device)
        if weights is not None:
            t.copy_(self.torch.from_numpy(netspec.pack_blob(spec, weights)))
        return t

    def broadcast_weights(self, blob, root=0):
        if self.active:
            self.dist.broadcast(blob, src=root)
        return blob

    def gather_stats(self, values):
        v = self.torch.tensor(np.asarray(values, np.float64), dtype=self.torch.float64, device=self.device)
        if self.active:
            out = [self.torch.zeros_like(v) for _ in range(self.world)]
            self.dist.all_gather(out, v)
            return self.torch.stack(out).cpu().numpy()
        return v.cpu().numpy()[None]

    def max_over_ranks(self, x):
        return float(self.gather_stats([float(x)]).max())

    def barrier(self):
        if self.active:
            self.dist.barrier()

    def close(self):
        if self.active:
            self.dist.destroy_process_group()
