"""world_size-2 tests of the N>1 host path (run on CPU): contiguous game sharding, weight-blob broadcast from
rank 0, gather of per-rank episode statistics, max-over-ranks timing over gloo (HostComm: the same four
operations NativeComm runs over RCCL through the C ABI on a GPU box), and the rendezvous that hands rank 0's
RCCL id to the other ranks."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from muzero_amd import dist as mzd
from muzero_amd import netspec


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _guarded(body, rank, q, *args):
    """Worker entry: an exception reaches the parent as text at once instead of as a 120 s queue timeout."""
    try:
        body(rank, *args, q)
    except BaseException:
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise


def _run_ranks(body, world, nports, *args):
    """Spawn `world` ranks of body(rank, world, *ports, *args, q) and return their sorted results.  The ports come from
    bind(0) and are released before the ranks bind them again, so another process can take one in between: a rendezvous
    that fails that way (address in use / nobody answered) is tried once more on fresh ports; anything else fails."""
    last = None
    for attempt in range(2):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ports = [_free_port() for _ in range(nports)]
        procs = [ctx.Process(target=_guarded, args=(body, r, q, world, *ports, *args)) for r in range(world)]
        for p in procs:
            p.start()
        res, err = [], None
        try:
            for _ in range(world):
                item = q.get(timeout=120)
                if item and item[0] == "error":
                    err = item[2]
                    break
                res.append(item)
        except Exception as e:   # queue.Empty: a rank never answered
            err = "no answer from a rank within 120 s: %r" % (e,)
        for p in procs:
            p.join(timeout=10 if err else 60)
            if p.is_alive():
                p.kill()
                p.join()
        if err is None:
            assert all(p.exitcode == 0 for p in procs)
            return sorted(res)
        last = err
        if not any(t in err for t in ("Address already in use", "EADDRINUSE", "no answer from a rank", "Connection refused",
                                      "Connection reset")):
            break
    raise AssertionError("rank failed:\n" + last)


def _worker(rank, world, port, total_games, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    comm = mzd.HostComm("gloo", torch.device("cpu"))
    assert comm.active and comm.rank == rank and comm.world == world
    spec = netspec.spec_tiny()
    weights = netspec.random_weights(spec, seed=7) if rank == 0 else None
    blob = comm.blob_tensor(spec, weights)
    if rank != 0:
        blob.fill_(-1.0)
    comm.broadcast_weights(blob, root=0)
    lo, hi = mzd.shard_range(total_games, rank, world)
    sims = (hi - lo) * 20                                 # every game of the shard runs n = 20 simulations
    stats = comm.gather_stats([hi - lo, sims, float(blob.double().sum())])
    tmax = comm.max_over_ranks(1.0 + rank)
    comm.barrier()
    q.put((rank, lo, hi, stats.tolist(), tmax, float(blob[:5].sum())))
    comm.close()


def test_two_rank_sharding_broadcast_and_gather():
    world, total = 2, 101
    res = _run_ranks(_worker, world, 1, total)
    (r0, lo0, hi0, st0, t0, b0), (r1, lo1, hi1, st1, t1, b1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 51, 51, 101)       # contiguous, complete, no overlap
    assert st0 == st1                                     # every rank sees the same gathered table
    st = np.array(st0)
    assert st[:, 0].sum() == total and st[:, 1].sum() == total * 20
    assert st[0, 2] == st[1, 2] and b0 == b1              # identical weights after the broadcast
    ref = netspec.pack_blob(netspec.spec_tiny(), netspec.random_weights(netspec.spec_tiny(), seed=7))
    assert abs(st[0, 2] - ref.astype(np.float64).sum()) < 1e-3
    assert t0 == t1 == 2.0                                # MAX over ranks


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 4096, 4099):
        for w in (1, 2, 4, 8):
            parts = [mzd.shard_range(total, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _id_worker(rank, world, port, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    store = mzd.rendezvous_store(rank, world)
    uid = mzd.exchange_unique_id(store, rank, lambda: bytes(range(128)))   # stand-in for mz_comm_unique_id
    q.put((rank, uid))


def test_unique_id_rendezvous_without_a_process_group():
    res = _run_ranks(_id_worker, 2, 1)
    assert res[0][1] == res[1][1] == bytes(range(128))


class _StandInCtx:
    """The mz_ctx methods NativeComm calls (capi.Context: comm_init / broadcast_weights / gather_stats / comm_barrier /
    comm_destroy), carried by HostComm's gloo collectives instead of RCCL, with a log of the calls."""

    def __init__(self, host, blob):
        self.host, self.blob, self.calls, self.uid = host, blob, [], None

    def comm_init(self, uid, rank, world):
        self.calls.append(("comm_init", rank, world))
        self.uid = bytes(uid)

    def broadcast_weights(self, root):
        self.calls.append(("broadcast_weights", root))
        self.host.broadcast_weights(self.blob, root)

    def gather_stats(self, values, world):
        self.calls.append(("gather_stats", len(values), world))
        return self.host.gather_stats(values)

    def comm_barrier(self):
        self.calls.append(("comm_barrier",))
        self.host.barrier()

    def comm_destroy(self):
        self.calls.append(("comm_destroy",))


def _native_worker(rank, world, port, port2, total_games, q):
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    host = mzd.HostComm("gloo", torch.device("cpu"))
    spec = netspec.spec_tiny()
    weights = netspec.random_weights(spec, seed=7) if rank == 0 else None
    blob = host.blob_tensor(spec, weights)
    if rank != 0:
        blob.fill_(-1.0)
    ctx = _StandInCtx(host, blob)
    store = torch.distributed.TCPStore("127.0.0.1", port2, world, rank == 0)    # the launcher's store in bench.py
    comm = mzd.NativeComm(ctx, rank, world, store=store, make_id=lambda: bytes(range(128)))
    # the very sequence of bench.py's N > 1 branch
    comm.broadcast_weights(0)
    comm.barrier()
    lo, hi = mzd.shard_range(total_games, rank, world)
    tmax = comm.max_over_ranks(1.0 + rank)
    stats = comm.gather_stats([hi - lo, (hi - lo) * 20, float(blob.double().sum())])
    comm.close()
    q.put((rank, stats.tolist(), tmax, ctx.calls, ctx.uid, comm.kind))
    host.close()


def test_native_comm_call_sequence_over_a_stand_in_context():
    """NativeComm (what bench.py runs for N > 1, over RCCL through the C ABI on a GPU box) driven at world size 2 over a
    stand-in context: the id reaches every rank, the calls arrive in the order the C ABI expects (init, broadcast,
    barrier, gathers, destroy), and its results equal HostComm's for the same worker body."""
    world, total = 2, 101
    res = _run_ranks(_native_worker, world, 2, total)
    (r0, st0, t0, calls0, uid0, kind0), (r1, st1, t1, calls1, uid1, kind1) = res
    assert uid0 == uid1 == bytes(range(128))
    want = [("comm_init", None, 2), ("broadcast_weights", 0), ("comm_barrier",), ("gather_stats", 1, 2),
            ("gather_stats", 3, 2), ("comm_destroy",)]
    for rank, calls in ((0, calls0), (1, calls1)):
        assert [c[0] for c in calls] == [w[0] for w in want]
        assert calls[0] == ("comm_init", rank, 2) and calls[1] == ("broadcast_weights", 0)
    assert kind0 == kind1 and kind0.startswith("rccl (C ABI")           # what bench.py prints as config.collectives
    assert st0 == st1 and t0 == t1 == 2.0
    st = np.array(st0)
    assert st[:, 0].sum() == total and st[0, 2] == st[1, 2]
    ref = netspec.pack_blob(netspec.spec_tiny(), netspec.random_weights(netspec.spec_tiny(), seed=7))
    assert abs(st[0, 2] - ref.astype(np.float64).sum()) < 1e-3
