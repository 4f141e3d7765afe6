#!/usr/bin/env python3
"""Headline benchmark: MCTS simulations/sec of batched Gumbel-MuZero self-play search, Go 9x9.

One "step" = one planned move for every game of the batch through the C ABI (mz_search_moves):
initialInference of all roots, root expansion + Gumbel top-m, then n x [select -> recurrentInference
of all leaves -> expand/backup/sequential halving], then storeSearchStatistics + selectAction.
Inputs (observations, legal masks) and outputs are resident in HBM during the timed region; the Gumbel
noise is drawn on the device (Philox, seed 7 - SURVEY.md §8d).  Games shard over GPUs (one process per
GPU, no data-path collective); RCCL - through the C ABI, mz_comm_init / mz_broadcast_weights /
mz_gather_stats - only broadcasts the weight blob before the timed region and gathers the per-rank
statistics after it; torch.distributed only hands rank 0's RCCL id to the other ranks.  If RCCL cannot start the run
fails (non-zero exit): no other transport is substituted.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0, "f16": 2500.0}   # MI355X dense MFMA peaks (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (spec factory name, n, m, c_visit, known bounds)   SURVEY.md §8a-0
    "go9": ("go9", 50, 16, 50, (-81.0, 81.0)),
    "go5": ("go5", 30, 16, 50, (-25.0, 25.0)),
    "tictactoe": ("tictactoe", 20, 4, 20, (-1.0, 1.0)),
    "pegsolitair": ("pegsolitair", 100, 16, 50, None),
}


def cpu_baseline(spec, weights, n, m, c_visit, bounds, games, seed, max_batch=1000):
    """The oracle search + fp32 torch-CPU forward, driven with the reference's batching policy (all
    games in lock-step, <= numParallelInferences = 1000 samples per forward, ModelController.java:86-89).
    Bounded sample: `games` games x one move."""
    import torch
    from oracle import search as osr
    from oracle.net import NetOracle
    from muzero_amd import netspec

    rng = np.random.default_rng(seed)
    A = spec.action_space
    obs = (rng.uniform(size=(games, spec.obs_planes, spec.height, spec.width)) < 0.3).astype(np.float32)
    legal = np.ones((games, A), np.uint8)
    u = rng.uniform(size=(games, A))
    gumbel = -np.log(-np.log(u))
    cells = netspec.action_cells_board(spec)
    net = NetOracle(spec, weights)
    ob = osr.OracleBatch(osr.make_config(A, m, c_visit, two_players=spec.two_players, known_bounds=bounds), games)
    chw = (spec.channels, spec.height, spec.width)
    hidden = torch.zeros((games, n + 1) + chw)

    def fwd_chunks(fn, count):
        outs = []
        for s in range(0, count, max_batch):
            outs.append(fn(slice(s, min(count, s + max_batch))))
        return [torch.cat([o[i] for o in outs]) for i in range(4)]

    t0 = time.perf_counter()
    st, lg, pol, val = fwd_chunks(lambda sl: net.initial_inference(obs[sl]), games)
    hidden[:, 0] = st
    ob.begin(legal, np.zeros(games, np.int32), n, lg.numpy(), pol.numpy(), val.numpy(), gumbel=gumbel)
    nodes = np.ones(games, np.int64)
    sims = 0
    ar = np.arange(games)
    while True:
        cnt, ps, ac, on = ob.select()
        if cnt == 0:
            break
        parent = hidden[ar, np.maximum(ps, 0)]
        st, lg, pol, val = fwd_chunks(lambda sl: net.recurrent_inference(parent[sl], ac[sl], cells), games)
        act = on.astype(bool)
        hidden[ar[act], nodes[act]] = st[act]
        nodes[act] += 1
        ob.expand_backup(on, lg.numpy(), pol.numpy(), val.numpy())
        sims += int(cnt)
    out = ob.finish()
    dt = time.perf_counter() - t0
    assert (out["status"] == 0).all()
    return sims / dt, sims, dt, torch.get_num_threads()


def fp32_leg(spec, weights, n, m, c_visit, bounds, games, obs, device, moves):
    """The same search on the exact-fp32 network path (MZ_PRECISION_FP32: the reference's own arithmetic,
    InitialInferenceListTranslator.java:44-112 is fp32 end to end; here on v_mfma_f32_16x16x4_f32) at the same batch:
    `moves` timed moves after a warm-up move, every kernel on one stream and timed with HIP events.  An extra, separately
    labelled figure - never `value`.  `python bench.py --precision fp32` runs the whole bench on this path
    (profiles/r3_fp32_kernel_stats.csv is rocprofv3's view of that)."""
    import torch
    from muzero_amd import capi, netspec
    ctx = capi.Context(capi.make_config(spec, n, m, c_visit, known_bounds=bounds, max_games=games,
                                        precision=capi.PRECISION_FP32, gumbel_seed=7), device=device)
    ctx.set_action_cells(netspec.action_cells_board(spec))
    ctx.load_weights(netspec.pack_blob(spec, weights))
    ctx.set_stream_split(False)
    legal = torch.ones((games, spec.action_space), dtype=torch.uint8, device=obs.device)
    o = obs[:games].contiguous()
    A, dev = spec.action_space, obs.device
    outs = dict(action=torch.zeros(games, dtype=torch.int32, device=dev),
                visit_counts=torch.zeros((games, A), dtype=torch.int32, device=dev),
                policy_target=torch.zeros((games, A), dtype=torch.float32, device=dev),
                root_value_target=torch.zeros(games, dtype=torch.float32, device=dev),
                sims_run=torch.zeros(games, dtype=torch.int32, device=dev),
                status=torch.zeros(games, dtype=torch.int32, device=dev))
    ctx.search_moves(games, o, legal, outputs=outs)
    ctx.synchronize()
    ctx.set_timing(True)
    sims, ms, cnt = 0, 0.0, 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(moves):
        ctx.search_moves(games, o, legal, outputs=outs)
        sims += ctx.last_stats()[0]
        t = ctx.get_timing()["recurrent"]
        ms += t[0]
        cnt += t[1]
    ctx.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(outs["status"].abs().sum().item()) == 0
    ctx.close()
    tf = netspec.flops_per_sample(spec)["recurrent"] * games / (ms / max(cnt, 1) / 1e3) / 1e12 if ms > 0 else 0.0
    return {"simulations_per_s": sims / dt, "games": games, "moves": moves, "dtype": "fp32",
            "recurrent_ms": ms / max(cnt, 1), "launches_timed": cnt, "achieved_tflops": tf, "peak_tflops": PEAK_TFLOPS["fp32"],
            "frac": tf / PEAK_TFLOPS["fp32"], "kernel": "k_net_fp32 (v_mfma_f32_16x16x4_f32 convs)"}


def find_traffic(workload, precision, games):
    """HBM bytes per launch of the dominant kernel chain from the PMC passes of tools/pmc.sh (FETCH_SIZE and WRITE_SIZE
    in separate rocprofv3 --pmc runs, gfx950 correction applied there).  Picks the newest committed
    profiles/*traffic_<workload>*.json of this precision, preferring one measured at this batch; a figure measured at
    another batch is scaled per sample and says so.  None (and a loud note on stderr) if there is none."""
    import glob
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", f"*traffic_{workload}*.json")):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if t.get("precision") not in (precision, "bf16" if precision == "f16" else precision) or "bytes_per_sample" not in t:
            continue
        key = (t.get("games") == games, os.path.basename(path))
        if best is None or key > best[0]:
            best = (key, path, t)
    if best is None:
        print(f"[bench] no profiles/*traffic_{workload}*.json for {precision}: roofline.traffic is null "
              "(run tools/pmc.sh on the GPU box and commit its traffic json)", file=sys.stderr, flush=True)
        return None, None
    _, path, t = best
    src = os.path.relpath(path, ROOT) + (f" (measured at {t.get('games')} games, scaled per sample)"
                                         if t.get("games") != games else f" (measured at {games} games)")
    return t["bytes_per_sample"] * games, src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="go9", choices=sorted(WORKLOADS))
    ap.add_argument("--games", type=int, default=16384, help="concurrent games per GPU (SURVEY.md §8d: B in {1024, 4096, 16384})")
    ap.add_argument("--precision", default="f16", choices=["fp32", "bf16", "f16"],
                    help="network arithmetic: fp32 exact kernels, or MFMA with bf16 / fp16 operands and fp32 accumulation")
    ap.add_argument("--cpu-games", type=int, default=1024, help="games in the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm-timeout", type=float, default=300.0, help="N > 1: seconds allowed for the RCCL set-up before the run is ended")
    ap.add_argument("--fp32-games", type=int, default=-1,
                    help="games of the extra exact-fp32 leg (the same search on MZ_PRECISION_FP32: v_mfma_f32_16x16x4_f32, "
                         "`steps` timed moves); -1 = the bench batch, 0 = skip")
    args = ap.parse_args()

    import torch
    from muzero_amd import capi, netspec
    from muzero_amd import dist as mzd

    world, rank, local_rank = mzd.world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    spec_name, n, m, c_visit, bounds = WORKLOADS[args.workload]
    spec = netspec.SPECS[spec_name]()
    B, A = args.games, spec.action_space
    precision = {"bf16": capi.PRECISION_BF16, "f16": capi.PRECISION_F16, "fp32": capi.PRECISION_FP32}[args.precision]
    ctx = capi.Context(capi.make_config(spec, n, m, c_visit, known_bounds=bounds, max_games=B, precision=precision,
                                        gumbel_seed=7 + 1000003 * rank),   # a stream of its own per shard
                       device=local_rank)
    ctx.set_action_cells(netspec.action_cells_board(spec))

    # weights: rank 0 draws them, RCCL broadcasts the fp32 blob over xGMI into every context's master copy,
    # every other rank repacks locally (mz_broadcast_weights)
    weights = netspec.random_weights(spec, seed=1) if rank == 0 or world == 1 else None
    if weights is not None:
        ctx.load_weights(netspec.pack_blob(spec, weights))
    comm = None
    # MZ_BENCH_FORCE_COMM=1: take the N > 1 branch at world size 1 too (a rehearsal of the rendezvous, mz_comm_init,
    # mz_broadcast_weights, mz_comm_barrier and mz_gather_stats calls on a one-GPU box)
    if world > 1 or os.environ.get("MZ_BENCH_FORCE_COMM") == "1":
        # The product's transport or nothing: RCCL through the C ABI.  A failure ends the job with a non-zero exit (the
        # launcher then stops the other ranks); a hang inside ncclCommInitRank / ncclBroadcast is not an exception, so a
        # watchdog ends the process if the set-up has not finished in time.  No other transport is ever substituted: a
        # scaling number over torch.distributed would not measure what DESIGN.md §7 describes.
        import threading

        def _give_up():
            print(f"[bench] rank {rank}: RCCL set-up (mz_comm_init / mz_broadcast_weights) did not finish within "
                  f"{args.comm_timeout} s", file=sys.stderr, flush=True)
            os._exit(4)

        dog = threading.Timer(args.comm_timeout, _give_up)
        dog.daemon = True
        dog.start()
        try:
            comm = mzd.NativeComm(ctx, rank, world)
            comm.broadcast_weights(0)
        except Exception as e:   # noqa: BLE001
            print(f"[bench] rank {rank}: native RCCL communicator failed: {e}", file=sys.stderr, flush=True)
            os._exit(3)
        dog.cancel()

    # synthetic games: random 0/1 planes of density 0.3, all actions legal, device Gumbel noise
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    obs = (torch.rand((B, spec.obs_planes, spec.height, spec.width), generator=gen, device=dev) < 0.3).float()
    legal = torch.ones((B, A), dtype=torch.uint8, device=dev)
    outs = dict(action=torch.zeros(B, dtype=torch.int32, device=dev),
                visit_counts=torch.zeros((B, A), dtype=torch.int32, device=dev),
                policy_target=torch.zeros((B, A), dtype=torch.float32, device=dev),
                root_value_target=torch.zeros(B, dtype=torch.float32, device=dev),
                sims_run=torch.zeros(B, dtype=torch.int32, device=dev),
                status=torch.zeros(B, dtype=torch.int32, device=dev))

    def sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()

    def step():
        ctx.search_moves(B, obs, legal, outputs=outs)   # gumbel=None + device_gumbel: k_root draws the noise

    for _ in range(args.warmup):
        step()
    sims_total = 0
    path_sum = 0.0
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step()
        s, mp = ctx.last_stats()
        sims_total += s
        path_sum += mp * s
    sync()
    dt = time.perf_counter() - t0
    assert int(outs["status"].abs().sum().item()) == 0

    # Per-kernel device times for the roofline (HIP events on the library's own stream).  The timed region above runs a
    # move as the product does: the two halves of the batch on two streams, one half's tree / head kernels under the
    # other half's tower - kernel durations overlap there and cannot be attributed.  So the same move is run
    # `steps` more times right here with every kernel back to back on ONE stream (mz_set_stream_split(0): undivided,
    # non-overlapping launches, identical results) and those launches are timed; `serial_ms_per_step` is that mode's
    # wall time per step.  rocprofv3 of `MZ_STREAMS=1 python bench.py` sees the same launches (profiles/).
    ctx.set_stream_split(False)
    ctx.set_timing(True)
    kernel_ms = {k: 0.0 for k in capi.T_NAMES}
    kernel_launches = {k: 0 for k in capi.T_NAMES}
    step()
    sync()
    t1 = time.perf_counter()
    for k in range(args.steps):
        step()
        for name, (ms, cnt) in ctx.get_timing().items():
            kernel_ms[name] += ms
            kernel_launches[name] += cnt
    sync()
    serial_dt = time.perf_counter() - t1
    ctx.set_timing(False)
    ctx.set_stream_split(True)

    # whole-job aggregate: MAX time over ranks, SUM of simulations (episode statistics gathered over RCCL)
    if comm is not None:
        dt = comm.max_over_ranks(dt)
        sims_all = int(comm.gather_stats([sims_total])[:, 0].sum())
    else:
        sims_all = sims_total

    if rank == 0:
        fl = netspec.flops_per_sample(spec)
        rec_ms, rec_n = kernel_ms["recurrent"], max(kernel_launches["recurrent"], 1)
        avg_s = rec_ms / rec_n / 1e3
        achieved = fl["recurrent"] * B / avg_s / 1e12 if avg_s > 0 else 0.0
        peak = PEAK_TFLOPS[args.precision]
        mean_path = path_sum / max(sims_total, 1)
        tree_bytes = A * (12 * (mean_path - 2) + 28 + 8 + 20 * (mean_path - 1))
        tree_ms = kernel_ms["tree_select"] + kernel_ms["tree_backup"]
        # HBM traffic of the dominant launch: PMC numbers (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes) are
        # collected offline by tools/pmc.sh and committed per sample; bench.py scales them to this run's batch.
        traffic, traffic_src = find_traffic(args.workload, args.precision, B)
        line = {
            "metric": "MCTS simulations/sec",
            "value": sims_all / dt,
            "unit": "simulations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"{args.workload} self-play search, {B} concurrent games per GPU, "
                                   f"n={n} simulations, m={m}, C={spec.channels}, R={spec.residuals}",
                       "games_per_gpu": B, "num_simulations": n, "parallelism": f"games sharded x{world}",
                       "gumbel": "device Philox4x32-10", "collectives": comm.kind if comm is not None else None},
            "roofline": {"bound": "mfma", "kernel": "recurrentInference", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "kernels": "k_tower(conv0 + blocks + broadcast block + blocks + rescale + head convs) -> k_heads",
                         "flops_per_launch": fl["recurrent"] * B, "avg_launch_ms": rec_ms / rec_n,
                         "launches_timed": rec_n,
                         "measured": "HIP events around every recurrentInference launch of `steps` further moves run "
                                     "on one stream (mz_set_stream_split(0)) right after the timed region"},
            "serial_ms_per_step": serial_dt / args.steps * 1e3,
            "tree_kernels": {"bound": "hbm", "mean_path_len": mean_path, "algorithmic_bytes_per_sim": tree_bytes,
                             "achieved_GBps": (tree_bytes * sims_total / (tree_ms / 1e3) / 1e9) if tree_ms else None,
                             "peak_GBps": 8000.0},
            "kernel_ms": {k: round(v, 3) for k, v in kernel_ms.items()},
        }
        if args.fp32_games != 0 and world == 1 and args.precision != "fp32":
            ctx.close()   # its 35 GB hidden-state pool is not needed any more
            ctx = None
            fg = B if args.fp32_games < 0 else min(args.fp32_games, B)
            line["fp32_path"] = fp32_leg(spec, weights, n, m, c_visit, bounds, fg, obs, local_rank, max(args.steps, 3))
        if not args.no_cpu_baseline and world == 1:   # timed beside the GPU at N = 1 only
            w_cpu = weights if weights is not None else netspec.random_weights(spec, seed=1)
            v, sims, secs, threads = cpu_baseline(spec, w_cpu, n, m, c_visit, bounds, args.cpu_games, seed=1000)
            line["cpu_baseline"] = {"value": v, "unit": "simulations/s", "cores": threads, "kind": "port",
                                    "sample": f"{args.cpu_games} games x 1 move ({sims} simulations, {secs:.1f} s): "
                                              "oracle search + fp32 torch-CPU forward, lock-step batches <= 1000"}
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.close()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
