#!/usr/bin/env python3
"""Micro-benchmark of the inference kernels alone (used under rocprofv3 for PMC passes)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import capi, netspec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="go9")
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--zero", action="store_true", help="all-zero weights and inputs: same instruction stream, least switching power")
    args = ap.parse_args()
    spec = netspec.SPECS[args.workload]()
    B = args.games
    prec = {"bf16": capi.PRECISION_BF16, "f16": capi.PRECISION_F16, "fp32": capi.PRECISION_FP32}[args.precision]
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=1, precision=prec, pool_slots=2 * B))
    blob = netspec.pack_blob(spec, netspec.random_weights(spec, seed=1))
    if args.zero:
        blob[:] = 0
    ctx.load_weights(blob)
    ctx.set_action_cells(netspec.action_cells_board(spec))
    rng = np.random.default_rng(0)
    obs = (rng.uniform(size=(B, spec.obs_planes, spec.height, spec.width)) < 0.3).astype(np.float32)
    if args.zero:
        obs[:] = 0
    s0 = np.arange(B, dtype=np.int32)
    ctx.initial_inference(obs, s0)
    act = rng.integers(0, spec.action_space, B).astype(np.int32)
    ctx.set_timing(True)
    ts = []
    for _ in range(args.iters):
        ctx.recurrent_inference(s0, act, s0 + B)
        ts.append(ctx.get_timing()["recurrent"][0])
    fl = netspec.flops_per_sample(spec)["recurrent"] * B
    t = float(np.median(ts))
    print(f"recurrent B={B}: median {t:.3f} ms  min {min(ts):.3f} ms  {fl / t / 1e9:.1f} TFLOP/s  {B / t * 1e3:.0f} samples/s")


if __name__ == "__main__":
    main()
