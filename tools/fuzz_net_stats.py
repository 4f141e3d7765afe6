#!/usr/bin/env python3
"""Error statistics of a network path (16-bit, or the exact fp32 one) against oracle/net.py over random weights, LayerNorm jitter and batches:
max over cases of |got - ref| / (1 + |ref|) (the quantity an rtol = atol bar bounds) per game and output, for the initial and
the recurrent step.  python3 tools/fuzz_net_stats.py [cases] [seed0] [f16|bf16|fp32]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import capi, netspec  # noqa: E402
from oracle.net import NetOracle  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import test_gpu_net_parity as T  # noqa: E402


def err(a, r):
    return float(np.max(np.abs(a - r) / (1.0 + np.abs(r))))


def sq(a, r):   # (sum of squares, count) of the same quantity: for the rms the error model of netspec.error_model predicts
    e = np.abs(a - r) / (1.0 + np.abs(r))
    return np.array([float(np.sum(e.astype(np.float64) ** 2)), float(e.size)])


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    prec = {"f16": capi.PRECISION_F16, "bf16": capi.PRECISION_BF16, "fp32": capi.PRECISION_FP32}[sys.argv[3] if len(sys.argv) > 3 else "f16"]
    specs = T._specs()
    names = ["tictactoe", "go5", "go9", "pegsolitair"]
    worst = {n: dict(h=0.0, l=0.0, p=0.0, v=0.0) for n in names}
    acc = {n: dict(h=np.zeros(2), l=np.zeros(2), p=np.zeros(2), v=np.zeros(2)) for n in names}
    rng0 = np.random.default_rng(seed0)
    for i in range(cases):
        name = names[i % 4]
        spec = specs[name]
        seed = seed0 + i
        B = int(rng0.integers(1, 20))
        jit = float(rng0.choice([0.0, 0.1, 0.3]))
        rng = np.random.default_rng(seed)
        w = netspec.random_weights(spec, seed=seed, ln_jitter=jit)
        ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=B, precision=prec, pool_slots=2 * B))
        ctx.load_weights(netspec.pack_blob(spec, w))
        cells = netspec.action_cells_board(spec)
        ctx.set_action_cells(cells)
        ref = NetOracle(spec, w)
        obs = H.synthetic_obs(rng, spec, B)
        s0 = np.arange(B, dtype=np.int32)
        lg, pol, val = ctx.initial_inference(obs, s0)
        rs, rl, rp, rv = ref.initial_inference(obs)
        hid = np.stack([ctx.read_hidden(s) for s in s0])
        act = rng.integers(0, spec.action_space, B).astype(np.int32)
        lg2, pol2, val2 = ctx.recurrent_inference(s0, act, s0 + B)
        rs2, rl2, rp2, rv2 = ref.recurrent_inference(hid, act, cells)
        hid2 = np.stack([ctx.read_hidden(s) for s in s0 + B])
        wv = worst[name]
        for a, r, k in ((hid, rs, "h"), (hid2, rs2, "h"), (lg, rl, "l"), (lg2, rl2, "l"), (pol, rp, "p"), (pol2, rp2, "p")):
            wv[k] = max(wv[k], err(a, r.numpy()))
            acc[name][k] += sq(a, r.numpy())
        for a, r in ((val, rv), (val2, rv2)):
            wv["v"] = max(wv["v"], err(a / spec.value_scale, r.numpy() / spec.value_scale))
            acc[name]["v"] += sq(a / spec.value_scale, r.numpy() / spec.value_scale)
        ctx.close()
        if i % 20 == 19:
            print(i + 1, "cases", flush=True)
    for n in names:
        print(n, "max", " ".join(f"{k}={v:.2e}" for k, v in worst[n].items()),
              "| rms", " ".join(f"{k}={np.sqrt(v[0] / max(v[1], 1)):.2e}" for k, v in acc[n].items()))


if __name__ == "__main__":
    main()
