#!/usr/bin/env python3
"""Randomised sweep of the network parity (tests/test_gpu_net_parity.py's _check over random weights, LayerNorm
jitter, batch sizes; all four games; fp32 and fp16-operand paths at the tests' own bars):
python3 tools/fuzz_net.py [cases] [seed0]."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import capi  # noqa: E402
from tests import test_gpu_net_parity as T  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
    rng = np.random.default_rng(seed0)
    specs = T._specs()
    names = ["tictactoe", "go5", "go9", "pegsolitair"]
    bad = 0
    t0 = time.time()
    for i in range(cases):
        name = names[i % 4]
        prec = capi.PRECISION_FP32 if rng.integers(0, 2) else capi.PRECISION_F16
        kw = dict(B=int(rng.integers(1, 20)), seed=seed0 + i, ln_jitter=float(rng.choice([0.0, 0.1, 0.3])))
        try:
            T._check(specs[name], prec, **kw)
        except AssertionError as e:   # noqa: PERF203
            bad += 1
            print(f"MISMATCH case {i} {name} precision {prec} {kw}: {str(e)[:400]}", flush=True)
        if i % 5 == 4:
            print(f"{i + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
