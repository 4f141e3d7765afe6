#!/usr/bin/env python3
"""Randomised sweep of the end-to-end check (tests/test_gpu_end_to_end.py::_e2e: mz_search_moves with the real networks,
the device records the network outputs it used, the oracle replays the search on them and must agree bit for bit):
python3 tools/fuzz_e2e.py [cases] [seed0]."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import capi, netspec  # noqa: E402
from tests import test_gpu_end_to_end as T  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    rng = np.random.default_rng(seed0)
    games = [("tictactoe", netspec.spec_tictactoe), ("go5", netspec.spec_go5), ("go9", netspec.spec_go9),
             ("pegsolitair", netspec.spec_pegsolitair)]
    bad = 0
    t0 = time.time()
    for i in range(cases):
        name, mk = games[i % 4]
        prec = [capi.PRECISION_FP32, capi.PRECISION_F16, capi.PRECISION_BF16][int(rng.integers(0, 3))]
        kw = dict(B=int(rng.integers(1, 40)), seed=seed0 + i, precision=prec, frac=float(rng.choice([0.0, 0.25])))
        try:
            T._e2e(mk(), name, **kw)
        except AssertionError as e:   # noqa: PERF203
            bad += 1
            print(f"MISMATCH case {i} {name} {kw}: {str(e)[:400]}", flush=True)
        if i % 5 == 4:
            print(f"{i + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
