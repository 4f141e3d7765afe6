#!/usr/bin/env python3
"""Randomised sweep of the injected-search parity (tests/test_gpu_search_parity.py's case runner over random games,
batch sizes, noise / selection modes and prior sharpness): python3 tools/fuzz_search.py [cases] [seed0].
Prints one line per 10 cases and every mismatch; exit status 1 if any case differs from the oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_search_parity as T  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    rng = np.random.default_rng(seed0)
    bad = 0
    t0 = time.time()
    for i in range(cases):
        name = str(rng.choice(["tictactoe", "go5", "go9", "pegsolitair"]))
        modes = [(False, True, 1.0), (True, True, 1.0), (True, False, 1.0), (False, False, 0.5), (True, True, 0.0),
                 (False, False, 1.0), (False, False, 0.0)]
        hybrid, gsel, temp = modes[int(rng.integers(len(modes)))]
        kw = dict(B=int(rng.integers(1, 129)), seed=seed0 + i, with_gumbel=bool(rng.integers(0, 4) > 0),
                  dirichlet_frac=float(rng.choice([0.0, 0.25, 0.5])), hybrid=hybrid, gumbel_sel=gsel, temperature=temp,
                  per_game_n=bool(rng.integers(0, 2)), special=bool(rng.integers(0, 2)),
                  sharp=float(rng.choice([0.05, 0.3, 1.0, 4.0, 20.0])))
        try:
            exp, got, ob, ctx, legal = T._run_case(name, **kw)
            T._assert_same(exp, got)
            ctx.close()
        except AssertionError as e:   # noqa: PERF203
            bad += 1
            print(f"MISMATCH case {i} {name} {kw}: {str(e)[:300]}", flush=True)
        if i % 10 == 9:
            print(f"{i + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
