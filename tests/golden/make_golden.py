#!/usr/bin/env python3
"""Generates tests/golden/search_*.npz: seeded inputs + the CPU oracle's outputs for whole-move searches.

The reference (Java 21 + DJL) cannot be executed in the build container (no JDK) and holds
no golden vectors for selected actions / visit counts, so these vectors come from the oracle restatement
(oracle/search_oracle.cpp), which is itself pinned to the reference's known-answer tests.  They freeze the
oracle's behaviour (regression pin) and give the HIP path a fixture that needs no oracle build.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import search as osr  # noqa: E402
from tests import helpers as H    # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make(name, B, seed, frac=0.0):
    c = H.SEARCH_CASES[name]
    A, n, m = c["A"], c["n"], c["m"]
    rng = np.random.default_rng(seed)
    legal = H.random_legal(rng, B, A)
    to_play = rng.integers(0, 2, B).astype(np.uint8)
    il, ip, iv = H.random_net_outputs(rng, (B,), A, c["vscale"])
    sl, sp, sv = H.random_net_outputs(rng, (B, n), A, c["vscale"])
    g = H.gumbel_noise(rng, B, A)
    dn = H.dirichlet_noise(rng, legal) if frac else np.zeros((B, A))
    cfg = osr.make_config(A, m, c["c_visit"], two_players=c["two_players"], known_bounds=c["bounds"],
                          root_exploration_fraction=frac)
    ob = osr.OracleBatch(cfg, B)
    out = ob.run_injected(legal, to_play.astype(np.int32), n, il, ip, iv, sl, sp, sv, gumbel=g,
                          dirichlet=dn if frac else None)
    assert (out["status"] == 0).all()
    np.savez_compressed(
        os.path.join(OUT, f"search_{name}.npz"), case=name, frac=frac, legal=legal, to_play=to_play,
        init_logits=il, init_policy=ip, init_value=iv, sim_logits=sl, sim_policy=sp, sim_value=sv, gumbel=g,
        dirichlet=dn, action=out["action"], visit_counts=out["visit_counts"], policy_target=out["policy_target"],
        root_value_target=out["root_value_target"], v_mix=out["v_mix"], sims_run=out["sims_run"])


if __name__ == "__main__":
    make("tictactoe", 16, 101)
    make("go5", 12, 102, frac=0.25)
    make("go9", 10, 103)
    make("pegsolitair", 8, 104)
    print("written to", OUT)
