"""Committed golden vectors (tests/golden/make_golden.py): the oracle must keep reproducing them (CPU), and
the HIP path must reproduce them through the C ABI (-m gpu)."""
import glob
import os

import numpy as np
import pytest

from oracle import search as osr
from tests import helpers as H

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "search_*.npz")))


def _load(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    z = _load(path)
    c = H.SEARCH_CASES[str(z["case"])]
    frac = float(z["frac"])
    B = z["legal"].shape[0]
    cfg = osr.make_config(c["A"], c["m"], c["c_visit"], two_players=c["two_players"], known_bounds=c["bounds"],
                          root_exploration_fraction=frac)
    ob = osr.OracleBatch(cfg, B)
    out = ob.run_injected(z["legal"], z["to_play"].astype(np.int32), c["n"], z["init_logits"], z["init_policy"],
                          z["init_value"], z["sim_logits"], z["sim_policy"], z["sim_value"], gumbel=z["gumbel"],
                          dirichlet=z["dirichlet"] if frac else None)
    np.testing.assert_array_equal(out["action"], z["action"])
    np.testing.assert_array_equal(out["visit_counts"], z["visit_counts"])
    np.testing.assert_array_equal(out["root_value_target"], z["root_value_target"])
    np.testing.assert_array_equal(out["v_mix"], z["v_mix"])
    np.testing.assert_allclose(out["policy_target"], z["policy_target"], rtol=0, atol=1.2e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_reproduces_golden(path):
    from muzero_amd import capi, netspec
    z = _load(path)
    c = H.SEARCH_CASES[str(z["case"])]
    frac = float(z["frac"])
    B = z["legal"].shape[0]
    spec = netspec.spec_tiny(two_players=c["two_players"], action_space=c["A"])
    ctx = capi.Context(capi.make_config(spec, c["n"], c["m"], c["c_visit"], known_bounds=c["bounds"],
                                        root_exploration_fraction=frac, max_games=B))
    out = ctx.search_moves_injected(B, z["legal"], z["init_logits"], z["init_policy"], z["init_value"],
                                    z["sim_logits"], z["sim_policy"], z["sim_value"], to_play=z["to_play"],
                                    gumbel=z["gumbel"], dirichlet=z["dirichlet"] if frac else None)
    assert (out["status"] == 0).all()
    np.testing.assert_array_equal(out["action"], z["action"])
    np.testing.assert_array_equal(out["visit_counts"], z["visit_counts"])
    np.testing.assert_array_equal(out["sims_run"], z["sims_run"])
    np.testing.assert_array_equal(out["root_value_target"], z["root_value_target"])
    np.testing.assert_array_equal(out["v_mix"], z["v_mix"])
    np.testing.assert_allclose(out["policy_target"], z["policy_target"], rtol=0, atol=1.2e-7)
    ctx.close()
