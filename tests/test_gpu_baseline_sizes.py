"""-m gpu: BASELINE.json's other workloads at their full batch (TicTacToe, PegSolitair and
Go 5x5 at 4096 concurrent games, real bf16 nets, one planned move) - the shape of test_bench_size_properties_go9.
The oracle cannot replay 4096 x n forwards in seconds, so these check what holds at any size: budget identities,
legality, normalisation, value range, and that games are independent units - the first k games searched alone
(k a multiple of the boards per workgroup tile, so every game keeps its place in its tile) give bit-identical results.
Covers the numChannels-256 wave grid (2 x 4 waves, 14 / 5 boards per tile) and the multi-k_bseg PegSolitair chain
beyond the <= 3 tiles of the oracle-checked tests."""
import numpy as np
import pytest

from muzero_amd import capi, netspec
from tests import helpers as H

pytestmark = pytest.mark.gpu

#            spec factory, case,    boards per tile, k
CASES = {
    "tictactoe": (netspec.spec_tictactoe, 14, 70),
    "go5": (netspec.spec_go5, 5, 50),
    "pegsolitair": (netspec.spec_pegsolitair, 5, 50),
}


@pytest.mark.parametrize("precision", [capi.PRECISION_BF16, capi.PRECISION_F16])
@pytest.mark.parametrize("name", ["tictactoe", "go5", "pegsolitair"])
def test_baseline_batch_properties_mfma(name, precision):
    factory, nb_tile, k = CASES[name]
    spec = factory()
    c = H.SEARCH_CASES[name]
    B, A, n, m = 4096, spec.action_space, c["n"], c["m"]
    rng = np.random.default_rng(31)
    w = netspec.pack_blob(spec, netspec.random_weights(spec, seed=1))
    obs = H.synthetic_obs(rng, spec, B)
    legal = np.ones((B, A), np.uint8)
    legal[7, : A // 2] = 0
    legal[11] = 0
    legal[11, A - 1] = 1                                 # one legal action: the shortcut inside a big batch
    legal[64:128] = H.random_legal(rng, 64, A, special=False)
    to_play = rng.integers(0, 2, B).astype(np.uint8)
    g = H.gumbel_noise(rng, B, A)

    def run(nb):
        ctx = capi.Context(capi.make_config(spec, n, m, c["c_visit"], known_bounds=c["bounds"], max_games=nb,
                                            precision=precision))
        ctx.load_weights(w)
        ctx.set_action_cells(netspec.action_cells_board(spec))
        out = ctx.search_moves(nb, obs[:nb], legal[:nb], to_play=to_play[:nb], gumbel=g[:nb])
        ctx.close()
        return out

    got = run(B)
    L = legal.sum(1)
    searched = L >= 2
    assert (got["status"] == 0).all()
    assert (got["sims_run"][searched] == n).all() and got["sims_run"][11] == 0
    assert got["action"][11] == A - 1 and got["policy_target"][11, A - 1] == 1.0
    assert (got["visit_counts"][searched].sum(1) == n).all()
    assert ((got["visit_counts"] > 0).sum(1) <= np.minimum(m, L)).all()
    assert (legal[np.arange(B), got["action"]] == 1).all()
    assert (got["visit_counts"][legal == 0] == 0).all() and (got["policy_target"][legal == 0] == 0).all()
    np.testing.assert_allclose(got["policy_target"].sum(1), 1.0, atol=1e-5)
    assert (got["visit_counts"][np.arange(B), got["selected_action"]][searched] > 0).all()
    assert np.isfinite(got["root_value_target"]).all() and np.isfinite(got["v_mix"]).all()
    if c["bounds"] is not None:
        assert (np.abs(got["root_value_target"]) <= c["bounds"][1] * (1 + 1e-5)).all()
    assert len(np.unique(got["action"])) > min(A, 8) // 2    # the nets see different boards
    assert k % nb_tile == 0
    sub = run(k)
    for key in ("action", "selected_action", "visit_counts", "policy_target", "root_value_target", "v_mix",
                "root_value_inference", "entropy"):
        np.testing.assert_array_equal(sub[key], got[key][:k])
