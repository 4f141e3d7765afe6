"""-m gpu: the HIP tree kernels against the CPU oracle, through the C ABI, with the network replaced
by identical pre-recorded outputs ("given identical network outputs and Gumbel noise, selected
actions and visit counts must be bit-exact", BASELINE.json north_star)."""
import numpy as np
import pytest

from muzero_amd import capi, netspec
from oracle import search as osr
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _run_case(name, B, seed, with_gumbel=True, dirichlet_frac=0.0, hybrid=False, gumbel_sel=True, temperature=1.0,
              per_game_n=False, special=True, sharp=1.0):
    c = H.SEARCH_CASES[name]
    A, n, m = c["A"], c["n"], c["m"]
    rng = np.random.default_rng(seed)
    legal = H.random_legal(rng, B, A, special)
    to_play = rng.integers(0, 2, B).astype(np.uint8)
    il, ip, iv = H.random_net_outputs(rng, (B,), A, c["vscale"], sharp)
    sl, sp, sv = H.random_net_outputs(rng, (B, n), A, c["vscale"], sharp)
    g = H.gumbel_noise(rng, B, A) if with_gumbel else None
    dn = H.dirichlet_noise(rng, legal) if dirichlet_frac else None
    expl = rng.integers(0, 2, B).astype(np.uint8)
    u = rng.uniform(size=B)
    nsims = rng.integers(max(2, n // 3), n + 1, B).astype(np.int32) if per_game_n else None

    ocfg = osr.make_config(A, m, c["c_visit"], two_players=c["two_players"], known_bounds=c["bounds"],
                           root_exploration_fraction=dirichlet_frac, hybrid=hybrid,
                           gumbel_action_selection=gumbel_sel, temperature_root=temperature)
    ob = osr.OracleBatch(ocfg, B)
    exp = ob.run_injected(legal, to_play.astype(np.int32), nsims if per_game_n else n, il, ip, iv, sl, sp, sv,
                          gumbel=g, dirichlet=dn, exploration_time=expl, u=u)

    spec = netspec.spec_tiny(two_players=c["two_players"], action_space=A, hw=(3, 3))
    cfg = capi.make_config(spec, n, m, c["c_visit"], known_bounds=c["bounds"],
                           root_exploration_fraction=dirichlet_frac, hybrid=hybrid,
                           gumbel_action_selection=gumbel_sel, temperature_root=temperature, max_games=B)
    ctx = capi.Context(cfg)
    got = ctx.search_moves_injected(B, legal, il, ip, iv, sl, sp, sv, to_play=to_play, num_simulations=nsims,
                                    gumbel=g, dirichlet=dn, exploration_time=expl, u=u)
    return exp, got, ob, ctx, legal


def _assert_same(exp, got, exact_targets=True):
    assert (got["status"] == 0).all() and (exp["status"] == 0).all()
    np.testing.assert_array_equal(got["selected_action"], exp["selected_action"])
    np.testing.assert_array_equal(got["action"], exp["action"])
    np.testing.assert_array_equal(got["visit_counts"], exp["visit_counts"])
    np.testing.assert_array_equal(got["sims_run"], exp["sims_run"])
    np.testing.assert_array_equal(got["playout_appended"], exp["playout_appended"])
    # fp64 sums are bit-exact, so their fp32 narrowing is too
    np.testing.assert_array_equal(got["root_value_target"], exp["root_value_target"])
    np.testing.assert_array_equal(got["v_mix"], exp["v_mix"])
    np.testing.assert_array_equal(got["root_value_inference"], exp["root_value_inference"])
    # exp()/log() outputs: ocml vs glibc, <= 1 ulp in fp64 before the fp32 narrowing
    np.testing.assert_allclose(got["policy_target"], exp["policy_target"], rtol=0, atol=1.2e-7)
    np.testing.assert_allclose(got["entropy"], exp["entropy"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got["legal_action_max_entropy"], exp["legal_action_max_entropy"], rtol=1e-6)
    np.testing.assert_allclose(got["root_entropy_value_target"], exp["root_entropy_value_target"], rtol=1e-6,
                               atol=1e-6)
    m = exp["playout_appended"].astype(bool)
    np.testing.assert_allclose(got["playout_policy"][m], exp["playout_policy"][m], rtol=0, atol=1.2e-7)


@pytest.mark.parametrize("name", ["tictactoe", "go5", "go9", "pegsolitair"])
def test_injected_search_bit_exact(name):
    exp, got, ob, ctx, legal = _run_case(name, B=96, seed=11)
    _assert_same(exp, got)
    # whole trees, bit for bit, for the edge-case games and a few random ones
    for g in [0, 2, 3, 4, 5, 6, 17, 63, 95]:
        n_nodes = ob.num_nodes(g)
        assert ctx.num_nodes(g) == n_nodes
        for slot in range(n_nodes):
            probs = H.tree_equal(ob.node(g, slot), ctx.node(g, slot))
            assert not probs, (name, g, slot, probs[:2])
        assert ctx.minmax(g) == ob.minmax(g)
    ctx.close()


@pytest.mark.parametrize("name", ["tictactoe", "go9"])
def test_injected_search_without_gumbel(name):
    exp, got, ob, ctx, _ = _run_case(name, B=64, seed=5, with_gumbel=False)
    _assert_same(exp, got)
    ctx.close()


@pytest.mark.parametrize("name", ["go5", "go9"])
def test_injected_search_with_dirichlet_noise(name):
    exp, got, ob, ctx, _ = _run_case(name, B=64, seed=7, dirichlet_frac=0.25)
    _assert_same(exp, got)
    ctx.close()


@pytest.mark.parametrize("hybrid,gumbel_sel,temperature", [(True, True, 1.0), (True, False, 1.0), (False, False, 0.5),
                                                           (True, True, 0.0)])
def test_select_action_modes(hybrid, gumbel_sel, temperature):
    exp, got, ob, ctx, _ = _run_case("go5", B=64, seed=3, hybrid=hybrid, gumbel_sel=gumbel_sel,
                                     temperature=temperature)
    _assert_same(exp, got)
    ctx.close()


def test_per_game_simulation_budgets():
    exp, got, ob, ctx, _ = _run_case("go9", B=64, seed=9, per_game_n=True)
    _assert_same(exp, got)
    ctx.close()


def test_ties_first_maximum_wins():
    # flat logits and no noise: every comparison is a tie, so only the child order decides
    exp, got, ob, ctx, _ = _run_case("go9", B=32, seed=2, with_gumbel=False, sharp=0.0)
    _assert_same(exp, got)
    ctx.close()


def test_full_size_properties_go9():
    """BASELINE size (4096 games, Go 9x9 shapes): size-independent properties instead of the oracle."""
    c = H.SEARCH_CASES["go9"]
    B, A, n = 4096, c["A"], c["n"]
    rng = np.random.default_rng(123)
    legal = np.ones((B, A), np.uint8)
    il, ip, iv = H.random_net_outputs(rng, (B,), A, c["vscale"])
    sl, sp, sv = H.random_net_outputs(rng, (B, n), A, c["vscale"])
    g = H.gumbel_noise(rng, B, A)
    spec = netspec.spec_tiny(action_space=A)
    ctx = capi.Context(capi.make_config(spec, n, c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B))
    got = ctx.search_moves_injected(B, legal, il, ip, iv, sl, sp, sv, gumbel=g)
    assert (got["status"] == 0).all()
    assert (got["sims_run"] == n).all()
    assert (got["visit_counts"].sum(1) == n).all()            # every simulation visits one root child
    assert ((got["visit_counts"] > 0).sum(1) <= c["m"]).all()  # only the m sampled candidates are visited
    np.testing.assert_allclose(got["policy_target"].sum(1), 1.0, atol=1e-5)
    assert (got["visit_counts"][np.arange(B), got["action"]] > 0).all()
    assert (np.abs(got["root_value_target"]) <= 81.0 + 1e-4).all()
    # games are independent: the first 64 rerun alone give identical results
    ctx2 = capi.Context(capi.make_config(spec, n, c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=64))
    sub = ctx2.search_moves_injected(64, legal[:64], il[:64], ip[:64], iv[:64], sl[:64], sp[:64], sv[:64], gumbel=g[:64])
    for k in ("action", "visit_counts", "policy_target", "root_value_target", "v_mix"):
        np.testing.assert_array_equal(sub[k], got[k][:64])
    ctx.close()
    ctx2.close()
