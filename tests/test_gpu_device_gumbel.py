"""-m gpu: Gumbel noise drawn on the device (mz_search_config.device_gumbel / gumbel_seed).

Reference: GumbelFunctions.drawGumble (GumbelFunctions.java:48-54) draws -log(-log(u)) per root child,
GumbelSearch.gumbelActionsStart (GumbelSearch.java:126-151) keeps the top-m of logit + g.  The noise
stream itself (ThreadLocalRandom there, Philox4x32-10 here) is not reproduced - north_star asks for
identical results "given identical ... Gumbel noise" - so the checks are: (1) the search that ran on
device-drawn noise is bit-identical to the oracle's search on that same noise read back; (2) the
noise is Gumbel(0, 1); (3) it is a pure function of (seed, move counter, game, action).
"""
import numpy as np
import pytest

from muzero_amd import capi, netspec
from oracle import search as osr
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _ctx(name, B, seed):
    c = H.SEARCH_CASES[name]
    spec = netspec.spec_tiny(two_players=c["two_players"], action_space=c["A"], hw=(3, 3))
    cfg = capi.make_config(spec, c["n"], c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B, gumbel_seed=seed)
    return capi.Context(cfg), c


@pytest.mark.parametrize("name", ["go9", "tictactoe"])
def test_device_noise_search_equals_oracle_on_the_same_noise(name):
    B = 96
    ctx, c = _ctx(name, B, seed=7)
    A, n = c["A"], c["n"]
    rng = np.random.default_rng(21)
    legal = H.random_legal(rng, B, A)
    to_play = rng.integers(0, 2, B).astype(np.uint8)
    il, ip, iv = H.random_net_outputs(rng, (B,), A, c["vscale"])
    sl, sp, sv = H.random_net_outputs(rng, (B, n), A, c["vscale"])
    got = ctx.search_moves_injected(B, legal, il, ip, iv, sl, sp, sv, to_play=to_play)   # gumbel=None -> device draws
    g = ctx.get_gumbel(B)
    assert (g[legal == 0] == 0).all()
    many = legal.sum(1) >= 2
    assert (g[many][legal[many] == 1] != 0).all()
    ocfg = osr.make_config(A, c["m"], c["c_visit"], two_players=c["two_players"], known_bounds=c["bounds"])
    exp = osr.OracleBatch(ocfg, B).run_injected(legal, to_play.astype(np.int32), n, il, ip, iv, sl, sp, sv, gumbel=g)
    for k in ("action", "selected_action", "visit_counts", "sims_run", "root_value_target", "v_mix"):
        np.testing.assert_array_equal(got[k], exp[k])
    # and it differs from the search without noise (withGumbel = false)
    ctx0 = capi.Context(capi.make_config(netspec.spec_tiny(two_players=c["two_players"], action_space=A, hw=(3, 3)),
                                         n, c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B))
    plain = ctx0.search_moves_injected(B, legal, il, ip, iv, sl, sp, sv, to_play=to_play)
    assert (plain["visit_counts"] != got["visit_counts"]).any()
    ctx.close()
    ctx0.close()


def test_device_noise_is_standard_gumbel_and_top_m_follows_softmax():
    name, B = "go9", 4096
    ctx, c = _ctx(name, B, seed=123)
    A, n, m = c["A"], c["n"], c["m"]
    rng = np.random.default_rng(5)
    legal = np.ones((B, A), np.uint8)
    # identical logits in every game: inclusion frequencies over games estimate the Gumbel-top-m probabilities
    lg = np.tile((1.5 * rng.standard_normal(A)).astype(np.float32), (B, 1))
    e = np.exp(lg - lg.max(-1, keepdims=True))
    p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    iv = np.zeros(B)
    sl, sp, sv = H.random_net_outputs(rng, (B, n), A, c["vscale"])
    got = ctx.search_moves_injected(B, legal, lg, p, iv, sl, sp, sv)
    g = ctx.get_gumbel(B).ravel()
    # Gumbel(0,1): mean = Euler-Mascheroni, variance = pi^2 / 6; 335872 samples
    assert abs(g.mean() - 0.5772156649) < 0.01
    assert abs(g.var() - np.pi ** 2 / 6) < 0.03
    assert abs(np.mean(g < 0.0) - np.exp(-1.0)) < 0.005      # CDF(0) = exp(-exp(0))
    assert np.isfinite(g).all()
    # first candidate drawn (arg-max of logit + g) ~ softmax(logits); the m visited root children contain it
    top1 = np.argmax(lg.astype(np.float64) + ctx.get_gumbel(B), axis=1)
    freq = np.bincount(top1, minlength=A) / B
    assert np.abs(freq - p[0]).max() < 0.03
    assert (got["visit_counts"][np.arange(B), top1] > 0).all()
    assert ((got["visit_counts"] > 0).sum(1) == m).all()
    ctx.close()


def test_device_noise_is_a_function_of_seed_counter_game_action():
    B = 32
    rng = np.random.default_rng(8)
    ctx, c = _ctx("go5", B, seed=99)
    A, n = c["A"], c["n"]
    legal = np.ones((B, A), np.uint8)
    il, ip, iv = H.random_net_outputs(rng, (B,), A, c["vscale"])
    sl, sp, sv = H.random_net_outputs(rng, (B, n), A, c["vscale"])
    run = lambda cx, b=B: cx.search_moves_injected(b, legal[:b], il[:b], ip[:b], iv[:b], sl[:b], sp[:b], sv[:b])
    a0 = run(ctx)
    g0 = ctx.get_gumbel(B)
    a1 = run(ctx)                      # the move counter advanced: fresh noise
    g1 = ctx.get_gumbel(B)
    assert (g0 != g1).all()
    ctx.seed_gumbel(99, 0)             # rewind: same noise, same search
    a2 = run(ctx)
    np.testing.assert_array_equal(ctx.get_gumbel(B), g0)
    np.testing.assert_array_equal(a2["visit_counts"], a0["visit_counts"])
    ctx.seed_gumbel(99, 1)
    run(ctx)
    np.testing.assert_array_equal(ctx.get_gumbel(B), g1)
    ctx.seed_gumbel(100, 0)            # another seed: other noise
    run(ctx)
    assert (ctx.get_gumbel(B) != g0).all()
    # a game's noise does not depend on the batch it runs in
    ctx.seed_gumbel(99, 0)
    run(ctx, 8)
    np.testing.assert_array_equal(ctx.get_gumbel(8), g0[:8])
    # an explicit noise array still wins over the device stream
    gx = H.gumbel_noise(rng, B, A)
    ctx.search_moves_injected(B, legal, il, ip, iv, sl, sp, sv, gumbel=gx)
    np.testing.assert_array_equal(ctx.get_gumbel(B), gx)
    assert a1 is not None
    ctx.close()
