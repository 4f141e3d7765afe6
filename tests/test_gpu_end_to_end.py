"""-m gpu: mz_search_moves with the real networks.  The device records the network outputs it used;
the CPU oracle replays the search on exactly those outputs and must reproduce actions and visit counts
bit for bit, and the recorded root outputs must match the fp32 CPU forward within tolerance."""
import numpy as np
import pytest

from muzero_amd import capi, netspec
from oracle import search as osr
from oracle.net import NetOracle
from tests import helpers as H
from tests.test_gpu_net_parity import error_bars

pytestmark = pytest.mark.gpu


def _e2e(spec, case, B, seed, precision=capi.PRECISION_FP32, frac=0.0):
    c = H.SEARCH_CASES[case]
    n, m = c["n"], c["m"]
    rng = np.random.default_rng(seed)
    w = netspec.random_weights(spec, seed=seed)
    ctx = capi.Context(capi.make_config(spec, n, m, c["c_visit"], known_bounds=c["bounds"],
                                        root_exploration_fraction=frac, max_games=B, precision=precision))
    ctx.load_weights(netspec.pack_blob(spec, w))
    ctx.set_action_cells(netspec.action_cells_board(spec))
    ctx.set_recording(True)
    obs = H.synthetic_obs(rng, spec, B)
    legal = H.random_legal(rng, B, spec.action_space)
    to_play = rng.integers(0, 2, B).astype(np.uint8)
    g = H.gumbel_noise(rng, B, spec.action_space)
    dn = H.dirichlet_noise(rng, legal) if frac else None
    got = ctx.search_moves(B, obs, legal, to_play=to_play, gumbel=g, dirichlet=dn)
    rec = ctx.get_recording()
    sims, mean_path = ctx.last_stats()
    assert sims == int(got["sims_run"].sum())
    assert 2.0 <= mean_path <= n + 1

    ocfg = osr.make_config(spec.action_space, m, c["c_visit"], two_players=spec.two_players, known_bounds=c["bounds"],
                           root_exploration_fraction=frac)
    ob = osr.OracleBatch(ocfg, B)
    exp = ob.run_injected(legal, to_play.astype(np.int32), n, rec["init_logits"], rec["init_policy"],
                          rec["init_value"], rec["sim_logits"], rec["sim_policy"], rec["sim_value"], gumbel=g,
                          dirichlet=dn)
    assert (got["status"] == 0).all()
    np.testing.assert_array_equal(got["action"], exp["action"])
    np.testing.assert_array_equal(got["visit_counts"], exp["visit_counts"])
    np.testing.assert_array_equal(got["sims_run"], exp["sims_run"])
    np.testing.assert_array_equal(got["root_value_target"], exp["root_value_target"])
    np.testing.assert_allclose(got["policy_target"], exp["policy_target"], rtol=0, atol=1.2e-7)

    # the recorded root outputs are the network's: compare with the CPU forward
    ref = NetOracle(spec, w)
    _, r_lg, r_pol, r_val = ref.initial_inference(obs)
    bars = error_bars(precision)       # the error model's bars (tests/test_gpu_net_parity.py)
    tol, vtol = bars["l"], bars["v"]
    np.testing.assert_allclose(rec["init_logits"], r_lg.numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(rec["init_policy"], r_pol.numpy(), rtol=bars["p"], atol=bars["p"])
    np.testing.assert_allclose(rec["init_value"] / spec.value_scale, r_val.numpy() / spec.value_scale, rtol=vtol,
                               atol=vtol)
    ctx.close()
    return got


def test_end_to_end_tictactoe_shape_fp32():
    _e2e(netspec.spec_tictactoe(), "tictactoe", B=32, seed=1)


def test_end_to_end_go9_fp32():
    _e2e(netspec.spec_go9(), "go9", B=16, seed=2)


def test_end_to_end_go5_dirichlet_fp32():
    _e2e(netspec.spec_go5(), "go5", B=16, seed=4, frac=0.25)


def test_end_to_end_pegsolitair_fp32():
    spec = netspec.spec_pegsolitair()
    _e2e(spec, "pegsolitair", B=8, seed=6)


def test_device_pointers_accepted():
    """Inputs and outputs already resident in HBM (torch CUDA tensors) give the same results as host arrays."""
    import torch
    spec = netspec.spec_tiny()
    c = H.SEARCH_CASES["tictactoe"]
    B = 32
    rng = np.random.default_rng(8)
    w = netspec.random_weights(spec, seed=8)
    ctx = capi.Context(capi.make_config(spec, c["n"], c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B))
    ctx.load_weights(netspec.pack_blob(spec, w))
    obs = H.synthetic_obs(rng, spec, B)
    legal = H.random_legal(rng, B, spec.action_space)
    g = H.gumbel_noise(rng, B, spec.action_space)
    host = ctx.search_moves(B, obs, legal, gumbel=g)
    dev = torch.device("cuda:0")
    out = dict(action=torch.zeros(B, dtype=torch.int32, device=dev),
               visit_counts=torch.zeros(B, spec.action_space, dtype=torch.int32, device=dev),
               policy_target=torch.zeros(B, spec.action_space, dtype=torch.float32, device=dev))
    ctx.search_moves(B, torch.from_numpy(obs).to(dev), torch.from_numpy(legal).to(dev),
                     gumbel=torch.from_numpy(g).to(dev), outputs=out)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out["action"].cpu().numpy(), host["action"])
    np.testing.assert_array_equal(out["visit_counts"].cpu().numpy(), host["visit_counts"])
    np.testing.assert_array_equal(out["policy_target"].cpu().numpy(), host["policy_target"])
    ctx.close()


def test_end_to_end_go9_bf16_mfma():
    # 20 games = 7 workgroup tiles of 3 boards, the last one ragged; search replay must stay bit-exact
    _e2e(netspec.spec_go9(), "go9", B=20, seed=5, precision=capi.PRECISION_BF16)


def test_end_to_end_pegsolitair_bf16_mfma():
    _e2e(netspec.spec_pegsolitair(), "pegsolitair", B=8, seed=7, precision=capi.PRECISION_BF16)


def test_end_to_end_tictactoe_bf16_mfma():
    # numChannels 256 grid (2 x 4 waves on 128 columns = 14 boards per workgroup); 33 games = 3 tiles, the last ragged
    _e2e(netspec.spec_tictactoe(), "tictactoe", B=33, seed=9, precision=capi.PRECISION_BF16)


def test_end_to_end_go5_bf16_mfma_dirichlet():
    # numChannels 256 with a broadcast block (5 boards per workgroup), root noise on
    _e2e(netspec.spec_go5(), "go5", B=11, seed=10, precision=capi.PRECISION_BF16, frac=0.25)


def test_end_to_end_go9_f16_mfma():
    _e2e(netspec.spec_go9(), "go9", B=20, seed=5, precision=capi.PRECISION_F16)


def test_end_to_end_go5_f16_mfma_dirichlet():
    _e2e(netspec.spec_go5(), "go5", B=11, seed=10, precision=capi.PRECISION_F16, frac=0.25)


def test_single_game_and_one_legal_action_shortcut_bf16():
    """Edge cases of PlanAction.planAction: a batch of one game, and games with a single legal action take the
    shortcut (PlanAction.java:99-101,147-171): no simulation, one-hot policy target, value target = root value."""
    spec = netspec.spec_go9()
    c = H.SEARCH_CASES["go9"]
    w = netspec.random_weights(spec, seed=11)
    rng = np.random.default_rng(11)
    B, A = 5, spec.action_space
    ctx = capi.Context(capi.make_config(spec, c["n"], c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B,
                                        precision=capi.PRECISION_BF16))
    ctx.load_weights(netspec.pack_blob(spec, w))
    ctx.set_action_cells(netspec.action_cells_board(spec))
    obs = H.synthetic_obs(rng, spec, B)
    legal = np.ones((B, A), np.uint8)
    legal[1] = 0
    legal[1, 37] = 1            # one option
    legal[3] = 0
    legal[3, A - 1] = 1         # only the pass move
    g = H.gumbel_noise(rng, B, A)
    got = ctx.search_moves(B, obs, legal, gumbel=g)
    assert (got["status"] == 0).all()
    assert got["action"][1] == 37 and got["action"][3] == A - 1
    assert got["sims_run"][1] == 0 and got["sims_run"][3] == 0
    assert got["sims_run"][0] == c["n"] and got["sims_run"][2] == c["n"] and got["sims_run"][4] == c["n"]
    assert got["policy_target"][1, 37] == 1.0 and got["policy_target"][1].sum() == 1.0
    assert got["root_value_target"][1] == got["root_value_inference"][1] == got["v_mix"][1]
    assert (got["visit_counts"][[0, 2, 4]].sum(1) == c["n"]).all()
    # the same first game alone gives the same result as inside the batch (games are independent units)
    one = ctx.search_moves(1, obs[:1], legal[:1], gumbel=g[:1])
    assert one["action"][0] == got["action"][0]
    np.testing.assert_array_equal(one["visit_counts"][0], got["visit_counts"][0])
    ctx.close()


def test_two_stream_search_is_bit_identical_to_one_stream():
    """mz_set_stream_split: a batch large enough to be cut (B / 2 >= 64 boards-per-tile units) is searched as two halves on
    two HIP streams by default.  Every output array must be byte-equal to the one-stream run of the same context: covers the
    second half's offsets (g0 in k_select / k_backup, g_base of the head features) with a ragged legal mask, one-option
    games in both halves and a batch size that is not a multiple of the workgroup tile (3 boards)."""
    spec = netspec.spec_go9()
    c = H.SEARCH_CASES["go9"]
    B, A, n = 500, spec.action_space, 12
    rng = np.random.default_rng(77)
    w = netspec.pack_blob(spec, netspec.random_weights(spec, seed=2))
    obs = H.synthetic_obs(rng, spec, B)
    legal = H.random_legal(rng, B, A, special=False)
    legal[5] = 0
    legal[5, 11] = 1            # one option, first half
    legal[B - 3] = 0
    legal[B - 3, A - 1] = 1     # one option, second half
    to_play = rng.integers(0, 2, B).astype(np.uint8)
    g = H.gumbel_noise(rng, B, A)
    ctx = capi.Context(capi.make_config(spec, n, c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B,
                                        precision=capi.PRECISION_F16))
    ctx.load_weights(w)
    ctx.set_action_cells(netspec.action_cells_board(spec))
    outs = []
    for split in (False, True, False):
        ctx.set_stream_split(split)
        outs.append(ctx.search_moves(B, obs, legal, to_play=to_play, gumbel=g))
    ctx.close()
    assert (outs[0]["status"] == 0).all() and (outs[0]["sims_run"][legal.sum(1) >= 2] == n).all()
    for key in outs[0]:
        for other in outs[1:]:
            a, b = np.asarray(outs[0][key]), np.asarray(other[key])
            assert a.tobytes() == b.tobytes(), key


@pytest.mark.parametrize("precision", [capi.PRECISION_F16, capi.PRECISION_BF16], ids=["f16", "bf16"])
def test_bench_size_properties_go9(precision):
    """The bench.py workload (16384 Go 9x9 games, real nets, one planned move) on the bench's own operand type (fp16) and on
    bf16: size-independent properties, and games are independent units - the first 99 games searched alone (same position
    in their workgroup tiles) give bit-identical actions, visit counts and targets."""
    spec = netspec.spec_go9()
    c = H.SEARCH_CASES["go9"]
    B, A, n = 16384, spec.action_space, c["n"]
    rng = np.random.default_rng(21)
    w = netspec.pack_blob(spec, netspec.random_weights(spec, seed=1))
    obs = H.synthetic_obs(rng, spec, B)
    legal = np.ones((B, A), np.uint8)
    legal[7, :40] = 0
    g = H.gumbel_noise(rng, B, A)

    def run(nb):
        ctx = capi.Context(capi.make_config(spec, n, c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=nb,
                                            precision=precision))
        ctx.load_weights(w)
        ctx.set_action_cells(netspec.action_cells_board(spec))
        out = ctx.search_moves(nb, obs[:nb], legal[:nb], gumbel=g[:nb])
        ctx.close()
        return out

    got = run(B)
    assert (got["status"] == 0).all()
    assert (got["sims_run"] == n).all()
    assert (got["visit_counts"].sum(1) == n).all()
    assert ((got["visit_counts"] > 0).sum(1) <= c["m"]).all()
    assert (legal[np.arange(B), got["action"]] == 1).all()
    assert (got["visit_counts"][legal == 0] == 0).all() and (got["policy_target"][legal == 0] == 0).all()
    np.testing.assert_allclose(got["policy_target"].sum(1), 1.0, atol=1e-5)
    assert np.isfinite(got["root_value_target"]).all() and (np.abs(got["root_value_target"]) <= 81.0 + 1e-3).all()
    assert len(np.unique(got["action"])) > 8      # the nets see different boards: not one constant answer
    sub = run(99)
    for key in ("action", "visit_counts", "policy_target", "root_value_target", "v_mix", "root_value_inference"):
        np.testing.assert_array_equal(sub[key], got[key][:99])
