"""The reference's own PegSolitair data fixture (games/pegsolitair/src/test/resources/games/buffer1000.zip: 10 recorded
episodes, 286 positions, legacy JSON buffer) as tests/golden/peg_actions.npz (made by tests/golden/make_peg_golden.py in the
build container; data only).  It pins the action-index convention that mz_set_action_cells' default assumes for A = 196
- index = direction * 49 + row * 7 + col, directions N, E, S, W (PegSolitairAdapter.java:19-38,
PegSolitairAction.java:57-61) - against moves and policy targets the reference itself wrote, and supplies real positions
and legal-mask shapes for the search."""
import os

import numpy as np
import pytest

from muzero_amd import netspec

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "peg_actions.npz")
DIRS = [(-1, 0), (0, 1), (1, 0), (0, -1)]


def _load():
    d = np.load(GOLD)
    planes = np.unpackbits(d["planes"], axis=1)[:, :49]
    legal = np.unpackbits(d["legal"], axis=1)[:, :196]
    support = np.unpackbits(d["target_support"], axis=1)[:, :196]
    return planes, legal, support, d["actions"].astype(np.int64), d["target_argmax"].astype(np.int64), d["game"].astype(np.int64)


def test_recorded_moves_and_targets_are_legal_under_the_index_convention():
    planes, legal, support, actions, tmax, game = _load()
    M = len(actions)
    assert M == 286 and len(np.unique(game)) == 10
    assert (planes[game != np.roll(game, 1)].sum(1) == 32).all()          # every episode starts from the 32-peg board
    assert (legal[np.arange(M), actions] == 1).all()                      # every move the reference played
    assert ((support & (1 - legal)).sum(1) == 0).all()                    # every non-zero policyTargets entry
    assert (support.sum(1) >= 1).all() and (legal[np.arange(M), tmax] == 1).all()
    # replaying a move under the convention gives the next recorded position's plane (jump over a peg into a hole)
    for i in range(M - 1):
        if game[i + 1] != game[i]:
            continue
        d, cell = divmod(int(actions[i]), 49)
        r, c = divmod(cell, 7)
        dr, dc = DIRS[d]
        nxt = planes[i].copy()
        assert nxt[r * 7 + c] == 1 and nxt[(r + dr) * 7 + c + dc] == 1 and nxt[(r + 2 * dr) * 7 + c + 2 * dc] == 0
        nxt[r * 7 + c] = nxt[(r + dr) * 7 + c + dc] = 0
        nxt[(r + 2 * dr) * 7 + c + 2 * dc] = 1
        np.testing.assert_array_equal(nxt, planes[i + 1])
    # a move removes one peg
    same = game[1:] == game[:-1]
    assert (planes[1:][same].sum(1) == planes[:-1][same].sum(1) - 1).all()


def test_default_action_cells_are_the_reference_encoding():
    """PegSolitairAction.encode sets plane = direction, cell = (row, col) of the jump's start: with the index convention
    above that is action -> plane * 49 + cell = the action index itself, which is what the library assumes by default."""
    spec = netspec.spec_pegsolitair()
    assert spec.action_space == 196 and spec.action_planes == 4 and spec.height * spec.width == 49
    np.testing.assert_array_equal(netspec.action_cells_board(spec), np.arange(196, dtype=np.int32))


@pytest.mark.gpu
def test_search_on_the_recorded_positions_matches_the_oracle():
    """mz_search_moves on all 286 real positions / legal masks of the fixture (A = 196, single player, n = 100, real fp16
    network with random weights): the oracle replays the search on the recorded network outputs bit for bit, positions with
    one legal move take the shortcut with the move the mask allows, and no illegal action is ever visited."""
    from muzero_amd import capi
    from oracle import search as osr
    from tests import helpers as H
    planes, legal, support, actions, tmax, game = _load()
    spec = netspec.spec_pegsolitair()
    c = H.SEARCH_CASES["pegsolitair"]
    B, A, n, m = len(actions), spec.action_space, c["n"], c["m"]
    rng = np.random.default_rng(5)
    legal = np.ascontiguousarray(legal.astype(np.uint8))
    obs = np.ascontiguousarray(planes.reshape(B, 1, 7, 7).astype(np.float32))
    ctx = capi.Context(capi.make_config(spec, n, m, c["c_visit"], known_bounds=c["bounds"], max_games=B,
                                        precision=capi.PRECISION_F16))
    ctx.load_weights(netspec.pack_blob(spec, netspec.random_weights(spec, seed=3)))
    ctx.set_action_cells(netspec.action_cells_board(spec))
    ctx.set_recording(True)
    g = H.gumbel_noise(rng, B, A)
    to_play = np.zeros(B, np.uint8)
    got = ctx.search_moves(B, obs, legal, to_play=to_play, gumbel=g)
    rec = ctx.get_recording()
    ctx.close()
    assert (got["status"] == 0).all()
    L = legal.sum(1)
    assert (got["sims_run"][L >= 2] == n).all() and (got["sims_run"][L == 1] == 0).all()
    assert (legal[np.arange(B), got["action"]] == 1).all()
    assert (got["visit_counts"][legal == 0] == 0).all() and (got["policy_target"][legal == 0] == 0).all()
    one = np.where(L == 1)[0]
    assert len(one) >= 10 and (got["action"][one] == legal[one].argmax(1)).all()
    ocfg = osr.make_config(A, m, c["c_visit"], two_players=spec.two_players, known_bounds=c["bounds"])
    exp = osr.OracleBatch(ocfg, B).run_injected(legal, to_play.astype(np.int32), n, rec["init_logits"], rec["init_policy"],
                                                rec["init_value"], rec["sim_logits"], rec["sim_policy"], rec["sim_value"],
                                                gumbel=g)
    np.testing.assert_array_equal(got["action"], exp["action"])
    np.testing.assert_array_equal(got["visit_counts"], exp["visit_counts"])
    np.testing.assert_array_equal(got["root_value_target"], exp["root_value_target"])
    np.testing.assert_allclose(got["policy_target"], exp["policy_target"], rtol=0, atol=1.2e-7)
