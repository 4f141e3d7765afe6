"""The remaining callers of the path against the oracle, and the experience hand-off (N3) end to end.

* fastRuleLearning root (Node.java:289-294, GumbelSearch.java:372-378) - CPU: every policy target the host mirror
  stores equals the oracle's restatement bit for bit; the last legal action is never played (Node.getRandomAction's
  exclusive bound, Node.java:188-191).
* PlanAction.playAfterJustWithInitialInference (PlanAction.java:174-193) - GPU: for every move the mirror made, the
  oracle, given the same network policy, legal actions and uniform draw, returns the same action and the same
  policy target bit for bit (HashMap child order, Kahan renormalisation, EnumeratedDistribution.sample()).
* N3 end to end - GPU: TicTacToe episodes are played on the device (Gumbel search, HYBRID play type: Gumbel and
  Dirichlet noise, temperature sampling), saved with saveState, decoded with google.protobuf, and EVERY number in the
  buffer is followed back to the mz_move_outputs of the planActions call that produced it (the mirror's MoveRecord).
Parity unpinned: the reference holds no vectors for these drivers."""
import json
import os
import subprocess
import zipfile

import numpy as np
import pytest

from muzero_amd import netspec
from oracle import search as osr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "muzero_amd", "host", "bin", "gamebuffer_tool")


def _run(tmp_path, n, seed, mode=None):
    args = [BIN, str(tmp_path), str(n), str(seed)]
    if mode:
        spec = netspec.spec_tictactoe()
        w = tmp_path / "ttt.f32"
        netspec.pack_blob(spec, netspec.random_weights(spec, seed=1)).tofile(w)
        args += [str(w), mode]
    out = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads((tmp_path / "buffer.json").read_text())


def f32(x):
    return np.asarray(x, np.float32)


def test_fast_rule_learning_targets_equal_the_oracle(tmp_path):
    ref = _run(tmp_path, 60, 3)
    moves = 0
    for g in ref["games"]:
        assert len(g["policy_targets"]) == len(g["actions"])
        for t, a in enumerate(g["actions"]):
            legal = np.array(g["legal_actions"][t], np.uint8)
            L = int(legal.sum())
            idx = np.nonzero(legal)[0]
            _, want = osr.fast_rule_learning_move(legal, 0)
            assert f32(g["policy_targets"][t]).tobytes() == want.tobytes()
            assert legal[a] == 1
            if L >= 2:
                assert a != idx[-1]          # nextInt(0, size - 1): the last child is never drawn
            else:
                assert a == idx[0] and want[a] == 1.0
            moves += 1
    assert moves > 300


@pytest.mark.gpu
def test_just_initial_inference_policy_equals_the_oracle(tmp_path):
    ref = _run(tmp_path, 200, 11, "initial")
    A, checked, actions = 9, 0, set()
    for g in ref["games"]:
        assert len(g["trace"]) == len(g["actions"])
        for t, rec in enumerate(g["trace"]):
            assert rec["just_initial"] == 1 and rec["move"] == t and rec["simulations"] == 0
            legal = np.array(rec["legal"], np.uint8)
            assert legal.tolist() == g["legal_actions"][t]
            pol = f32(rec["network_policy"])
            assert pol.size == A and abs(pol.sum() - 1) < 1e-5
            a, pt = osr.play_after_just_initial_inference(legal, pol, np.zeros(A, np.float32), rec["u"])
            assert a == rec["action"] == g["actions"][t]
            assert pt.tobytes() == f32(rec["policy_target"]).tobytes() == f32(g["policy_targets"][t]).tobytes()
            assert g["root_value_targets"][t] == 0 and g["v_mix"][t] == 0      # a fresh root (GumbelSearch.java:77-81)
            checked += 1
            actions.add(a)
    assert checked > 1000 and len(actions) == 9


@pytest.mark.gpu
def test_device_played_episodes_reach_the_replay_buffer_unchanged(tmp_path):
    pytest.importorskip("google.protobuf")
    from tests.test_game_buffer_proto import _messages, _bits
    n = 96
    ref = _run(tmp_path, n, 5, "gpu")
    raw = (tmp_path / "buffer.dat").read_bytes()
    z = zipfile.ZipFile(tmp_path / "MuZero-TicTacToe-7-protobuf.zip")       # GameBufferIO.saveState's container
    assert z.read("MuZero-TicTacToe-7.dat") == raw
    m = _messages()()
    m.ParseFromString(raw)
    assert len(m.game_protos) == n == m.counter
    searched = shortcut = 0
    for g, r in zip(m.game_protos, ref["games"]):
        tr = r["trace"]
        T = len(tr)
        assert 5 <= T <= 9 and [x["move"] for x in tr] == list(range(T))
        # every per-move field of the buffer is the mz_move_outputs value of that move's mz_search_moves call
        assert list(g.actions) == [x["action"] for x in tr]
        assert f32(g.root_value_targets).tobytes() == f32([x["root_value_target"] for x in tr]).tobytes()
        assert f32(g.v_mix).tobytes() == f32([x["v_mix"] for x in tr]).tobytes()
        assert f32(g.root_values_from_initial_inference).tobytes() == f32([x["root_value_inference"] for x in tr]).tobytes()
        assert f32(g.entropies).tobytes() == f32([x["entropy"] for x in tr]).tobytes()
        assert f32(g.maxEntropies).tobytes() == f32([x["max_entropy"] for x in tr]).tobytes()
        assert len(g.policy_targets) == T
        for pt, x in zip(g.policy_targets, tr):
            assert f32(pt.policy).tobytes() == f32(x["policy_target"]).tobytes()
        pl = [x for x in tr if x["playout_appended"]]
        assert len(g.playout_policy) == len(pl)
        for pp, x in zip(g.playout_policy, pl):
            assert f32(pp.policy).tobytes() == f32(x["playout_policy"]).tobytes()
        se = [x for x in tr if x["simulations"] > 0]                         # GumbelSearch.storeSearchStatistics only
        assert f32(g.root_entropy_value_targets).tobytes() == f32([x["root_entropy_value_target"] for x in se]).tobytes()
        searched += len(se)
        shortcut += T - len(se)
        # and what the environment plugin contributes: legal masks, observations, rewards
        assert [[int(b) for b in la.legal_action] for la in g.legal_actions] == r["legal_actions"]
        assert len(g.observations) == T + 1
        for o, ro in zip(g.observations, r["observations"]):
            assert _bits(o.observationPartA, 9) == ro["a"] and _bits(o.observationPartB, 9) == ro["b"]
        assert f32(g.rewards).tobytes() == f32(r["rewards"]).tobytes()
        for x, la in zip(tr, r["legal_actions"]):
            assert la[x["action"]] == 1
            p = f32(x["policy_target"])
            assert abs(p.sum() - 1) < 1e-5 and (p[np.array(la) == 0] == 0).all()
            assert x["simulations"] in (0, 20)
    assert searched > 5 * n and shortcut >= 0
    assert m.SerializeToString(deterministic=True) == raw
