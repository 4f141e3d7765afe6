"""Host side above the C ABI (muzero_amd/host): the C++ mirror of the reference's self-play interfaces.

CPU part: the reference's own plugin tests restated against the mirror (EnvironmentTest.java:44-57,
GameTest.java:47-72) - see muzero_amd/host/host_rules_test.cpp.
GPU part: PlayService.playNewGames over the C ABI, the counterpart of PlayTest.java:62-72
(testMultiGameStarter_Basic: 1000 new TicTacToe games are played to the end).
"""
import json
import os
import subprocess

import numpy as np
import pytest

from muzero_amd import netspec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "muzero_amd", "host", "bin")


def _binary(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):   # built by __graft_entry__.build(); plain g++, no GPU needed
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "muzero_amd", "host"), "-s"])
    return path


def test_host_rules_match_reference_plugin_tests():
    out = subprocess.run([_binary("host_rules_test")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok" in out.stdout


def _play(tmp_path, n, play_type, precision, seed=1, mode="play"):
    spec = netspec.spec_tictactoe()
    blob = netspec.pack_blob(spec, netspec.random_weights(spec, seed=1))
    wfile = tmp_path / "ttt.f32"
    blob.astype(np.float32).tofile(wfile)
    out = subprocess.run([_binary("selfplay_main"), str(wfile), str(n), str(seed), play_type, precision, mode],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("play_type,precision,n", [("PLAYOUT", "bf16", 1000), ("HYBRID", "fp32", 256)])
def test_play_new_games_to_the_end(tmp_path, play_type, precision, n):
    r = _play(tmp_path, n, play_type, precision)
    assert r["games"] == n                       # PlayTest.java:70
    assert r["not_done"] == 0                    # every episode reached a terminal position or maxMoves
    assert r["illegal_actions"] == 0             # every action was legal where it was played
    assert r["bad_lengths"] == 0                 # GameDTO lists grow in step (GameDTO.java:42-77)
    assert r["bad_policy_targets"] == 0          # targets are distributions over the legal actions
    assert 5 * n <= r["moves"] <= 9 * n
    assert r["wins_a"] + r["wins_b"] + r["draws"] == n
    # 20 simulations per searched move; moves with one legal action take the shortcut (PlanAction.java:99-101)
    assert r["simulations"] == 20 * (r["moves"] - r["one_option_moves"])


@pytest.mark.gpu
def test_reanalyse_replays_the_original_moves(tmp_path):
    """PlayService.reanalyseGames (PlayService.java:53-76): the search runs again on every position of a finished
    episode, the original moves are kept (GumbelSearch.java:365-370), no playout policy is added, and the values of the
    initial inference repeat exactly (same weights, same positions)."""
    n = 128
    r = _play(tmp_path, n, "HYBRID", "fp32", mode="reanalyse")
    assert r["games"] == n and r["not_done"] == 0 and r["illegal_actions"] == 0
    assert r["replay_action_mismatch"] == 0
    assert r["replay_length_mismatch"] == 0
    assert r["replay_playout_grew"] == 0
    assert r["replay_values_equal"] == n
    assert r["replay_targets_equal"] < n          # fresh Gumbel / Dirichlet noise: the policy targets are new
    assert r["simulations"] == 2 * 20 * (r["moves"] - r["one_option_moves"])


@pytest.mark.gpu
def test_just_replay_with_initial_inference(tmp_path):
    """PlayService.justReplayGamesWithInitialInference (PlayService.java:78-85, PlanAction.java:34-65)."""
    n = 64
    r = _play(tmp_path, n, "PLAYOUT", "fp32", mode="justreplay")
    assert r["replay_action_mismatch"] == 0 and r["replay_length_mismatch"] == 0
    assert r["replay_values_equal"] == n


@pytest.mark.gpu
def test_just_initial_inference_policy(tmp_path):
    """PlanAction.playAfterJustWithInitialInference (PlanAction.java:174-193): moves drawn from the renormalised priors."""
    n = 256
    r = _play(tmp_path, n, "PLAYOUT", "bf16", mode="initial")
    assert r["games"] == n and r["not_done"] == 0 and r["illegal_actions"] == 0
    assert r["bad_lengths"] == 0 and r["bad_policy_targets"] == 0
    assert r["simulations"] == 0
