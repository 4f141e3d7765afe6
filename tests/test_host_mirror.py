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


def _play(tmp_path, n, play_type, precision, seed=1):
    spec = netspec.spec_tictactoe()
    blob = netspec.pack_blob(spec, netspec.random_weights(spec, seed=1))
    wfile = tmp_path / "ttt.f32"
    blob.astype(np.float32).tofile(wfile)
    out = subprocess.run([_binary("selfplay_main"), str(wfile), str(n), str(seed), play_type, precision],
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
