"""-m gpu: two contexts on one device driven from two host threads at once (one JVM thread per context).
Each thread's results must be those of the same calls made alone: contexts share nothing but the device (per-context
streams, buffers and error strings; the LDS opt-in state of net.h is per (kernel, device) under a lock)."""
import threading

import numpy as np
import pytest

from muzero_amd import capi, netspec
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _make(spec, case, B, seed, precision):
    c = H.SEARCH_CASES[case]
    ctx = capi.Context(capi.make_config(spec, c["n"], c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B,
                                        precision=precision))
    ctx.load_weights(netspec.pack_blob(spec, netspec.random_weights(spec, seed=seed)))
    ctx.set_action_cells(netspec.action_cells_board(spec))
    rng = np.random.default_rng(seed)
    args = dict(obs=H.synthetic_obs(rng, spec, B), legal=H.random_legal(rng, B, spec.action_space),
                to_play=rng.integers(0, 2, B).astype(np.uint8), gumbel=H.gumbel_noise(rng, B, spec.action_space))
    return ctx, args


def _moves(ctx, B, a, reps):
    out = []
    for _ in range(reps):
        got = ctx.search_moves(B, a["obs"], a["legal"], to_play=a["to_play"], gumbel=a["gumbel"])
        out.append((got["action"].copy(), got["visit_counts"].copy(), got["root_value_target"].copy()))
    return out


def test_two_contexts_two_threads_same_results_as_alone():
    jobs = [(netspec.spec_go9(), "go9", 48, 21, capi.PRECISION_F16),
            (netspec.spec_tictactoe(), "tictactoe", 200, 22, capi.PRECISION_FP32)]
    ctxs = [_make(*j) for j in jobs]
    alone = [_moves(ctx, j[2], a, 2) for (ctx, a), j in zip(ctxs, jobs)]
    together = [None, None]
    errors = []

    def run(i):
        try:
            together[i] = _moves(ctxs[i][0], jobs[i][2], ctxs[i][1], 2)
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=110)
    assert not errors, errors
    for i in range(2):
        for (a0, v0, r0), (a1, v1, r1) in zip(alone[i], together[i]):
            np.testing.assert_array_equal(a0, a1)
            np.testing.assert_array_equal(v0, v1)
            np.testing.assert_array_equal(r0, r1)
    for ctx, _ in ctxs:
        ctx.close()
