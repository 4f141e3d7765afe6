"""-m gpu: mz_search_moves_injected_noalloc - the entry point for callers that cannot allocate (graph capture, tight
serving loops) - against mz_search_moves_injected, through the C ABI.  Its trees and per-game state live in a
file-scope __device__ array of the library (one per GPU, shared by all calls, keeping the last call's contents), so
the check runs two calls in a row on one context, the second with a smaller batch and other data, and asks for
bit-identical outputs each time (integer outputs and fp32 narrowings of bit-exact fp64 sums; the exp/log-based
targets come from the same kernels, so they are bit-identical here too)."""
import numpy as np
import pytest

from muzero_amd import capi, netspec
from tests import helpers as H

pytestmark = pytest.mark.gpu

OUT_DTYPES = dict(action=np.int32, selected_action=np.int32, visit_counts=np.int32, policy_target=np.float32,
                  root_value_target=np.float32, v_mix=np.float32, root_value_inference=np.float32, entropy=np.float32,
                  legal_action_max_entropy=np.float32, root_entropy_value_target=np.float32, playout_policy=np.float32,
                  playout_appended=np.uint8, sims_run=np.int32, status=np.int32)


def _inputs(rng, c, B, frac):
    A, n = c["A"], c["n"]
    legal = H.random_legal(rng, B, A)
    il, ip, iv = H.random_net_outputs(rng, (B,), A, c["vscale"])
    sl, sp, sv = H.random_net_outputs(rng, (B, n), A, c["vscale"])
    return dict(legal=legal, to_play=rng.integers(0, 2, B).astype(np.uint8), il=il, ip=ip, iv=iv, sl=sl, sp=sp, sv=sv,
                g=H.gumbel_noise(rng, B, A), dn=H.dirichlet_noise(rng, legal) if frac else None,
                expl=rng.integers(0, 2, B).astype(np.uint8), u=rng.uniform(size=B),
                nsims=rng.integers(max(2, n // 3), n + 1, B).astype(np.int32))


def test_noalloc_entry_point_equals_the_allocating_one_twice_in_a_row():
    import torch
    dev = torch.device("cuda:0")
    c = H.SEARCH_CASES["go9"]
    A, n, frac = c["A"], c["n"], 0.25
    spec = netspec.spec_tiny(two_players=True, action_space=A, hw=(3, 3))
    Bmax = 96
    cfg = capi.make_config(spec, n, c["m"], c["c_visit"], known_bounds=c["bounds"], root_exploration_fraction=frac,
                           hybrid=True, gumbel_action_selection=False, max_games=Bmax)
    ctx = capi.Context(cfg)
    assert ctx.noalloc_scratch_bytes(Bmax) <= capi.NOALLOC_SCRATCH_BYTES
    rng = np.random.default_rng(41)
    for B, per_game_n in ((Bmax, False), (37, True)):      # the second call is smaller and reuses the same array
        x = _inputs(rng, c, B, frac)
        nsims = x["nsims"] if per_game_n else None
        want = ctx.search_moves_injected(B, x["legal"], x["il"], x["ip"], x["iv"], x["sl"], x["sp"], x["sv"],
                                         to_play=x["to_play"], num_simulations=nsims, gumbel=x["g"], dirichlet=x["dn"],
                                         exploration_time=x["expl"], u=x["u"])
        t = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if v is not None else None) for k, v in x.items()}
        outs = {k: torch.full((B, A) if k in ("visit_counts", "policy_target", "playout_policy") else (B,), 77,
                              dtype=getattr(torch, np.dtype(dt).name), device=dev) for k, dt in OUT_DTYPES.items()}
        torch.cuda.synchronize()
        ctx.search_moves_injected_noalloc(B, t["legal"], t["il"], t["ip"], t["iv"], t["sl"], t["sp"], t["sv"], n, outs,
                                          to_play=t["to_play"], num_simulations=t["nsims"] if per_game_n else None,
                                          gumbel=t["g"], dirichlet=t["dn"], exploration_time=t["expl"], u=t["u"])
        ctx.synchronize()
        assert (want["status"] == 0).all()
        for k in OUT_DTYPES:
            got = outs[k].cpu().numpy()
            if k == "playout_policy":                      # only defined where appended (neither path writes the rest)
                m = want["playout_appended"].astype(bool)
                assert got[m].tobytes() == want[k][m].tobytes(), k
            else:
                assert got.tobytes() == want[k].tobytes(), k
    ctx.close()


def test_noalloc_entry_point_rejects_what_it_cannot_serve():
    import torch
    dev = torch.device("cuda:0")
    c = H.SEARCH_CASES["pegsolitair"]                        # A = 196, n = 100: 0.65 MB of trees per game
    A, n = c["A"], c["n"]
    spec = netspec.spec_tiny(two_players=False, action_space=A, hw=(3, 3))
    B = 64
    ctx = capi.Context(capi.make_config(spec, n, c["m"], c["c_visit"], max_games=B))
    assert ctx.noalloc_scratch_bytes(B) > capi.NOALLOC_SCRATCH_BYTES
    rng = np.random.default_rng(2)
    x = _inputs(rng, c, B, 0.0)
    t = {k: (torch.from_numpy(np.ascontiguousarray(v)).to(dev) if v is not None else None) for k, v in x.items()}
    outs = dict(action=torch.zeros(B, dtype=torch.int32, device=dev))
    with pytest.raises(capi.MuZeroError, match="bytes of scratch"):
        ctx.search_moves_injected_noalloc(B, t["legal"], t["il"], t["ip"], t["iv"], t["sl"], t["sp"], t["sv"], n, outs,
                                          gumbel=t["g"])
    fits = 16
    assert ctx.noalloc_scratch_bytes(fits) <= capi.NOALLOC_SCRATCH_BYTES
    with pytest.raises(capi.MuZeroError, match="own device only"):   # a host buffer would need staging
        ctx.search_moves_injected_noalloc(fits, x["legal"][:fits].copy(), t["il"], t["ip"], t["iv"], t["sl"], t["sp"],
                                          t["sv"], n, outs, gumbel=t["g"])
    ctx.search_moves_injected_noalloc(fits, t["legal"], t["il"], t["ip"], t["iv"], t["sl"], t["sp"], t["sv"], n, outs,
                                      gumbel=t["g"])
    ctx.synchronize()
    want = ctx.search_moves_injected(fits, x["legal"][:fits], x["il"][:fits], x["ip"][:fits], x["iv"][:fits],
                                     x["sl"][:fits], x["sp"][:fits], x["sv"][:fits], gumbel=x["g"][:fits])
    np.testing.assert_array_equal(outs["action"].cpu().numpy()[:fits], want["action"])
    ctx.close()
