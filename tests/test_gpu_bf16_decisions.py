"""-m gpu: what the bf16 network path costs in the currency north_star uses - decisions.

(1) The same >= 1024 Go 9x9 positions, weights and noise searched through the fp32 context (the reference's
    arithmetic) and the bf16 MFMA context (the default): fraction of identical selected actions, L1 distance of root
    visit counts, root value targets.  Random weights give nearly flat priors and small value differences between
    children, i.e. MANY near-ties - a harder case for agreement than a trained net.
(2) A depth-5 unroll (initialInference + 5 chained recurrentInference steps, hidden states never leaving the device)
    against oracle/net.py chained in fp32 - errors compound as they do along a search path (the single-step test of
    test_gpu_net_parity.py restarts the reference from the device state).
Parity unpinned: the reference holds no activations; oracle/net.py is the restatement.
The measured numbers are printed (pytest -s) and recorded in DESIGN.md §2.
"""
import numpy as np
import pytest

from muzero_amd import capi, netspec
from oracle.net import NetOracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


#                 same selected_action, same candidate set, visit L1 mean, d rvt max / mean, d root inference max
BOUNDS = {capi.PRECISION_BF16: (0.95, 0.92, 4.0, 3.5e-2, 6e-3, 3.5e-2),
          capi.PRECISION_F16: (0.99, 0.99, 1.0, 3.5e-2, 1e-3, 6e-3)}   # (a changed decision moves that game's target)


@pytest.mark.parametrize("prec16", [capi.PRECISION_BF16, capi.PRECISION_F16])
def test_bf16_and_fp32_searches_agree_on_decisions_go9(prec16):
    spec = netspec.spec_go9()
    c = H.SEARCH_CASES["go9"]
    B, A, n = 1024, spec.action_space, c["n"]
    rng = np.random.default_rng(77)
    blob = netspec.pack_blob(spec, netspec.random_weights(spec, seed=3))
    obs = H.synthetic_obs(rng, spec, B)
    legal = np.ones((B, A), np.uint8)
    g = H.gumbel_noise(rng, B, A)
    res = {}
    for prec in (capi.PRECISION_FP32, prec16):
        ctx = capi.Context(capi.make_config(spec, n, c["m"], c["c_visit"], known_bounds=c["bounds"], max_games=B,
                                            precision=prec))
        ctx.load_weights(blob)
        ctx.set_action_cells(netspec.action_cells_board(spec))
        res[prec] = ctx.search_moves(B, obs, legal, gumbel=g)
        ctx.close()
    f, b = res[capi.PRECISION_FP32], res[prec16]
    tag = "bf16" if prec16 == capi.PRECISION_BF16 else "fp16"
    same_sel = float(np.mean(f["selected_action"] == b["selected_action"]))
    l1 = np.abs(f["visit_counts"] - b["visit_counts"]).sum(1)
    dv = np.abs(f["root_value_target"] - b["root_value_target"]) / 81.0
    dvi = np.abs(f["root_value_inference"] - b["root_value_inference"]) / 81.0
    dpt = np.abs(f["policy_target"] - b["policy_target"]).sum(1)
    # the m candidates are decided by logit + g at the root: how often is that set identical
    same_cand = float(np.mean(((f["visit_counts"] > 0) == (b["visit_counts"] > 0)).all(1)))
    print(f"\n{tag} vs fp32 search, {B} Go 9x9 games, n={n}: identical selected_action {same_sel:.4f}, "
          f"identical candidate set {same_cand:.4f}, visit-count L1 mean {l1.mean():.3f} (of {2 * n}) max {l1.max()}, "
          f"|d root_value_target|/81 mean {dv.mean():.5f} max {dv.max():.5f}, "
          f"|d root_value_inference|/81 mean {dvi.mean():.5f} max {dvi.max():.5f}, policy-target L1 mean {dpt.mean():.4f}")
    b_sel, b_cand, b_l1, b_dv, b_dvm, b_dvi = BOUNDS[prec16]
    assert same_sel >= b_sel
    assert same_cand >= b_cand
    assert l1.mean() <= b_l1
    assert dv.max() <= b_dv and dv.mean() <= b_dvm
    assert dvi.max() <= b_dvi


@pytest.mark.parametrize("prec16", [capi.PRECISION_BF16, capi.PRECISION_F16])
@pytest.mark.parametrize("name,factory", [("go9", netspec.spec_go9), ("tictactoe", netspec.spec_tictactoe)])
def test_bf16_depth5_unroll_against_fp32_oracle(name, factory, prec16):
    spec = factory()
    B, D = 12, 5
    rng = np.random.default_rng(13)
    w = netspec.random_weights(spec, seed=13, ln_jitter=0.1)
    cells = netspec.action_cells_board(spec)
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=B, precision=prec16, pool_slots=(D + 1) * B))
    bf = prec16 == capi.PRECISION_BF16
    #      hidden max, hidden mean, logits max, policy max, value max   (every depth: the rescale keeps errors from growing)
    tol = (1.2e-2, 2.5e-3, 5e-2, 3e-3, 3e-2) if bf else (2e-3, 4e-4, 8e-3, 1e-3, 6e-3)
    ctx.load_weights(netspec.pack_blob(spec, w))
    ctx.set_action_cells(cells)
    ref = NetOracle(spec, w)
    obs = H.synthetic_obs(rng, spec, B)
    slots = np.arange(B, dtype=np.int32)
    lg, pol, val = ctx.initial_inference(obs, slots)
    r_state, r_lg, r_pol, r_val = ref.initial_inference(obs)
    rows = []
    for d in range(D + 1):
        if d > 0:
            action = rng.integers(0, spec.action_space, B).astype(np.int32)
            lg, pol, val = ctx.recurrent_inference(slots + (d - 1) * B, action, slots + d * B)
            r_state, r_lg, r_pol, r_val = ref.recurrent_inference(r_state, action, cells)   # the oracle's OWN state
        hid = np.stack([ctx.read_hidden(s) for s in slots + d * B])
        eh = np.abs(hid - r_state.numpy())
        el = np.abs(lg - r_lg.numpy())
        ep = np.abs(pol - r_pol.numpy())
        ev = np.abs(val - r_val.numpy()) / spec.value_scale
        rows.append((d, eh.max(), eh.mean(), el.max(), ep.max(), ev.max(), ev.mean()))
    print(f"\n{name} {'bf16' if bf else 'fp16'} unroll vs fp32 oracle (depth, hidden max, hidden mean, logits max, policy max, value max, value mean):")
    for r in rows:
        print("  %d  %.4f %.5f  %.4f  %.5f  %.4f %.4f" % r)
    for d, hmax, hmean, lmax, pmax, vmax, vmean in rows:
        assert hmax <= tol[0] and hmean <= tol[1] and lmax <= tol[2] and pmax <= tol[3] and vmax <= tol[4], (d, rows)
    ctx.close()
