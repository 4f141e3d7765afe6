"""-m gpu: initialInference / recurrentInference of the HIP path against the fp32 torch-CPU restatement
of the reference blocks (oracle/net.py), through the C ABI's inference-level entry points.

Tolerances (north_star: "network outputs must match the reference DJL forward within a stated
fp32/bf16 tolerance"; the reference's own cross-backend precedent is allClose(1e-2, 1e-2),
onnx/.../BlockTestHelper.java:36):
  fp32 path: |diff| <= 2e-4 + 2e-4*|ref| on hidden state, logits, policy; value/value_scale likewise.
  16-bit MFMA paths: bars from the error model of DESIGN.md §2 (error_bars below), not from the last outlier:
  every error is proportional to the unit roundoff u of the operand type (2^-11 fp16, 2^-8 bf16), with one
  constant per output that holds for all four games and both types (profiles/r3_fuzz_net.txt).
"""
import numpy as np
import pytest

from muzero_amd import capi, netspec
from oracle.net import NetOracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

# Error model of the 16-bit MFMA paths.  Each conv rounds its activations and weights to the operand
# type: relative errors uniform in [-u, u], so a conv output carries 0.82 u (= u sqrt(2/3)) of its own rms as noise,
# whatever its K.  LayerNorm passes relative noise on unchanged, and the residual stream grows with depth as fast as the
# noise the blocks add, so the tower's output state carries the noise of an EFFECTIVE 6-7 convs, 2.1 u relative rms, at any
# depth (measured: the same for 6, 12 and 16 blocks).  The min-max rescale divides by the state's range, 2 sqrt(2 ln N) = 8-9
# standard deviations for N = C*H*W values: hidden state 0.25 u.  Heads (fp32 arithmetic on that state, one more LayerNorm each):
# logits 0.9 u, value / value_scale 1.3 u, policy (softmax slope <= 1/4) 0.25 u.  Those are rms figures of |diff| / (1 + |ref|);
# tools/fuzz_net_stats.py measures 0.22-0.27, 0.7-0.9, 1.0-1.4 and <= 0.12 u on both operand types and all four games.  The
# distribution has long tails (boards whose state has a small range): worst / rms is 9-16 over 200 networks x 2 steps, so the
# bars are K_TAIL = 20 times the rms figure; the fp16 bars are additionally capped at the reference's own cross-backend bar 1e-2.
UNIT_ROUNDOFF = {capi.PRECISION_F16: 2.0 ** -11, capi.PRECISION_BF16: 2.0 ** -8}
RMS_IN_U = dict(h=0.25, l=0.9, p=0.25, v=1.3)
K_TAIL = 20.0


def error_bars(precision):
    if precision == capi.PRECISION_FP32:
        return dict(h=2e-4, l=2e-4, p=2e-4, v=2e-4)
    cap = 1e-2 if precision == capi.PRECISION_F16 else np.inf
    return {k: min(K_TAIL * c * UNIT_ROUNDOFF[precision], cap) for k, c in RMS_IN_U.items()}


# fp16: h 2.4e-3, l 8.8e-3, p 2.4e-3, v 1e-2 (capped from 1.27e-2); bf16: h 2.0e-2, l 7.0e-2, p 2.0e-2, v 0.10
TOL = {p: error_bars(p) for p in (capi.PRECISION_FP32, capi.PRECISION_BF16, capi.PRECISION_F16)}

def _specs():
    return {
        "tiny_bcast": netspec.spec_tiny(),                                    # 12 channels, broadcast block at idx 1
        "tiny_single": netspec.spec_tiny(two_players=False, action_space=20, hw=(4, 5), obs=2, act=3,
                                         residuals=3, broadcast_every_n=0, channels=8, bottleneck=4),
        # shapes only the fp32 path takes: channel counts in multiples of 16 = its Q16 order and MFMA convs (net.h) with one and
        # two row tiles, a ragged input-channel count, two column tiles, a broadcast block; a net whose bottleneck is not a
        # multiple of 16 (everything on the vector ALU in [channel][cell] order); and a Q16 net whose C-wide LayerNorms
        # (176 x 70 / 4 = 3080 float4 items) are beyond the register path of ln_relu while its bottleneck ones are inside it
        "tiny_fp32_mfma": netspec.spec_tiny(action_space=21, hw=(4, 5), obs=5, act=2, residuals=3, broadcast_every_n=2,
                                            channels=32, bottleneck=16),
        "tiny_fp32_mixed": netspec.spec_tiny(two_players=False, action_space=10, hw=(3, 3), obs=3, act=1, residuals=2,
                                             broadcast_every_n=0, channels=16, bottleneck=8),
        "fp32_q16_wide": netspec.spec_tiny(action_space=70, hw=(3, 23), obs=3, act=1, residuals=2, broadcast_every_n=2,
                                           channels=176, bottleneck=64),
        "tictactoe": netspec.spec_tictactoe(),
        "go5": netspec.spec_go5(),
        "go9": netspec.spec_go9(),
        "pegsolitair": netspec.spec_pegsolitair(),
    }


def _check(spec, precision, B, seed, ln_jitter=0.1, weights=None):
    rng = np.random.default_rng(seed)
    w = weights if weights is not None else netspec.random_weights(spec, seed=seed, ln_jitter=ln_jitter)
    blob = netspec.pack_blob(spec, w)
    cells = netspec.action_cells_board(spec)
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=B, precision=precision, pool_slots=2 * B))
    assert ctx.num_params() == netspec.num_params(spec) == blob.size
    ctx.load_weights(blob)
    ctx.set_action_cells(cells)
    ref = NetOracle(spec, w)
    tol = TOL[precision]

    obs = H.synthetic_obs(rng, spec, B)
    slots0 = np.arange(B, dtype=np.int32)
    lg, pol, val = ctx.initial_inference(obs, slots0)
    r_state, r_lg, r_pol, r_val = ref.initial_inference(obs)
    hid = np.stack([ctx.read_hidden(s) for s in slots0])
    np.testing.assert_allclose(hid, r_state.numpy(), rtol=tol["h"], atol=tol["h"])
    np.testing.assert_allclose(lg, r_lg.numpy(), rtol=tol["l"], atol=tol["l"])
    np.testing.assert_allclose(pol, r_pol.numpy(), rtol=tol["p"], atol=tol["p"])
    np.testing.assert_allclose(val / spec.value_scale, r_val.numpy() / spec.value_scale, rtol=tol["v"], atol=tol["v"])
    assert np.abs(pol.sum(1) - 1).max() < 1e-5
    assert hid.min() >= 0.0 and hid.max() <= 1.0 + 1e-6          # RescaleBlockExt range

    # recurrent step from the device-resident hidden states, including the last action (Go: pass -> zero plane)
    action = rng.integers(0, spec.action_space, B).astype(np.int32)
    action[0] = spec.action_space - 1
    slots1 = slots0 + B
    lg2, pol2, val2 = ctx.recurrent_inference(slots0, action, slots1)
    # reference step from the DEVICE hidden state so that errors do not compound across the two calls
    r_state2, r_lg2, r_pol2, r_val2 = ref.recurrent_inference(hid, action, cells)
    hid2 = np.stack([ctx.read_hidden(s) for s in slots1])
    np.testing.assert_allclose(hid2, r_state2.numpy(), rtol=tol["h"], atol=tol["h"])
    np.testing.assert_allclose(lg2, r_lg2.numpy(), rtol=tol["l"], atol=tol["l"])
    np.testing.assert_allclose(pol2, r_pol2.numpy(), rtol=tol["p"], atol=tol["p"])
    np.testing.assert_allclose(val2 / spec.value_scale, r_val2.numpy() / spec.value_scale, rtol=tol["v"],
                               atol=tol["v"])
    ctx.close()


@pytest.mark.parametrize("name", ["tiny_bcast", "tiny_single", "tiny_fp32_mfma", "tiny_fp32_mixed", "fp32_q16_wide", "tictactoe",
                                  "go5", "go9", "pegsolitair"])
def test_fp32_forward_matches_oracle(name):
    _check(_specs()[name], capi.PRECISION_FP32, B=6, seed=3)


def test_fp32_forward_zero_input():
    # the reference's BlockTest also feeds zeros (BlockTest.java:35-147)
    spec = netspec.spec_tiny()
    w = netspec.random_weights(spec, seed=5, ln_jitter=0.1)
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=2, pool_slots=4))
    ctx.load_weights(netspec.pack_blob(spec, w))
    obs = np.zeros((2, spec.obs_planes, spec.height, spec.width), np.float32)
    lg, pol, val = ctx.initial_inference(obs, np.array([0, 1], np.int32))
    r = NetOracle(spec, w).initial_inference(obs)
    np.testing.assert_allclose(lg, r[1].numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(val, r[3].numpy(), rtol=2e-4, atol=2e-4)
    ctx.close()


def test_weight_blob_size_is_checked():
    spec = netspec.spec_tiny()
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=1, pool_slots=2))
    with pytest.raises(capi.MuZeroError):
        ctx.load_weights(np.zeros(17, np.float32))
    with pytest.raises(capi.MuZeroError):       # inference before weights are loaded
        ctx.initial_inference(np.zeros((1, 3, 3, 3), np.float32), np.array([0], np.int32))
    ctx.close()


@pytest.mark.parametrize("name", ["go9", "pegsolitair", "tictactoe", "go5"])
def test_bf16_mfma_forward_matches_oracle(name):
    # B=7: two workgroup tiles for Go 9x9 (3 boards each) plus a ragged last tile
    _check(_specs()[name], capi.PRECISION_BF16, B=7, seed=3)


@pytest.mark.parametrize("name", ["go9", "pegsolitair", "tictactoe", "go5"])
def test_f16_mfma_forward_matches_oracle(name):
    _check(_specs()[name], capi.PRECISION_F16, B=7, seed=3)


def _scaled_affine(spec, seed, gain, shift):
    w = netspec.random_weights(spec, seed=seed, ln_jitter=0.1)
    for name in w:
        if ".ln" in name and name.split(".")[0] in ("rep", "dyn"):
            if name.endswith(".g"):
                w[name] = (w[name] * gain).astype(np.float32)
            elif name.endswith(".b"):
                w[name] = (w[name] + shift).astype(np.float32)
    return w


@pytest.mark.parametrize("name", ["go9", "tictactoe"])
def test_f16_large_layernorm_affine_takes_a_larger_activation_scale(name):
    """The fp16 path stores gamma / S and beta / S and clamps to [0, 1] as its ReLU (net_mfma.hip: ln_apply_pk, ln_scale): S is
    256 for the usual affine and grows with it.  gamma x 6, beta + 2 (S = 1024) must meet the same bars."""
    spec = _specs()[name]
    _check(spec, capi.PRECISION_F16, B=7, seed=21, weights=_scaled_affine(spec, 21, 6.0, 2.0))


def test_f16_refuses_an_affine_its_scaled_activations_cannot_hold():
    spec = _specs()["go9"]
    blob = netspec.pack_blob(spec, _scaled_affine(spec, 22, 300.0, 0.0))
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=2, precision=capi.PRECISION_F16, pool_slots=4))
    with pytest.raises(capi.MuZeroError, match="too large for the fp16-operand path"):
        ctx.load_weights(blob)
    ctx.close()
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=2, precision=capi.PRECISION_BF16, pool_slots=4))
    ctx.load_weights(blob)      # the bf16 path has no such limit
    ctx.close()


@pytest.mark.parametrize("env,name", [({"MZ_FUSE": "0"}, "go9"), ({"MZ_FUSE": "1"}, "go9"),
                                      ({"MZ_FUSE": "0"}, "pegsolitair"), ({"MZ_FUSE": "1"}, "go5"),
                                      ({"MZ_FUSE": "1"}, "pegsolitair"),   # k_seg -> k_bseg -> k_bseg (default: k_tower2)
                                      ({"MZ_GEOM": "1"}, "go9"), ({"MZ_GEOM": "1", "MZ_FUSE": "0"}, "pegsolitair")])
def test_bf16_alternative_launch_chains_and_geometry(env, name, monkeypatch):
    """The diagnostic variants kept for A/B runs meet the same tolerance: separate k_seg / k_bcast
    launches (MZ_FUSE=0), broadcast block fused with the blocks behind it (MZ_FUSE=1; the default is the one-launch
    tower), and 4 waves x 2 column tiles per workgroup (MZ_GEOM=1).  Both are read when the context is created."""
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    _check(_specs()[name], capi.PRECISION_BF16, B=7, seed=4)


def test_bf16_mfma_matches_fp32_path_statistically():
    """Same weights through both HIP paths: the bf16 error stays at the bf16 rounding level."""
    spec = netspec.spec_go9()
    rng = np.random.default_rng(12)
    w = netspec.random_weights(spec, seed=12, ln_jitter=0.1)
    blob = netspec.pack_blob(spec, w)
    B = 12
    obs = H.synthetic_obs(rng, spec, B)
    res = []
    for prec in (capi.PRECISION_FP32, capi.PRECISION_BF16):
        ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=B, precision=prec, pool_slots=2 * B))
        ctx.load_weights(blob)
        ctx.set_action_cells(netspec.action_cells_board(spec))
        slots = np.arange(B, dtype=np.int32)
        out0 = ctx.initial_inference(obs, slots)
        act = rng.integers(0, spec.action_space, B).astype(np.int32)
        out1 = ctx.recurrent_inference(slots, act, slots + B)
        res.append((out0, out1, np.stack([ctx.read_hidden(s) for s in slots + B])))
        ctx.close()
    for i in range(2):
        np.testing.assert_allclose(res[1][i][0], res[0][i][0], rtol=3e-2, atol=3e-2)   # logits
        np.testing.assert_allclose(res[1][i][1], res[0][i][1], rtol=2e-2, atol=2e-2)   # policy
        dv = np.abs(res[1][i][2] - res[0][i][2]) / 81
        assert dv.max() < 8e-2 and dv.mean() < 2.5e-2, dv
    err = np.abs(res[1][2] - res[0][2])
    assert err.max() < 3e-2 and err.mean() < 4e-3, (err.max(), err.mean())
