"""-m gpu: the weight hand-over and multi-GPU entry points of the C ABI.

* mz_weights_device_ptr + mz_commit_weights (the in-place path an RCCL broadcast lands in) must give
  exactly the network mz_load_weights gives - bit for bit, both precisions.
* mz_comm_* / mz_broadcast_weights / mz_gather_stats run at world size 1 on the one GPU of the test box
  (librccl bound at run time); with >= 2 GPUs visible a two-context / two-thread run broadcasts from
  device 0 to device 1 inside one process, which also covers the per-device LDS opt-in of the kernels.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from muzero_amd import capi, netspec
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _hip():
    return C.CDLL("libamdhip64.so")


def _memcpy_h2d(dst, arr):
    hip = _hip()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(dst, arr.ctypes.data, arr.nbytes, 1) == 0
    assert hip.hipDeviceSynchronize() == 0


def _infer(ctx, spec, obs, actions):
    B = obs.shape[0]
    lg0, p0, v0 = ctx.initial_inference(obs, np.arange(B))
    lg1, p1, v1 = ctx.recurrent_inference(np.arange(B), actions, np.arange(B, 2 * B))
    hid = ctx.read_hidden(B)
    return [lg0, p0, v0, lg1, p1, v1, hid]


@pytest.mark.parametrize("precision", [capi.PRECISION_FP32, capi.PRECISION_BF16])
@pytest.mark.parametrize("game", ["go9", "tictactoe"])
def test_commit_through_device_pointer_equals_load(precision, game):
    spec = netspec.SPECS[game]()
    B = 5
    rng = np.random.default_rng(3)
    blob = netspec.pack_blob(spec, netspec.random_weights(spec, seed=4))
    obs = H.synthetic_obs(rng, spec, B)
    actions = rng.integers(0, spec.action_space, B).astype(np.int32)
    cfg = capi.make_config(spec, 4, 2, 50, max_games=B, precision=precision, pool_slots=2 * B)
    a = capi.Context(cfg)
    a.set_action_cells(netspec.action_cells_board(spec))
    a.load_weights(blob)
    want = _infer(a, spec, obs, actions)
    a.close()

    b = capi.Context(cfg)
    b.set_action_cells(netspec.action_cells_board(spec))
    ptr, nbytes = b.weights_device_ptr()
    assert nbytes == blob.nbytes and b.num_params() == blob.size
    with pytest.raises(capi.MuZeroError, match="weights not loaded"):
        b.initial_inference(obs, np.arange(B))
    _memcpy_h2d(ptr, blob)          # what ncclBroadcast does on a non-root rank
    b.commit_weights()
    got = _infer(b, spec, obs, actions)
    for w, g in zip(want, got):
        assert w.tobytes() == g.tobytes()
    b.close()


def test_recurrent_inference_rejects_aliasing_slots():
    spec = netspec.spec_tiny()
    ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=4, pool_slots=8))
    ctx.load_weights(netspec.pack_blob(spec, netspec.random_weights(spec, seed=1)))
    obs = H.synthetic_obs(np.random.default_rng(0), spec, 4)
    ctx.initial_inference(obs, np.arange(4))
    acts = np.zeros(4, np.int32)
    with pytest.raises(capi.MuZeroError, match="distinct"):
        ctx.recurrent_inference([0, 1, 2, 3], acts, [4, 4, 5, 6])
    with pytest.raises(capi.MuZeroError, match="src_slot of another"):
        ctx.recurrent_inference([0, 1, 2, 3], acts, [1, 4, 5, 6])
    ctx.recurrent_inference([0, 1, 2, 3], acts, [0, 4, 5, 6])   # overwriting one's own parent is allowed
    ctx.close()


def test_comm_world_size_one():
    spec = netspec.spec_go5()
    blob = netspec.pack_blob(spec, netspec.random_weights(spec, seed=2))
    cfg = capi.make_config(spec, 4, 2, 50, max_games=2, precision=capi.PRECISION_BF16, pool_slots=2)
    ctx = capi.Context(cfg)
    with pytest.raises(capi.MuZeroError, match="mz_comm_init"):
        ctx.broadcast_weights(0)
    uid = capi.comm_unique_id()
    assert len(uid) == capi.COMM_ID_BYTES and any(uid)
    ctx.comm_init(uid, 0, 1)
    with pytest.raises(capi.MuZeroError, match="no weights"):
        ctx.broadcast_weights(0)
    ctx.load_weights(blob)
    ctx.broadcast_weights(0)
    got = ctx.gather_stats([3.0, 5.5, -1.0], 1)
    np.testing.assert_array_equal(got, [[3.0, 5.5, -1.0]])
    ctx.comm_barrier()
    obs = H.synthetic_obs(np.random.default_rng(0), spec, 2)
    lg, _, _ = ctx.initial_inference(obs, [0, 1])
    assert np.isfinite(lg).all()
    ctx.comm_destroy()
    ctx.close()


def test_two_devices_in_one_process_broadcast():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    spec = netspec.spec_go9()
    blob = netspec.pack_blob(spec, netspec.random_weights(spec, seed=6))
    B = 4
    obs = H.synthetic_obs(np.random.default_rng(1), spec, B)
    cfg = capi.make_config(spec, 4, 2, 50, max_games=B, precision=capi.PRECISION_BF16, pool_slots=B)
    ctxs = [capi.Context(cfg, device=d) for d in (0, 1)]
    ctxs[0].load_weights(blob)
    uid = capi.comm_unique_id()
    outs, errs = [None, None], []

    def run(r):
        try:
            ctxs[r].comm_init(uid, r, 2)
            ctxs[r].broadcast_weights(0)
            outs[r] = ctxs[r].initial_inference(obs, np.arange(B))
            st = ctxs[r].gather_stats([float(r + 1)], 2)
            assert st.ravel().tolist() == [1.0, 2.0]
            ctxs[r].comm_destroy()
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    [t.start() for t in ths]
    [t.join(120) for t in ths]
    assert not errs, errs
    for a, b in zip(outs[0], outs[1]):
        assert a.tobytes() == b.tobytes()
    [c.close() for c in ctxs]
