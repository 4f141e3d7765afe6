"""Shape of the representation / dynamics / prediction nets and the canonical weight blob.

The blob is this library's own documented format (import of DJL .params checkpoints
is "next" row N2): a flat little-endian fp32 array holding the tensors listed by `tensor_list()` in
order, each row-major with the shape given there.  The block structure follows the reference:
  RepresentationBlock.java:38-40, DynamicsBlock.java:40-42, StartResidualBlock.java:53-62,
  ResidualTower.java:30-40, BottleneckResidualBlock.java:55-79, BroadcastResidualBlock.java:56-77,
  PredictionBlock.java:46-84 (paths under platform/.../agent/d_model/djl/blocks/).
"""
from dataclasses import dataclass, asdict

import numpy as np


@dataclass(frozen=True)
class NetSpec:
    height: int
    width: int
    obs_planes: int          # O
    action_planes: int       # a
    channels: int            # C  numChannels
    bottleneck: int          # Cb numBottleneckChannels
    residuals: int           # R  numResiduals
    broadcast_every_n: int   # block i is a broadcast block iff i % n == n-1 (ResidualTower.java:33)
    action_space: int        # A
    two_players: bool        # tanh on the value head (PredictionBlock.java:55-57)
    value_scale: float       # valueSpan / 2 (InitialInferenceListTranslator.java:94)

    @property
    def hw(self):
        return self.height * self.width

    def is_broadcast(self, i):
        n = self.broadcast_every_n
        return n > 0 and i % n == n - 1

    def to_dict(self):
        return asdict(self)


# BASELINE.json configs.  PegSolitair's yml is stale; Cb=64 / broadcastEveryN=8 are chosen.
def spec_tictactoe():
    return NetSpec(3, 3, 3, 1, 256, 128, 6, 8, 9, True, 1.0)


def spec_go5():
    return NetSpec(5, 5, 17, 1, 256, 128, 12, 8, 26, True, 25.0)


def spec_go9():
    return NetSpec(9, 9, 17, 1, 128, 64, 12, 8, 82, True, 81.0)


def spec_pegsolitair():
    return NetSpec(7, 7, 1, 4, 128, 64, 16, 8, 196, False, 25.0)


def spec_tiny(two_players=True, action_space=9, hw=(3, 3), obs=3, act=1, residuals=2, broadcast_every_n=2,
              channels=12, bottleneck=6, value_scale=1.0):
    """application-test.yml sized net (numChannels: 12) for CPU-side tests."""
    return NetSpec(hw[0], hw[1], obs, act, channels, bottleneck, residuals, broadcast_every_n, action_space,
                   two_players, value_scale)


SPECS = {"tictactoe": spec_tictactoe, "go5": spec_go5, "go9": spec_go9, "pegsolitair": spec_pegsolitair}


def _tower(prefix, s):
    C, Cb, HW = s.channels, s.bottleneck, s.hw
    out = []
    for i in range(s.residuals):
        p = f"{prefix}.b{i}"
        if s.is_broadcast(i):
            out += [(p + ".ln1.g", (C, HW)), (p + ".ln1.b", (C, HW)), (p + ".c1", (C, C)),
                    (p + ".ln2.g", (C, HW)), (p + ".ln2.b", (C, HW)), (p + ".bw", (HW, HW)), (p + ".bb", (HW,)),
                    (p + ".ln3.g", (C, HW)), (p + ".ln3.b", (C, HW)), (p + ".c2", (C, C))]
        else:
            out += [(p + ".ln1.g", (C, HW)), (p + ".ln1.b", (C, HW)), (p + ".c1", (Cb, C)),
                    (p + ".ln2.g", (Cb, HW)), (p + ".ln2.b", (Cb, HW)), (p + ".c2", (Cb, Cb, 3, 3)),
                    (p + ".ln3.g", (Cb, HW)), (p + ".ln3.b", (Cb, HW)), (p + ".c3", (Cb, Cb, 3, 3)),
                    (p + ".ln4.g", (Cb, HW)), (p + ".ln4.b", (Cb, HW)), (p + ".c4", (C, Cb))]
    return out


def tensor_list(s):
    """[(name, shape)] in blob order."""
    C, HW, A = s.channels, s.hw, s.action_space
    t = [("rep.conv0", (C, s.obs_planes, 3, 3))]
    t += _tower("rep", s)
    t += [("dyn.conv0", (C, C + s.action_planes, 3, 3))]
    t += _tower("dyn", s)
    t += [("pred.p.conv", (2, C)), ("pred.p.ln.g", (2, HW)), ("pred.p.ln.b", (2, HW)),
          ("pred.p.fc.w", (A, 2 * HW)), ("pred.p.fc.b", (A,)),
          ("pred.v.conv", (1, C)), ("pred.v.ln.g", (1, HW)), ("pred.v.ln.b", (1, HW)),
          ("pred.v.fc1.w", (C, HW)), ("pred.v.fc1.b", (C,)), ("pred.v.fc2.w", (1, C)), ("pred.v.fc2.b", (1,))]
    return t


def num_params(s):
    return int(sum(int(np.prod(shape)) for _, shape in tensor_list(s)))


def random_weights(s, seed=1, ln_jitter=0.0):
    """SURVEY.md §8d synthetic weights: conv/linear ~ N(0, 1/fan_in), LN gamma=1, beta=0 (optionally
    jittered so that tests exercise the affine terms), biases ~ N(0, 0.01)."""
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in tensor_list(s):
        leaf = name.split(".")[-1]
        if leaf == "g":
            v = np.ones(shape, np.float32)
            if ln_jitter:
                v += (ln_jitter * rng.standard_normal(shape)).astype(np.float32)
        elif leaf == "b" and ".ln" in name:
            v = np.zeros(shape, np.float32)
            if ln_jitter:
                v += (ln_jitter * rng.standard_normal(shape)).astype(np.float32)
        elif leaf in ("b", "bb"):
            v = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        w[name] = v
    return w


def pack_blob(s, weights):
    parts = []
    for name, shape in tensor_list(s):
        v = np.ascontiguousarray(weights[name], np.float32)
        assert v.shape == tuple(shape), (name, v.shape, shape)
        parts.append(v.ravel())
    return np.concatenate(parts)


def unpack_blob(s, blob):
    blob = np.asarray(blob, np.float32)
    w, off = {}, 0
    for name, shape in tensor_list(s):
        n = int(np.prod(shape))
        w[name] = blob[off:off + n].reshape(shape)
        off += n
    assert off == blob.size
    return w


def action_cells_board(s, with_pass=False):
    """action index -> flattened cell in the [a, H, W] action planes, or -1 for an all-zero plane.
    TicTacToeAction.java:57-61 / GoAction.java:67-73 (the pass move encodes as zeros)."""
    cells = np.arange(s.action_space, dtype=np.int32)
    cells[cells >= s.hw * s.action_planes] = -1
    if with_pass:
        cells[s.hw:] = -1
    return cells


def flops_per_sample(s):
    """Algorithmic FLOPs = 2 x MACs of conv / linear / broadcast layers only."""
    C, Cb, HW, A = s.channels, s.bottleneck, s.hw, s.action_space
    tower = 0
    for i in range(s.residuals):
        tower += (HW * 2 * C * C + C * HW * HW) if s.is_broadcast(i) else HW * (2 * C * Cb + 18 * Cb * Cb)
    heads = HW * (2 * C + C) + HW * C + 2 * HW * A + C
    rec = 9 * (C + s.action_planes) * C * HW + tower + heads
    ini = 9 * s.obs_planes * C * HW + tower + heads
    return {"recurrent": 2 * rec, "initial": 2 * ini}
