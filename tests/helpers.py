"""Seeded synthetic inputs shared by the parity tests (same inputs go to the HIP path and the oracle)."""
import numpy as np

from muzero_amd import netspec

# search configs at BASELINE.json's shapes: (A, n, m, c_visit, bounds, two_players)
SEARCH_CASES = {
    "tictactoe": dict(A=9, n=20, m=4, c_visit=20, bounds=(-1.0, 1.0), two_players=True, vscale=1.0),
    "go5": dict(A=26, n=30, m=16, c_visit=50, bounds=(-25.0, 25.0), two_players=True, vscale=25.0),
    "go9": dict(A=82, n=50, m=16, c_visit=50, bounds=(-81.0, 81.0), two_players=True, vscale=81.0),
    "pegsolitair": dict(A=196, n=100, m=16, c_visit=50, bounds=None, two_players=False, vscale=25.0),
}


def random_net_outputs(rng, shape, A, vscale, sharp=1.0):
    """fp32 logits, fp32 softmax (computed in fp32 like libtorch would), fp64 value = vscale * fp32 v."""
    lg = (sharp * rng.standard_normal(shape + (A,))).astype(np.float32)
    e = np.exp(lg - lg.max(-1, keepdims=True), dtype=np.float32)
    p = (e / e.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
    v = np.tanh(rng.standard_normal(shape)).astype(np.float32)
    return lg, p, vscale * v.astype(np.float64)


def random_legal(rng, B, A, special=True):
    """Legal masks: mostly dense, some sparse, plus the edge cases the reference handles
    (one legal action -> shortcut; two; hash-bucket collisions; all legal)."""
    legal = (rng.uniform(size=(B, A)) < 0.7).astype(np.uint8)
    for g in range(B):
        if legal[g].sum() < 2:
            legal[g, rng.choice(A, 2, replace=False)] = 1
    if special and B >= 8:
        legal[0] = 1                                   # everything legal
        legal[1] = 0
        legal[1, A // 2] = 1                           # one legal action -> shortcut
        legal[2] = 0
        legal[2, [0, A - 1]] = 1                       # two legal actions
        legal[3] = 0
        legal[3, rng.choice(A, 3, replace=False)] = 1  # three
        legal[4] = 0
        idx = [i for i in range(A) if i % 16 == 5][:12]
        if len(idx) >= 2:
            legal[4, idx] = 1                          # same HashMap bucket at T=16 (insertion order matters)
        else:
            legal[4, :2] = 1
        legal[5] = 0
        legal[5, :min(A, 13)] = 1                      # 13 entries force T=32
        legal[6] = (rng.uniform(size=A) < 0.15).astype(np.uint8)
        if legal[6].sum() < 2:
            legal[6, :2] = 1
    return legal


def gumbel_noise(rng, B, A):
    u = rng.uniform(size=(B, A))
    u[u == 0] = 0.5
    return -np.log(-np.log(u))


def dirichlet_noise(rng, legal, alpha=0.3):
    """A Dirichlet sample over each game's legal actions, stored by ACTION index (0 elsewhere)."""
    B, A = legal.shape
    out = np.zeros((B, A), np.float64)
    for g in range(B):
        idx = np.nonzero(legal[g])[0]
        out[g, idx] = rng.dirichlet(np.full(len(idx), alpha))
    return out


def synthetic_obs(rng, spec, B, density=0.3):
    return (rng.uniform(size=(B, spec.obs_planes, spec.height, spec.width)) < density).astype(np.float32)


def tree_equal(a, b):
    """Bitwise comparison of two node dumps (oracle.search.OracleBatch.node / capi.Context.node)."""
    problems = []
    for k in ("parent_slot", "action", "to_play", "visit_count"):
        if a[k] != b[k]:
            problems.append((k, a[k], b[k]))
    for k in ("value", "vmix", "improved_value", "q_sum"):
        if np.float64(a[k]).tobytes() != np.float64(b[k]).tobytes():
            problems.append((k, a[k], b[k]))
    # the entropy twins go through log(): libm on the host vs ocml on the GPU, <= 1 ulp each
    for k in ("v_entropy_mix", "improved_entropy_value", "entropy_q_sum", "entropy_reward"):
        if not np.isclose(a[k], b[k], rtol=1e-12, atol=1e-13):
            problems.append((k, a[k], b[k]))
    for k in ("child_action", "child_visit", "child_slot"):
        if not np.array_equal(a[k], b[k]):
            problems.append((k, a[k], b[k]))
    for k in ("child_prior", "child_qsum"):
        if a[k].tobytes() != b[k].tobytes():
            problems.append((k, a[k], b[k]))
    return problems
