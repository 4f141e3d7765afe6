"""ctypes front end of oracle/libmzoracle.so (the CPU restatement of the reference search).

TEST INFRASTRUCTURE ONLY - see oracle/search_oracle.cpp for the reference citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmzoracle.so")
_lib = None


class OracleConfig(C.Structure):
    _fields_ = [
        ("action_space", C.c_int32),
        ("initial_gumbel_m", C.c_int32),
        ("c_visit", C.c_int32),
        ("two_players", C.c_int32),
        ("known_bounds", C.c_int32),
        ("hybrid", C.c_int32),
        ("gumbel_action_selection", C.c_int32),
        ("reserved", C.c_int32),
        ("c_scale", C.c_double),
        ("discount", C.c_double),
        ("bound_min", C.c_double),
        ("bound_max", C.c_double),
        ("root_exploration_fraction", C.c_double),
        ("temperature_root", C.c_double),
    ]


def build(force=False):
    src = os.path.join(_HERE, "search_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libmzoracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.mzo_batch_new.restype = C.c_void_p
        _lib.mzo_batch_new.argtypes = [C.POINTER(OracleConfig), C.c_int]
        _lib.mzo_batch_free.argtypes = [C.c_void_p]
        _lib.mzo_batch_begin.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5
        _lib.mzo_batch_select.argtypes = [C.c_void_p] * 4
        _lib.mzo_batch_select.restype = C.c_int
        _lib.mzo_batch_expand_backup.argtypes = [C.c_void_p] * 5
        _lib.mzo_batch_finish.argtypes = [C.c_void_p] * 17
        _lib.mzo_game_error.restype = C.c_char_p
        _lib.mzo_game_error.argtypes = [C.c_void_p, C.c_int]
        _lib.mzo_tree_num_nodes.argtypes = [C.c_void_p, C.c_int]
        _lib.mzo_tree_node.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 11
        _lib.mzo_minmax.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.mzo_extra_phase_visits.argtypes = [C.c_int, C.c_int, C.c_void_p]
        _lib.mzo_sh_trace.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _lib.mzo_hashmap_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.mzo_softmax.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
        _lib.mzo_draw_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.mzo_java_sum.argtypes = [C.c_void_p, C.c_int]
        _lib.mzo_java_sum.restype = C.c_double
        _lib.mzo_entropy.argtypes = [C.c_void_p, C.c_int]
        _lib.mzo_entropy.restype = C.c_double
        _lib.mzo_fast_rule_learning_move.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=dtype)


def make_config(action_space, initial_gumbel_m, c_visit, c_scale=1.0, discount=1.0, two_players=True,
                known_bounds=None, root_exploration_fraction=0.0, hybrid=False, gumbel_action_selection=True,
                temperature_root=1.0):
    cfg = OracleConfig()
    cfg.action_space = action_space
    cfg.initial_gumbel_m = initial_gumbel_m
    cfg.c_visit = c_visit
    cfg.two_players = 1 if two_players else 0
    cfg.known_bounds = 0 if known_bounds is None else 1
    cfg.hybrid = 1 if hybrid else 0
    cfg.gumbel_action_selection = 1 if gumbel_action_selection else 0
    cfg.c_scale = c_scale
    cfg.discount = discount
    cfg.bound_min, cfg.bound_max = (0.0, 0.0) if known_bounds is None else known_bounds
    cfg.root_exploration_fraction = root_exploration_fraction
    cfg.temperature_root = temperature_root
    return cfg


class OracleBatch:
    """B independent reference searches advanced in lock-step (the reference's batching policy:
    one batched forward per simulation step, ModelController.java:365-387)."""

    def __init__(self, cfg, B):
        self.cfg = cfg
        self.B = B
        self.A = cfg.action_space
        self._h = lib().mzo_batch_new(C.byref(cfg), B)

    def close(self):
        if self._h:
            lib().mzo_batch_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def begin(self, legal, to_play, n_sims, logits, policy, value, gumbel=None, dirichlet=None):
        B, A = self.B, self.A
        legal = _arr(legal, np.uint8)
        to_play = _arr(to_play, np.int32)
        logits = _arr(logits, np.float32)
        policy = _arr(policy, np.float32)
        value = _arr(value, np.float64)
        gumbel = _arr(gumbel, np.float64)
        dirichlet = _arr(dirichlet, np.float64)
        assert legal.shape == (B, A) and logits.shape == (B, A) and policy.shape == (B, A)
        if np.isscalar(n_sims):
            ns, nd = None, int(n_sims)
        else:
            ns, nd = _arr(n_sims, np.int32), 0
        lib().mzo_batch_begin(self._h, _p(legal), _p(to_play), _p(ns), nd, _p(logits), _p(policy), _p(value),
                              _p(gumbel), _p(dirichlet))

    def select(self):
        ps = np.empty(self.B, np.int32)
        ac = np.empty(self.B, np.int32)
        on = np.empty(self.B, np.uint8)
        cnt = lib().mzo_batch_select(self._h, _p(ps), _p(ac), _p(on))
        return cnt, ps, ac, on

    def expand_backup(self, active, logits, policy, value):
        active = _arr(active, np.uint8)
        logits = _arr(logits, np.float32)
        policy = _arr(policy, np.float32)
        value = _arr(value, np.float64)
        lib().mzo_batch_expand_backup(self._h, _p(active), _p(logits), _p(policy), _p(value))

    def finish(self, exploration_time=None, u=None):
        B, A = self.B, self.A
        out = dict(
            action=np.empty(B, np.int32), visit_counts=np.empty((B, A), np.int32),
            policy_target=np.empty((B, A), np.float32), root_value_target=np.empty(B, np.float32),
            v_mix=np.empty(B, np.float32), root_value_inference=np.empty(B, np.float32),
            entropy=np.empty(B, np.float32), legal_action_max_entropy=np.empty(B, np.float32),
            root_entropy_value_target=np.empty(B, np.float32), playout_policy=np.empty((B, A), np.float32),
            playout_appended=np.empty(B, np.uint8), selected_action=np.empty(B, np.int32),
            sims_run=np.empty(B, np.int32), status=np.empty(B, np.int32))
        et = _arr(exploration_time, np.uint8)
        uu = _arr(u, np.float64)
        lib().mzo_batch_finish(self._h, _p(et), _p(uu), _p(out["action"]), _p(out["visit_counts"]),
                               _p(out["policy_target"]), _p(out["root_value_target"]), _p(out["v_mix"]),
                               _p(out["root_value_inference"]), _p(out["entropy"]),
                               _p(out["legal_action_max_entropy"]), _p(out["root_entropy_value_target"]),
                               _p(out["playout_policy"]), _p(out["playout_appended"]), _p(out["selected_action"]),
                               _p(out["sims_run"]), _p(out["status"]))
        return out

    def error(self, g):
        return lib().mzo_game_error(self._h, g).decode()

    def num_nodes(self, g):
        return lib().mzo_tree_num_nodes(self._h, g)

    def node(self, g, slot):
        A = self.A
        ps, ac, tp, vc = (np.zeros(1, np.int32) for _ in range(4))
        sc = np.zeros(8, np.float64)
        ca, cv, cs = (np.zeros(A, np.int32) for _ in range(3))
        cp, cl, cq = (np.zeros(A, np.float64) for _ in range(3))
        n = lib().mzo_tree_node(self._h, g, slot, _p(ps), _p(ac), _p(tp), _p(vc), _p(sc), _p(ca), _p(cp), _p(cl),
                                _p(cv), _p(cq), _p(cs))
        if n < 0:
            raise IndexError(slot)
        return dict(parent_slot=int(ps[0]), action=int(ac[0]), to_play=int(tp[0]), visit_count=int(vc[0]),
                    value=sc[0], vmix=sc[1], improved_value=sc[2], q_sum=sc[3], v_entropy_mix=sc[4],
                    improved_entropy_value=sc[5], entropy_q_sum=sc[6], entropy_reward=sc[7],
                    child_action=ca[:n].copy(), child_prior=cp[:n].copy(), child_logit=cl[:n].copy(),
                    child_visit=cv[:n].copy(), child_qsum=cq[:n].copy(), child_slot=cs[:n].copy())

    def minmax(self, g):
        mn, mx = C.c_double(), C.c_double()
        lib().mzo_minmax(self._h, g, C.byref(mn), C.byref(mx))
        return mn.value, mx.value

    def run_injected(self, legal, to_play, n_sims, init_logits, init_policy, init_value, sim_logits, sim_policy,
                     sim_value, gumbel=None, dirichlet=None, exploration_time=None, u=None):
        """Whole move with pre-recorded network outputs: sim_* are [B, S, A] / [B, S], entry s is what
        recurrentInference returns for the leaf of simulation s of that game."""
        self.begin(legal, to_play, n_sims, init_logits, init_policy, init_value, gumbel, dirichlet)
        S = sim_value.shape[1]
        s = 0
        while True:
            cnt, ps, ac, on = self.select()
            if cnt == 0:
                break
            assert s < S, "more simulation steps than recorded network outputs"
            self.expand_backup(on, sim_logits[:, s], sim_policy[:, s], sim_value[:, s])
            s += 1
        return self.finish(exploration_time, u)


# --- known-answer helpers ---------------------------------------------------------------------
def extra_phase_visits(budget, m):
    out = np.zeros(64, np.int32)
    n = lib().mzo_extra_phase_visits(budget, m, _p(out))
    return out[:n].tolist()


def phases_for(m):
    return lib().mzo_phases_for(m)


def sh_count_next(n, m, k):
    return lib().mzo_sh_count_next(n, m, k)


def sh_trace(n, m, k, max_steps=4096):
    out = np.zeros((max_steps, 4), np.int32)
    c = lib().mzo_sh_trace(n, m, k, _p(out), max_steps)
    return out[:c].copy()


def hashmap_order(keys):
    keys = np.ascontiguousarray(keys, np.int32)
    out = np.zeros(len(keys), np.int32)
    n = lib().mzo_hashmap_order(_p(keys), len(keys), _p(out))
    if n < 0:
        raise ValueError("treeified bin")
    return out[:n].tolist()


def softmax(raw, temperature=1.0):
    raw = np.ascontiguousarray(raw, np.float64)
    out = np.zeros_like(raw)
    lib().mzo_softmax(_p(raw), len(raw), float(temperature), _p(out))
    return out


def draw_actions(actions, x, n):
    actions = np.ascontiguousarray(actions, np.int32)
    x = np.ascontiguousarray(x, np.float64)
    out = np.zeros(max(n, 1), np.int32)
    r = lib().mzo_draw_actions(_p(actions), _p(x), len(x), n, _p(out))
    if r < 0:
        raise ValueError("n > len")
    return out[:r].tolist()


def java_sum(v):
    v = np.ascontiguousarray(v, np.float64)
    return lib().mzo_java_sum(_p(v), len(v))


def entropy(p):
    p = np.ascontiguousarray(p, np.float32)
    return lib().mzo_entropy(_p(p), len(p))


def play_after_just_initial_inference(legal, policy, logits, u):
    """PlanAction.playAfterJustWithInitialInference -> (action, policy_target float32[A])."""
    legal = np.ascontiguousarray(legal, np.uint8)
    policy = np.ascontiguousarray(policy, np.float32)
    logits = np.ascontiguousarray(logits, np.float32)
    A = legal.size
    pt = np.zeros(A, np.float32)
    rvt, vmix = C.c_float(), C.c_float()
    L = lib()
    L.mzo_play_after_just_initial_inference.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                                        C.c_void_p, C.c_void_p, C.c_void_p]
    a = L.mzo_play_after_just_initial_inference(_p(legal), _p(policy), _p(logits), A, float(u), _p(pt), C.byref(rvt),
                                                C.byref(vmix))
    if a < 0:
        raise ValueError("no legal action")
    return a, pt


def fast_rule_learning_move(legal, r):
    """fastRuleLearning root -> (action, policy_target float32[A]); r = the nextInt(0, L - 1) draw."""
    legal = np.ascontiguousarray(legal, np.uint8)
    pt = np.zeros(legal.size, np.float32)
    a = lib().mzo_fast_rule_learning_move(_p(legal), legal.size, int(r), _p(pt))
    if a < 0:
        raise ValueError("bad draw or no legal action")
    return a, pt
