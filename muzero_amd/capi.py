"""ctypes binding of libmuzero_hip.so (include/muzero_hip.h) - the same entry points a Java FFM /
JNI adapter binds.  No CPU fallback: loading fails loudly if the HIP library is
missing, and mz_create fails if no GPU is present.

Arrays may be numpy arrays (host memory) or torch CUDA tensors / raw integer device addresses
(device memory); the library detects which.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MZ_LIB", os.path.join(_HERE, "libmuzero_hip.so"))  # MZ_LIB: diagnostic builds

PRECISION_FP32 = 0
PRECISION_BF16 = 1
PRECISION_F16 = 2
T_NAMES = ("initial", "recurrent", "tree_root", "tree_select", "tree_backup", "tree_finish")


class NetConfig(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("obs_planes", C.c_int32), ("action_planes", C.c_int32),
                ("channels", C.c_int32), ("bottleneck", C.c_int32), ("residuals", C.c_int32),
                ("broadcast_every_n", C.c_int32), ("action_space", C.c_int32), ("two_players", C.c_int32),
                ("value_scale", C.c_double)]


class SearchConfig(C.Structure):
    _fields_ = [("num_simulations", C.c_int32), ("initial_gumbel_m", C.c_int32), ("c_visit", C.c_int32),
                ("known_bounds", C.c_int32), ("hybrid", C.c_int32), ("gumbel_action_selection", C.c_int32),
                ("c_scale", C.c_double), ("discount", C.c_double), ("bound_min", C.c_double),
                ("bound_max", C.c_double), ("root_exploration_fraction", C.c_double),
                ("temperature_root", C.c_double), ("gumbel_seed", C.c_uint64), ("device_gumbel", C.c_int32),
                ("reserved0", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("net", NetConfig), ("search", SearchConfig), ("max_games", C.c_int32),
                ("max_simulations", C.c_int32), ("precision", C.c_int32), ("pool_slots", C.c_int32)]


class MoveOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "action", "selected_action", "visit_counts", "policy_target", "root_value_target", "v_mix",
        "root_value_inference", "entropy", "legal_action_max_entropy", "root_entropy_value_target",
        "playout_policy", "playout_appended", "sims_run", "status")]


EXPORTS = (
    "mz_create", "mz_destroy", "mz_last_error", "mz_set_action_cells", "mz_load_weights", "mz_num_params",
    "mz_weights_device_ptr", "mz_commit_weights", "mz_initial_inference", "mz_recurrent_inference",
    "mz_read_hidden", "mz_search_moves", "mz_search_moves_injected", "mz_set_recording", "mz_get_recording",
    "mz_tree_num_nodes", "mz_tree_node", "mz_tree_minmax", "mz_set_timing", "mz_get_timing", "mz_get_last_stats",
    "mz_synchronize", "mz_version", "mz_comm_unique_id", "mz_comm_init", "mz_comm_destroy", "mz_broadcast_weights",
    "mz_gather_stats", "mz_comm_barrier", "mz_get_recording_games", "mz_get_gumbel", "mz_seed_gumbel",
    "mz_child_order", "mz_noalloc_scratch_bytes", "mz_search_moves_injected_noalloc", "mz_set_stream_split")
NOALLOC_SCRATCH_BYTES = 16 << 20
COMM_ID_BYTES = 128

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.mz_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(vp)]
        L.mz_destroy.argtypes = [vp]
        L.mz_destroy.restype = None
        L.mz_last_error.argtypes = [vp]
        L.mz_last_error.restype = C.c_char_p
        L.mz_set_action_cells.argtypes = [vp, vp]
        L.mz_load_weights.argtypes = [vp, vp, C.c_size_t]
        L.mz_num_params.argtypes = [vp]
        L.mz_num_params.restype = C.c_size_t
        L.mz_weights_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.mz_commit_weights.argtypes = [vp]
        L.mz_initial_inference.argtypes = [vp, C.c_int] + [vp] * 5
        L.mz_recurrent_inference.argtypes = [vp, C.c_int] + [vp] * 6
        L.mz_read_hidden.argtypes = [vp, C.c_int32, vp]
        L.mz_search_moves.argtypes = [vp, C.c_int] + [vp] * 8 + [C.POINTER(MoveOutputs)]
        L.mz_search_moves_injected.argtypes = [vp, C.c_int] + [vp] * 9 + [C.c_int] + [vp] * 4 + [C.POINTER(MoveOutputs)]
        L.mz_set_recording.argtypes = [vp, C.c_int]
        L.mz_get_recording.argtypes = [vp] * 7
        L.mz_tree_num_nodes.argtypes = [vp, C.c_int]
        L.mz_tree_node.argtypes = [vp, C.c_int, C.c_int] + [vp] * 11
        L.mz_tree_minmax.argtypes = [vp, C.c_int, vp, vp]
        L.mz_set_timing.argtypes = [vp, C.c_int]
        L.mz_get_timing.argtypes = [vp, vp, vp]
        L.mz_get_last_stats.argtypes = [vp, vp, vp]
        L.mz_synchronize.argtypes = [vp]
        L.mz_version.restype = C.c_char_p
        L.mz_comm_unique_id.argtypes = [vp]
        L.mz_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
        L.mz_comm_destroy.argtypes = [vp]
        L.mz_broadcast_weights.argtypes = [vp, C.c_int]
        L.mz_gather_stats.argtypes = [vp, vp, C.c_int, vp]
        L.mz_comm_barrier.argtypes = [vp]
        L.mz_get_recording_games.argtypes = [vp]
        L.mz_get_gumbel.argtypes = [vp, vp]
        L.mz_seed_gumbel.argtypes = [vp, C.c_uint64, C.c_uint64]
        L.mz_child_order.argtypes = [vp, C.c_int, vp]
        L.mz_set_stream_split.argtypes = [vp, C.c_int]
        L.mz_noalloc_scratch_bytes.argtypes = [vp, C.c_int]
        L.mz_noalloc_scratch_bytes.restype = C.c_size_t
        L.mz_search_moves_injected_noalloc.argtypes = L.mz_search_moves_injected.argtypes
        _lib = L
    return _lib


def comm_unique_id():
    """ncclGetUniqueId through the C ABI: rank 0 calls it and hands the 128 bytes to the other ranks."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    if lib().mz_comm_unique_id(C.cast(buf, C.c_void_p)) != 0:
        raise MuZeroError(lib().mz_last_error(None).decode())
    return buf.raw


def child_order(legal):
    """Root child order (HashMap iteration order of the legal actions, Node.java:295-304) -> action indices."""
    legal = np.ascontiguousarray(legal, np.uint8)
    order = np.empty(legal.size, np.int32)
    L = lib().mz_child_order(_ptr(legal), legal.size, _ptr(order))
    if L < 0:
        raise MuZeroError("mz_child_order failed")
    return order[:L].copy()


class MuZeroError(RuntimeError):
    """Counterpart of the reference's unchecked MuZeroException (platform/.../common/MuZeroException.java)."""


def _ptr(a):
    """numpy array -> host address; torch tensor -> data_ptr(); int -> as is; None -> NULL."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        return a.data_ptr()
    raise TypeError(type(a))


def _np(a, dtype):
    if a is None or not isinstance(a, (np.ndarray, list, tuple)):
        return a
    return np.ascontiguousarray(a, dtype=dtype)


def make_config(spec, num_simulations, initial_gumbel_m, c_visit, c_scale=1.0, discount=1.0, known_bounds=None,
                root_exploration_fraction=0.0, hybrid=False, gumbel_action_selection=True, temperature_root=1.0,
                max_games=1, max_simulations=None, precision=PRECISION_FP32, pool_slots=0, gumbel_seed=None):
    cfg = Config()
    n = cfg.net
    n.height, n.width = spec.height, spec.width
    n.obs_planes, n.action_planes = spec.obs_planes, spec.action_planes
    n.channels, n.bottleneck, n.residuals = spec.channels, spec.bottleneck, spec.residuals
    n.broadcast_every_n, n.action_space = spec.broadcast_every_n, spec.action_space
    n.two_players = 1 if spec.two_players else 0
    n.value_scale = spec.value_scale
    s = cfg.search
    s.num_simulations, s.initial_gumbel_m, s.c_visit = num_simulations, initial_gumbel_m, c_visit
    s.known_bounds = 0 if known_bounds is None else 1
    s.hybrid = 1 if hybrid else 0
    s.gumbel_action_selection = 1 if gumbel_action_selection else 0
    s.c_scale, s.discount = c_scale, discount
    s.bound_min, s.bound_max = (0.0, 0.0) if known_bounds is None else known_bounds
    s.root_exploration_fraction = root_exploration_fraction
    s.temperature_root = temperature_root
    s.device_gumbel = 0 if gumbel_seed is None else 1   # gumbel=None then means: draw the noise on the device
    s.gumbel_seed = 0 if gumbel_seed is None else gumbel_seed
    cfg.max_games = max_games
    cfg.max_simulations = max_simulations if max_simulations is not None else num_simulations
    cfg.precision = precision
    cfg.pool_slots = pool_slots
    return cfg


class Context:
    """One mz_ctx (one GPU). Not thread-safe, like the reference's single model thread."""

    def __init__(self, cfg, device=0):
        self.cfg = cfg
        self.A = cfg.net.action_space
        self._h = C.c_void_p()
        rc = lib().mz_create(C.byref(cfg), device, C.byref(self._h))
        if rc != 0:
            raise MuZeroError(lib().mz_last_error(None).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().mz_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise MuZeroError(lib().mz_last_error(self._h).decode())

    # --- weights -------------------------------------------------------------------------------
    def num_params(self):
        return lib().mz_num_params(self._h)

    def load_weights(self, blob):
        blob = _np(blob, np.float32)
        n = blob.size if isinstance(blob, np.ndarray) else blob.numel()
        self._check(lib().mz_load_weights(self._h, _ptr(blob), n))

    def weights_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(lib().mz_weights_device_ptr(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def commit_weights(self):
        self._check(lib().mz_commit_weights(self._h))

    def set_action_cells(self, cells):
        cells = np.ascontiguousarray(cells, np.int32)
        assert cells.size == self.A
        self._check(lib().mz_set_action_cells(self._h, _ptr(cells)))

    # --- inference level (ModelService.initialInference / recurrentInference) --------------------
    def initial_inference(self, obs, dst_slot):
        obs = _np(obs, np.float32)
        dst_slot = np.ascontiguousarray(dst_slot, np.int32)
        B = dst_slot.size
        logits = np.empty((B, self.A), np.float32)
        policy = np.empty((B, self.A), np.float32)
        value = np.empty(B, np.float64)
        self._check(lib().mz_initial_inference(self._h, B, _ptr(obs), _ptr(dst_slot), _ptr(logits), _ptr(policy),
                                               _ptr(value)))
        return logits, policy, value

    def recurrent_inference(self, src_slot, action, dst_slot):
        src_slot = np.ascontiguousarray(src_slot, np.int32)
        action = np.ascontiguousarray(action, np.int32)
        dst_slot = np.ascontiguousarray(dst_slot, np.int32)
        B = dst_slot.size
        logits = np.empty((B, self.A), np.float32)
        policy = np.empty((B, self.A), np.float32)
        value = np.empty(B, np.float64)
        self._check(lib().mz_recurrent_inference(self._h, B, _ptr(src_slot), _ptr(action), _ptr(dst_slot),
                                                 _ptr(logits), _ptr(policy), _ptr(value)))
        return logits, policy, value

    def read_hidden(self, slot):
        n = self.cfg.net
        out = np.empty((n.channels, n.height, n.width), np.float32)
        self._check(lib().mz_read_hidden(self._h, int(slot), _ptr(out)))
        return out

    # --- move level (PlanAction.planAction for B games) -------------------------------------------
    def _alloc_outputs(self, B):
        A = self.A
        o = dict(
            action=np.empty(B, np.int32), selected_action=np.empty(B, np.int32),
            visit_counts=np.empty((B, A), np.int32), policy_target=np.empty((B, A), np.float32),
            root_value_target=np.empty(B, np.float32), v_mix=np.empty(B, np.float32),
            root_value_inference=np.empty(B, np.float32), entropy=np.empty(B, np.float32),
            legal_action_max_entropy=np.empty(B, np.float32), root_entropy_value_target=np.empty(B, np.float32),
            playout_policy=np.empty((B, A), np.float32), playout_appended=np.empty(B, np.uint8),
            sims_run=np.empty(B, np.int32), status=np.empty(B, np.int32))
        return o

    @staticmethod
    def _outputs_struct(o):
        mo = MoveOutputs()
        for name, _ in MoveOutputs._fields_:
            setattr(mo, name, _ptr(o.get(name)))
        return mo

    def search_moves(self, B, obs, legal, to_play=None, num_simulations=None, gumbel=None, dirichlet=None,
                     exploration_time=None, u=None, outputs=None):
        """outputs: dict of preallocated numpy arrays / torch CUDA tensors (subset allowed), or None for
        freshly allocated numpy arrays of everything."""
        o = self._alloc_outputs(B) if outputs is None else outputs
        mo = self._outputs_struct(o)
        self._check(lib().mz_search_moves(
            self._h, B, _ptr(_np(obs, np.float32)), _ptr(_np(legal, np.uint8)), _ptr(_np(to_play, np.uint8)),
            _ptr(_np(num_simulations, np.int32)), _ptr(_np(gumbel, np.float64)), _ptr(_np(dirichlet, np.float64)),
            _ptr(_np(exploration_time, np.uint8)), _ptr(_np(u, np.float64)), C.byref(mo)))
        return o

    def search_moves_injected(self, B, legal, init_logits, init_policy, init_value, sim_logits, sim_policy, sim_value,
                              to_play=None, num_simulations=None, gumbel=None, dirichlet=None, exploration_time=None,
                              u=None, outputs=None):
        o = self._alloc_outputs(B) if outputs is None else outputs
        mo = self._outputs_struct(o)
        sim_value = _np(sim_value, np.float64)
        S = sim_value.shape[1]
        self._check(lib().mz_search_moves_injected(
            self._h, B, _ptr(_np(legal, np.uint8)), _ptr(_np(to_play, np.uint8)),
            _ptr(_np(num_simulations, np.int32)), _ptr(_np(init_logits, np.float32)),
            _ptr(_np(init_policy, np.float32)), _ptr(_np(init_value, np.float64)),
            _ptr(_np(sim_logits, np.float32)), _ptr(_np(sim_policy, np.float32)), _ptr(sim_value), S,
            _ptr(_np(gumbel, np.float64)), _ptr(_np(dirichlet, np.float64)), _ptr(_np(exploration_time, np.uint8)),
            _ptr(_np(u, np.float64)), C.byref(mo)))
        return o

    def noalloc_scratch_bytes(self, B):
        return lib().mz_noalloc_scratch_bytes(self._h, B)

    def search_moves_injected_noalloc(self, B, legal, init_logits, init_policy, init_value, sim_logits, sim_policy,
                                      sim_value, S, outputs, to_play=None, num_simulations=None, gumbel=None,
                                      dirichlet=None, exploration_time=None, u=None):
        """Device pointers only (torch CUDA tensors / raw addresses); asynchronous: synchronize() before reading."""
        mo = self._outputs_struct(outputs)
        self._check(lib().mz_search_moves_injected_noalloc(
            self._h, B, _ptr(legal), _ptr(to_play), _ptr(num_simulations), _ptr(init_logits), _ptr(init_policy),
            _ptr(init_value), _ptr(sim_logits), _ptr(sim_policy), _ptr(sim_value), S, _ptr(gumbel), _ptr(dirichlet),
            _ptr(exploration_time), _ptr(u), C.byref(mo)))

    def set_recording(self, enable):
        self._check(lib().mz_set_recording(self._h, 1 if enable else 0))

    def get_recording(self):
        B, A, S = lib().mz_get_recording_games(self._h), self.A, self.cfg.max_simulations
        r = dict(init_logits=np.empty((B, A), np.float32), init_policy=np.empty((B, A), np.float32),
                 init_value=np.empty(B, np.float64), sim_logits=np.empty((B, S, A), np.float32),
                 sim_policy=np.empty((B, S, A), np.float32), sim_value=np.empty((B, S), np.float64))
        self._check(lib().mz_get_recording(self._h, _ptr(r["init_logits"]), _ptr(r["init_policy"]),
                                           _ptr(r["init_value"]), _ptr(r["sim_logits"]), _ptr(r["sim_policy"]),
                                           _ptr(r["sim_value"])))
        return r

    def get_gumbel(self, B):
        """Gumbel noise (by action index) the last search of B games used - drawn on the device or passed in."""
        g = np.empty((B, self.A), np.float64)
        self._check(lib().mz_get_gumbel(self._h, _ptr(g)))
        return g

    def seed_gumbel(self, seed, move_counter=0):
        self._check(lib().mz_seed_gumbel(self._h, seed, move_counter))

    # --- multi-GPU (RCCL through the C ABI) -------------------------------------------------------
    def comm_init(self, unique_id, rank, world):
        assert len(unique_id) == COMM_ID_BYTES
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        self._check(lib().mz_comm_init(self._h, C.cast(buf, C.c_void_p), rank, world))

    def comm_destroy(self):
        self._check(lib().mz_comm_destroy(self._h))

    def broadcast_weights(self, root=0):
        self._check(lib().mz_broadcast_weights(self._h, root))

    def gather_stats(self, values, world):
        v = np.ascontiguousarray(values, np.float64)
        out = np.empty((world, v.size), np.float64)
        self._check(lib().mz_gather_stats(self._h, _ptr(v), v.size, _ptr(out)))
        return out

    def comm_barrier(self):
        self._check(lib().mz_comm_barrier(self._h))

    # --- introspection ----------------------------------------------------------------------------
    def num_nodes(self, g):
        return lib().mz_tree_num_nodes(self._h, g)

    def node(self, g, slot):
        A = self.A
        ps, ac, tp, vc = (np.zeros(1, np.int32) for _ in range(4))
        sc = np.zeros(8, np.float64)
        ca, cv, cs = (np.zeros(A, np.int32) for _ in range(3))
        cp, cl, cq = (np.zeros(A, np.float64) for _ in range(3))
        n = lib().mz_tree_node(self._h, g, slot, _ptr(ps), _ptr(ac), _ptr(tp), _ptr(vc), _ptr(sc), _ptr(ca), _ptr(cp),
                               _ptr(cl), _ptr(cv), _ptr(cq), _ptr(cs))
        if n < 0:
            raise MuZeroError(lib().mz_last_error(self._h).decode())
        return dict(parent_slot=int(ps[0]), action=int(ac[0]), to_play=int(tp[0]), visit_count=int(vc[0]),
                    value=sc[0], vmix=sc[1], improved_value=sc[2], q_sum=sc[3], v_entropy_mix=sc[4],
                    improved_entropy_value=sc[5], entropy_q_sum=sc[6], entropy_reward=sc[7],
                    child_action=ca[:n].copy(), child_prior=cp[:n].copy(), child_logit=cl[:n].copy(),
                    child_visit=cv[:n].copy(), child_qsum=cq[:n].copy(), child_slot=cs[:n].copy())

    def minmax(self, g):
        mn, mx = C.c_double(), C.c_double()
        self._check(lib().mz_tree_minmax(self._h, g, C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    # --- measurement ------------------------------------------------------------------------------
    def set_timing(self, enable):
        self._check(lib().mz_set_timing(self._h, 1 if enable else 0))

    def set_stream_split(self, enable):
        self._check(lib().mz_set_stream_split(self._h, 1 if enable else 0))

    def get_timing(self):
        ms = np.zeros(len(T_NAMES), np.float64)
        n = np.zeros(len(T_NAMES), np.int64)
        self._check(lib().mz_get_timing(self._h, _ptr(ms), _ptr(n)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(T_NAMES)}

    def last_stats(self):
        s, p = C.c_int64(), C.c_double()
        self._check(lib().mz_get_last_stats(self._h, C.byref(s), C.byref(p)))
        return s.value, p.value

    def synchronize(self):
        self._check(lib().mz_synchronize(self._h))
