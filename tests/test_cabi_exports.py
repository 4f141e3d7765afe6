"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/muzero_hip.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from muzero_amd import capi, netspec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "muzero_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mz_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    L = capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/muzero_hip.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_struct_layouts_match_header():
    # sizes the C compiler gives the same structs (LP64, natural alignment)
    assert C.sizeof(capi.NetConfig) == 48
    assert C.sizeof(capi.SearchConfig) == 88
    assert C.sizeof(capi.Config) == 48 + 88 + 16
    assert C.sizeof(capi.MoveOutputs) == 14 * 8


def test_create_fails_loudly_without_gpu_or_on_bad_config():
    import torch
    spec = netspec.spec_tiny()
    if not torch.cuda.is_available():
        with pytest.raises(capi.MuZeroError, match="no HIP device"):
            capi.Context(capi.make_config(spec, 4, 2, 50, max_games=1))
    bad = capi.make_config(spec, 4, 1, 50, max_games=1)   # initialGumbelM < 2 (SequentialHalving needs k >= 2)
    with pytest.raises(capi.MuZeroError, match="initial_gumbel_m"):
        capi.Context(bad)


def test_param_count_formula_matches_tensor_list():
    # BASELINE.md §4 parameter counts (LN affine tensors included)
    assert abs(netspec.num_params(netspec.spec_go9()) - (1.68e6 + 1.81e6 + 27e3)) / 3.5e6 < 0.02
    fl = netspec.flops_per_sample(netspec.spec_go9())
    assert abs(fl["recurrent"] / 1e6 - 191.8) < 0.2 and abs(fl["initial"] / 1e6 - 170.8) < 0.2
    fl = netspec.flops_per_sample(netspec.spec_tictactoe())
    assert abs(fl["recurrent"] / 1e6 - 49.6) < 0.2


def test_child_order_helper_matches_the_oracle_hashmap():
    """mz_child_order (host-side helper for replaying a reference Dirichlet vector, which the reference applies
    by child index, Node.java:353-358) against the oracle's java.util.HashMap emulation."""
    import numpy as np
    from oracle import search as osr
    rng = np.random.default_rng(0)
    for A in (9, 26, 82, 196):
        masks = [np.ones(A, np.uint8)]
        for _ in range(40):
            masks.append((rng.uniform(size=A) < rng.uniform(0.05, 0.95)).astype(np.uint8))
        m = np.zeros(A, np.uint8)
        m[[i for i in range(A) if i % 16 == 5][:12]] = 1   # one bucket at T = 16: the ninth entry forces a resize
        masks.append(m)
        for legal in masks:
            keys = np.nonzero(legal)[0]
            if len(keys) == 0:
                continue
            assert capi.child_order(legal).tolist() == osr.hashmap_order(keys)
