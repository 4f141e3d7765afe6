#!/usr/bin/env python3
"""Diagnostic: compare the raw residual stream after n blocks (MZ_DEBUG_STOP=n) with the torch oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from muzero_amd import capi, netspec  # noqa: E402
from oracle.net import NetOracle  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "go9"
n = int(os.environ.get("MZ_DEBUG_STOP", "0"))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 7
spec = netspec.SPECS[name]()
w = netspec.random_weights(spec, seed=3, ln_jitter=0.1)
ctx = capi.Context(capi.make_config(spec, 4, 2, 50, max_games=B, precision=capi.PRECISION_BF16, pool_slots=2 * B))
ctx.load_weights(netspec.pack_blob(spec, w))
ctx.set_action_cells(netspec.action_cells_board(spec))
rng = np.random.default_rng(0)
obs = (rng.uniform(size=(B, spec.obs_planes, spec.height, spec.width)) < 0.3).astype(np.float32)
slots = np.arange(B, dtype=np.int32)
try:
    ctx.initial_inference(obs, slots)
except Exception as e:  # outputs are garbage in this mode
    print("note:", e)
hid = np.stack([ctx.read_hidden(s) for s in slots])
o = NetOracle(spec, w)
with torch.no_grad():
    x = o._c3(torch.as_tensor(obs), "rep.conv0")
    for i in range(n):
        p = f"rep.b{i}"
        x = o._broadcast(x, p) if spec.is_broadcast(i) else o._bottleneck(x, p)
x = x.numpy()
err = np.abs(hid - x)
print(f"{name} after {n} blocks: nan={np.isnan(hid).sum()} max|ref|={np.abs(x).max():.3f} max err={np.nanmax(err):.4f} mean err={np.nanmean(err):.5f}")
bad = np.argwhere(~(err < 0.05 + 0.05 * np.abs(x)))
print("bad count", len(bad), "of", err.size)
if len(bad):
    print("first bad (b,ch,y,x):", bad[:8].tolist())
    print("boards with bad:", sorted(set(bad[:, 0].tolist())), "channels(min,max):", bad[:, 1].min(), bad[:, 1].max())
    cells = sorted(set((bad[:, 2] * spec.width + bad[:, 3]).tolist()))
    print("cells:", cells[:100])
