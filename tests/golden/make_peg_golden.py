#!/usr/bin/env python3
"""Builds tests/golden/peg_actions.npz from the one data fixture the reference holds for a plugin on the search path:
games/pegsolitair/src/test/resources/games/buffer1000.zip (1000 PegSolitair episodes in the legacy JSON buffer format:
`actions`, `policyTargets`, `rewards` per game).  Run in the build container (the reference is not on the GPU box):

    python3 tests/golden/make_peg_golden.py [/root/reference]

Every episode is replayed on the 33-cell English board with this repository's own restatement of the rules
(Board.java:27-36,55-77,113-124: start = all pegs, hole at (4,4); a jump from p over a peg into a hole two cells on) under
the action-index convention of PegSolitairAdapter.java:19-38 / PegSolitairAction.java:57-61:
    index = direction * 49 + (row - 1) * 7 + (col - 1),  direction in enum order N, E, S, W  (N = row - 1, E = col + 1).
Only DATA goes into the fixture: per position the peg plane, the legal mask this restatement derives, the recorded action
and the support of the recorded policy target.  tests/test_peg_fixture.py asserts that the reference's recorded moves and
targets agree with the convention (pins mz_set_action_cells' default for A = 196), and a -m gpu test searches those real
positions."""
import json
import os
import sys
import zipfile

import numpy as np

DIRS = [(-1, 0), (0, 1), (1, 0), (0, -1)]   # Direction.java: N, E, S, W; Point.pointIn: N = row - 1, E = col + 1, S = row + 1, W = col - 1


def in_range(r, c):     # NeighborMap.inRange: the 7x7 square without its four 2x2 corners (rows / cols 1..7)
    return 1 <= r <= 7 and 1 <= c <= 7 and (3 <= r <= 5 or 3 <= c <= 5)


def legal_mask(pegs):
    m = np.zeros(196, np.uint8)
    for d, (dr, dc) in enumerate(DIRS):
        for r in range(1, 8):
            for c in range(1, 8):
                r2, c2, r3, c3 = r + dr, c + dc, r + 2 * dr, c + 2 * dc
                if in_range(r, c) and in_range(r2, c2) and in_range(r3, c3) and (r, c) in pegs and (r2, c2) in pegs \
                        and (r3, c3) not in pegs:
                    m[d * 49 + (r - 1) * 7 + (c - 1)] = 1
    return m


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    z = zipfile.ZipFile(os.path.join(ref, "games/pegsolitair/src/test/resources/games/buffer1000.zip"))
    data = json.loads(z.read("buffer1000.json"))["data"]
    planes, legals, actions, support, game_of, tmax = [], [], [], [], [], []
    for gi, g in enumerate(data):
        pegs = {(r, c) for r in range(1, 8) for c in range(1, 8) if in_range(r, c)} - {(4, 4)}
        for t, a in enumerate(g["actions"]):
            plane = np.zeros(49, np.uint8)
            for (r, c) in pegs:
                plane[(r - 1) * 7 + (c - 1)] = 1
            planes.append(plane)
            legals.append(legal_mask(pegs))
            actions.append(a)
            pt = np.asarray(g["policyTargets"][t], np.float32) if t < len(g["policyTargets"]) else np.zeros(196, np.float32)
            support.append((pt > 0).astype(np.uint8))
            tmax.append(int(np.argmax(pt)) if pt.any() else -1)
            game_of.append(gi)
            d, cell = divmod(a, 49)
            r, c = cell // 7 + 1, cell % 7 + 1
            dr, dc = DIRS[d]
            pegs -= {(r, c), (r + dr, c + dc)}       # Board.applyJump (no legality check here: the test does that)
            pegs |= {(r + 2 * dr, c + 2 * dc)}
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "peg_actions.npz")
    np.savez_compressed(out, planes=np.packbits(np.array(planes), axis=1), legal=np.packbits(np.array(legals), axis=1),
                        actions=np.array(actions, np.int16), target_support=np.packbits(np.array(support), axis=1),
                        target_argmax=np.array(tmax, np.int16), game=np.array(game_of, np.int16))
    print(out, len(actions), "positions of", len(data), "games", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
