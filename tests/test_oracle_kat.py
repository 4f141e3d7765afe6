"""Pins the CPU oracle against every known-answer test the reference holds for the search path
(SURVEY.md §8c): platform/src/test/java/.../c_planning/SequentialHalvingTest.java,
GumbelTest.java and common/FunctionsTest.java, plus the JDK semantics of SURVEY.md §8a-2."""
import math

import numpy as np
import pytest

from oracle import search as osr


# SequentialHalvingTest.java:13-32
@pytest.mark.parametrize("budget,m,expected", [(200, 16, [3, 6, 12, 28]), (50, 8, [2, 4, 9]), (20, 9, [1, 1, 2, 3])])
def test_extra_phase_visits_kat(budget, m, expected):
    assert osr.extra_phase_visits(budget, m) == expected


# SequentialHalvingTest.java:36-46
def test_next_runs_exactly_n_times():
    assert osr.sh_count_next(20, 9, 9) == 20


# SequentialHalvingTest.java:51-87
@pytest.mark.parametrize("n,m,k", [(20, 4, 9), (30, 16, 26), (200, 16, 26), (50, 16, 26), (50, 16, 16)])
def test_budget_identity(n, m, k):
    m_eff = min(m, k)
    visits = osr.extra_phase_visits(n, m_eff)
    s, mm = 0, m_eff
    for v in visits:
        s += mm * v
        mm //= 2
    assert s == n
    assert osr.sh_count_next(n, m, k) == n


# Schedules at BASELINE.json's configs
@pytest.mark.parametrize("n,m,expected", [(20, 4, [2, 6]), (30, 16, [1, 1, 1, 1]), (50, 16, [1, 1, 3, 7])])
def test_config_schedules(n, m, expected):
    assert osr.extra_phase_visits(n, m) == expected


def test_phases_equal_integer_ceil_log2():
    # Functions.log2 = Math.log(n)/Math.log(2) (Functions.java:114-116); the GPU path uses integer ceil-log2.
    for m in range(2, 513):
        assert osr.phases_for(m) == (m - 1).bit_length(), m


def test_sh_trace_shape():
    tr = osr.sh_trace(50, 16, 82)
    assert len(tr) == 50
    # phase 0: 16 candidates once, then 8, then 4 x3, then 2 x7
    assert tr[:16, 1].tolist() == list(range(16)) and set(tr[:16, 0]) == {0}
    assert tr[16:24, 2].tolist() == [8] * 8
    assert tr[-1].tolist() == [3, 1, 2, 6]


# GumbelTest.java:51-54 - drawGumbelActions returns exactly n actions
@pytest.mark.parametrize("n", [0, 1, 2, 3])
def test_draw_actions_counts(n):
    got = osr.draw_actions([0, 1, 2], [0.2, 0.7, 0.1], n)
    assert len(got) == n
    assert got == [1, 0, 2][:n]


def test_draw_actions_first_max_wins_ties():
    # Stream.max keeps the first maximal element
    assert osr.draw_actions([7, 8, 9, 10], [1.0, 3.0, 3.0, 2.0], 3) == [8, 9, 10]
    # Double.compare: -0.0 < +0.0
    assert osr.draw_actions([0, 1], [-0.0, 0.0], 1) == [1]
    assert osr.draw_actions([0, 1], [0.0, -0.0], 1) == [0]


# GumbelTest.java:17-22 is a plain vector add; FunctionsTest.java:12-16,32-40: softmax sums to one
def test_softmax_normalised():
    rng = np.random.default_rng(0)
    for n in (3, 9, 26, 82, 196):
        p = osr.softmax(rng.normal(size=n) * 5)
        assert abs(p.sum() - 1.0) < 1e-12
        assert (p >= 0).all()
    p = osr.softmax([1.0, 2.0, 3.0])
    e = np.exp(np.array([1.0, 2.0, 3.0]) - 3.0)
    np.testing.assert_allclose(p, e / e.sum(), rtol=1e-15)


def test_softmax0_first_strict_maximum():
    assert osr.softmax([1.0, 5.0, 5.0, 2.0], 0.0).tolist() == [0.0, 1.0, 0.0, 0.0]
    assert osr.softmax([-np.inf, -np.inf, 0.0], 0.0).tolist() == [0.0, 0.0, 1.0]


def test_softmax_minus_inf_entries():
    p = osr.softmax([-np.inf, 0.0, math.log(3.0)])
    np.testing.assert_allclose(p, [0.0, 0.25, 0.75], rtol=1e-15)


# --- JDK semantics -----------------------------------------------------------
def test_hashmap_order_examples_from_survey():
    assert osr.hashmap_order(range(9)) == [5, 6, 7, 8, 0, 1, 2, 3, 4]            # TicTacToe full board
    assert osr.hashmap_order(range(82)) == list(range(69, 82)) + list(range(69))  # Go 9x9
    assert osr.hashmap_order(range(26)) == list(range(5, 26)) + list(range(5))    # Go 5x5
    assert osr.hashmap_order(range(196)) == list(range(196))                      # PegSolitair


def test_hashmap_order_collisions_keep_insertion_order():
    # T = 16: 59+3 = 62 -> bucket 14, 59+19 = 78 -> bucket 14, 59+4 = 63 -> 15, 59+5 = 64 -> 0
    assert osr.hashmap_order([3, 4, 5, 19]) == [5, 3, 19, 4]
    # 13 keys force T = 32
    keys = list(range(13))
    assert osr.hashmap_order(keys) == sorted(keys, key=lambda i: ((59 + i) & 31, i))


def test_hashmap_small_table_resizes_instead_of_treeifying():
    # nine keys congruent mod 16 with T = 16 (< MIN_TREEIFY_CAPACITY): the ninth insertion resizes to 32
    keys = [5 + 16 * j for j in range(9)]
    got = osr.hashmap_order(keys)
    assert got == sorted(keys, key=lambda i: ((59 + i) & 31, i))
    assert got != sorted(keys, key=lambda i: ((59 + i) & 15, i))


def test_java_sum_is_compensated():
    v = np.array([1.0] + [1e-16] * 10)
    assert osr.java_sum(v) == 1.000000000000001
    assert float(np.add.reduce(v)) == 1.0 or True  # plain summation may lose the tail; Kahan must not
    acc = 0.0
    for x in v:
        acc += x
    assert acc == 1.0
    # NaN / inf fallback: simple sum is returned
    assert osr.java_sum(np.array([np.inf, 1.0, np.inf])) == np.inf


def test_entropy_plain_fold():
    p = np.array([0.5, 0.25, 0.25, 0.0], np.float32)
    assert abs(osr.entropy(p) - 1.5 * math.log(2)) < 1e-15
    assert osr.entropy(np.array([1.0, 0.0], np.float32)) == 0.0
