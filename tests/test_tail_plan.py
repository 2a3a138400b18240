"""Host logic of ssa_k_verify's end game, on the CPU (no device): the launch plan ssa_api.hip computes for a slice.

The kernel trusts the plan: a piece wave runs the windows [lo, hi) of its descriptors and parks the accumulator; the last
piece adds [e]G and compares.  So the plan must cover every ladder window of every pass exactly once, in order, never let
a piece span the two passes of the subgroup check (src/signature.rs:182-184 is pass 0, :196-198 pass 1), and describe a
grid of [first pieces][ordinary workgroups][second pieces] ... that holds every lane exactly once."""
import itertools

import pytest

LADDER_STEPS = 50        # windows of [h]P (ssa_kernels.hpp)
LADDER_STEPS_Q = 43      # windows of [q]P: the width-5 NAF of q has 44 digits (qnaf.inc)
VP_MAX = 8


@pytest.fixture(scope="module")
def ssa():
    import schnorr_sig_amd as m
    return m


def _covered(plan, torsion):
    """the (pass, window) sequence the pieces run, in order"""
    seq = []
    for pas, first, last, lo, hi in plan["pieces"]:
        n_steps = LADDER_STEPS if pas else LADDER_STEPS_Q
        assert 0 <= lo < hi <= n_steps
        assert first == (lo == 0) and last == (hi == n_steps)
        seq += [(pas, w) for w in range(lo, hi)]
    return seq


@pytest.mark.parametrize("torsion", [False, True])
@pytest.mark.parametrize("pieces,gens,uniform", [(2, 1, False), (3, 1, False), (5, 1, False), (5, 2, False), (8, 1, False),
                                                  (4, 1, True), (7, 3, True), (8, 2, True)])
def test_pieces_cover_every_window_once(ssa, torsion, pieces, gens, uniform):
    waves = 2048
    n = 1 << 20
    plan = ssa.debug_tail_plan(waves, n, check_torsion=torsion, pieces=pieces, gens=gens, uniform=uniform)
    assert 2 <= plan["n_pieces"] <= VP_MAX
    want = ([(0, w) for w in range(LADDER_STEPS_Q)] if torsion else []) + [(1, w) for w in range(LADDER_STEPS)]
    assert _covered(plan, torsion) == want
    # never more pieces than asked for, plus the pass boundary
    assert plan["n_pieces"] <= pieces + (1 if torsion else 0)
    # the whole-pass descriptors of the ordinary workgroups
    assert plan["whole"] == [(0, True, True, 0, LADDER_STEPS_Q), (1, True, True, 0, LADDER_STEPS)]


@pytest.mark.parametrize("n", [1 << 20, (1 << 20) - 1, (1 << 20) + 1, 131072 + 64, 147456, 200001, 294912, 3 * 131072 + 37])
def test_grid_holds_every_lane_once(ssa, n):
    waves = 2048
    plan = ssa.debug_tail_plan(waves, n)
    groups = (n + 63) // 64
    assert plan["n_pieces"] == 5
    assert plan["tail_groups"] % 4 == 0 and plan["tail_groups"] >= waves
    # ordinary workgroups hold lanes [0, 256 * main_blocks), the tail groups the rest (rounded up to a workgroup)
    assert plan["main_blocks"] * 4 + plan["tail_groups"] >= groups
    assert plan["main_blocks"] * 4 + plan["tail_groups"] - groups < 4
    assert plan["main_blocks"] * 256 <= n
    assert plan["grid_blocks"] == plan["main_blocks"] + plan["n_pieces"] * plan["tail_groups"] // 4


def test_geometric_and_uniform_cuts(ssa):
    """halving pieces: each is about half of what is left (by instruction count: a window is 5 doublings + 1 addition,
    the table build stands in front of the first piece, the comb and the comparison behind the last); equal pieces: the
    same number of windows give or take the table build and the comb"""
    geo = ssa.debug_tail_plan(2048, 1 << 20, pieces=5)["pieces"]
    widths = [hi - lo for _, _, _, lo, hi in geo]
    assert sum(widths) == LADDER_STEPS and widths == sorted(widths, reverse=True)
    assert 20 <= widths[0] <= 27 and widths[-1] <= 4          # 1/2 of ~56 window-equivalents minus the table build; 1/16
    uni = ssa.debug_tail_plan(2048, 1 << 20, pieces=5, uniform=True)["pieces"]
    widths = [hi - lo for _, _, _, lo, hi in uni]
    assert sum(widths) == LADDER_STEPS and max(widths) - min(widths) <= 5


def test_no_end_game_when_it_cannot_pay(ssa):
    waves = 2048
    for n in (1, 64, 7680, 65536, 131008):                   # under one generation of waves: one ordinary launch
        plan = ssa.debug_tail_plan(waves, n)
        assert plan["n_pieces"] == 0 and plan["grid_blocks"] == (n + 255) // 256
    one = ssa.debug_tail_plan(waves, 131009)                 # exactly one generation: all of it runs in pieces
    assert one["n_pieces"] == 5 and one["main_blocks"] == 0 and one["tail_groups"] == 2048
    assert ssa.debug_tail_plan(waves, 1 << 20, pieces=0)["n_pieces"] == 0
    assert ssa.debug_tail_plan(waves, 1 << 20, pieces=1)["n_pieces"] == 0
    assert ssa.debug_tail_plan(0, 1 << 20)["n_pieces"] == 0   # (occupancy query failed: no plan)
    # a launch must have min_main generations of ordinary workgroups beside its tail
    assert ssa.debug_tail_plan(waves, 3 * 131072, min_main=2)["n_pieces"] == 5
    assert ssa.debug_tail_plan(waves, 3 * 131072 - 64, min_main=2)["n_pieces"] == 0
    # more pieces than the kernel's descriptor array: clamped
    for torsion, pieces in itertools.product((False, True), (9, 20, 255)):
        plan = ssa.debug_tail_plan(waves, 1 << 20, check_torsion=torsion, pieces=pieces)
        assert 2 <= plan["n_pieces"] <= VP_MAX
