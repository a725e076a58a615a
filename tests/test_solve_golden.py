"""Host-side position solve against vectors produced by the reference's own solve.

tests/golden/solve_golden.json holds pair tables and what the reference's compute_image_overlap2 ->
filter_translations -> compute_maximum_spanning_tree -> compute_final_position made of them
(generator: tests/golden/make_golden_solve.py).  Flags, tree and positions must be identical;
replacement columns are medians of the same numbers and must be equal to the last bit.
"""
import json
import os

import numpy as np
import pytest

from image_stitcher_amd import solve

GOLD = os.path.join(os.path.dirname(__file__), "golden", "solve_golden.json")
with open(GOLD) as f:
    CASES = json.load(f)["cases"]

EXC = {"ValueError": ValueError, "RuntimeError": RuntimeError, "AssertionError": AssertionError}


def load_case(case) -> solve.TranslationGrid:
    def arr(rows):
        return np.array([[np.nan if v is None else v for v in r] for r in rows], dtype=np.float64)
    left, top = solve.neighbour_indices(case["row"], case["col"])
    return solve.TranslationGrid(np.asarray(case["row"]), np.asarray(case["col"]), {"left": left, "top": top},
                                 {"left": arr(case["left_first"]), "top": arr(case["top_first"])},
                                 None if "stage_x" not in case else np.asarray(case["stage_x"]),
                                 None if "stage_y" not in case else np.asarray(case["stage_y"]))


def check_against_expected(res: solve.PositionSolve, exp) -> None:
    assert res.overlap_left == exp["overlap_left"]
    assert res.overlap_top == exp["overlap_top"]
    assert res.repeatability == exp["glue"]["repeatability"]
    for d in ("left", "top"):
        for k, flags in ((1, res.filters.valid1), (2, res.filters.valid2), (3, res.filters.valid3)):
            assert flags[d].tolist() == exp[f"{d}_valid{k}"], f"{d}_valid{k}"
        np.testing.assert_array_equal(res.filters.second[d], np.asarray(exp[f"{d}_second"], dtype=np.float64))
    tree = sorted([e.f, e.t, e.direction, e.weight] for e in res.tree)
    assert tree == exp["tree"]
    assert res.y_pos.dtype == np.int32 and res.x_pos.dtype == np.int32
    assert res.y_pos.tolist() == exp["y_pos"]
    assert res.x_pos.tolist() == exp["x_pos"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_solve_matches_reference(case):
    grid = load_case(case)
    exp = case["expected"]
    if "raises" in exp:
        with pytest.raises(EXC[exp["raises"]]) as info:
            solve.solve_positions(grid, case["size_y"], case["size_x"], case["pou"], case["ncc_threshold"])
        if exp["message"]:
            assert str(info.value)[:60] == exp["message"][:60]
        return
    res = solve.solve_positions(grid, case["size_y"], case["size_x"], case["pou"], case["ncc_threshold"])
    check_against_expected(res, exp)


def test_golden_cases_exercise_every_branch():
    """The vectors would prove little if nothing was ever rejected, replaced or tied."""
    ok = [c for c in CASES if "raises" not in c["expected"]]
    assert any(not all(c["expected"]["left_valid3"][i] for i in range(len(c["row"])) if c["col"][i] > 0) for c in ok)
    assert any(c["expected"]["left_valid1"] != c["expected"]["left_valid2"] for c in ok)
    assert any(c["expected"]["top_valid2"] != c["expected"]["top_valid3"] for c in ok)
    assert any(len({e[3] for e in c["expected"]["tree"]}) < len(c["expected"]["tree"]) for c in ok)
    assert any(len(c["expected"]["tree"]) < len(c["row"]) - 1 for c in ok)      # a tile placed from the stage
    assert {c["expected"]["raises"] for c in CASES if "raises" in c["expected"]} == set(EXC)


def test_repeatability_by_hand():
    """tile_registration.py:1007-1044 is inline code with no function to call, so the golden cases
    take this number as an input; pin it on a table small enough to work by hand.

    2 x 3 grid, 100 px tiles, overlap 10 %: left pairs step 90 +/- along x.
      left x: 88, 91 (row 0), 90, 93 (row 1)  -> r1 = ceil((93 - 88) / 2) = 3
      left y within row 0: 0, 4 -> 4; row 1: 1, 2 -> 1       -> r2 = ceil(4 / 2) = 2
      top y: 90, 90, 91 -> r1 = ceil(1 / 2) = 1; top x within a column: single values -> r2 = 0
    """
    row, col = np.array([0, 0, 0, 1, 1, 1]), np.array([0, 1, 2, 0, 1, 2])
    left, top = solve.neighbour_indices(row, col)
    nan = [np.nan] * 3
    lf = np.array([nan, [0.9, 0, 88], [0.9, 4, 91], nan, [0.9, 1, 90], [0.9, 2, 93]], dtype=np.float64)
    tf = np.array([nan, nan, nan, [0.9, 90, 1], [0.9, 90, -2], [0.9, 91, 0]], dtype=np.float64)
    grid = solve.TranslationGrid(row, col, {"left": left, "top": top}, {"left": lf, "top": tf})
    ol, ot, has_left, has_top = solve.estimate_overlaps(grid, 100, 100, 3, 0.5)
    assert (ol, ot) == (100 - 90.5, 10.0)
    assert solve.estimate_repeatability(grid, 100, 100, ol, ot, has_left, has_top, 3, 0.5) == 3.0
    # a pair below the NCC threshold, or outside the overlap window, does not count
    lf[5] = [0.4, 2, 93]
    assert solve.estimate_repeatability(grid, 100, 100, ol, ot, has_left, has_top, 3, 0.5) == 2.0


def test_pixel_size_guess_by_hand():
    """tile_registration.py:1161-1174: first tile with a left neighbour, |dx| mm * 1000 / 2084."""
    row, col = np.array([0, 0, 1, 1]), np.array([0, 1, 0, 1])
    left, top = solve.neighbour_indices(row, col)
    z = np.full((4, 3), np.nan)
    grid = solve.TranslationGrid(row, col, {"left": left, "top": top}, {"left": z, "top": z},
                                 np.array([1.0, 2.5, 1.0, 2.5]), np.zeros(4))
    assert solve.estimate_pixel_size_um(grid) == 1.5 * 1000.0 / 2084.0
    grid.stage_x = None
    assert solve.estimate_pixel_size_um(grid) is None


def test_grid_from_pairs_files_results_under_the_second_tile():
    from image_stitcher_amd import synth
    pair_idx, pair_dir = synth.grid_pairs(2, 3)
    res = np.arange(pair_idx.shape[0] * 3, dtype=np.float64).reshape(-1, 3)
    status = np.zeros(pair_idx.shape[0], dtype=np.int8)
    status[2] = 1
    row, col = np.repeat(np.arange(2), 3), np.tile(np.arange(3), 2)
    grid = solve.grid_from_pairs(row, col, pair_idx, pair_dir, res, status)
    for p, ((a, b), d) in enumerate(zip(pair_idx, pair_dir)):
        got = grid.first[solve.DIRECTIONS[d]][b]
        if p == 2:
            assert np.isnan(got).all()
        else:
            assert got.tolist() == res[p].tolist()
    assert np.isnan(grid.first["left"][[0, 3]]).all() and np.isnan(grid.first["top"][:3]).all()
    with pytest.raises(ValueError):
        solve.grid_from_pairs(row, col, [[0, 2]], [0], [[1, 2, 3]])
    with pytest.raises(ValueError):
        solve.neighbour_indices([0, 0], [1, 1])
