"""Grid level parity: device pair results through the host position solve (needs an MI355X).

The expected positions of the ``pairs_of_*`` cases come from the reference run end to end in the
development container: its compute_single_translation on the seeded tiles, then its own filters,
spanning tree and position accumulation (tests/golden/make_golden.py and make_golden_solve.py).
"""
import json
import os

import numpy as np
import pytest

from image_stitcher_amd import solve, synth

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "solve_golden.json")) as f:
    END_TO_END = [c for c in json.load(f)["cases"] if "registration_grid" in c]


@pytest.fixture(scope="module")
def reg():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from image_stitcher_amd import registration
    return registration


@pytest.mark.parametrize("case", END_TO_END, ids=[c["name"] for c in END_TO_END])
def test_positions_match_reference_end_to_end(reg, golden, case):
    spec = next(g for g in golden["grids"] if g["name"] == case["registration_grid"])
    g = synth.make_grid(spec["seed"], spec["rows"], spec["cols"], spec["tile_h"], spec["tile_w"],
                        overlap=spec["overlap"], jitter=spec["jitter"])
    out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, spec["edge_width"])
    grid = solve.grid_from_pairs(case["row"], case["col"], g.pair_idx, g.pair_dir, out.cpu().numpy(),
                                 status.cpu().numpy())
    res = solve.solve_positions(grid, case["size_y"], case["size_x"], case["pou"], case["ncc_threshold"])
    exp = case["expected"]
    assert res.y_pos.tolist() == exp["y_pos"] and res.x_pos.tolist() == exp["x_pos"]
    assert len(res.tree) == len(exp["tree"])     # which of the equally good (NCC 1 +- 1e-16) pairs, is not pinned
    for d in ("left", "top"):
        assert res.filters.valid3[d].tolist() == exp[f"{d}_valid3"]
    # and those are the planted positions
    planted = g.true_yx - g.true_yx.min(axis=0)
    assert np.array_equal(np.stack([res.y_pos, res.x_pos], axis=1), planted)


def test_grid_with_a_featureless_tile_is_placed_from_the_stage(reg):
    """6 x 6 grid with one blank tile: its four pairs are rejected on the device, the tree spans
    the other 35 tiles and the blank one lands where its stage coordinates say.  The same pair
    table computed by the CPU oracle must give the same solve."""
    from oracle import registration as oreg
    rows = cols = 6
    g = synth.make_grid(77, rows, cols, 256, 256, overlap=0.125, jitter=3)
    blank = 2 * cols + 3
    g.tiles[blank] = 5000
    out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    touched = (g.pair_idx == blank).any(axis=1)
    assert (status[touched] == 1).all() and (status[~touched] == 0).all()
    px_um = 0.752
    stage_x = 10.0 + g.true_yx[:, 1] * px_um / 1000.0
    stage_y = 20.0 + g.true_yx[:, 0] * px_um / 1000.0
    row, col = np.repeat(np.arange(rows), cols), np.tile(np.arange(cols), rows)
    grid = solve.grid_from_pairs(row, col, g.pair_idx, g.pair_dir, out, status, stage_x, stage_y)
    res = solve.solve_positions(grid, 256, 256)
    assert res.not_in_tree == [blank] and len(res.tree) == rows * cols - 2
    planted = g.true_yx - g.true_yx.min(axis=0)
    got = np.stack([res.y_pos, res.x_pos], axis=1)
    keep = np.arange(rows * cols) != blank
    assert np.array_equal(got[keep], planted[keep])
    # the blank tile: stage offset from tile 0 times the guessed scale (first left step / 2084 px)
    scale = 1000.0 / res.pixel_size_um
    want = np.round(np.array([(stage_y[blank] - stage_y[0]) * scale, (stage_x[blank] - stage_x[0]) * scale])
                    + got[0])
    assert np.array_equal(got[blank], want.astype(np.int64))
    # oracle pair table -> identical solve
    cpu, cpu_status = oreg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32)
    assert np.array_equal(cpu_status, status)
    ref = solve.solve_positions(solve.grid_from_pairs(row, col, g.pair_idx, g.pair_dir, cpu, cpu_status,
                                                      stage_x, stage_y), 256, 256)
    assert np.array_equal(ref.y_pos, res.y_pos) and np.array_equal(ref.x_pos, res.x_pos)


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
def test_register_solve_fuse_reassembles_the_scene(reg, mode):
    """The whole path on one grid: tiles cut from a scene at jittered positions, pairs registered
    on the device, positions solved, tiles fused at those positions.  Every covered mosaic pixel
    must be the scene pixel it was cut from (overlapping tiles agree there, so the blend of equal
    values is that value)."""
    from image_stitcher_amd import fusion
    rows, cols, t, jitter, overlap = 4, 5, 512, 4, 0.10
    seed = 91
    g = synth.make_grid(seed, rows, cols, t, t, overlap=overlap, jitter=jitter)
    scene = synth.make_scene(seed, (rows - 1) * g.pitch_y + t + 2 * jitter, (cols - 1) * g.pitch_x + t + 2 * jitter)
    out, status = reg.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 64)
    row, col = np.repeat(np.arange(rows), cols), np.tile(np.arange(cols), rows)
    res = solve.solve_positions(solve.grid_from_pairs(row, col, g.pair_idx, g.pair_dir, out.cpu().numpy(),
                                                      status.cpu().numpy()), t, t)
    yx = np.stack([res.y_pos, res.x_pos], axis=1).astype(np.int32)
    mh, mw = int(yx[:, 0].max()) + t, int(yx[:, 1].max()) + t
    mosaic = fusion.fuse_tiles(g.tiles, yx, np.zeros(rows * cols, dtype=np.int32), 1, mh, mw, mode=mode)
    mosaic = mosaic.cpu().numpy()[0]
    y0, x0 = g.true_yx.min(axis=0)
    covered = np.zeros((mh, mw), dtype=bool)
    for y, x in yx:
        covered[y:y + t, x:x + t] = True
    want = scene[y0:y0 + mh, x0:x0 + mw]
    assert covered.mean() > 0.99
    assert np.array_equal(mosaic[covered], want[covered])
    assert (mosaic[~covered] == 0).all()
