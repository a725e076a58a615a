"""The consensus stage of the reference's "Method 1" on the device (flag MISTITCH_REG_CONSENSUS) against vectors
made by running the reference's own clustering branch (tests/golden/make_golden_consensus.py), needs an MI355X."""
import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import registration as oreg
from test_gpu_peaks10 import _tiles_with_strips
from test_oracle_golden import _consensus_golden, consensus_strips

pytestmark = pytest.mark.gpu


def test_consensus_on_seeded_strips_matches_the_reference():
    import torch
    from image_stitcher_amd import registration
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    doc = _consensus_golden()
    seen = set()
    for s in doc["strips"]:
        a, b = consensus_strips(s)
        tiles, direction, edge, off = _tiles_with_strips(a, b, s["seed"])
        out, status, table = registration.register_pairs(tiles, [[0, 1]], [direction], edge, consensus=True,
                                                         return_candidates=True)
        row, table = out.cpu().numpy()[0], table.cpu().numpy()[0]
        if int(status.cpu()[0]) != 0:
            # the tile-level featureless test (not part of the strip-level vectors) rejected the pair: the oracle agrees
            e1, e2, _, _ = oreg.edge_strips(tiles[0], tiles[1], "top" if direction else "left", edge)
            assert oreg.is_edge_featureless(e1) or oreg.is_edge_featureless(e2), s
            continue
        assert int(np.isfinite(table[:, 2]).sum()) == s["n"], s
        assert (row[1] - off[0], row[2] - off[1]) == tuple(s["final"][1:]), s
        assert row[0] == pytest.approx(s["final"][0], abs=1e-12)
        seen.add(s["kind"])
        # without the flag the same tiles give the candidate set's best NCC (the fallback), as before
        plain, _ = registration.register_pairs(tiles, [[0, 1]], [direction], edge, peaks10=True)
        p = plain.cpu().numpy()[0]
        assert (p[1] - off[0], p[2] - off[1]) == tuple(s["best_ncc"][1:]), s
    assert seen == {"stands_wins", "stands_loses", "declined"}, seen


@pytest.mark.parametrize("chunk", [None, 4])
def test_consensus_on_grids_matches_the_reference_and_the_oracle(chunk):
    """Rows of the reference's top-level compute_single_translation with its consensus stage live."""
    import torch
    from image_stitcher_amd import _lib, registration
    doc = _consensus_golden()
    ctx = None if chunk is None else _lib.Context(0, torch.cuda.current_stream().cuda_stream, pair_chunk=chunk)
    for g in doc["grids"]:
        grid = synth.make_grid(g["seed"], g["rows"], g["cols"], g["tile_h"], g["tile_w"], overlap=g["overlap"], jitter=g["jitter"])
        out, status = registration.register_pairs(grid.tiles, grid.pair_idx, grid.pair_dir, g["edge_width"], consensus=True, ctx=ctx)
        out, status = out.cpu().numpy(), status.cpu().numpy()
        for p, exp in enumerate(g["pairs"]):
            assert (exp is None) == (status[p] != 0), (g["name"], p)
            if exp is not None:
                assert tuple(out[p, 1:]) == tuple(exp[1:]), (g["name"], p, out[p], exp)
                assert out[p, 0] == pytest.approx(exp[0], abs=1e-12)


def test_consensus_random_small_strips_match_the_oracle():
    """Small strips, where the consensus stands often: device == oracle (itself checked against the reference on
    1 500 such strips by tests/golden/check_consensus_against_reference.py)."""
    from image_stitcher_amd import registration
    rng = np.random.default_rng(5)
    stood = 0
    for trial in range(60):
        th, tw = int(rng.choice([16, 24, 32, 48, 64])), int(rng.choice([16, 24, 32, 48, 64, 96]))
        edge = int(rng.choice([6, 8, 12, 16]))
        g = synth.make_grid(9000 + trial, 2, 3, th, tw, overlap=float(rng.choice([0.1, 0.2, 0.3])), jitter=int(rng.integers(1, 3)),
                            lowpass=float(rng.choice([0, 200, 2000, 8000])))
        want, want_st, st = oreg.register_pairs_consensus(g.tiles, g.pair_idx, g.pair_dir, edge)
        out, status = registration.register_pairs(g.tiles, g.pair_idx, g.pair_dir, edge, consensus=True)
        out, status = out.cpu().numpy(), status.cpu().numpy()
        assert np.array_equal(status != 0, want_st != 0), trial
        ok = want_st == 0
        assert np.array_equal(out[ok, 1:], want[ok, 1:]), (trial, out, want)
        assert np.allclose(out[ok, 0], want[ok, 0], rtol=0, atol=1e-9), trial
        stood += sum(1 for v in st if v)
    assert stood > 0


def test_consensus_with_subpixel_keeps_the_list_order():
    """Method 1 first, the subpixel estimate second, the discrete search last: on identical overlaps (NCC 1.0 at the
    integer shift) neither later entry can be strictly better."""
    from image_stitcher_amd import registration
    g = synth.make_grid(131, 3, 3, 256, 256, overlap=0.125, jitter=3)
    want, want_st, _ = oreg.register_pairs_consensus(g.tiles, g.pair_idx, g.pair_dir, 32)
    out, st = registration.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32, consensus=True, subpixel=True)
    assert np.array_equal(st.cpu().numpy(), want_st)
    assert np.array_equal(out.cpu().numpy()[:, 1:], want[:, 1:])
    assert np.array_equal(want[:, 1:].astype(np.int64), g.expected_shift)
