"""The 160-candidate walk (flag MISTITCH_REG_PEAKS10) against vectors from the reference's own
``generate_translation_candidates`` / best-NCC fallback (tests/golden/make_golden_m1.py), needs an MI355X."""
import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import registration as oreg
from test_oracle_golden import _m1_golden, m1_strips

pytestmark = pytest.mark.gpu


def _tiles_with_strips(a, b, seed):
    """Two tiles whose facing edge strips are ``a`` and ``b``: (tiles, direction, edge, (off_y, off_x))."""
    h, w = a.shape
    rng = np.random.default_rng(seed)
    if h >= w:
        tw = 3 * w
        t1 = rng.integers(1000, 31000, size=(h, tw)).astype(np.uint16)
        t2 = rng.integers(1000, 31000, size=(h, tw)).astype(np.uint16)
        t1[:, tw - w:] = a
        t2[:, :w] = b
        return np.stack([t1, t2]), 0, w, (0, tw - w)
    th = 3 * h
    t1 = rng.integers(1000, 31000, size=(th, w)).astype(np.uint16)
    t2 = rng.integers(1000, 31000, size=(th, w)).astype(np.uint16)
    t1[th - h:, :] = a
    t2[:h, :] = b
    return np.stack([t1, t2]), 1, h, (th - h, 0)


@pytest.mark.parametrize("fft", ["f64", "f32"])
def test_candidate_table_and_fallback_match_the_reference(fft):
    import torch
    from image_stitcher_amd import registration
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    doc = _m1_golden()
    for s in doc["strips"]:
        a, b = m1_strips(s)
        tiles, direction, edge, off = _tiles_with_strips(a, b, s["seed"])
        out, status, table = registration.register_pairs(tiles, [[0, 1]], [direction], edge, fft=fft, peaks10=True,
                                                         return_candidates=True)
        row, table = out.cpu().numpy()[0], table.cpu().numpy()[0]
        assert int(status.cpu()[0]) == 0 and table.shape == (160, 3)
        if fft == "f32" and s["smooth"]:
            continue   # near-equal peaks of a smooth scene: complex64 need not rank them as complex128 does
        kept = np.isfinite(table[:, 2])
        assert np.array_equal(table[kept, :2], np.array(s["translations"])), s["name"]
        assert np.allclose(table[kept, 2], np.array(s["scores"]), rtol=0, atol=1e-12), s["name"]
        score, y, x = s["fallback"]
        assert (row[1], row[2]) == (y + off[0], x + off[1]), s["name"]
        assert row[0] == pytest.approx(score, abs=1e-12)
        # the default walk on the same tiles: the reference's two-peak discrete search
        plain, _ = registration.register_pairs(tiles, [[0, 1]], [direction], edge, fft=fft)
        d = s["discrete"]
        p = plain.cpu().numpy()[0]
        assert (p[1], p[2]) == (d[1] + off[0], d[2] + off[1]) and p[0] == pytest.approx(d[0], abs=1e-12)
    for g in doc["grids"]:
        grid = synth.make_grid(g["seed"], g["rows"], g["cols"], g["tile_h"], g["tile_w"], overlap=g["overlap"], jitter=g["jitter"])
        out, status, table = registration.register_pairs(grid.tiles, grid.pair_idx, grid.pair_dir, g["edge_width"], fft=fft,
                                                         peaks10=True, return_candidates=True)
        out, status, table = out.cpu().numpy(), status.cpu().numpy(), table.cpu().numpy()
        for p, exp in enumerate(g["pairs"]):
            assert (exp is None) == (status[p] != 0)
            if exp is None:
                assert np.isnan(table[p]).all()
                continue
            kept = np.isfinite(table[p, :, 2])
            assert int(kept.sum()) == exp["n"], (g["name"], p)
            assert float(table[p, kept, 2].sum()) == pytest.approx(exp["score_sum"], abs=1e-9)
            assert (out[p, 1], out[p, 2]) == tuple(exp["fallback"][1:]) and out[p, 0] == pytest.approx(exp["fallback"][0], abs=1e-12)
            first = table[p, kept][0]
            assert first[:2].tolist() == exp["first"][0] and first[2] == pytest.approx(exp["first"][1], abs=1e-12)


def test_peaks10_with_subpixel_and_passes():
    """The wide walk next to the subpixel candidate (which then only wins when strictly better), in passes
    smaller than the call, equals the oracle's combination."""
    import torch
    from image_stitcher_amd import _lib, registration
    from oracle import subpixel as osp
    g = synth.make_grid(131, 3, 3, 256, 256, overlap=0.125, jitter=3)
    small = _lib.Context(0, torch.cuda.current_stream().cuda_stream, pair_chunk=5)
    want, want_st, _ = oreg.register_pairs_peaks10(g.tiles, g.pair_idx, g.pair_dir, 32)
    for ctx in (None, small):
        out, st = registration.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32, peaks10=True, ctx=ctx)
        assert np.array_equal(st.cpu().numpy(), want_st)
        assert np.array_equal(out.cpu().numpy()[:, 1:], want[:, 1:])
        assert np.allclose(out.cpu().numpy()[:, 0], want[:, 0], rtol=0, atol=1e-12)
    sub, st = registration.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32, peaks10=True, subpixel=True)
    sub = sub.cpu().numpy()
    # identical overlaps score 1.0 on the integer shift already: the subpixel estimate cannot be strictly better
    assert np.array_equal(sub[:, 1:], want[:, 1:])
    assert np.array_equal(want[:, 1:].astype(np.int64), g.expected_shift)


def test_candidate_tables_of_a_call_large_enough_for_the_long_row_blocks():
    """A call of many pairs sums its candidates in workgroups of 64 rows, a small one in workgroups of 32: 144 pairs
    of 128 x 256 strips with the ten-peak set are enough for the former (several row blocks per strip, cut at the wrap
    rows of both row lags).  Every kept row of every pair's table against the oracle's."""
    import torch
    from image_stitcher_amd import registration
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    g = synth.make_grid(977, 9, 9, 256, 256, overlap=0.3, jitter=6)
    assert len(g.pair_idx) == 144
    out, status, table = registration.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 128, peaks10=True, return_candidates=True)
    out, status, table = out.cpu().numpy(), status.cpu().numpy(), table.cpu().numpy()
    want, want_st, tables = oreg.register_pairs_peaks10(g.tiles, g.pair_idx, g.pair_dir, 128)
    assert np.array_equal(status, want_st)
    for p, t in enumerate(tables):
        if t is None:
            continue
        translations, scores = t
        kept = np.isfinite(table[p, :, 2])
        e1_off = out[p, 1:] - want[p, 1:]          # (both in tile coordinates: zero)
        assert not e1_off.any()
        strip = table[p, kept]
        assert np.array_equal(strip[:, :2], np.asarray(translations, dtype=np.float64).reshape(-1, 2)), p
        assert np.allclose(strip[:, 2], np.asarray(scores, dtype=np.float64), rtol=0, atol=1e-12), p
    assert np.allclose(out[:, 0], want[:, 0], rtol=0, atol=1e-12)
    assert np.array_equal(out[:, 1:].astype(np.int64), g.expected_shift)
