"""Tile ingest: band planning on the CPU, the streamed upload + registration on the GPU."""
import numpy as np
import pytest

from image_stitcher_amd import ingest, synth


def test_bands_release_every_pair_once_and_only_when_its_tiles_are_up():
    for rows, cols, per in ((5, 4, 2), (3, 7, 1), (4, 4, 4), (6, 3, 5)):
        pair_idx, _ = synth.grid_pairs(rows, cols)
        bands = ingest.plan_bands(rows, cols, pair_idx, per)
        assert bands[0].tile0 == 0 and bands[-1].tile1 == rows * cols
        assert all(a.tile1 == b.tile0 for a, b in zip(bands[:-1], bands[1:]))
        released = np.concatenate([b.pairs for b in bands])
        assert sorted(released.tolist()) == list(range(len(pair_idx)))
        for b in bands:
            assert b.tile1 - b.tile0 <= per * cols
            assert (pair_idx[b.pairs] < b.tile1).all()          # nothing that is still on the host
            assert (pair_idx[b.pairs][:, 1] >= b.tile0).all()   # and nothing an earlier band could have released
    with pytest.raises(ValueError):
        ingest.plan_bands(2, 2, [[1, 0]], 1)
    with pytest.raises(ValueError):
        ingest.plan_bands(2, 2, [[0, 1]], 0)


@pytest.mark.gpu
@pytest.mark.parametrize("per", [1, 2, 5])
def test_streamed_registration_equals_the_resident_one(per):
    import torch
    from image_stitcher_amd import registration
    g = synth.make_grid(41, 5, 4, 256, 320, overlap=0.125, jitter=3)
    ff = synth.make_flatfield(256, 320, amplitude=0.25)
    calls = []

    def fill(lo, hi, dst):
        assert dst.shape == (hi - lo, 256, 320) and dst.dtype == np.uint16
        calls.append((lo, hi))
        np.copyto(dst, g.tiles[lo:hi])

    for flat in (None, ff):
        calls.clear()
        tiles, out, status = ingest.register_streaming(fill, 5, 4, 256, 320, g.pair_idx, g.pair_dir, edge_width=32,
                                                       rows_per_band=per, flatfield=flat)
        want, want_st = registration.register_pairs(g.tiles, g.pair_idx, g.pair_dir, 32, flatfield=flat)
        torch.cuda.synchronize()
        assert calls == [(lo, min(20, lo + 4 * per)) for lo in range(0, 20, 4 * per)]
        assert torch.equal(tiles.cpu(), torch.from_numpy(g.tiles.view(np.int16)).view(torch.uint16))
        assert torch.equal(status, want_st) and torch.equal(out, want)
    assert np.array_equal(out.cpu().numpy()[:, 1:].astype(np.int64), g.expected_shift)


def test_mosaic_bands_cover_the_rows_and_list_straddling_tiles_twice():
    y = np.array([0, 3, 100, 190, 200, 390])
    bands = ingest.plan_mosaic_bands(y, 100, 490, 128)
    assert [(a, b) for a, b, _ in bands] == [(0, 128), (128, 256), (256, 384), (384, 490)]
    assert [idx.tolist() for _, _, idx in bands] == [[0, 1, 2], [2, 3, 4], [3, 4], [5]]
    assert [(a, b) for a, b, _ in ingest.plan_mosaic_bands(y, 100, 490, 200, rows=(100, 450))] == [(100, 300), (300, 450)]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["overwrite", "feather"])
@pytest.mark.parametrize("band_rows", [64, 100, 1000])
def test_streamed_fusion_equals_the_one_shot_mosaic(mode, band_rows):
    from image_stitcher_amd import fusion
    g = synth.make_grid(43, 4, 3, 96, 136, overlap=0.15, jitter=3)
    ch = 2
    other = synth.make_grid(44, 4, 3, 96, 136, overlap=0.15, jitter=3)
    tiles = np.concatenate([g.tiles, other.tiles])
    yx = np.tile((g.true_yx - g.true_yx.min(axis=0)).astype(np.int32), (ch, 1))
    plane = np.repeat(np.arange(ch), 12).astype(np.int32)
    ffs = np.stack([synth.make_flatfield(96, 136, amplitude=0.3 - 0.1 * c) for c in range(ch)])
    mh, mw = int(yx[:, 0].max()) + 96, int(yx[:, 1].max()) + 136
    want = fusion.fuse_tiles(tiles, yx, plane, ch, mh, mw, flatfields=ffs, tile_ff=plane, mode=mode).cpu().numpy()
    got = np.full((ch, mh, mw), 0xABCD, dtype=np.uint16)
    seen = []

    filled = []

    def fill(idx, dst):
        assert (np.diff(idx) > 0).all() and dst.shape == (len(idx), 96, 136)
        filled.extend(idx.tolist())
        np.copyto(dst, tiles[idx])

    def drain(r0, r1, src):
        seen.append((r0, r1))
        got[:, r0:r1] = src

    n = ingest.fuse_streaming(fill, drain, yx, plane, ch, 96, 136, mh, mw, band_rows=band_rows, flatfields=ffs,
                              tile_ff=plane, mode=mode)
    assert n == len(seen) == -(-mh // band_rows)
    assert seen == [(r, min(mh, r + band_rows)) for r in range(0, mh, band_rows)]
    assert np.array_equal(got, want)
    assert sorted(filled) == list(range(len(tiles)))      # every tile decoded and sent once, straddling or not


@pytest.mark.gpu
def test_host_to_host_stitch_reassembles_the_scenes():
    """Two channels cut from two scenes at the same jittered positions; channel 0 registers.  The
    streamed path must hand back mosaics whose covered pixels are the scenes'."""
    rows, cols, t, jitter = 3, 4, 256, 3
    g0 = synth.make_grid(61, rows, cols, t, t, overlap=0.125, jitter=jitter)
    sh, sw = (rows - 1) * g0.pitch_y + t + 2 * jitter, (cols - 1) * g0.pitch_x + t + 2 * jitter
    scenes = [synth.make_scene(61, sh, sw), synth.make_scene(62, sh, sw)]
    tiles = np.stack([sc[y:y + t, x:x + t] for sc in scenes for y, x in g0.true_yx])
    assert np.array_equal(tiles[:rows * cols], g0.tiles)
    n = rows * cols
    plane = np.repeat(np.arange(2), n).astype(np.int32)
    grid_index = np.tile(np.arange(n), 2)
    bands = {}

    def drain(r0, r1, src):
        bands[r0] = src.copy()

    pos, mh, mw, out, status = ingest.stitch_streaming(
        ingest.fill_from_array(g0.tiles), lambda idx, dst: np.copyto(dst, tiles[idx]), drain, rows, cols, t, t, 2, plane,
        grid_index, edge_width=32, band_rows=200)
    assert (status == 0).all() and np.array_equal(out[:, 1:].astype(np.int64), g0.expected_shift)
    mosaic = np.concatenate([bands[r] for r in sorted(bands)], axis=1)
    assert mosaic.shape == (2, mh, mw)
    y0, x0 = g0.true_yx.min(axis=0)
    covered = np.zeros((mh, mw), dtype=bool)
    for y, x in zip(pos.y_pos, pos.x_pos):
        covered[y:y + t, x:x + t] = True
    for c in range(2):
        want = scenes[c][y0:y0 + mh, x0:x0 + mw]
        assert np.array_equal(mosaic[c][covered], want[covered])
        assert (mosaic[c][~covered] == 0).all()
