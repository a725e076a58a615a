"""The mosaic born in chunk order (mistitch_fuse_chunked) against the chunk layout of the plain mosaic
(mistitch_chunk_layout of mistitch_fuse), bit for bit: both modes, with and without flatfields, mosaics whose sides
are no multiples of the chunk shape, tiles cut by the mosaic's edges, row bands, blocks with more tiles than the
per-block list holds, tiles narrower than a group, no tiles at all.  The plain mosaic is itself checked against the
oracle elsewhere (tests/test_gpu_fusion.py)."""
import numpy as np
import pytest

from tests.fusion_cases import overlapping_scene

pytestmark = pytest.mark.gpu


def _both(tiles, yx, plane, n_planes, mh, mw, chunk, mode, ffs=None, rows=None):
    import torch
    from image_stitcher_amd import fusion, omezarr
    tff = plane if ffs is not None else None
    plain = fusion.fuse_tiles(tiles, yx, plane, n_planes, mh, mw, flatfields=ffs, tile_ff=tff, mode=mode)
    want = omezarr.chunk_layout(plain, *chunk)
    if rows is None:
        got = fusion.fuse_tiles(tiles, yx, plane, n_planes, mh, mw, flatfields=ffs, tile_ff=tff, mode=mode, chunks=chunk)
    else:   # the bands in turn into one array that starts as 0xA5A5: everything is written exactly by its band
        got = torch.full(tuple(want.shape), 0x25A5, dtype=torch.int16, device="cuda").view(torch.uint16)
        for r0, r1 in rows:
            fusion.fuse_tiles(tiles, yx, plane, n_planes, mh, mw, flatfields=ffs, tile_ff=tff, mode=mode, chunks=chunk,
                              rows=(r0, r1), out=got)
    return got.cpu().view(torch.int16).numpy(), want.cpu().view(torch.int16).numpy()


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
@pytest.mark.parametrize("with_ff", [False, True])
@pytest.mark.parametrize("chunk", [(64, 64), (100, 136), (256, 2048), (37, 8)])
def test_chunked_equals_layout_of_plain(mode, with_ff, chunk):
    from image_stitcher_amd import synth
    tiles, yx, plane, mh, mw = overlapping_scene(5, 3, 4, 96, 160, n_planes=2, jitter=3, overlap=0.15)
    ffs = np.stack([synth.make_flatfield(96, 160, amplitude=0.3 - 0.1 * c) for c in range(2)]) if with_ff else None
    got, want = _both(tiles, yx, plane, 2, mh, mw, chunk, mode, ffs)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
def test_wide_mosaic_several_blocks_and_bands(mode):
    """A mosaic several 2048-pixel blocks wide and several 32-row blocks tall, fused in three bands."""
    from image_stitcher_amd import synth
    tiles, yx, plane, mh, mw = overlapping_scene(9, 2, 5, 120, 1100, n_planes=1, jitter=4, overlap=0.1)
    ffs = synth.make_flatfield(120, 1100)[None]
    for chunk in [(128, 1104), (64, 512)]:
        got, want = _both(tiles, yx, plane, 1, mh, mw, chunk, mode, ffs)
        assert np.array_equal(got, want)
        bands = [(0, 70), (70, 71), (71, mh)]
        got, want = _both(tiles, yx, plane, 1, mh, mw, chunk, mode, ffs, rows=bands)
        assert np.array_equal(got, want)


def test_tiles_beyond_every_edge_crowded_blocks_and_tiny_tiles():
    rng = np.random.default_rng(2)
    # 50 tiles on one spot (more than a block's list holds), tiles sticking out on all four sides
    n, th, tw, mh, mw = 50, 40, 56, 90, 150
    tiles = rng.integers(0, 65536, size=(n, th, tw), dtype=np.uint16)
    yx = np.stack([rng.integers(-20, mh - 10, size=n), rng.integers(-30, mw - 10, size=n)], axis=1).astype(np.int32)
    plane = np.zeros(n, dtype=np.int32)
    for mode in ("overwrite", "feather"):
        for chunk in [(32, 64), (90, 152), (16, 8)]:
            got, want = _both(tiles, yx, plane, 1, mh, mw, chunk, mode)
            assert np.array_equal(got, want), (mode, chunk)
    # tiles narrower than a group of 8 pixels: every group is resolved pixel by pixel
    tiles = rng.integers(0, 65536, size=(12, 9, 5), dtype=np.uint16)
    yx = np.stack([rng.integers(0, 20, size=12), rng.integers(0, 30, size=12)], axis=1).astype(np.int32)
    for mode in ("overwrite", "feather"):
        got, want = _both(tiles, yx, np.zeros(12, dtype=np.int32), 1, 30, 37, (16, 16), mode)
        assert np.array_equal(got, want), mode


def test_no_tiles_and_refusals():
    import torch
    from image_stitcher_amd import _lib, fusion
    empty = np.zeros((0, 16, 16), dtype=np.uint16)
    got = torch.full((2, 2, 3, 16, 24), 7, dtype=torch.int16, device="cuda").view(torch.uint16)
    fusion.fuse_tiles(empty, np.zeros((0, 2)), np.zeros(0), 2, 30, 70, chunks=(16, 24), rows=(5, 30), out=got)
    host = got.cpu().view(torch.int16).numpy()
    assert (host[:, 0, :, :5] == 7).all() and (host[:, 0, :, 5:] == 0).all() and (host[:, 1] == 0).all()
    with pytest.raises(_lib.MistitchError, match="multiple of 8"):
        fusion.fuse_tiles(np.zeros((1, 16, 16), dtype=np.uint16), [[0, 0]], [0], 1, 16, 16, chunks=(16, 12))
