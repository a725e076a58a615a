"""HIP fusion / flatfield / projection kernels against the oracle (needs an MI355X)."""
import numpy as np
import pytest

from image_stitcher_amd import synth
from oracle import fusion as ofu
from fusion_cases import reference_fixture_grid, overlapping_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fu():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from image_stitcher_amd import fusion
    return fusion


def _np(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("shape", [(3, 3, 64, 64, 1), (2, 4, 96, 200, 2), (4, 3, 128, 520, 3), (1, 1, 33, 17, 1)])
def test_overwrite_matches_oracle(fu, shape):
    rows, cols, th, tw, planes = shape
    tiles, yx, plane, mh, mw = overlapping_scene(7 + rows, rows, cols, th, tw, n_planes=planes)
    ref = ofu.fuse_overwrite(tiles, yx, plane, planes, mh, mw)
    got = _np(fu.fuse_tiles(tiles, yx, plane, planes, mh, mw))
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("ffdt", [np.float32, np.float64])
@pytest.mark.parametrize("mode", ["overwrite", "feather"])
def test_flatfield_fusion_matches_oracle(fu, ffdt, mode):
    tiles, yx, plane, mh, mw = overlapping_scene(21, 3, 3, 72, 136, n_planes=2)
    ffs = np.stack([synth.make_flatfield(72, 136, dtype=ffdt, amplitude=0.5),
                    synth.make_flatfield(72, 136, dtype=ffdt, amplitude=0.1)])
    tile_ff = np.where(plane == 0, 0, 1).astype(np.int32)
    tile_ff[::5] = -1  # some tiles have no flatfield: left untouched like a missing channel
    oracle = ofu.fuse_overwrite if mode == "overwrite" else ofu.fuse_feather
    ref = oracle(tiles, yx, plane, 2, mh, mw, flatfields=ffs, tile_ff=tile_ff)
    got = _np(fu.fuse_tiles(tiles, yx, plane, 2, mh, mw, flatfields=ffs, tile_ff=tile_ff, mode=mode))
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
@pytest.mark.parametrize("case", ["odd_value_in_plane_37", "odd_value_in_an_unused_plane", "planes_beyond_63"])
def test_flatfield_range_check_over_many_planes(fu, mode, case):
    """The float32 division is scaling-free only when every value of every USED plane is an ordinary one; which
    planes are used is worked out on the device from tile_ff (planes 0..63 one by one, any plane beyond them
    switches to the full division).  Whatever path is taken, results equal the oracle's."""
    tiles, yx, plane, mh, mw = overlapping_scene(77, 3, 3, 40, 72, n_planes=1)
    n_ff = 72 if case == "planes_beyond_63" else 40
    ffs = np.stack([synth.make_flatfield(40, 72, dtype=np.float32, amplitude=0.05 + 0.01 * (k % 30)) for k in range(n_ff)])
    tile_ff = np.array([0, 37, 5, -1, 33, 37, 31, 32, 0], dtype=np.int32)
    if case == "odd_value_in_plane_37":
        ffs[37, 11, 13], ffs[37, 39, 71], ffs[37, 0, 0] = 1e-30, np.inf, -2.0
    elif case == "odd_value_in_an_unused_plane":
        ffs[36, 11, 13], ffs[38, 0, 0], ffs[1, 5, 5] = 1e-30, np.inf, np.nan
    else:
        tile_ff[[2, 6]] = [64, 71]
        ffs[71, 3, 3] = 1e-30
    oracle = ofu.fuse_overwrite if mode == "overwrite" else ofu.fuse_feather
    with np.errstate(all="ignore"):
        ref = oracle(tiles, yx, plane, 1, mh, mw, flatfields=ffs, tile_ff=tile_ff)
    got = _np(fu.fuse_tiles(tiles, yx, plane, 1, mh, mw, flatfields=ffs, tile_ff=tile_ff, mode=mode))
    assert np.array_equal(got, ref)


def test_feather_matches_oracle_and_properties(fu):
    tiles, yx, plane, mh, mw = overlapping_scene(31, 3, 4, 80, 112, n_planes=1, overlap=0.25)
    ref = ofu.fuse_feather(tiles, yx, plane, 1, mh, mw)
    got = _np(fu.fuse_tiles(tiles, yx, plane, 1, mh, mw, mode="feather"))
    diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert diff.max() <= 1          # stated tolerance: 1 LSB
    assert diff.max() == 0          # and in fact identical: same float32 operation order
    over = _np(fu.fuse_tiles(tiles, yx, plane, 1, mh, mw, mode="overwrite"))
    cover = np.zeros((1, mh, mw), dtype=int)
    for t in range(len(tiles)):
        y, x = yx[t]
        cover[0, y:y + 80, x:x + 112] += 1
    assert np.array_equal(got[cover == 1], over[cover == 1])
    assert (got[cover == 0] == 0).all()


def test_row_bands_compose(fu):
    """Chunk-row sharding: fusing disjoint row bands separately equals one full pass."""
    import torch
    tiles, yx, plane, mh, mw = overlapping_scene(41, 4, 3, 64, 96, n_planes=2)
    full = _np(fu.fuse_tiles(tiles, yx, plane, 2, mh, mw))
    out = torch.zeros((2, mh, mw), dtype=torch.int16, device="cuda").view(torch.uint16)
    cuts = [0, 37, 38, 150, mh]
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        fu.fuse_tiles(tiles, yx, plane, 2, mh, mw, rows=(r0, r1), out=out)
    assert np.array_equal(_np(out), full)


def test_many_tiles_on_one_block_use_the_slow_path(fu):
    """More tiles on one output block than the in-LDS list holds."""
    rng = np.random.default_rng(3)
    n = 50
    tiles = rng.integers(0, 65536, size=(n, 24, 40), dtype=np.uint16)
    yx = rng.integers(0, 12, size=(n, 2)).astype(np.int32)
    plane = np.zeros(n, dtype=np.int32)
    for mode, oracle in (("overwrite", ofu.fuse_overwrite), ("feather", ofu.fuse_feather)):
        ref = oracle(tiles, yx, plane, 1, 40, 56)
        got = _np(fu.fuse_tiles(tiles, yx, plane, 1, 40, 56, mode=mode))
        assert np.array_equal(got, ref), mode


@pytest.mark.parametrize("with_ff", [False, True])
def test_crowded_block_within_the_list(fu, with_ff):
    """17 to 32 tiles on one output block: list indices beyond 15 reach the batched paths (two sources)
    or must fall back to the pixel-by-pixel one (three or four sources); tile edges cut groups everywhere."""
    rng = np.random.default_rng(11)
    n, th, tw = 28, 40, 72
    tiles = rng.integers(0, 65536, size=(n, th, tw), dtype=np.uint16)
    # a loose 4 x 7 arrangement with large random offsets: between one and five tiles over most pixels
    yx = np.array([[r * 22 + rng.integers(0, 16), c * 40 + rng.integers(0, 30)] for r in range(4) for c in range(7)],
                  dtype=np.int32)
    plane = np.zeros(n, dtype=np.int32)
    mh, mw = int(yx[:, 0].max()) + th, int(yx[:, 1].max()) + tw
    ffs = synth.make_flatfield(th, tw, amplitude=0.4)[None] if with_ff else None
    tile_ff = np.zeros(n, dtype=np.int32) if with_ff else None
    for mode, oracle in (("overwrite", ofu.fuse_overwrite), ("feather", ofu.fuse_feather)):
        ref = oracle(tiles, yx, plane, 1, mh, mw, flatfields=ffs, tile_ff=tile_ff)
        got = _np(fu.fuse_tiles(tiles, yx, plane, 1, mh, mw, flatfields=ffs, tile_ff=tile_ff, mode=mode))
        assert np.array_equal(got, ref), mode


def test_reference_known_answers(fu):
    names, x_mm, y_mm, zs, tiles = reference_fixture_grid(3, 3, 1000)
    infos = [fu.TileInfo(n, x, y, z, "DAPI") for n, x, y, z in zip(names, x_mm, y_mm, zs)]
    calls = []
    im = fu.stitch_region(tiles, infos, ["DAPI"], 1.0, update_progress=lambda d, t: calls.append((d, t)))
    assert tuple(im.shape) == (1, 1, 1, 3000, 3000)
    im = _np(im)
    probes = {(0, 0): 0, (1000, 0): 3, (1500, 0): 3, (2000, 0): 6, (0, 1000): 1, (0, 1500): 1,
              (0, 2000): 2, (2999, 2999): 8}
    for (y, x), v in probes.items():
        assert im[0, 0, 0, y, x] == v
    assert calls == [(i, 9) for i in range(9)]


def test_stitch_region_z_and_channels(fu):
    names, x_mm, y_mm, zs, tiles = reference_fixture_grid(2, 2, 50, num_z=3)
    infos = [fu.TileInfo(n, x, y, z, "DAPI") for n, x, y, z in zip(names, x_mm, y_mm, zs)]
    # 50 um fields on a 1 mm pitch would leave gaps; use 20 um pixels so tiles abut
    px = 20.0
    im = _np(fu.stitch_region(tiles, infos, ["DAPI"], px, selected_z=[1]))
    assert im.shape == (1, 1, 1, 100, 100)
    assert im[0, 0, 0, 0, 0] == 1 and im[0, 0, 0, 0, 50] == 4 and im[0, 0, 0, 50, 0] == 7 and im[0, 0, 0, 99, 99] == 10
    im = _np(fu.stitch_region(tiles, infos, ["DAPI"], px))
    assert im.shape == (1, 1, 3, 100, 100)
    assert im[0, 0, 2, 60, 60] == 11


def test_apply_flatfield_vectors_from_reference(fu, golden):
    f = golden["flatfield"]
    h, w = f["shape"]
    ramp = (np.arange(h * w, dtype=np.uint32).reshape(h, w) * 11 % 65536).astype(np.uint16)
    ramp[0, :8] = [0, 1, 65535, 65534, 40000, 50000, 60000, 32767]
    for dt, exp in f["results"].items():
        ff = synth.make_flatfield(h, w, dtype=np.dtype(dt), amplitude=f["amplitude"])
        got = fu.apply_flatfield_to_image(ramp, ff)
        assert got.dtype == np.uint16
        assert np.array_equal(got, ofu.apply_flatfield(ramp, ff))
        assert int(got.astype(np.int64).sum()) == exp["sum"]
        assert got[0].tolist() == exp["first_row"] and got[-1].tolist() == exp["last_row"]
    with pytest.raises(ValueError):
        fu.apply_flatfield_to_image(ramp, np.ones((h, w + 1), dtype=np.float32))
    # odd sizes exercise the scalar tail
    odd = ramp[:13, :11].copy()
    ffo = synth.make_flatfield(13, 11, amplitude=0.3)
    stack = np.stack([odd, odd[::-1].copy()])
    assert np.array_equal(fu.apply_flatfield_to_image(stack, ffo),
                          np.stack([ofu.apply_flatfield(s, ffo) for s in stack]))


def test_mip(fu):
    tiles = [np.array([[1, 2], [3, 4]], dtype=np.uint16), np.array([[5, 1], [7, 3]], dtype=np.uint16),
             np.array([[2, 6], [1, 0]], dtype=np.uint16)]
    assert np.array_equal(fu.compute_mip(tiles), np.array([[5, 6], [7, 4]]))
    rng = np.random.default_rng(0)
    stack = rng.integers(0, 65536, size=(5, 37, 53), dtype=np.uint16)
    assert np.array_equal(fu.compute_mip(list(stack)), ofu.compute_mip(list(stack)))
    with pytest.raises(ValueError):
        fu.compute_mip([])


@pytest.mark.parametrize("shape,levels", [((1, 2, 1, 300, 517), 4), ((3, 64, 256), 7), ((130, 1031), 6), ((2, 7, 9), 3)])
def test_pyramid_levels_match_block_means(shape, levels):
    """Every level bit-exact against np.mean over 2^l blocks (float64, trimmed)."""
    from image_stitcher_amd import fusion
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 65536, size=shape).astype(np.uint16)
    got = fusion.generate_pyramid(img, levels)
    want = ofu.generate_pyramid(img, levels)
    assert len(got) == len(want) == levels
    for l in range(levels):
        assert got[l].shape == want[l].shape, l
        assert got[l].dtype == want[l].dtype, l
        assert np.array_equal(got[l], want[l]), l


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
def test_plate_of_wells_as_planes(fu, mode):
    """A multi-well plate (many small independent grids): every (well, channel) is one plane of
    a single fuse call, sized for the largest well; pairs of all wells go through one
    registration call.  Each plane must equal that well fused on its own by the oracle."""
    from image_stitcher_amd import registration, solve
    from oracle import registration as oreg
    wells, rows, cols, th, tw, ch = 6, 2, 3, 96, 128, 2
    rng = np.random.default_rng(5)
    tiles, yx, plane, ffidx, pair_idx, pair_dir = [], [], [], [], [], []
    dims = []
    per_well = rows * cols
    for w in range(wells):
        for c in range(ch):
            g = synth.make_grid(300 + 10 * w + c, rows, cols, th, tw, overlap=0.125, jitter=2 + w % 2)
            if c == 0:
                well_yx = (g.true_yx - g.true_yx.min(axis=0)).astype(np.int32)
                dims.append((int(well_yx[:, 0].max()) + th, int(well_yx[:, 1].max()) + tw))
                pair_idx.append(g.pair_idx + len(tiles))
                pair_dir.append(g.pair_dir)
            else:   # other channels of a well are cut at the same positions from another scene
                scene = synth.make_scene(900 + 10 * w + c, (rows - 1) * g.pitch_y + th + 8, (cols - 1) * g.pitch_x + tw + 8)
                for t in range(per_well):
                    y0, x0 = well_yx[t]
                    g.tiles[t] = scene[y0:y0 + th, x0:x0 + tw]
            tiles.extend(g.tiles)
            yx.extend(well_yx)
            plane.extend([w * ch + c] * per_well)
            ffidx.extend([c] * per_well)
    tiles, yx = np.stack(tiles), np.stack(yx)
    pair_idx, pair_dir = np.concatenate(pair_idx), np.concatenate(pair_dir)
    mh, mw = max(d[0] for d in dims), max(d[1] for d in dims)
    ffs = np.stack([synth.make_flatfield(th, tw, amplitude=0.3 - 0.1 * c) for c in range(ch)])
    # registration of every well's pairs in one call, then one solve per well
    out, status = registration.register_pairs(tiles, pair_idx, pair_dir, 16, flatfield=None)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    exp_out, exp_status = oreg.register_pairs(tiles, pair_idx, pair_dir, 16)
    assert np.array_equal(status, exp_status) and np.array_equal(out[:, 1:], exp_out[:, 1:])
    ppw = len(pair_idx) // wells
    row, col = np.repeat(np.arange(rows), cols), np.tile(np.arange(cols), rows)
    for w in range(wells):
        sl = slice(w * ppw, (w + 1) * ppw)
        first_tile = w * ch * per_well
        res = solve.solve_positions(solve.grid_from_pairs(row, col, pair_idx[sl] - first_tile, pair_dir[sl], out[sl],
                                                          status[sl]), th, tw, pou=8)
        assert np.array_equal(np.stack([res.y_pos, res.x_pos], axis=1), yx[first_tile:first_tile + per_well])
    got = _np(fu.fuse_tiles(tiles, yx, plane, wells * ch, mh, mw, flatfields=ffs, tile_ff=ffidx, mode=mode))
    fuse = ofu.fuse_overwrite if mode == "overwrite" else ofu.fuse_feather
    for w in range(wells):
        sl = slice(w * ch * per_well, (w + 1) * ch * per_well)
        h, wd = dims[w]
        alone = fuse(tiles[sl], yx[sl], [p - w * ch for p in plane[sl]], ch, h, wd, flatfields=ffs, tile_ff=ffidx[sl])
        assert np.array_equal(got[w * ch:(w + 1) * ch, :h, :wd], alone), w
        assert (got[w * ch:(w + 1) * ch, h:] == 0).all() and (got[w * ch:(w + 1) * ch, :, wd:] == 0).all()


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
def test_tiles_sticking_out_on_every_side_are_clipped(fu, mode):
    """Tile origins left of and above the mosaic (negative coordinates) as well as tiles past its right
    and lower edge: the mosaic equals the window of a larger oracle canvas."""
    tiles, yx, plane, mh, mw = overlapping_scene(51, 3, 4, 64, 96, n_planes=2)
    oracle = ofu.fuse_overwrite if mode == "overwrite" else ofu.fuse_feather
    canvas = oracle(tiles, yx, plane, 2, mh, mw)
    oy, ox, wh, ww = 37, 53, mh - 37 - 21, mw - 53 - 30        # a window strictly inside the canvas
    shifted = yx.copy()
    shifted[:, 0] -= oy
    shifted[:, 1] -= ox
    assert (shifted < 0).any()
    got = _np(fu.fuse_tiles(tiles, shifted, plane, 2, wh, ww, mode=mode))
    assert np.array_equal(got, canvas[:, oy:oy + wh, ox:ox + ww])


def _acquisition(seed, rows, cols, th, tw, num_z, channels, rgb=False, lead1=False):
    """Files R0_{fov}_{z}_{channel}.tiff of an overlapping grid with stage positions in mm at 1000 um per pixel."""
    rng = np.random.default_rng(seed)
    names, xs, ys, keys, tiles = [], [], [], [], []
    for fov in range(rows * cols):
        r, c = divmod(fov, cols)
        y, x = r * (th - 9) + int(rng.integers(0, 4)), c * (tw - 13) + int(rng.integers(0, 4))
        for z in range(num_z):
            for ch in channels:
                names.append(f"R0_{fov}_{z}_{ch}.tiff")
                xs.append(float(x))
                ys.append(float(y))
                keys.append((fov, z, ch))
                shape = (th, tw, 3) if rgb else ((1, th, tw) if lead1 else (th, tw))
                tiles.append(rng.integers(0, 65536, size=shape).astype(np.uint16))
    return names, xs, ys, keys, tiles


@pytest.mark.parametrize("apply_mip", [False, True])
@pytest.mark.parametrize("kind", ["mono", "rgb", "lead1"])
def test_stitch_region_rgb_squeeze_and_mip_match_the_reference_loop(fu, kind, apply_mip):
    """RGB tiles fan out to the {base}_R/_G/_B planes, (1, H, W) tiles are squeezed, and with apply_mip every
    (fov, channel) z-stack is projected and placed at z = 0: all against the restated reference loop."""
    rgb, lead1 = kind == "rgb", kind == "lead1"
    channels = ["BF_LED_matrix_full"] if rgb else ["Fluorescence_488_nm_Ex", "Fluorescence_561_nm_Ex"]
    mono = ["BF_R", "BF_G", "BF_B"] if rgb else channels
    names, xs, ys, keys, tiles = _acquisition(3, 2, 3, 40, 56, 3, channels, rgb=rgb, lead1=lead1)
    infos = [fu.TileInfo(n, x, y, k[1], k[2]) for n, x, y, k in zip(names, xs, ys, keys)]
    for selected in (None, [1, 2]):
        calls, ref_calls = [], []
        got = _np(fu.stitch_region(np.stack(tiles), infos, mono, 1000.0, selected_z=selected, apply_mip=apply_mip,
                                   update_progress=lambda d, t: calls.append((d, t))))
        ref = ofu.stitch_region(tiles, keys, names, xs, ys, mono, 1000.0, selected_z=selected, apply_mip=apply_mip,
                                progress=ref_calls)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref)
        assert calls == ref_calls
    # flatfields keyed by plane index, one plane left without
    ffs = {0: synth.make_flatfield(40, 56, amplitude=0.4), (2 if rgb else 1): synth.make_flatfield(40, 56, amplitude=0.1)}
    got = _np(fu.stitch_region(np.stack(tiles), infos, mono, 1000.0, flatfields=ffs, apply_flatfield=True, apply_mip=apply_mip))
    ref = ofu.stitch_region(tiles, keys, names, xs, ys, mono, 1000.0, apply_mip=apply_mip, flatfields=ffs)
    assert np.array_equal(got, ref)


def test_mip_known_answers_of_the_reference_tests(fu):
    """The arrays of the reference's test_compute_mip (stitcher_test.py:167-249), 3-D tiles included."""
    t3 = [np.array([[[1, 10, 100], [2, 20, 200]], [[3, 30, 300], [4, 40, 400]]], dtype=np.uint16),
          np.array([[[5, 15, 105], [6, 25, 205]], [[7, 35, 305], [8, 45, 405]]], dtype=np.uint16)]
    want = np.array([[[5, 15, 105], [6, 25, 205]], [[7, 35, 305], [8, 45, 405]]])
    assert np.array_equal(fu.compute_mip(t3), want)
    assert np.array_equal(fu.compute_mip([np.ones((3, 2), dtype=np.uint16)] * 2), np.ones((3, 2)))
    assert np.array_equal(fu.compute_mip([np.zeros((2, 2, 3), dtype=np.uint16)] * 2), np.zeros((2, 2, 3)))
    # a constant-tile acquisition as the reference's fixture writes it: value = fov * num_z + z, so the
    # projection of a field of view is its highest z level
    names, x_mm, y_mm, zs, tiles = reference_fixture_grid(2, 2, 50, num_z=3)
    infos = [fu.TileInfo(n, x, y, z, "DAPI") for n, x, y, z in zip(names, x_mm, y_mm, zs)]
    im = _np(fu.stitch_region(tiles, infos, ["DAPI"], 20.0, apply_mip=True))
    assert im.shape == (1, 1, 1, 100, 100)
    assert im[0, 0, 0, 0, 0] == 2 and im[0, 0, 0, 0, 50] == 5 and im[0, 0, 0, 50, 0] == 8 and im[0, 0, 0, 99, 99] == 11


@pytest.mark.parametrize("static_workspace", [False, True])
@pytest.mark.parametrize("mode", ["overwrite", "feather"])
@pytest.mark.parametrize("with_ff", [False, True])
def test_fusing_no_tiles_zeroes_the_band_and_nothing_else(fu, mode, static_workspace, with_ff):
    """Zero tiles (a NULL tile array): the band is what the reference's zero-initialised output holds
    (stitcher.py:154-181), rows outside the band keep the caller's contents; both entry points."""
    import torch
    mh, mw, planes = 300, 1037, 2
    out = torch.full((planes, mh, mw), 0x5a5a, dtype=torch.int16, device="cuda").view(torch.uint16)
    empty = torch.empty((0, 64, 64), dtype=torch.int16, device="cuda").view(torch.uint16)
    assert empty.data_ptr() == 0
    ffs = np.ones((1, 64, 64), dtype=np.float32) if with_ff else None
    tff = np.zeros((0,), dtype=np.int32) if with_ff else None
    got = fu.fuse_tiles(empty, np.zeros((0, 2), np.int32), np.zeros((0,), np.int32), planes, mh, mw, flatfields=ffs,
                        tile_ff=tff, mode=mode, rows=(40, 173), out=out, static_workspace=static_workspace)
    got = _np(got)
    assert (got[:, 40:173] == 0).all()
    assert (got[:, :40] == 0x5a5a).all() and (got[:, 173:] == 0x5a5a).all()


def test_stitch_region_of_a_z_level_no_tile_has(fu):
    """selected_z names a level without tiles: the reference's loop places nothing and returns zeros."""
    infos = [fu.TileInfo(f"R0_{i}_0_c.tiff", 0.1 * i, 0.0, 0, "c") for i in range(3)]
    tiles = np.full((3, 32, 32), 7, dtype=np.uint16)
    out = _np(fu.stitch_region(tiles, infos, ["c"], 1.0, selected_z=[5]))
    assert out.shape[:3] == (1, 1, 1) and (out == 0).all()


@pytest.mark.parametrize("mode", ["overwrite", "feather"])
@pytest.mark.parametrize("with_ff", [False, True])
def test_a_tile_array_smaller_than_one_16_byte_load(fu, mode, with_ff):
    """One 2 x 3 tile (12 bytes; its flatfield plane 24 bytes): no lane may read 16 bytes at the array's start."""
    tiles = np.array([[[10, 20, 30], [40, 50, 60]]], dtype=np.uint16)
    yx, plane = np.array([[1, 2]], np.int32), np.zeros(1, np.int32)
    ffs = np.array([[[1.0, 2.0, 0.5], [4.0, 1.0, 1.0]]], dtype=np.float32) if with_ff else None
    tff = np.zeros(1, np.int32) if with_ff else None
    oracle = ofu.fuse_overwrite if mode == "overwrite" else ofu.fuse_feather
    ref = oracle(tiles, yx, plane, 1, 5, 9, flatfields=ffs, tile_ff=tff)
    got = _np(fu.fuse_tiles(tiles, yx, plane, 1, 5, 9, flatfields=ffs, tile_ff=tff, mode=mode))
    assert np.array_equal(got, ref)
