"""One acquisition sharded over two "ranks", run one after the other on ONE GPU through the HIP path
(needs an MI355X): what each rank of a node does, minus the collective.  The rows gathered from the
two ranks and the mosaic assembled from their bands must equal the unsharded calls bit for bit."""
import numpy as np
import pytest

from image_stitcher_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("with_ff", [False, True])
def test_sharded_ranks_equal_the_unsharded_call(world, with_ff):
    import torch
    from image_stitcher_amd import dist as mdist
    from image_stitcher_amd import fusion, registration, solve
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    rows, cols, tile, edge, ch = 5, 4, 512, 64, 2
    g0 = synth.make_grid(71, rows, cols, tile, tile, overlap=0.10, jitter=3)
    g1 = synth.make_grid(72, rows, cols, tile, tile, overlap=0.10, jitter=3)
    g1_tiles = np.stack([synth.make_scene(73, 4000, 4000)[y:y + tile, x:x + tile] for y, x in g0.true_yx])
    acq = np.stack([g0.tiles, g1_tiles])                       # [C, N, H, W]; channel 0 is registered
    ffs = np.stack([synth.make_flatfield(tile, tile, amplitude=0.25 - 0.1 * c) for c in range(ch)]) if with_ff else None
    whole, whole_st = registration.register_pairs(acq[0], g0.pair_idx, g0.pair_dir, edge,
                                                  flatfield=ffs[0] if with_ff else None)
    # every rank's part, through the same entry points, then the reorder the all-gather ends with
    parts, shards = [], []
    for rank in range(world):
        rs = mdist.RegistrationShard(g0.pair_idx, rows, cols, world, rank, by=("pairs" if world == 3 else "grid_rows"))
        local_ids = rs.tile_ids
        out, st = registration.register_pairs(acq[0][local_ids], rs.local_pairs(local_ids), g0.pair_dir[rs.pair_ids], edge,
                                              flatfield=ffs[0] if with_ff else None)
        parts.append((out, st))
        shards.append(rs)
        assert len(local_ids) < rows * cols                    # a rank holds its band and one halo row only
    g_rows, g_st = mdist.scatter_to_global([s.pair_ids for s in shards], torch.cat([p[0] for p in parts]),
                                           torch.cat([p[1] for p in parts]), len(g0.pair_idx))
    assert torch.equal(g_st, whole_st) and torch.equal(g_rows, whole)
    # replicated solve, then the bands
    row, col = np.repeat(np.arange(rows), cols), np.tile(np.arange(cols), rows)
    res = solve.solve_positions(solve.grid_from_pairs(row, col, g0.pair_idx, g0.pair_dir, g_rows.cpu().numpy(),
                                                      g_st.cpu().numpy()), tile, tile)
    yx = np.stack([res.y_pos, res.x_pos], axis=1).astype(np.int32)
    assert np.array_equal(yx, g0.true_yx - g0.true_yx.min(axis=0))
    mh, mw = int(yx[:, 0].max()) + tile, int(yx[:, 1].max()) + tile
    n = rows * cols
    plane = np.repeat(np.arange(ch), n).astype(np.int32)
    flat = acq.reshape(ch * n, tile, tile)
    for mode in ("overwrite", "feather"):
        full = fusion.fuse_tiles(flat, np.tile(yx, (ch, 1)), plane, ch, mh, mw, flatfields=ffs,
                                 tile_ff=plane if with_ff else None, mode=mode)
        bands = []
        for rank in range(world):
            fs = mdist.FusionShard(yx[:, 0], tile, mh, world, rank, align=256)
            ids = fs.tile_ids                                  # only the tiles that touch the band are resident
            local = acq[:, ids].reshape(ch * len(ids), tile, tile)
            lp = np.repeat(np.arange(ch), len(ids)).astype(np.int32)
            band = fusion.fuse_tiles(local, np.tile(fs.local_yx(yx), (ch, 1)), lp, ch, fs.rows, mw, flatfields=ffs,
                                     tile_ff=lp if with_ff else None, mode=mode)
            bands.append(band)
            assert len(ids) < n
        assert torch.equal(torch.cat(bands, dim=1), full), mode
