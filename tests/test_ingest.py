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
