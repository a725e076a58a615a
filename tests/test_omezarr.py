"""The writer side of fusion (image-stitcher_amd/omezarr.py): chunk layout on the device against NumPy
slicing and padding, the store read back chunk file by chunk file, and the attribute documents against
what stitcher.py:489-523 and :596-645 put into them.  zarr, ome-zarr and dask are not importable here:
the on-disk format is checked by this repository's own plain reader of the Zarr v2 directory layout
(parity with those libraries' readers unpinned)."""
import json
import os

import numpy as np
import pytest

from image_stitcher_amd import omezarr


def test_chunk_shape_and_level_count_rules():
    assert omezarr.reference_chunks(2048, 2048) == (1, 1, 1, 2048, 2048)
    assert omezarr.reference_chunks(3000, 20000) == (1, 1, 1, 3000, 12288)      # parameters.py:695-701
    # parameters.py:1036-1041: max(1, ceil(log2(max(w, h) / 1024 * max_dimension)))
    assert omezarr.default_pyramid_levels(1000, 900) == 1
    assert omezarr.default_pyramid_levels(37071, 37070) == 6
    assert omezarr.default_pyramid_levels(4096, 4096, max_dimension=3) == 4


def test_attribute_documents():
    ms = omezarr.multiscales_attrs("A1_t0", 3, 0.752, dz_um=1.5)
    assert len(ms) == 1 and ms[0]["version"] == "0.4" and ms[0]["name"] == "A1_t0"
    assert [a["name"] for a in ms[0]["axes"]] == ["t", "c", "z", "y", "x"]
    assert [a.get("unit") for a in ms[0]["axes"]] == ["second", None, "micrometer", "micrometer", "micrometer"]
    assert [d["path"] for d in ms[0]["datasets"]] == ["0", "1", "2"]
    for level, d in enumerate(ms[0]["datasets"]):
        (tr,) = d["coordinateTransformations"]
        assert tr == {"type": "scale", "scale": [1, 1, 1.5, 0.752 * 2 ** level, 0.752 * 2 ** level]}
    om = omezarr.omero_attrs("A1_t0", ["Fluorescence_488_nm_Ex", "BF_LED_matrix_full"], [0x1FFF00, 0xFFFFFF])
    assert om["id"] == 1 and om["version"] == "0.4" and om["name"] == "A1_t0"
    assert [c["label"] for c in om["channels"]] == ["Fluorescence_488_nm_Ex", "BF_LED_matrix_full"]
    assert [c["color"] for c in om["channels"]] == ["1FFF00", "FFFFFF"]
    assert om["channels"][0]["window"] == {"start": 0, "end": 65535, "min": 0, "max": 65535}
    assert om["channels"][0]["active"] is True and om["channels"][0]["coefficient"] == 1 and om["channels"][0]["family"] == "linear"


def _np_chunks(img, ch, cw):
    planes, h, w = img.shape
    ncy, ncx = -(-h // ch), -(-w // cw)
    padded = np.zeros((planes, ncy * ch, ncx * cw), dtype=img.dtype)
    padded[:, :h, :w] = img
    return padded.reshape(planes, ncy, ch, ncx, cw).transpose(0, 1, 3, 2, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,chunk", [((3, 300, 520), (128, 256)),     # 16-byte groups, edge chunks on both axes
                                          ((2, 257, 515), (100, 99)),      # nothing aligned: element path
                                          ((1, 64, 64), (64, 64)),          # a single chunk
                                          ((2, 130, 1000), (512, 4096)),    # chunks larger than the image
                                          ((1, 96, 1016), (32, 248)),       # aligned rows, an edge cut inside a 16-byte group
                                          ((2, 257, 515), (64, 128)),       # odd row length: 16-byte loads at 2-byte alignment
                                          ((1, 70, 37), (16, 8))])          # one 16-byte group per chunk row
def test_chunk_layout_matches_numpy(shape, chunk):
    import torch
    rng = np.random.default_rng(7)
    img = rng.integers(0, 65536, size=shape, dtype=np.uint16)
    dev = torch.from_numpy(img.view(np.int16)).cuda().view(torch.uint16)
    got = omezarr.chunk_layout(dev, *chunk).cpu().view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(got, _np_chunks(img, *chunk))
    f = rng.standard_normal(shape)
    gotf = omezarr.chunk_layout(torch.from_numpy(f).cuda(), *chunk).cpu().numpy()
    assert np.array_equal(gotf, _np_chunks(f, *chunk))


@pytest.mark.gpu
def test_written_store_reads_back(tmp_path):
    import torch
    rng = np.random.default_rng(11)
    n_c, n_z, h, w = 2, 3, 301, 530
    mosaic = rng.integers(0, 65536, size=(n_c, n_z, h, w), dtype=np.uint16)
    dev = torch.from_numpy(mosaic.view(np.int16)).cuda().view(torch.uint16)
    path = str(tmp_path / "A1_stitched.ome.zarr")
    info = omezarr.write_ome_zarr(path, dev, ["a", "b"], [0xFF0000, 0x00FF00], px_um=0.5, tile_h=128, tile_w=256,
                                  n_levels=3, name="A1_t0")
    assert info["chunks"] == (1, 1, 1, 128, 256)
    assert json.load(open(os.path.join(path, ".zgroup"))) == {"zarr_format": 2}
    attrs = json.load(open(os.path.join(path, ".zattrs")))
    assert attrs["multiscales"][0]["datasets"][2]["coordinateTransformations"][0]["scale"] == [1, 1, 1.0, 2.0, 2.0]
    assert len(attrs["omero"]["channels"]) == 2
    for level in range(3):
        meta = json.load(open(os.path.join(path, str(level), ".zarray")))
        assert meta["shape"] == [1, n_c, n_z, h >> level, w >> level] and meta["chunks"] == [1, 1, 1, 128, 256]
        assert meta["dtype"] == ("<u2" if level == 0 else "<f8") and meta["compressor"] is None and meta["order"] == "C"
        assert meta["dimension_separator"] == "/" and meta["zarr_format"] == 2
        got = omezarr.read_level(path, level)
        if level == 0:
            want = mosaic[None]
        else:   # dask.array.coarsen(np.mean, ..., trim_excess=True): float64 block means (stitcher.py:652-668)
            s = 2 ** level
            hh, ww = (h // s) * s, (w // s) * s
            want = mosaic[:, :, :hh, :ww].astype(np.float64).reshape(n_c, n_z, hh // s, s, ww // s, s).mean(axis=(3, 5))[None]
        assert got.shape == want.shape and np.array_equal(got, want)
        # every chunk file is one whole chunk
        f = os.path.join(path, str(level), "0", "1", "2", "0", "0")
        assert os.path.getsize(f) == 128 * 256 * (2 if level == 0 else 8)


def test_bitshuffle_definition_on_hand_worked_blocks():
    """The NumPy statement of Blosc's bitshuffle filter (the device kernel's check) on blocks small enough to work by
    hand.  c-blosc / numcodecs are not importable: parity with their output unpinned."""
    # eight 1-byte elements, element i = 1 << i: bit k is set in element k only -> row k = the byte with bit k set
    one = np.array([1 << i for i in range(8)], dtype=np.uint8)
    assert omezarr.bitshuffle_numpy(one, 1, 8).tolist() == [1 << k for k in range(8)]
    # sixteen uint16 elements, all 0x0100 except element 9 = 0x8003: rows of 2 bytes, row 8 b + k
    v = np.full(16, 0x0100, dtype=np.uint16)
    v[9] = 0x8003
    out = omezarr.bitshuffle_numpy(v, 2, 32).reshape(16, 2)
    assert out[0].tolist() == [0, 0b10] and out[1].tolist() == [0, 0b10]              # bits 0, 1 of the low byte: element 9
    assert out[8].tolist() == [0xFF, 0xFF & ~0b10]                                    # bit 0 of the high byte: all but 9
    assert out[15].tolist() == [0, 0b10] and not out[2:8].any() and not out[9:15].any()
    # 11 elements: the first 8 are transposed, the last 3 copied; a second, short block of 5 elements is copied whole
    w = np.arange(16, dtype=np.uint16) * 257
    got = omezarr.bitshuffle_numpy(w, 2, 22)
    assert np.array_equal(got[:16], omezarr.bitshuffle_numpy(w[:8], 2, 16)) and np.array_equal(got[16:22], w[8:11].view(np.uint8))
    assert np.array_equal(got[22:], w[11:].view(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("n_elems,elem_size,block_bytes", [(128 * 256, 2, 128 * 256 * 2),   # one chunk, one block
                                                           (1 << 20, 2, 1 << 18),          # whole 256 KiB blocks: 16-byte path
                                                           (3 * 4096 + 2048 + 77, 2, 8192),  # short last block with a tail
                                                           (5000, 8, 4096),                 # float64 level, 512-element blocks
                                                           (40000, 8, 1 << 18),             # float64, one short block
                                                           (1000, 4, 256), (777, 1, 100), (4096 + 13, 2, 1 << 16),
                                                           (7, 2, 64), (8, 2, 64)])
def test_bitshuffle_on_the_device_matches_the_definition(n_elems, elem_size, block_bytes):
    import torch
    rng = np.random.default_rng(n_elems)
    raw = rng.integers(0, 256, size=n_elems * elem_size, dtype=np.uint8)
    got = omezarr.bitshuffle(torch.from_numpy(raw).cuda(), elem_size, block_bytes).cpu().numpy()
    assert np.array_equal(got, omezarr.bitshuffle_numpy(raw, elem_size, block_bytes))


@pytest.mark.gpu
def test_bitshuffle_of_a_laid_out_chunk():
    """A chunk as the writer hands it out (chunk_layout), filtered block by block as the codec would."""
    import torch
    rng = np.random.default_rng(3)
    img = rng.integers(0, 4096, size=(1, 300, 520), dtype=np.uint16)
    laid = omezarr.chunk_layout(torch.from_numpy(img.view(np.int16)).cuda().view(torch.uint16), 128, 256)
    chunk = laid[0, 1, 1]
    got = omezarr.bitshuffle(chunk, 2, 1 << 14).cpu().numpy()
    assert np.array_equal(got, omezarr.bitshuffle_numpy(_np_chunks(img, 128, 256)[0, 1, 1], 2, 1 << 14))


@pytest.mark.gpu
def test_store_of_a_mosaic_born_in_chunk_order_equals_the_plain_one(tmp_path):
    """fuse_tiles(chunks=...) -> write_ome_zarr(chunked_hw=...) writes the files that the plain mosaic gives: level 0
    without a layout pass, the pyramid read from the chunk-ordered mosaic."""
    import filecmp
    import torch
    from image_stitcher_amd import fusion
    from tests.fusion_cases import overlapping_scene
    tiles, yx, plane, mh, mw = overlapping_scene(21, 3, 3, 128, 256, n_planes=2, jitter=3, overlap=0.1)
    plain = fusion.fuse_tiles(tiles, yx, plane, 2, mh, mw, mode="feather").reshape(2, 1, mh, mw)
    born = fusion.fuse_tiles(tiles, yx, plane, 2, mh, mw, mode="feather", chunks=(128, 256))
    born = born.reshape(2, 1, *born.shape[1:])
    a, b = str(tmp_path / "plain.zarr"), str(tmp_path / "born.zarr")
    kw = dict(channel_names=["a", "b"], channel_colors=[1, 2], px_um=0.5, tile_h=128, tile_w=256, n_levels=3)
    omezarr.write_ome_zarr(a, plain, **kw)
    omezarr.write_ome_zarr(b, born, chunked_hw=(mh, mw), **kw)
    for level in range(3):
        la, lb = omezarr.read_level(a, level), omezarr.read_level(b, level)
        assert la.shape == lb.shape and np.array_equal(la, lb)
    cmp = filecmp.dircmp(a, b)
    assert not cmp.left_only and not cmp.right_only
    assert filecmp.cmp(os.path.join(a, "0", "0", "1", "0", "1", "1"), os.path.join(b, "0", "0", "1", "0", "1", "1"), shallow=False)
    # every file of every level, byte for byte (the pyramid levels of the chunk-ordered mosaic are born in chunk order
    # too: their padding must be the zeros the layout pass writes)
    n_files = 0
    for root, _, files in os.walk(a):
        for name in files:
            other = os.path.join(b, os.path.relpath(os.path.join(root, name), a))
            assert filecmp.cmp(os.path.join(root, name), other, shallow=False), other
            n_files += 1
    assert n_files > 3 * 2 + 5


@pytest.mark.gpu
@pytest.mark.parametrize("shape,chunks,levels", [((2, 300, 520), (128, 256), 4), ((1, 1, 257, 263), (64, 64), 5),
                                                 ((3, 128, 256), (128, 256), 3), ((1, 70, 40), (16, 8), 7)])
def test_pyramid_levels_born_in_chunk_order_equal_the_laid_out_levels(shape, chunks, levels):
    """generate_pyramid(level_chunks=...) against chunk_layout of the row-major levels, padding included, from a
    row-major mosaic and from a chunk-ordered one."""
    import torch
    from image_stitcher_amd import fusion
    rng = np.random.default_rng(shape[-1])
    img = torch.from_numpy(rng.integers(0, 65536, size=shape, dtype=np.uint16).view(np.int16)).cuda().view(torch.uint16)
    ch, cw = chunks
    h, w = shape[-2:]
    plain = fusion.generate_pyramid(img, levels)
    for born in (fusion.generate_pyramid(img, levels, level_chunks=chunks),
                 fusion.generate_pyramid(omezarr.chunk_layout(img.reshape(-1, h, w), ch, cw).reshape(shape[:-2] + (-(-h // ch), -(-w // cw), ch, cw)),
                                         levels, chunked_hw=(h, w), level_chunks=chunks) if cw % 8 == 0 else None):
        if born is None:
            continue
        assert len(born) == levels
        for l in range(1, levels):
            lh, lw = h >> l, w >> l
            assert tuple(born[l].shape[-4:]) == (-(-lh // ch), -(-lw // cw), ch, cw)
            if lh == 0 or lw == 0:
                continue
            want = omezarr.chunk_layout(plain[l].reshape(-1, lh, lw), ch, cw)
            assert torch.equal(born[l].reshape(want.shape), want), (l, shape)
